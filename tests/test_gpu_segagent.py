"""GPU tests of SegAgent (reference unscreen/binseg/agent.py + deeplab/*).  Pre / post processing: bit-exact against the
oracle (index mask exact GIVEN the same logits).  Network: the HIP convs against the folded plan on torch-CPU float32."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nv():
    assert torch.cuda.is_available()
    from video_unscreen_amd import _native
    _native.load()
    return _native


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def frame_of(H, W, seed=0):
    from video_unscreen_amd.synth import SyntheticClip
    return SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)["frame"].numpy()


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 962), (256, 256, 962), (1920, 1080, 962), (200, 1920, 962), (2160, 3840, 912)])
def test_segagent_pre_bit_exact(nv, oracle_lib, H, W, L):
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    th, tw = nv.segagent_target_size(H, W, L)
    assert (th, tw) == oracle_lib.segagent_target_size(H, W, L)
    locs = nv.grid_crop_locations(th, tw)
    assert np.array_equal(locs, oracle_lib.grid_crop_locations(th, tw))
    d = dev(img)
    for i in sorted(set([0, 1, len(locs) // 2, len(locs) - 2, len(locs) - 1])):
        got = nv.segagent_pre(d, th, tw, locs[i]).cpu().numpy()
        want = oracle_lib.segagent_pre(img, th, tw, locs[i])
        assert np.array_equal(got, want), (i, np.abs(got - want).max())          # same float32 operations: bit-exact
    # a mirrored crop is the mirror of its partner
    a = nv.segagent_pre(d, th, tw, locs[0]).cpu().numpy()
    b = nv.segagent_pre(d, th, tw, locs[1]).cpu().numpy()
    assert np.array_equal(a[:, ::-1], b)


def test_im2col_and_upsample_nc(nv, oracle_lib):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(45, 61, 4, generator=g)
    x[..., 3] = 0
    OH, OW = (45 + 6 - 7) // 2 + 1, (61 + 6 - 7) // 2 + 1
    cols = torch.empty(OH, OW, 160, device="cuda")
    nv.im2col(x.cuda(), 3, 7, 7, 2, 3, cols)
    ref = F.unfold(x[..., :3].permute(2, 0, 1)[None], 7, padding=3, stride=2)[0]             # (3*49, L), rows c*49 + tap
    ref = ref.view(3, 49, OH, OW).permute(2, 3, 1, 0).reshape(OH, OW, 147)                    # -> tap*3 + c
    got = cols.cpu()
    assert torch.equal(got[..., :147], ref) and not got[..., 147:].any()
    src = torch.randn(17, 23, 8, generator=g)
    dst = torch.full((33, 47, 16), 7.0, device="cuda")
    nv.upsample_nc(src.cuda(), dst, 4)
    want = oracle_lib.upsample_nc(src.numpy(), 33, 47)
    out = dst.cpu().numpy()
    assert np.array_equal(out[..., 4:12], want)                                               # same float32 operations
    assert (out[..., :4] == 7).all() and (out[..., 12:] == 7).all()
    t = F.interpolate(src.permute(2, 0, 1)[None], size=(33, 47), mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
    assert np.abs(want - t.numpy()).max() < 1e-5                                              # and it is torch's formula


@pytest.mark.parametrize("B,H,W,C", [(1, 237, 237, 128), (3, 45, 61, 64), (2, 8, 9, 4), (1, 1, 1, 8), (12, 237, 237, 128)])
def test_maxpool3x3s2_matches_torch(nv, B, H, W, C):
    """The ResNet stems' MaxPool2d(3, 2, 1) (usc_maxpool3x3s2_nhwc_f32) against torch's on the same values: max is exact, so bit for bit;
    images stacked along y must not see each other's rows."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    x = torch.randn(B, H, W, C, generator=g)
    want = F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()
    OH, OW = want.shape[1:3]
    got = torch.full((B * OH, OW, C), 7.0, device="cuda")
    nv.maxpool3x3s2(x.view(B * H, W, C).cuda(), got, batch=B)
    assert torch.equal(got.cpu().view(B, OH, OW, C), want)
    with pytest.raises(nv.NativeError):
        nv.maxpool3x3s2(x.view(B * H, W, C).cuda(), torch.empty((B * OH + 1, OW, C), device="cuda"), batch=B)


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 962), (256, 256, 962), (300, 200, 600)])
def test_segagent_post_bit_exact(nv, oracle_lib, H, W, L):
    """Random smooth logits for every crop: merged mask and float64 scores identical to the oracle's."""
    import scipy.ndimage as ndi
    th, tw = nv.segagent_target_size(H, W, L)
    locs = nv.grid_crop_locations(th, tw)
    rng = np.random.default_rng(5)
    lg = np.zeros((len(locs), 129, 129, 32), np.float32)
    lg[..., :2] = ndi.gaussian_filter(rng.normal(0, 4, (len(locs), 129, 129, 2)), (0, 6, 6, 0)).astype(np.float32) * 6
    lg[0, :5, :5, 0] = 200.0            # saturated softmax: exp underflow path
    lg[1, :5, :5, 1] = -300.0
    want, want_s = oracle_lib.segagent_post(lg, locs, 513, 513, th, tw, H, W, want_score=True)
    got, got_s = nv.segagent_post(dev(lg), locs, 513, 513, th, tw, H, W, want_score=True)
    assert 0.1 < (want > 0).mean() < 0.9
    assert np.array_equal(got_s.cpu().numpy(), want_s)
    assert np.array_equal(got.cpu().numpy(), want)
    # scores are probabilities
    assert np.abs(want_s.sum(-1) - 1).max() < 1e-6


def test_deeplab_hip_convs_match_torch_plan(nv):
    from video_unscreen_amd.binseg.deeplab_net import DeepLabNetTorch, HipDeepLabNet, deeplab_synthetic_state_dict
    sd = deeplab_synthetic_state_dict(0)
    for (H, W) in ((513, 513), (97, 129)):
        g = torch.Generator().manual_seed(H)
        x = torch.randn(1, 3, H, W, generator=g)
        want = DeepLabNetTorch(sd)(x)[0].permute(1, 2, 0).numpy()
        net = HipDeepLabNet(sd, H, W)
        x4 = torch.cat([x[0].permute(1, 2, 0), torch.zeros(H, W, 1)], -1).contiguous().cuda()
        got = net(x4).cpu().numpy()
        assert got.shape[:2] == want.shape[:2]
        err = np.abs(got[..., :2] - want).max() / np.abs(want).max()
        assert err < 1e-4, err                        # f32 convs, different summation order: relative 1e-4
        assert not got[..., 2:].any()
        got2 = net(x4).cpu().numpy()
        assert np.array_equal(got, got2)              # fixed accumulation order: run-to-run identical


def test_deeplab_hip_net_matches_reference_golden(nv):
    """HipDeepLabNet + the device's align_corners=False upsample against the output of the REFERENCE model itself
    (unscreen/binseg/deeplab/_deeplab.py:29-60 incl. its final F.interpolate; tests/golden/gen_golden.py, keyed-deterministic
    weights), same seed as tests/test_oracle_cpu.py::test_deeplab_plan_and_segagent_geometry_match_reference_golden."""
    import os
    from video_unscreen_amd.binseg.deeplab_net import HipDeepLabNet, deeplab_synthetic_state_dict
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "deeplab_97x129.npz"))
    g = torch.Generator().manual_seed(int(z["seed"]))
    x = torch.randn(1, 3, 97, 129, generator=g)
    net = HipDeepLabNet(deeplab_synthetic_state_dict(0), 97, 129)
    x4 = torch.cat([x[0].permute(1, 2, 0), torch.zeros(97, 129, 1)], -1).contiguous().cuda()
    low = net(x4)                                                       # (h/4, w/4, 32) logits, classes in channels 0, 1
    up = torch.zeros((97, 129, low.shape[2]), device="cuda")
    nv.upsample_nc(low.contiguous(), up, 0)
    got = up.cpu().numpy()[..., :2].transpose(2, 0, 1)[None]
    assert got.shape == z["y"].shape
    err = np.abs(got - z["y"]).max() / np.abs(z["y"]).max()
    assert err < 1e-4, err


@pytest.mark.parametrize("H,W", [(270, 480), (256, 256)])
def test_segagent_vs_oracle_chain(nv, oracle_lib, H, W):
    """Whole agent vs oracle pre -> folded plan on torch-CPU -> oracle post (mask flips only where the two class scores are
    within the conv round-off), and bit-exact when the oracle post is fed the device logits."""
    from video_unscreen_amd.binseg import SegAgent
    from video_unscreen_amd.binseg.deeplab_net import DeepLabNetTorch
    agent = SegAgent(model_path=None, input_long_side=600, stride_ratio=0.5, flip=True)
    frame = frame_of(H, W)
    got = agent.forward(frame)
    th, tw = oracle_lib.segagent_target_size(H, W, 600)
    locs = oracle_lib.grid_crop_locations(th, tw)
    cpu_net = DeepLabNetTorch(agent.state_dict)
    lg = np.zeros((len(locs), 129, 129, 2), np.float32)
    for i, loc in enumerate(locs):
        x = torch.from_numpy(oracle_lib.segagent_pre(frame, th, tw, loc)[..., :3]).permute(2, 0, 1)[None]
        lg[i] = cpu_net(x)[0].permute(1, 2, 0).numpy()
    want, score = oracle_lib.segagent_post(lg, locs, 513, 513, th, tw, H, W, want_score=True)
    assert got.shape == (H, W) and got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
    assert 0.02 < (want > 0).mean() < 0.98
    diff = got != want
    assert diff.mean() < 5e-4, diff.mean()
    assert (np.abs(score[..., 1] - score[..., 0])[diff] < 1e-3).all()
    mask_dev, lgd, locs_d = agent.forward_device(dev(frame), return_logits=True)
    want2 = oracle_lib.segagent_post(lgd.cpu().numpy(), locs_d, 513, 513, th, tw, H, W)
    assert np.array_equal(mask_dev.cpu().numpy(), want2)


def test_segagent_crop_batching_is_bit_identical(nv):
    """Crops through batched network forwards (4 per forward; the 18-crop case ends with a batch of 2) give the logits and
    the mask of crop-by-crop execution bit for bit."""
    from video_unscreen_amd.binseg import SegAgent
    frame = dev(frame_of(256, 256))
    a1 = SegAgent(model_path=None, input_long_side=962, crop_batch=1)
    a4 = SegAgent(model_path=None, input_long_side=962, crop_batch=4)
    m1, l1, locs1 = a1.forward_device(frame, return_logits=True)
    m4, l4, locs4 = a4.forward_device(frame, return_logits=True)
    assert len(locs1) == 18 and np.array_equal(locs1, locs4)
    assert torch.equal(l1, l4) and torch.equal(m1, m4)
