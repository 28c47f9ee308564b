"""GPU tests of ISegAgent's no-BRS forward (reference unscreen/iseg/agent.py + iseg/brs/*).
Pixel stages: bit-exact against the oracle.  Network and whole agent: against the outputs of the REFERENCE's own Python on
keyed-deterministic weights (tests/golden/iseg_90x160.npz): probabilities within 1e-4, mask exact given the same probabilities."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def nv():
    assert torch.cuda.is_available()
    from video_unscreen_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLD, "iseg_90x160.npz"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _prm():
    from video_unscreen_amd.iseg.net import IsegPlan, iseg_synthetic_state_dict
    p = IsegPlan(iseg_synthetic_state_dict(0))
    return torch.cat([p.rgb_w1.reshape(-1), p.rgb_b1, p.rgb_w2.reshape(-1), p.rgb_b2]).numpy()


@pytest.mark.parametrize("H,W,L,flip", [(90, 160, 128, True), (1080, 1920, 800, True), (1920, 1080, 800, False), (256, 256, 800, True), (2160, 3840, 800, True)])
def test_iseg_pre_bit_exact(nv, oracle_lib, H, W, L, flip):
    rng = np.random.default_rng(H + W)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    in_h, in_w = nv.get_target_size(H, W, L, 1)
    pts = np.array([[in_h * 0.4, in_w * 0.5], [in_h * 0.7, in_w * 0.2], [-1, -1], [3, 5], [-1, -1], [-1, -1]], np.float32).astype(np.int32).astype(np.float32)
    want = oracle_lib.iseg_pre(img, in_h, in_w, pts, _prm(), flip=flip)
    got = nv.iseg_pre(dev(img), in_h, in_w, pts, _prm(), flip=flip).cpu().numpy()
    assert got.shape == want.shape and np.array_equal(got, want)             # same float32 operations in the same order


def test_dwconv3x3_matches_torch(nv):
    g = torch.Generator().manual_seed(4)
    for (b, h, w, c) in ((1, 17, 23, 8), (2, 30, 41, 160), (2, 113, 200, 64)):
        x = torch.randn(b, c, h, w, generator=g)
        wt = torch.randn(c, 1, 3, 3, generator=g)
        want = F.conv2d(x.double(), wt.double(), padding=1, groups=c).float()
        src = x.permute(0, 2, 3, 1).reshape(b * h, w, c).contiguous().cuda()
        dst = torch.full_like(src, float("nan"))
        nv.dwconv3x3(src, wt.view(c, 9).t().contiguous().cuda(), dst, batch=b)
        got = dst.view(b, h, w, c).permute(0, 3, 1, 2).cpu()
        assert float((got - want).abs().max()) < 1e-5 * float(want.abs().max())


@pytest.mark.parametrize("H,W,in_h,in_w,flip", [(90, 160, 72, 128, True), (1080, 1920, 450, 800, True), (225, 400, 450, 800, False), (450, 800, 450, 800, True)])
def test_iseg_post_bit_exact(nv, oracle_lib, H, W, in_h, in_w, flip):
    import scipy.ndimage as ndi
    rng = np.random.default_rng(in_h)
    nimg = 2 if flip else 1
    lh, lw = (in_h + 3) // 4, (in_w + 3) // 4
    lg = np.zeros((nimg * lh, lw, 32), np.float32)
    lg[..., 0] = ndi.gaussian_filter(rng.normal(0, 1, (nimg * lh, lw)), 4).astype(np.float32) * 40
    lg[..., 1:] = 1e9                                                        # padding channels must never be read
    want_p, want_m = oracle_lib.iseg_post(lg, in_h, in_w, flip, H, W, 0.5)
    got_p, got_m = nv.iseg_post(dev(lg), in_h, in_w, flip, H, W, 0.5)
    assert 0.1 < (want_m > 0).mean() < 0.9
    assert np.array_equal(got_p.cpu().numpy(), want_p)
    assert np.array_equal(got_m.cpu().numpy(), want_m)
    # the probabilities are torch's formula (float rounding of exp aside)
    t = torch.from_numpy(lg[..., 0]).view(nimg, 1, lh, lw)
    up = F.interpolate(t, size=(in_h, in_w), mode="bilinear", align_corners=True)
    ref = torch.sigmoid(0.5 * (up[:1] + up[1:].flip(3)) if flip else up)[0, 0].numpy()
    assert np.abs(want_p - ref).max() < 2e-6


def test_iseg_hip_net_matches_reference_golden(nv, z):
    """HipIsegNet on the reference's rgb_conv output against the reference model's logits (both images of the flip pair)."""
    from video_unscreen_amd.iseg.net import iseg_synthetic_state_dict
    from video_unscreen_amd.iseg.net_hip import HipIsegNet
    net = HipIsegNet(iseg_synthetic_state_dict(0), 72, 128, batch=2)
    x = torch.from_numpy(z["x"])                                             # (2, 3, 72, 128)
    x4 = torch.cat([x, torch.zeros(2, 1, 72, 128)], 1).permute(0, 2, 3, 1).reshape(2 * 72, 128, 4).contiguous().cuda()
    low = net(x4)                                                            # (2 * 18, 32, 32) logits at 1/4 resolution
    up = F.interpolate(low[..., 0].view(2, 1, net.out_h, net.out_w).cpu(), size=(72, 128), mode="bilinear", align_corners=True).numpy()
    err = np.abs(up - z["logits"]).max() / np.abs(z["logits"]).max()
    assert err < 1e-4, err
    assert torch.equal(net(x4), low)                                         # run-to-run identical


def test_iseg_agent_vs_reference_golden(nv, oracle_lib, z):
    """The drop-in agent end to end against the reference's own ISegAgent.forward (preprocess, model, flip pair, sigmoid,
    postprocess) on the same frame and clicks."""
    from unscreen.iseg import ISegAgent
    agent = ISegAgent(model_path=None, with_brs=False, input_long_side=int(z["input_long_side"]), prob_thresh=0.5, with_flip=True)
    clicks = [(bool(p), int(y), int(x)) for p, y, x in z["clicks"]]
    frame = z["frame"].copy()
    got = agent.forward(frame, clicks)
    assert np.array_equal(frame, z["frame"])                                 # the frame is not modified
    assert got.shape == z["mask"].shape and got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
    mask_d, dbg = agent.forward_device(dev(z["frame"]), clicks, return_debug=True)
    assert np.array_equal(dbg["points"], z["points"][0])
    prob = dbg["prob"].cpu().numpy()
    assert np.abs(prob - z["prob"]).max() < 1e-4, float(np.abs(prob - z["prob"]).max())
    flips = got != z["mask"]
    assert flips.mean() < 2e-3
    if flips.any():                                                          # a pixel may differ only where the probability sits on the threshold
        full = oracle_lib.inv_pad_resize(z["prob"], z["mask"].shape)
        assert np.abs(full[flips] - 0.5).max() < 1e-4
    # and exact given the same probabilities: the oracle's post stage on the device logits
    want_p, want_m = oracle_lib.iseg_post(dbg["logits"].cpu().numpy(), 72, 128, True, 90, 160, 0.5)
    assert np.array_equal(prob, want_p) and np.array_equal(mask_d.cpu().numpy(), want_m)
    # second golden case: no flip, one negative click, portrait frame
    agent2 = ISegAgent(model_path=None, input_long_side=128, with_flip=False)
    got2 = agent2.forward(z["frame2"].copy(), [(False, 80, 45)])
    assert (got2 != z["mask2"]).mean() < 2e-3
    with pytest.raises(NotImplementedError):
        ISegAgent(model_path=None, with_brs=True)


def test_iseg_agent_full_size_vs_cpu_plan(nv, oracle_lib):
    """1080p frame at the default input_long_side = 800 (450 x 800 network input): device path against oracle pre -> folded plan on
    torch-CPU -> oracle post."""
    from video_unscreen_amd.iseg import ISegAgent
    from video_unscreen_amd.iseg.net import IsegNetTorch
    from video_unscreen_amd.synth import SyntheticClip
    H, W = 1080, 1920
    frame = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)["frame"].numpy()
    clicks = [(True, 540, 960), (True, 700, 1100), (False, 50, 60), (False, 1000, 1800)]
    agent = ISegAgent(model_path=None)
    mask_d, dbg = agent.forward_device(dev(frame), clicks, return_debug=True)
    in_h, in_w = dbg["in_size"]
    assert (in_h, in_w) == (450, 800)
    x4 = oracle_lib.iseg_pre(frame, in_h, in_w, dbg["points"], agent.model_for(in_h, in_w).rgb_params(), flip=True)
    assert np.array_equal(dbg["x4"].cpu().numpy(), x4)
    x = torch.from_numpy(x4.reshape(2, in_h, in_w, 4)[..., :3]).permute(0, 3, 1, 2).contiguous()
    low = IsegNetTorch(agent.state_dict).features(x)                         # (2, 1, 113, 200)
    got_low = dbg["logits"][..., 0].view(2, 1, low.shape[2], low.shape[3]).cpu()
    err = float((got_low - low).abs().max() / low.abs().max())
    assert err < 2e-4, err
    lg = np.zeros((2 * low.shape[2], low.shape[3], 1), np.float32)
    lg[..., 0] = low.numpy().reshape(-1, low.shape[3])
    want_p, want_m = oracle_lib.iseg_post(lg, in_h, in_w, True, H, W, 0.5)
    assert np.abs(dbg["prob"].cpu().numpy() - want_p).max() < 1e-4
    assert 0.02 < (want_m > 0).mean() < 0.98
    assert (mask_d.cpu().numpy() != want_m).mean() < 1e-3
