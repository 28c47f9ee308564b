"""GPU tests of the human-parsing segmentation stage (reference unscreen/binseg/agent_human.py + human_parse.py).
Pre / post processing: bit-exact against the oracle (float32 with the same operation order; index mask exact GIVEN the
same logits).  Network: the HIP convs against the folded plan on torch-CPU float32, relative tolerance 1e-4."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nv():
    assert torch.cuda.is_available()
    from video_unscreen_amd import _native
    _native.load()
    return _native


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840), (256, 256), (1920, 1080), (33, 47), (473, 473)])
def test_humanseg_pre_bit_exact(nv, oracle_lib, H, W):
    img = np.random.default_rng(H * 7 + W).integers(0, 256, (H, W, 3), dtype=np.uint8)
    want = oracle_lib.humanseg_pre(img)                                   # (3, 473, 473)
    got = nv.humanseg_pre(dev(img)).cpu().numpy()                         # (473, 473, 4)
    assert np.array_equal(got[..., :3].transpose(2, 0, 1), want)          # float32, bit-exact
    assert np.all(got[..., 3] == 0)


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840), (256, 256), (1920, 1080), (33, 47)])
def test_humanseg_post_bit_exact(nv, oracle_lib, H, W):
    rng = np.random.default_rng(H + W)
    import scipy.ndimage as ndi
    lg = ndi.gaussian_filter(rng.normal(size=(20, 119, 119)), (0, 6, 6)).astype(np.float32) * 20
    lg[0] = lg[1:].max(0) + ndi.gaussian_filter(rng.normal(size=(119, 119)), 8).astype(np.float32) * 30   # ~half background
    want = oracle_lib.humanseg_post(lg, H, W)
    buf = np.zeros((119, 119, 32), np.float32)
    buf[..., :20] = lg.transpose(1, 2, 0)
    buf[..., 20:] = 1e9                                                   # padding channels must never be read
    got = nv.humanseg_post(dev(buf), 20, H, W).cpu().numpy()
    assert 0.02 < (want > 0).mean() < 0.98
    assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ"      # index mask: bit-exact


@pytest.mark.parametrize("H,W", [(97, 129), (473, 473)])
def test_schp_hip_convs_match_torch_plan(nv, H, W):
    from video_unscreen_amd.binseg.schp_net import HipSchpNet, SchpNetTorch, schp_synthetic_state_dict
    sd = schp_synthetic_state_dict(0)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 3, H, W, generator=g)
    want = SchpNetTorch(sd)(x)[0].permute(1, 2, 0)                        # (h, w, 20) on the CPU, float32
    net = HipSchpNet(sd, H, W)
    x4 = torch.cat([x, torch.zeros(1, 1, H, W)], 1)[0].permute(1, 2, 0).contiguous().cuda()
    got = net(x4).cpu()
    assert got.shape[:2] == want.shape[:2]
    err = float((got[..., :20] - want).abs().max() / want.abs().max())
    assert err < 1e-4, err                                               # 150+ stacked f32 convs, different summation order
    assert float(got[..., 20:].abs().max()) == 0.0                        # padded output channels stay zero
    again = net(x4).cpu()
    assert torch.equal(again, got)                                       # deterministic


def test_schp_hip_net_matches_reference_golden(nv):
    """HipSchpNet on the device against the output of the REFERENCE module itself (unscreen/binseg/human_parse.py:248-265, run by
    tests/golden/gen_golden.py on keyed-deterministic weights): same input seed, same tolerance as the CPU plan's test
    (tests/test_oracle_cpu.py::test_schp_plan_matches_reference_golden)."""
    import os
    from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "schp_97x129.npz"))
    g = torch.Generator().manual_seed(int(z["seed"]))
    x = torch.randn(1, 3, 97, 129, generator=g)
    net = HipSchpNet(schp_synthetic_state_dict(0), 97, 129)
    x4 = torch.cat([x, torch.zeros(1, 1, 97, 129)], 1)[0].permute(1, 2, 0).contiguous().cuda()
    got = net(x4).cpu().numpy()[..., :20].transpose(2, 0, 1)[None]
    assert got.shape == z["y"].shape
    err = np.abs(got - z["y"]).max() / np.abs(z["y"]).max()
    assert err < 1e-4, err                                               # CPU plan vs reference: 2e-5; + the HIP summation order


@pytest.mark.parametrize("H,W", [(270, 480), (1080, 1920)])
def test_humanseg_agent_vs_oracle_chain(nv, oracle_lib, H, W):
    """Whole agent vs oracle pre -> folded plan on torch-CPU -> oracle post.  The mask is an argmax of float logits (person iff
    some class > 0 beats the background class), so a pixel may only differ where that decision is a tie at the conv stack's
    round-off: every differing pixel must have |max_{c>0} logit - logit_0| < 1e-4 in the oracle's full-resolution logits, and
    there are at most 64 of them (the parity report of this build counts 2 - 16 at these sizes); given the SAME logits the
    post-processing is bit-exact (test above)."""
    from video_unscreen_amd.binseg import HumanSegAgent
    from video_unscreen_amd.binseg.schp_net import SchpNetTorch
    from video_unscreen_amd.synth import SyntheticClip
    agent = HumanSegAgent(model_path=None)
    frame = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)["frame"].numpy()
    got = agent.forward(frame)
    x = torch.from_numpy(oracle_lib.humanseg_pre(frame))[None]
    logits = SchpNetTorch(agent.state_dict)(x)[0].numpy()
    want = oracle_lib.humanseg_post(logits, H, W)
    assert got.shape == (H, W) and got.dtype == np.uint8 and set(np.unique(got)) <= {0, 255}
    assert 0.02 < (want > 0).mean() < 0.98
    diff = got != want
    assert diff.sum() <= 64, int(diff.sum())
    if diff.any():                                    # the differing pixels are ties: person-vs-background margin below the round-off
        up = oracle_lib.upsample_align_corners(logits, 473, 473)                         # (473, 473, 20)
        full = oracle_lib.warp_affine(up, oracle_lib.humanseg_affine(W, H, inv=True), H, W)      # the reference's H x W x 20 tensor
        margin = np.abs(full[..., 1:].max(axis=-1) - full[..., 0])
        assert margin[diff].max() < 1e-4, float(margin[diff].max())
    # and bit-exact when fed the device logits themselves
    mask_dev, lg = agent.forward_device(dev(frame), return_logits=True)
    want2 = oracle_lib.humanseg_post(lg.cpu().numpy()[..., :20].transpose(2, 0, 1), H, W)
    assert np.array_equal(mask_dev.cpu().numpy(), want2)


@pytest.mark.parametrize("plan", [1, 4])
def test_humanseg_batched_forward_is_bit_identical(nv, plan):
    """Two / three frames through one batched network forward: logits and masks equal the single-frame results bit for bit
    (tile shape and K partition of every layer depend on the per-image shape and the agent's plan_batch, not on the batch)."""
    from video_unscreen_amd.binseg import HumanSegAgent
    from video_unscreen_amd.synth import SyntheticClip
    agent = HumanSegAgent(model_path=None)
    agent.set_plan_batch(plan)
    clip = SyntheticClip(270, 480, 3, config_id=2, device="cuda")
    frames = [clip.frame(t)["frame"] for t in range(3)]
    single = []
    for f in frames:
        m, lg = agent.forward_device(f, return_logits=True)
        single.append((m.clone(), lg.clone()))                  # the logits buffer is reused by the next forward
    for n in (2, 3):
        masks = agent.forward_device_batch(frames[:n])
        net = agent.model_for(max(n, plan))                     # a short group rides in the planned-batch network
        for b in range(n):
            assert torch.equal(net.image(net.logits32, b), single[b][1]), (n, b)
            assert torch.equal(masks[b], single[b][0])


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("s,h,w,batch", [(1, 30, 30, 1), (2, 30, 30, 2), (3, 17, 23, 1), (6, 30, 30, 4), (9, 33, 35, 2)])
def test_upsample_taps_accum_equals_conv_of_upsampled(nv, s, h, w, batch, align):
    """usc_upsample_taps_accum_f32: sum_taps shift_tap(up(p W_tap)) == conv3x3(up(p), padding 1) - the identity that takes the
    PSP priors out of the parsing net's 4096-channel bottleneck (human_parse.py:83-111)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(100 + s)
    cin, cout = 8, 12
    p = torch.randn(batch, cin, s, s, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g)
    want = F.conv2d(F.interpolate(p.double(), size=(h, w), mode="bilinear", align_corners=align), wt.double(), padding=1).float()
    # q = p [W_0 .. W_8]: channels tap * cout + co, taps in (ky, kx) row-major order
    wq = wt.permute(2, 3, 0, 1).reshape(9 * cout, cin)
    q = torch.einsum("bchw,oc->bhwo", p.double(), wq.double()).float().reshape(batch * s, s, 9 * cout).contiguous().cuda()
    dst = torch.full((batch * h, w, cout), float("nan"), device="cuda")
    nv.upsample_taps_accum(q, dst, batch=batch, align_corners=align)
    got = dst.view(batch, h, w, cout).permute(0, 3, 1, 2).cpu()
    assert float((got - want).abs().max()) < 2e-5 * float(want.abs().max())
    base = torch.randn(batch * h, w, cout, generator=g).cuda()
    acc = base.clone()
    nv.upsample_taps_accum(q, acc, batch=batch, accumulate=True, align_corners=align)
    assert float((acc - (base + dst)).abs().max()) < 1e-5 * float(want.abs().max())
