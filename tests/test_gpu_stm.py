"""GPU tests of STMAgent (reference unscreen/stm/agent.py + model.py).  Pre / post processing: bit-exact against the oracle
(mask exact GIVEN the same decoder logits).  Network: the HIP convs / memory read against the folded plan on torch-CPU."""
import math

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


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 960), (256, 256, 960), (1920, 1080, 960), (135, 241, 320)])
def test_stm_pre_bit_exact(nv, oracle_lib, H, W, L):
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = rng.integers(0, 256, (H, W), dtype=np.uint8)
    in_h, in_w = nv.get_target_size(H, W, L, 16)
    assert np.array_equal(nv.stm_pre(dev(img), dev(mask), in_h, in_w).cpu().numpy(), oracle_lib.stm_pre(img, mask, in_h, in_w))
    assert np.array_equal(nv.stm_pre(dev(img), None, in_h, in_w).cpu().numpy(), oracle_lib.stm_pre(img, None, in_h, in_w))


def test_pack_and_row_softmax(nv):
    g = torch.Generator().manual_seed(0)
    src = torch.randn(70, 24, generator=g).cuda()                       # (rows, cols) row-major
    # (K = col, N = row): element (k, n) = src[n, k]
    dst = torch.full((32 // 4, 96, 4), 9.0, device="cuda")
    nv.pack_kn(src, 24, 70, 1, 24, dst)
    full = torch.zeros(32, 96)
    full[:24, :70] = src.cpu().t()
    assert torch.equal(dst.cpu(), full.view(8, 4, 96).permute(0, 2, 1))
    # (K = row, N = col)
    dst = torch.full((96 // 4, 32, 4), 9.0, device="cuda")
    nv.pack_kn(src, 70, 24, 24, 1, dst)
    full = torch.zeros(96, 32)
    full[:70, :24] = src.cpu()
    assert torch.equal(dst.cpu(), full.view(24, 4, 32).permute(0, 2, 1))
    x = torch.randn(50, 96, generator=g) * 5
    y = x.clone().cuda()
    nv.row_softmax(y, 70, math.sqrt(128))
    want = F.softmax(x[:, :70] / math.sqrt(128), dim=1)
    assert (y.cpu()[:, :70] - want).abs().max() < 1e-6 and not y.cpu()[:, 70:].any()


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 960), (256, 256, 960), (135, 241, 320)])
def test_stm_post_bit_exact(nv, oracle_lib, H, W, L):
    import scipy.ndimage as ndi
    in_h, in_w = nv.get_target_size(H, W, L, 16)
    rng = np.random.default_rng(3)
    p2 = np.zeros((in_h // 4, in_w // 4, 32), np.float32)
    p2[..., :2] = ndi.gaussian_filter(rng.normal(0, 4, (in_h // 4, in_w // 4, 2)), (5, 5, 0)).astype(np.float32) * 8
    p2[:4, :4, 0] = 90.0           # saturated: the 1e-7 clamp of the soft aggregation
    p2[:4, 4:8, 1] = 90.0
    want, want_p = oracle_lib.stm_post(p2, in_h, in_w, H, W, want_prob=True)
    got, got_p = nv.stm_post(dev(p2), in_h, in_w, H, W)
    assert 0.1 < (want > 0).mean() < 0.9
    assert np.array_equal(got_p.cpu().numpy(), want_p)
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(nv.stm_mask_from_prob(dev(want_p), H, W).cpu().numpy(), want)
    # the probabilities are the reference's formula (float rounding of exp / log aside)
    t = torch.from_numpy(p2[..., :2]).permute(2, 0, 1)[None]
    ps = F.softmax(F.interpolate(t, scale_factor=4, mode="bilinear", align_corners=False), 1)[:, 1]
    em = torch.stack([1 - ps[0], ps[0]], 0)[None].clamp(1e-7, 1 - 1e-7)
    ref = F.softmax(torch.log(em / (1 - em)), 1)[0].permute(1, 2, 0).numpy()
    assert np.abs(want_p - ref).max() < 2e-6


def _inputs(H, W, seed):
    g = torch.Generator().manual_seed(seed)
    f0, f1 = torch.randn(1, 3, H, W, generator=g), torch.randn(1, 3, H, W, generator=g)
    m = (F.avg_pool2d(torch.rand(1, 1, H, W, generator=g), 9, 1, 4) > 0.5).float()
    return f0, f1, torch.cat([1 - m, m], 1)


def _x8(frame, masks=None):
    H, W = frame.shape[2:]
    x = torch.zeros(H, W, 8)
    x[..., :3] = frame[0].permute(1, 2, 0)
    if masks is not None:
        x[..., 3], x[..., 4] = masks[0, 1], masks[0, 0]
    return x.cuda()


@pytest.mark.parametrize("H,W", [(64, 96), (272, 480), (544, 960)])          # 544 x 960: the network input of a 1080p / 4K frame
def test_stm_hip_net_matches_torch_plan(nv, H, W):
    from video_unscreen_amd.stm.net import StmNetTorch, stm_synthetic_state_dict
    from video_unscreen_amd.stm.net_hip import HipStmNet
    sd = stm_synthetic_state_dict(0)
    cpu = StmNetTorch(sd)
    net = HipStmNet(sd, H, W, capacity=3)
    f0, f1, masks = _inputs(H, W, H)
    k4, v4 = cpu.memorize(f0, masks)
    net.x8.copy_(_x8(f0, masks))
    net.memorize(0)
    hw = net.hw
    gk = net.kbank[:hw].cpu().numpy().reshape(net.h16, net.w16, -1).transpose(2, 0, 1)
    gv = net.vbank[:hw].cpu().numpy().reshape(net.h16, net.w16, -1).transpose(2, 0, 1)
    assert np.abs(gk - k4[0].numpy()).max() < 1e-4 * np.abs(k4.numpy()).max()        # f32 convs, other summation order
    assert np.abs(gv - v4[0].numpy()).max() < 1e-4 * np.abs(v4.numpy()).max()
    # segment against one and against two memory frames (the second: the flipped first, so the read differs)
    k2, v2 = cpu.memorize(f1, masks.flip(3))
    net.x8.copy_(_x8(f1, masks.flip(3)))
    net.memorize(1)
    for T, (keys, values) in ((1, (k4.unsqueeze(2), v4.unsqueeze(2))),
                              (2, (torch.stack([k4, k2], 2), torch.stack([v4, v2], 2)))):
        want = cpu.decoder_logits(f1, keys, values)[0].permute(1, 2, 0).numpy()
        net.x8.copy_(_x8(f1))
        got = net.segment(T).cpu().numpy()
        err = np.abs(got[..., :2] - want).max() / np.abs(want).max()
        assert err < 2e-4, (T, err)
        assert not got[..., 2:].any()
        got2 = net.segment(T).cpu().numpy()
        assert np.array_equal(got, got2)                                              # run-to-run identical


def test_stm_hip_net_matches_reference_golden(nv):
    """HipStmNet on the device against the REFERENCE model's own memorize() / segment() outputs (unscreen/stm/model.py:185-219;
    tests/golden/gen_golden.py, keyed-deterministic weights): key / value of the memorised frame, and the class probabilities of
    segment() for one and for two memory frames (the second entry built from the first exactly as the generator did), through
    the device's own post stage (usc_stm_post: x4 upsample, softmax, soft aggregation).  Tolerances as in
    tests/test_oracle_cpu.py::test_stm_plan_matches_reference_golden."""
    import os
    from video_unscreen_amd.stm.net import stm_synthetic_state_dict
    from video_unscreen_amd.stm.net_hip import HipStmNet
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "stm_64x96.npz"))
    g = torch.Generator().manual_seed(int(z["seed"]))
    f0, f1 = torch.randn(1, 3, 64, 96, generator=g), torch.randn(1, 3, 64, 96, generator=g)
    m = (torch.rand(1, 1, 64, 96, generator=g) > 0.5).float()
    net = HipStmNet(stm_synthetic_state_dict(0), 64, 96, capacity=3)
    net.x8.copy_(_x8(f0, torch.cat([1 - m, m], 1)))
    net.memorize(0)
    hw = net.hw
    gk = net.kbank[:hw].cpu().numpy().reshape(net.h16, net.w16, -1).transpose(2, 0, 1)[None]
    gv = net.vbank[:hw].cpu().numpy().reshape(net.h16, net.w16, -1).transpose(2, 0, 1)[None]
    assert np.abs(gk - z["k4"]).max() < 1e-4 * np.abs(z["k4"]).max()
    assert np.abs(gv - z["v4"]).max() < 1e-4 * np.abs(z["v4"]).max()
    # second memory entry: keys * 0.5 + 0.1, values flipped along y - dim 3 of (1, C, T, h, w) - (gen_golden.py builds logit2 from exactly this)
    net.kbank[hw:2 * hw].copy_(net.kbank[:hw] * 0.5 + 0.1)
    net.vbank[hw:2 * hw].copy_(net.vbank[:hw].view(net.h16, net.w16, -1).flip(0).reshape(hw, -1))
    for T, name in ((1, "logit1"), (2, "logit2")):
        net.x8.copy_(_x8(f1))
        p2 = net.segment(T)
        mask, prob = nv.stm_post(p2, 64, 96, 64, 96)
        ref = torch.softmax(torch.from_numpy(z[name]), 1)[0].permute(1, 2, 0).numpy()
        assert np.abs(prob.cpu().numpy() - ref).max() < 2e-4, (T, float(np.abs(prob.cpu().numpy() - ref).max()))
        assert (mask.cpu().numpy() != (ref[..., 1] > ref[..., 0]) * 255).mean() < 1e-3


@pytest.mark.parametrize("H,W", [(270, 480)])
def test_stm_agent_vs_oracle_chain(nv, oracle_lib, H, W):
    """STMAgent.forward on 3 frames vs oracle pre -> folded plan on torch-CPU -> oracle post, step by step (the mask may flip
    only where the two probabilities are within the conv round-off), and exact when the oracle post gets the device logits."""
    from video_unscreen_amd.stm import STMAgent
    from video_unscreen_amd.stm.net import StmNetTorch
    from video_unscreen_amd.synth import SyntheticClip
    agent = STMAgent(model_path=None, memory_step=2, memory_capacity=10, input_long_side=480)
    clip = SyntheticClip(H, W, 3, config_id=2, device="cpu")
    fr = [clip.frame(t) for t in range(3)]
    frames = [f["frame"].numpy() for f in fr]
    mask0 = fr[0]["seg"].numpy()
    got = agent.forward(frames, mask0)
    assert len(got) == 3 and all(m.shape == (H, W) and m.dtype == np.uint8 for m in got)
    in_h, in_w = oracle_lib.get_target_size(H, W, 480, 16)
    cpu = StmNetTorch(agent.state_dict)
    x0 = oracle_lib.stm_pre(frames[0], mask0, in_h, in_w)
    prob = np.stack([x0[..., 4], x0[..., 3]], -1)
    # frame 0: argmax of the resized one-hot
    p2 = np.zeros((in_h // 4, in_w // 4, 2), np.float32)
    want0 = np.asarray(nv.stm_mask_from_prob(dev(prob), H, W).cpu())
    assert np.array_equal(got[0], want0)
    assert (got[0] != (mask0 > 127) * 255).mean() < 0.02              # a down-up resize of the mask: only edge pixels move
    nchw = lambda a: torch.from_numpy(np.ascontiguousarray(a)).permute(2, 0, 1)[None]          # noqa: E731
    keys = values = None
    for t in (1, 2):
        xp = oracle_lib.stm_pre(frames[t - 1], None, in_h, in_w)
        k, v = cpu.memorize(nchw(xp[..., :3]), nchw(prob))
        if t == 1:
            ik, iv = k.unsqueeze(2), v.unsqueeze(2)
        else:
            ik, iv = torch.cat([keys, k.unsqueeze(2)], 2), torch.cat([values, v.unsqueeze(2)], 2)
        xq = oracle_lib.stm_pre(frames[t], None, in_h, in_w)
        p2 = cpu.decoder_logits(nchw(xq[..., :3]), ik, iv)[0].permute(1, 2, 0).numpy()
        want, prob = oracle_lib.stm_post(p2, in_h, in_w, H, W, want_prob=True)
        assert 0.02 < (want > 0).mean() < 0.98
        assert (got[t] != want).mean() < 2e-3, (t, (got[t] != want).mean())
        if t - (1 % 2) == 0:                                          # the reference's update condition (agent.py:166)
            keys, values = ik, iv
    masks_d, dbg = agent.forward_device([dev(f) for f in frames], dev(mask0), return_debug=True)
    assert [d["T"] for d in dbg] == [1, 2]
    for t in (1, 2):
        want = oracle_lib.stm_post(dbg[t - 1]["p2"].cpu().numpy(), in_h, in_w, H, W)
        assert np.array_equal(masks_d[t].cpu().numpy(), want)


def test_stm_agent_memory_policies(nv):
    """Longer clips: the reference's update rule (bank frozen after t = 1, agent.py:166) and the repaired rule with a small
    capacity (oldest entry dropped, slots compacted): memory sizes per step, determinism, and the compaction keeping the
    right entries (the 6-frame result's first 4 masks equal the 4-frame result's)."""
    from video_unscreen_amd.stm import STMAgent
    from video_unscreen_amd.synth import SyntheticClip
    H, W = 135, 240
    clip = SyntheticClip(H, W, 6, config_id=2, device="cuda")
    fr = [clip.frame(t) for t in range(6)]
    frames, mask0 = [f["frame"] for f in fr], fr[0]["seg"]
    quirk = STMAgent(model_path=None, memory_step=2, memory_capacity=10, input_long_side=240)
    masks, dbg = quirk.forward_device(frames, mask0, return_debug=True)
    assert [d["T"] for d in dbg] == [1, 2, 2, 2, 2]
    fixed = STMAgent(model_path=None, memory_step=1, memory_capacity=3, input_long_side=240, fix_memory_step=True)
    m6, d6 = fixed.forward_device(frames, mask0, return_debug=True)
    assert [d["T"] for d in d6] == [1, 2, 3, 3, 3]
    m6b, _ = fixed.forward_device(frames, mask0, return_debug=True)
    assert all(torch.equal(a, b) for a, b in zip(m6, m6b))
    m4, _ = fixed.forward_device(frames[:4], mask0, return_debug=True)
    assert all(torch.equal(a, b) for a, b in zip(m6[:4], m4))
    with pytest.raises(RuntimeError):                          # memory_step = 1 with the reference's rule: its NameError
        STMAgent(model_path=None, memory_step=1, memory_capacity=3, input_long_side=240).forward_device(frames[:3], mask0)
