"""GPU parity tests of the green-mode stages (tools/unscreen/green.py:70-126): colour conversion, colour filtering,
colour correction, box filter - through the C ABI, against the CPU oracle.  Byte results must be bit-exact."""
import numpy as np
import pytest
import torch

from green_ref import OracleColorFilter, green_screen_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nv():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from video_unscreen_amd import _native
    _native.load()
    return _native


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def all_colours():
    """Every (b, g, r) on a 52-step lattice plus the 8 cube corners and 4096 random triples."""
    ax = np.arange(0, 256, 5, dtype=np.uint8)
    lat = np.stack(np.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3)
    rnd = np.random.default_rng(5).integers(0, 256, (4096, 3), dtype=np.uint8)
    cor = np.array([[b, g, r] for b in (0, 255) for g in (0, 255) for r in (0, 255)], np.uint8)
    px = np.concatenate([lat, rnd, cor])
    pad = (-len(px)) % 64
    px = np.concatenate([px, np.zeros((pad, 3), np.uint8)])
    return px.reshape(-1, 64, 3)


def test_cvt_color_bit_exact(nv, oracle_lib):
    img = all_colours()
    d = dev(img)
    assert np.array_equal(host(nv.cvt_color(d, nv.BGR2HSV)), oracle_lib.bgr2hsv(img))
    assert np.array_equal(host(nv.cvt_color(d, nv.HSV2BGR)), oracle_lib.hsv2bgr(img))
    assert np.array_equal(host(nv.cvt_color(d, nv.BGR2GRAY)), oracle_lib.bgr2gray(img))
    assert np.array_equal(host(nv.cvt_color(d, nv.BGR2LAB)), oracle_lib.bgr2lab(img))
    assert np.array_equal(host(nv.cvt_color(d, nv.LAB2BGR)), oracle_lib.lab2bgr(img))      # all 2^24 (L, a, b) triples
    # known OpenCV values (8-bit Lab): white, black, pure green
    lab = host(nv.cvt_color(dev(np.array([[[255, 255, 255], [0, 0, 0], [0, 255, 0]]], np.uint8)), nv.BGR2LAB))
    assert lab.tolist() == [[[255, 128, 128], [0, 128, 128], [224, 42, 211]]]


def _mixture(rng, K, lo, hi):
    m = rng.uniform(lo, hi, K)
    s = rng.uniform(1.5, 25, K)
    w = rng.random(K) + 0.05
    return m, s, w / w.sum()


@pytest.mark.parametrize("H,W", [(540, 960), (33, 47)])
def test_gmm_alpha_bit_exact(nv, oracle_lib, H, W):
    rng = np.random.default_rng(3)
    mix = [_mixture(rng, k, 0, 255) for k in (3, 5, 5, 10, 10, 10)]
    luts_o = np.stack([oracle_lib.gmm_lut(*m) for m in mix])
    luts_n = np.stack([nv.gmm_lut(*m) for m in mix])
    assert np.array_equal(luts_o, luts_n)                          # same float32 host arithmetic
    # the table is the reference's formula: compare with a float64 evaluation (get_prob_by_gmm, agent.py:201-230)
    x = np.arange(256)[None, :]
    for (m, s, w), lut in zip(mix, luts_o):
        ref = (w[:, None] / (s[:, None] * np.sqrt(2 * np.pi)) * np.exp(-0.5 * ((x - m[:, None]) / s[:, None]) ** 2)).sum(0)
        assert np.abs(lut - ref).max() <= 1e-6 * max(ref.max(), 1e-3)
    hsv = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    hsv[..., 0] %= 180
    want_a, want_p = oracle_lib.gmm_alpha(hsv, luts_o[:3], luts_o[3:], want_prob=True)
    got_a, got_p = nv.gmm_alpha(dev(hsv), dev(luts_n), want_prob=True)
    assert np.array_equal(host(got_p), want_p)                     # float32, correctly rounded ops only: bit-exact
    assert np.array_equal(host(got_a), want_a)
    # adaptive threshold + close/open
    mask = (rng.random((H, W)) > 0.5).astype(np.uint8) * 255
    from video_unscreen_amd.colorfiltering import ColorFilteringAgent
    got = host(ColorFilteringAgent().postprocess_device(dev(want_a.copy()), dev(mask)))
    assert np.array_equal(got, oracle_lib.cf_postprocess(want_a, mask))
    # empty consistent area: mean of nothing is nan, nothing is zeroed
    z = np.zeros_like(mask)
    got = host(nv.cf_threshold(dev(want_a.copy()), dev(z)))
    assert np.array_equal(got, want_a)


@pytest.mark.parametrize("H,W", [(1080, 1920), (256, 256), (135, 241)])
def test_color_filtering_agent_matches_oracle_flow(nv, oracle_lib, H, W):
    """Whole ColorFilteringAgent.forward (training frame with 3 EM rounds, then an inference-only frame) against the
    oracle-side restatement with identically seeded sklearn mixtures: alpha and background bit-exact."""
    from video_unscreen_amd.colorfiltering import ColorFilteringAgent
    frame, coarse = green_screen_clip(H, W, 11)
    agent = ColorFilteringAgent(random_state=0)
    ref = OracleColorFilter(oracle_lib, random_state=0)
    for iters in (3, 0):
        alpha, bg, conf = agent.forward(frame.copy(), coarse.copy(), iters=iters)
        want_alpha, want_bg = ref.forward(frame.copy(), coarse.copy(), iters)
        assert np.array_equal(bg, want_bg)
        assert np.array_equal(alpha, want_alpha), f"{(alpha != want_alpha).sum()} px differ"
        assert isinstance(conf, float)
    assert agent.is_trained()
    # the screen is removed, the person kept
    assert alpha[5, 5] == 0 and alpha[int(H * 0.55), W // 2] > 200
    # degenerate masks come back untouched, as in the reference (agent.py:303-308)
    empty = np.zeros((H, W), np.uint8)
    a, b, c = agent.forward(frame, empty, iters=0)
    assert np.array_equal(a, empty) and np.array_equal(b, frame) and c == 1.0
    full = np.full((H, W), 255, np.uint8)
    a, b, c = agent.forward(frame, full, iters=0)
    assert np.array_equal(a, full) and not b.any() and c == 1.0


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840), (256, 256), (135, 241), (700, 500)])
def test_color_correct_bit_exact(nv, oracle_lib, H, W):
    from video_unscreen_amd.utils import color_correct
    frame, coarse = green_screen_clip(H, W, 21)
    rng = np.random.default_rng(22)
    alpha = coarse.copy()
    alpha[rng.random((H, W)) < 0.3] = 0
    alpha = np.where(coarse > 0, rng.integers(1, 256, (H, W)), 0).astype(np.uint8)
    bg_color = frame[0, 0]
    want, it_o = oracle_lib.color_correct(frame, alpha, bg_color, want_iters=True)
    got, it_n = nv.color_correct(dev(frame), dev(alpha), bg_color, want_iters=True)
    assert it_n == it_o and it_o >= 1
    assert np.array_equal(host(got), want)
    assert np.array_equal(color_correct(frame, alpha, bg_color), want)
    # an all-zero alpha: the masked mean is nan, the loop is skipped, the result is zero
    z = np.zeros((H, W), np.uint8)
    got, it_n = nv.color_correct(dev(frame), dev(z), bg_color, want_iters=True)
    assert it_n == 0 and not host(got).any()


@pytest.mark.parametrize("H,W,cn", [(540, 960, 1), (270, 480, 3), (33, 47, 3), (1, 9, 1), (7, 1, 3)])
def test_box_filter_bit_exact(nv, oracle_lib, H, W, cn):
    rng = np.random.default_rng(31)
    img = rng.integers(0, 256, (H, W, cn) if cn == 3 else (H, W), dtype=np.uint8)
    for k in (3, 5, 1):
        assert np.array_equal(host(nv.box_filter(dev(img), k)), oracle_lib.box_filter(img, k))
    # normalised box: constants are preserved, the mean of a 0/255 checkerboard rounds like (s * 7282 + ...) >> 16
    c = np.full_like(img, 77)
    assert np.array_equal(host(nv.box_filter(dev(c), 3)), c)


def test_harmonization_agent(nv, oracle_lib):
    from video_unscreen_amd.harmonization import HarmonizationAgent
    ha = HarmonizationAgent()
    rng = np.random.default_rng(41)
    alpha = (rng.random((270, 480)) > 0.6).astype(np.uint8) * 255
    th, tw = oracle_lib.get_target_size(270, 480, 1920)
    want = oracle_lib.resize_linear(alpha, th, tw)
    for _ in range(3):
        want = oracle_lib.box_filter(want, 3)
    want = oracle_lib.resize_linear(want, 270, 480)
    assert np.array_equal(ha.alpha_smoothing(alpha), want)
    bg = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    th, tw = oracle_lib.get_target_size(270, 480, 480)
    want = bg if (th, tw) == (270, 480) else oracle_lib.resize_linear(bg, th, tw)
    for _ in range(3):
        want = oracle_lib.box_filter(want, 3)
    want = oracle_lib.resize_linear(want, 270, 480)
    assert np.array_equal(ha.background_blurring(bg), want)
    th, tw = oracle_lib.get_target_size(270, 480, 240)
    small = oracle_lib.resize_linear(bg, th, tw)
    assert np.allclose(ha.get_means(bg), small.reshape(-1, 3).mean(axis=0), rtol=0, atol=1e-9)
    m = oracle_lib.resize_linear(alpha, th, tw) > 0
    assert np.allclose(ha.get_means(bg, alpha > 0), small[m].reshape(-1, 3).mean(axis=0), rtol=0, atol=1e-9)
    # foreground_toning (agent.py:39-76) against the oracle chain: Lab, region means, clamped shifts in float64, clip, truncation, Lab -> BGR
    fg = rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    for kw in ({}, dict(toning_ratio=(1.0, 1.0, 1.0), max_shift=40, min_shift=-40)):
        ratio, hi, lo = kw.get("toning_ratio", (0.5, 0.05, 0.05)), kw.get("max_shift", 15), kw.get("min_shift", -30)
        fl, bl = oracle_lib.bgr2lab(fg), oracle_lib.bgr2lab(bg)
        fm = oracle_lib.resize_linear(fl, th, tw)[m].reshape(-1, 3).mean(axis=0)
        bm = oracle_lib.resize_linear(bl, th, tw).reshape(-1, 3).mean(axis=0)
        f = fl.astype(np.float64)
        for i in range(3):
            f[:, :, i] = f[:, :, i] + max(min(ratio[i] * (bm[i] - fm[i]), hi), lo)
        want = oracle_lib.lab2bgr(np.clip(f, 0, 255).astype(np.uint8))
        assert np.array_equal(ha.foreground_toning(fg, bg, alpha, **kw), want)


GREEN_CFG = {
    "colorfiltering_update_duration": 30, "colorfiltering_train_iters": 3, "fg_exist_thr": 0.001,
    "binseg": {"model_path": "synthetic", "input_long_side": 600, "stride_ratio": 0.5, "flip": True},
    "colorfiltering": {"input_long_side": 960, "bg_ncomp": [3, 5, 5], "fg_ncomp": [10, 10, 10], "max_num_samples": 10000,
                       "color_prior_winsize": 30, "use_opencv_gmm": False},
    "objectremoval": {"score_map_center": {"landscape": [0.5, 0.5], "portrait": [0.6, 0.5]}, "saliency_thr": 0.005,
                      "consensus_thr": 0.5},
    "trimap": {"input_long_side": 960, "kernelsize": 3, "iters": 5, "color_winsize": [10, 100, 180]},
    "vmatting": {"model_path": "synthetic", "input_long_side": 960},
}


class _FixedSeg:
    """Stands in for SegAgent where the test wants a known coarse mask (SegAgent itself: tests/test_gpu_segagent.py)."""

    def __init__(self, masks):
        self.masks, self.calls = masks, 0

    def forward_device(self, frame):
        self.calls += 1
        return dev(self.masks[min(self.calls - 1, len(self.masks) - 1)])


@pytest.mark.parametrize("H,W,real", [(270, 480, False), (256, 256, True)])
def test_green_pipeline_stage_by_stage(nv, oracle_lib, H, W, real):
    """Three frames of the green loop (train, track, track): every stage's device output is checked against the oracle
    applied to the previous stage's device output - byte stages bit-exact, the matting alpha within 1 level.
    256 x 256 = BASELINE.json configs[0] as a whole loop (reference tools/unscreen/green.py:70-138 with configs/green.json): the
    untracked frame goes through the real SegAgent (DeepLabv3+ over 18 crops at long side 962) and is compared with the CPU chain
    (oracle/chains.py:seg_agent; the mask may flip only where the two class scores tie)."""
    import copy
    from video_unscreen_amd.pipeline_green import GreenPipeline
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    cfg = copy.deepcopy(GREEN_CFG)
    if real:
        cfg["binseg"]["input_long_side"] = 962                       # configs/green.json
    clip = SyntheticClip(H, W, 3, config_id=1, device="cpu")
    frames = [clip.frame(t) for t in range(3)]
    pipe = GreenPipeline(cfg, H, W, segagent=None if real else _FixedSeg([frames[0]["seg"].numpy()]), cf_random_state=0)
    ref_cf = OracleColorFilter(oracle_lib, random_state=0)
    cpu_net = MattingNet(synthetic_state_dict(0), device="cpu", dtype=torch.float32)
    alpha_pre = np.zeros((H, W), np.uint8)
    tracking = False
    for t, f in enumerate(frames):
        frame = f["frame"].numpy()
        out = pipe.step(dev(frame))
        assert out["tracking"] == tracking
        seg = host(out["segmask"])
        if real and not tracking:
            from oracle import chains
            from video_unscreen_amd.binseg.deeplab_net import DeepLabNetTorch
            want_seg = chains.seg_agent(oracle_lib, frame, DeepLabNetTorch(pipe.segagent.state_dict), 962)
            assert 0.02 < (want_seg > 0).mean() < 0.98 and (seg != want_seg).mean() < 5e-4, float((seg != want_seg).mean())
        else:
            assert np.array_equal(seg, alpha_pre if tracking else f["seg"].numpy())
        iters = 3 if (t % 30 == 0 or not tracking) else 0
        assert out["cf_iters"] == iters
        want_cf, want_bg = ref_cf.forward(frame.copy(), seg.copy(), iters)
        assert np.array_equal(host(out["alphacf"]), want_cf)
        bg_color = want_bg[0, 0]
        assert np.array_equal(out["bg_color"], bg_color)
        want_or = oracle_lib.remove_invalid_objects(cfg, want_cf.copy(), None if tracking else seg.copy())
        assert np.array_equal(host(out["alphaor"]), want_or)
        want_tri = oracle_lib.trimap_withbg(want_or, frame, bg_color, tuple(cfg["trimap"]["color_winsize"]))
        assert np.array_equal(host(out["trimap"]), want_tri)
        in_h, in_w = oracle_lib.vmat_input_size(H, W, 960)
        x7 = oracle_lib.vmat_pre(frame, alpha_pre, want_tri, in_h, in_w)
        pred = cpu_net(torch.from_numpy(x7)[None]).numpy()[0, 0]
        want_mat = oracle_lib.vmat_post(pred, want_tri)
        got_mat = host(out["alpha_mat"])
        d = np.abs(got_mat.astype(int) - want_mat.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3          # truncation of a float alpha that agrees to ~1e-6
        want_alpha = oracle_lib.color_correct(frame, got_mat, bg_color)
        assert np.array_equal(host(out["alpha"]), want_alpha)
        bgp = want_bg.copy()
        want_fg, want_bgp = oracle_lib.get_fg(frame, want_alpha, bgp, paste_thr=128)
        assert np.array_equal(host(out["fg"]), want_fg)
        assert np.array_equal(host(out["bg"]), want_bgp)
        alpha_pre = want_alpha
        tracking = bool((want_alpha >= 128).sum() > cfg["fg_exist_thr"] * H * W)
        assert pipe.tracking_flag == tracking
    assert pipe.tracking_count == 2
    # a frame with nothing in it: zero alpha / fg, the frame itself as background (green.py:82-85)
    pipe.reset()
    pipe.segagent = _FixedSeg([np.zeros((H, W), np.uint8)])
    out = pipe.step(dev(frames[0]["frame"].numpy()))
    assert not host(out["alpha"]).any() and not host(out["fg"]).any() and np.array_equal(host(out["bg"]), frames[0]["frame"].numpy())
    assert pipe.tracking_flag is False


def test_green_tool_end_to_end(tmp_path):
    """tools/unscreen/green.py on a small synthetic green-screen dataset on disk, SegAgent (DeepLabv3+) included."""
    import json
    import os
    import subprocess
    import sys
    from video_unscreen_amd.binseg.deeplab_net import deeplab_synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils.fileio import read_img, save_img
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    H, W, n = 270, 480, 3
    root = tmp_path / "data"
    src = root / "src_img" / "3"
    src.mkdir(parents=True)
    clip = SyntheticClip(H, W, n, config_id=1, device="cpu")
    for t in range(n):
        save_img(clip.frame(t)["frame"].numpy(), str(src / f"{t:06d}.jpg"))
    torch.save({"state_dict": synthetic_state_dict(0)}, str(tmp_path / "unet.pth"))
    torch.save(deeplab_synthetic_state_dict(0), str(tmp_path / "deeplab.pth"))
    cfg = json.load(open(os.path.join(ROOT, "configs", "green.json")))
    cfg["vmatting"]["model_path"] = str(tmp_path / "unet.pth")
    cfg["binseg"]["model_path"] = str(tmp_path / "deeplab.pth")
    cfg["binseg"]["input_long_side"] = 600
    json.dump(cfg, open(tmp_path / "green.json", "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "unscreen", "green.py"), "--cfg", str(tmp_path / "green.json"),
                        "--video_id", "3", "--data_root", str(root)], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "use tracking" in r.stdout
    dst = root / "test_img" / "3"
    for t in range(n):
        for mode in ("alphamask", "fg", "bg"):
            assert read_img(str(dst / f"{mode}_{t:06d}.jpg")).shape == (H, W, 3)
