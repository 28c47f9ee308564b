"""GPU parity tests (-m gpu): every HIP stage, called through the C ABI (video_unscreen_amd._native -> ctypes ->
libunscreen_hip.so), against the CPU oracle on the same seeded inputs.  Integer / byte stages must be bit-exact;
the float alpha must be within 1e-3 (BASELINE.json north_star) - tolerances are stated at each assert."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch

pytestmark = pytest.mark.gpu

ALPHA_TOL = 1e-3          # north_star: alpha / foreground within 1e-3 L-inf of the reference CPU path


@pytest.fixture(scope="module")
def nv():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from video_unscreen_amd import _native
    _native.load()                     # raises if the HIP extension is missing: no fallback
    return _native


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def blob_mask(H, W, seed, q=0.6, sigma=None):
    rng = np.random.default_rng(seed)
    sigma = sigma if sigma is not None else max(2, min(H, W) // 40)
    x = ndi.gaussian_filter(rng.random((H, W)), sigma)
    return (x > np.quantile(x, q)).astype(np.uint8) * 255


def rand_img(H, W, seed, smooth=True):
    rng = np.random.default_rng(seed)
    if not smooth:
        return rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    base = ndi.gaussian_filter(rng.random((H, W, 3)) * 255, (3, 3, 0))
    img = (base - base.min()) / (base.max() - base.min() + 1e-9) * 255 + rng.normal(0, 2, (H, W, 3))
    return np.clip(np.round(img), 0, 255).astype(np.uint8)


SIZES = [(1080, 1920), (2160, 3840), (256, 256), (270, 480), (135, 241), (64, 96), (1920, 1080), (33, 47), (1, 7), (5, 1)]


@pytest.mark.parametrize("H,W", SIZES)
def test_trimap_bit_exact(nv, oracle_lib, H, W):
    for seed, (L, k, it) in enumerate([(960, 3, 5), (960, 3, 5), (64, 4, 2), (100, 5, 3)]):
        if seed >= 2 and H * W > 600000:
            continue
        m = blob_mask(H, W, 10 + seed)
        if seed == 1:                                   # grey masks: thresholds 127 / 128 matter
            m = np.random.default_rng(seed).integers(0, 256, (H, W), dtype=np.uint8)
        want = oracle_lib.trimap(m, L, k, it)
        got = host(nv.trimap(dev(m), L, k, it))
        assert np.array_equal(got, want), f"{(H, W, L, k, it)}: {(got != want).sum()} pixels differ"   # bit-exact
    for m in (np.zeros((H, W), np.uint8), np.full((H, W), 255, np.uint8)):
        assert np.array_equal(host(nv.trimap(dev(m))), oracle_lib.trimap(m))


@pytest.mark.parametrize("H,W", [(1080, 1920), (270, 480), (65, 130), (33, 47), (3, 200)])
@pytest.mark.parametrize("k,it", [(3, 5), (4, 2), (5, 10), (3, 2), (7, 3), (1, 3), (3, 0)])
def test_morph_bit_exact(nv, oracle_lib, H, W, k, it):
    if H * W > 600000 and (k, it) not in ((3, 5), (4, 2)):
        pytest.skip("large size covered for the reference's own parameter pairs")
    g = np.random.default_rng(k * 100 + it).integers(0, 256, (H, W), dtype=np.uint8)
    for op, name in ((nv.DILATE, "dilate"), (nv.ERODE, "erode")):
        assert np.array_equal(host(nv.morph(dev(g), k, it, op)), oracle_lib.morph(g, k, it, name))   # bit-exact


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840), (270, 480), (135, 241), (33, 47), (2, 3), (70, 501), (40, 249), (17, 248), (50, 1000)])
def test_bgdiff_mask_bit_exact(nv, oracle_lib, H, W):
    frame = rand_img(H, W, 1)
    bg_always = rand_img(H, W, 2)
    bg_i = np.clip(bg_always.astype(int) + np.random.default_rng(3).integers(-3, 4, bg_always.shape), 0, 255).astype(np.uint8)
    fgm = blob_mask(H, W, 4)
    frame = np.where(fgm[..., None] > 0, frame, np.clip(bg_always.astype(int) + np.random.default_rng(5).integers(-20, 21, frame.shape), 0, 255)).astype(np.uint8)
    seg1 = blob_mask(H, W, 6)
    seg3 = np.stack([seg1, np.roll(seg1, 1, 1), np.roll(seg1, 1, 0)], -1)
    for seg in (seg3, seg1):
        for beta, thr in ((0.1, 25), (0.37, 3)):
            wb, wa = oracle_lib.bgdiff_mask(frame, bg_i, bg_always, seg, beta, thr)
            gb, ga = nv.bgdiff_mask(dev(frame), dev(bg_i), dev(bg_always), dev(seg), beta, thr)
            assert np.array_equal(host(gb), wb)            # fused background: bit-exact (f32 blend, truncation)
            assert np.array_equal(host(ga), wa)            # gated alpha: bit-exact
    wb, wa = oracle_lib.bgdiff_mask(frame, bg_i, None, seg1, 1.0, 25)          # bg.py:85-92: one background image, no fusion
    gb, ga = nv.bgdiff_mask(dev(frame), dev(bg_i), None, dev(seg1), 1.0, 25)
    assert np.array_equal(host(gb), wb) and np.array_equal(host(ga), wa)


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 960), (2160, 3840, 960), (256, 256, 960), (270, 480, 960),
                                   (135, 241, 64), (64, 128, 64), (1920, 1080, 960), (50, 70, 70), (33, 47, 96)])
def test_vmat_pre_bit_exact(nv, oracle_lib, H, W, L):
    frame = rand_img(H, W, 11, smooth=False)
    ap = np.random.default_rng(12).integers(0, 256, (H, W), dtype=np.uint8)
    tri = np.random.default_rng(13).choice(np.array([0, 128, 255], np.uint8), (H, W))
    tri = ndi.median_filter(tri, 5)
    in_h, in_w = oracle_lib.vmat_input_size(H, W, L)
    want_nchw = oracle_lib.vmat_pre(frame, ap, tri, in_h, in_w, "nchw")
    got_nhwc = nv.vmat_pre(dev(frame), dev(ap), dev(tri), in_h, in_w, layout=nv.NHWC)
    assert got_nhwc.shape == (1, 7, in_h, in_w)
    assert np.array_equal(host(got_nhwc)[0], want_nchw)    # float32, bit-exact (same op order, no contraction)
    got_nchw = nv.vmat_pre(dev(frame), dev(ap), dev(tri), in_h, in_w, layout=nv.NCHW)
    assert np.array_equal(host(got_nchw)[0], want_nchw)


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 960), (2160, 3840, 960), (256, 256, 960), (135, 241, 64), (64, 128, 64),
                                   (1920, 1080, 960), (33, 47, 96), (544, 960, 960)])
def test_vmat_post_bit_exact(nv, oracle_lib, H, W, L):
    in_h, in_w = oracle_lib.vmat_input_size(H, W, L)
    rng = np.random.default_rng(21)
    pred = ndi.gaussian_filter(rng.random((in_h, in_w)), 2).astype(np.float32)
    pred = ((pred - pred.min()) / (pred.max() - pred.min())).astype(np.float32)
    pred[0, 0] = 1.0
    tri = ndi.median_filter(rng.choice(np.array([0, 128, 255], np.uint8), (H, W)), 5)
    w8, wf = oracle_lib.vmat_post(pred, tri, want_float=True)
    g8, gf = nv.vmat_post(dev(pred), dev(tri), want_float=True)
    assert np.array_equal(host(gf), wf)                    # float alpha: bit-exact
    assert np.array_equal(host(g8), w8)                    # uint8 alpha (truncation): bit-exact
    assert np.array_equal(host(nv.vmat_post(dev(pred), dev(tri))), w8)


@pytest.mark.parametrize("H,W,L", [(1080, 1920, 960), (2160, 3840, 960), (135, 241, 64), (33, 47, 96), (544, 960, 960), (1, 3, 64)])
def test_vmat_post_fg_one_launch_bit_exact(nv, oracle_lib, H, W, L):
    """The back half's fused launch (postprocess + paste + get_fg) against the oracle's two functions, and against the two kernels."""
    in_h, in_w = oracle_lib.vmat_input_size(H, W, L)
    rng = np.random.default_rng(51)
    pred = ndi.gaussian_filter(rng.random((in_h, in_w)), 2).astype(np.float32)
    pred = ((pred - pred.min()) / (pred.max() - pred.min())).astype(np.float32)
    tri = ndi.median_filter(rng.choice(np.array([0, 128, 255], np.uint8), (H, W)), 5) if min(H, W) >= 5 else rng.choice(np.array([0, 128, 255], np.uint8), (H, W))
    frame = rand_img(H, W, 52, smooth=False)
    bg = rand_img(H, W, 53, smooth=False)
    w8, wf = oracle_lib.vmat_post(pred, tri, want_float=True)
    for paste in (1, 0, 128):
        want_fg, _ = oracle_lib.get_fg(frame, w8, bg, paste)
        g8, gf, gfg = nv.vmat_post_fg(dev(pred), dev(tri), dev(frame), dev(bg), paste_thr=paste, want_float=True)
        assert np.array_equal(host(g8), w8) and np.array_equal(host(gf), wf)          # bit-exact
        assert np.array_equal(host(gfg), want_fg), f"paste {paste}: {(host(gfg) != want_fg).sum()} bytes differ"
        a2 = nv.vmat_post(dev(pred), dev(tri))
        assert np.array_equal(host(nv.get_fg(dev(frame), a2, dev(bg), paste)), host(gfg))
    g8, gf, gfg = nv.vmat_post_fg(dev(pred), dev(tri), dev(frame), dev(bg), paste_thr=1)
    assert gf is None and np.array_equal(host(g8), w8)


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840), (270, 480), (33, 47), (1, 3), (7, 1)])
def test_get_fg_get_bg_blend_bit_exact(nv, oracle_lib, H, W):
    frame = rand_img(H, W, 31, smooth=False)
    bg = rand_img(H, W, 32, smooth=False)
    alpha = np.random.default_rng(33).integers(0, 256, (H, W), dtype=np.uint8)
    alpha[np.random.default_rng(34).random((H, W)) < 0.3] = 0
    alpha[np.random.default_rng(35).random((H, W)) < 0.2] = 255
    for paste in (0, 1, 128):
        wf, wb = oracle_lib.get_fg(frame, alpha, bg, paste)
        gf, gb = nv.get_fg(dev(frame), dev(alpha), dev(bg), paste, want_bg=True)
        assert np.array_equal(host(gf), wf), f"paste {paste}: {(host(gf) != wf).sum()} bytes differ"   # bit-exact
        assert np.array_equal(host(gb), wb)
    assert np.array_equal(host(nv.get_bg(dev(alpha), dev(bg))), oracle_lib.get_bg(alpha, bg))           # bit-exact
    m3 = np.stack([alpha, np.roll(alpha, 1, 0), np.roll(alpha, 1, 1)], -1)
    for m in (alpha, m3):
        assert np.array_equal(host(nv.blend(dev(frame), dev(m), dev(bg))), oracle_lib.blend(frame, m, bg))   # bit-exact


def test_get_fg_exhaustive_hsv_colours(nv, oracle_lib):
    """All 2^24 BGR colours once as frame (alpha 255 -> pure HSV round trip) and once as background."""
    v = np.arange(256, dtype=np.uint8)
    for chunk in range(0, 256, 64):
        b, g, r = np.meshgrid(v[chunk:chunk + 64], v, v, indexing="ij")
        img = np.stack([b, g, r], -1).reshape(64 * 256, 256, 3)
        other = np.roll(img, 17, axis=1)[::-1].copy()
        alpha = ((np.arange(img.shape[0])[:, None] * 7 + np.arange(256)[None, :] * 13) % 256).astype(np.uint8)
        for (f, bgd) in ((img, other), (other, img)):
            wf, _ = oracle_lib.get_fg(f, alpha, bgd, 0)
            assert np.array_equal(host(nv.get_fg(dev(f), dev(alpha), dev(bgd), 0)), wf)       # bit-exact


@pytest.mark.parametrize("n", [1, 15, 16, 17, 1000, 2073600, 8294400])
def test_count_ge_exact(nv, oracle_lib, n):
    m = np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8)
    for thr in (128, 1, 255):
        assert int(nv.count_ge(dev(m), thr).item()) == oracle_lib.count_ge(m, thr)
    assert int(nv.count_ge(dev(np.zeros(n, np.uint8)), 128).item()) == 0


def test_agents_numpy_api(nv, oracle_lib, cfg):
    """The reference-shaped NumPy API (drop-in agents / utils) gives the oracle's answers."""
    from video_unscreen_amd.trimap import TrimapAgent
    from video_unscreen_amd.utils import dilate_mask, erode_mask, exist_foreground, get_bg, get_fg
    H, W = 120, 200
    m = blob_mask(H, W, 41)
    agent = TrimapAgent(**cfg["trimap"])
    assert np.array_equal(agent.forward(m.copy()), oracle_lib.trimap(m, 960, 3, 5))
    assert np.array_equal(dilate_mask(m, 4, 2), oracle_lib.morph(m, 4, 2, "dilate"))
    assert np.array_equal(erode_mask(m, 3, 2), oracle_lib.morph(m, 3, 2, "erode"))
    assert exist_foreground(m, 0.001) == oracle_lib.exist_foreground(m, 0.001)
    assert exist_foreground(np.zeros_like(m), 0.001) is False
    frame, bg = rand_img(H, W, 42), rand_img(H, W, 43)
    assert np.array_equal(get_fg(frame, m, bg), oracle_lib.get_fg(frame, m, bg, 0)[0])
    assert np.array_equal(get_bg(m, bg), oracle_lib.get_bg(m, bg))


@pytest.mark.parametrize("H,W", [(270, 480), (1080, 1920), (2160, 3840)])      # 3840 x 2160: BASELINE.json configs[3] as a whole path
def test_matting_agent_and_pipeline_vs_oracle(nv, oracle_lib, cfg, H, W):
    """Whole frames: the device-resident loop against the oracle loop (bg_offline.py:145-176).  Index masks
    bit-exact; float alpha within ALPHA_TOL of the CPU path; uint8 alpha within 1 LSB (a truncation can flip)."""
    import __graft_entry__ as ge
    from video_unscreen_amd.pipeline import BgStepPipeline
    from video_unscreen_amd.synth import SyntheticClip
    nframes = 3 if H < 1000 else 2
    clip = SyntheticClip(H, W, nframes, config_id=2, device="cpu")
    pipe = BgStepPipeline(cfg, H, W, device="cuda:0")
    bg_always = clip.bg_always
    alpha_pre = None
    for t in range(nframes):
        f = clip.frame(t)
        out = pipe.step(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bg_always.cuda(), want_float=True)
        want = ge.oracle_frame(oracle_lib, cfg, pipe, f["frame"].numpy(), f["seg"].numpy(), f["bg"].numpy(),
                               bg_always.numpy(), alpha_pre)
        assert np.array_equal(host(out["bg"]), want["bg"])                      # bit-exact
        assert np.array_equal(host(out["gate"]), want["alpha_gate"])            # bit-exact
        assert np.array_equal(host(out["alphaor"]), want["alphaor"])            # bit-exact
        assert np.array_equal(host(out["trimap"]), want["trimap"])              # bit-exact
        dpred = np.abs(host(out["pred"])[0, 0] - want["pred"]).max()
        assert dpred < ALPHA_TOL, f"network output differs from the CPU path by {dpred}"
        da = np.abs(host(out["alpha_f32"]) - want["alpha_f32"]).max()
        assert da < ALPHA_TOL, f"float alpha differs by {da}"
        d8 = np.abs(host(out["alpha"]).astype(int) - want["alpha"].astype(int)).max()
        assert d8 <= 1
        # foreground: given the SAME alpha the HIP composite is bit-exact
        fg_same, _ = oracle_lib.get_fg(f["frame"].numpy(), host(out["alpha"]), want["bg"], paste_thr=1)
        assert np.array_equal(host(out["fg"]), fg_same)
        alpha_pre = host(out["alpha"])          # keep both chains on the same previous alpha
        pipe.alpha_pre = out["alpha"]


def test_pipelined_run_equals_step(nv, cfg):
    """run() (two streams + host worker) must produce exactly what the synchronous step() produces."""
    from video_unscreen_amd.pipeline import BgStepPipeline
    from video_unscreen_amd.synth import SyntheticClip
    H, W, n = 270, 480, 6
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    frames = [clip.frame(t) for t in range(n)]
    bga = clip.bg_always.cuda()
    ins = [(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga) for f in frames]
    p1 = BgStepPipeline(cfg, H, W)
    ref = []
    for i in ins:
        o = p1.step(*i)
        ref.append((o["alpha"].clone(), o["fg"].clone(), o["trimap"].clone()))
    p2 = BgStepPipeline(cfg, H, W)
    got = []
    p2.run(iter(ins), on_result=lambda k, o: got.append((o["alpha"].clone(), o["fg"].clone(), o["trimap"].clone())))
    torch.cuda.synchronize()
    assert len(got) == n
    for (a, f, t), (a2, f2, t2) in zip(ref, got):
        # every kernel of the path has a fixed summation order (no atomics in the convolutions): the two loop forms agree exactly
        assert torch.equal(t, t2) and torch.equal(a, a2) and torch.equal(f, f2)


def test_pipelined_run_with_batched_segmentation_is_identical(nv, cfg):
    """run() with the human-parsing net segmenting 3 frames per forward (7 frames: groups of 3, 3, 1) gives exactly the
    frames of the per-frame loop step(): masks, trimaps, alphas and foregrounds (the HIP convs have a fixed summation order and
    the batched layers keep the K partition of the pipeline's plan, whatever the batch)."""
    from video_unscreen_amd.binseg import HumanSegAgent
    from video_unscreen_amd.pipeline import BgStepPipeline
    from video_unscreen_amd.synth import SyntheticClip
    H, W, n = 270, 480, 7
    clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
    frames = [clip.frame(t) for t in range(n)]
    bga = clip.bg_always.cuda()
    ins = [(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga) for f in frames]
    agent = HumanSegAgent(model_path=None)
    keys = ("seg", "trimap", "alpha", "fg")
    outs = []
    for mode in ("step", "run"):
        pipe = BgStepPipeline(cfg, H, W, seg_agent=agent, seg_batch=3)       # both plan the conv K partitions for 3 frames
        got = []
        if mode == "run":
            pipe.run(iter(ins), on_result=lambda k, o: got.append(tuple(o[key].clone() for key in keys)))
        else:
            for i in ins:
                o = pipe.step(*i)
                got.append(tuple(o[key].clone() for key in keys))
        torch.cuda.synchronize()
        assert len(got) == n
        outs.append(got)
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert 0.02 < float((outs[0][0][0] > 0).float().mean()) < 0.98          # the masks come from the network, not the inputs


@pytest.mark.parametrize("H,W", [(1080, 1920), (270, 480), (33, 47), (2, 5)])
def test_inrange_and_trimap_withbg_bit_exact(nv, oracle_lib, H, W):
    """is_pixel_inrange (fgfuncs.py:9-65) and generate_trimap_withbg (trimap/agent.py:63-101): green-mode path."""
    rng = np.random.default_rng(H + W)
    bgc = np.array([60, 200, 60], np.uint8)
    img = rand_img(H, W, 51, smooth=False)
    green = np.clip(bgc.astype(int) + rng.integers(-12, 13, (H, W, 3)), 0, 255).astype(np.uint8)
    fgm = blob_mask(H, W, 52, q=0.6)
    img = np.where(fgm[..., None] > 0, img, green)
    bgimg = np.clip(green.astype(int) + rng.integers(-3, 4, (H, W, 3)), 0, 255).astype(np.uint8)
    for win in ((10, 100, 180), (20, 20, 120), (0, 0, 0)):
        for bg in (bgc, bgimg):
            want = oracle_lib.is_pixel_inrange(img, bg, win)
            got = host(nv.is_pixel_inrange(dev(img), dev(bg) if bg.ndim == 3 else bg, win)) > 0
            assert np.array_equal(got, want)                                     # bit-exact
    # trimap with background: both branches of the fuzzy-ratio test, plus the empty mask
    mask_consistent = fgm.copy()                                               # few fuzzy pixels -> ensemble branch
    mask_inconsistent = blob_mask(H, W, 53, q=0.2)                             # covers lots of green -> trust-mask branch
    for m in (mask_consistent, mask_inconsistent, np.zeros((H, W), np.uint8)):
        for bg in (bgc, bgimg):
            want = oracle_lib.trimap_withbg(m, img, bg, (10, 100, 180), 960, 3, 5)
            got = host(nv.trimap_withbg(dev(m), dev(img), dev(bg) if bg.ndim == 3 else bg, (10, 100, 180), 960, 3, 5))
            assert np.array_equal(got, want), f"{(got != want).sum()} pixels differ"   # bit-exact
    from video_unscreen_amd.trimap import TrimapAgent
    from video_unscreen_amd.utils import is_pixel_inrange
    agent = TrimapAgent()
    assert np.array_equal(agent.forward(mask_consistent.copy(), img, bgc), oracle_lib.trimap_withbg(mask_consistent, img, bgc))
    empty = np.zeros((H, W), np.uint8)
    assert agent.forward(empty, img, bgc) is empty                              # reference returns the input object
    assert np.array_equal(is_pixel_inrange(img, bgc, (10, 100, 180)), oracle_lib.is_pixel_inrange(img, bgc, (10, 100, 180)))


def test_vmatting_agent_numpy_api(nv, oracle_lib, cfg):
    """VMattingAgent.forward (NumPy in / out, reference vmatting/agent.py:134-163) against the oracle chain."""
    import __graft_entry__ as ge  # noqa: F401
    from video_unscreen_amd.vmatting import VMattingAgent
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    H, W = 270, 480
    rng = np.random.default_rng(5)
    frame = rand_img(H, W, 61)
    alpha_pre = rng.integers(0, 256, (H, W), dtype=np.uint8)
    tri = oracle_lib.trimap(blob_mask(H, W, 62), 960, 3, 5)
    agent = VMattingAgent("synthetic", input_long_side=960)
    got = agent.forward(frame.copy(), alpha_pre.copy(), tri.copy())
    in_h, in_w = oracle_lib.vmat_input_size(H, W, 960)
    x7 = oracle_lib.vmat_pre(frame, alpha_pre, tri, in_h, in_w)
    pred = MattingNet(synthetic_state_dict(0), device="cpu")(torch.from_numpy(x7)[None]).numpy()[0, 0]
    want = oracle_lib.vmat_post(pred, tri)
    assert got.shape == (H, W) and got.dtype == np.uint8
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1          # 1 LSB: a *255 truncation may flip
    assert np.array_equal(got[tri == 0], want[tri == 0]) and np.array_equal(got[tri == 255], want[tri == 255])


@pytest.mark.parametrize("H,W", [(1080, 1920), (135, 241), (33, 47)])
def test_replace_geometry_and_composites_bit_exact(nv, oracle_lib, H, W):
    """Replace mode (tools/replace/replace.py:59-82): resize, shift_fg, rescale_fg, the float64 blend, composite_fgbg,
    fuse_fgbg - all uint8 / fixed point or float64: bit-exact."""
    img = rand_img(H, W, 71, smooth=False)
    g = np.ascontiguousarray(img[..., 1])
    for (dh, dw) in ((H // 2, W // 2), (H * 2, W + 7), (max(H // 3, 1), max(W // 3, 1)), (H, W)):
        if dh * dw > 3_000_000:
            continue
        assert np.array_equal(host(nv.resize(dev(img), dh, dw, 1)), oracle_lib.resize_linear(img, dh, dw))
        assert np.array_equal(host(nv.resize(dev(g), dh, dw, 1)), oracle_lib.resize_linear(g, dh, dw))
        assert np.array_equal(host(nv.resize(dev(img), dh, dw, 0)), oracle_lib.resize_nearest(img, dh, dw))
        assert np.array_equal(host(nv.resize(dev(img), dh, dw, 2)), oracle_lib.resize_cubic(img, dh, dw))
    from video_unscreen_amd.utils import composite_fgbg, fuse_fgbg, rescale_fg, shift_fg
    for (dx, dy) in ((-1.725405405405406, 8.128108108108108), (-0.2072, -2.6028), (3, 0), (0, 0)):
        assert np.array_equal(shift_fg(img, dx, dy), oracle_lib.shift_fg(img, dx, dy))
        assert np.array_equal(shift_fg(g, dx, dy), oracle_lib.shift_fg(g, dx, dy))
    for sc in (1.2, 1.1, 1.0):
        assert np.array_equal(rescale_fg(img, sc), oracle_lib.rescale_fg(img, sc))
        assert np.array_equal(rescale_fg(g, sc), oracle_lib.rescale_fg(g, sc))
    alpha = np.random.default_rng(72).integers(0, 256, (H, W), dtype=np.uint8)
    alpha[::3] = 255
    bg = rand_img(max(H // 2, 2), W + 11, 73, smooth=False)
    assert np.array_equal(composite_fgbg(img, alpha, bg), oracle_lib.composite_fgbg(img, alpha, bg))
    bg2 = rand_img(H, W, 74, smooth=False)
    assert np.array_equal(fuse_fgbg(img, bg2, alpha), oracle_lib.blend(img, alpha, bg2))
    # the replace frame: shift + zoom of fg and 3-channel mask, float64 blend, truncation
    m3 = np.stack([alpha, alpha, alpha], -1)
    M = np.float32([[1, 0, -0.2072], [0, 1, -2.6028]]).astype(np.float64)
    fg_s = nv.rescale_fg(nv.warp_affine(dev(img), M, H, W), 1.2)
    m_s = nv.rescale_fg(nv.warp_affine(dev(m3), M, H, W), 1.2)
    got = host(nv.blend(fg_s, m_s, dev(bg2)))
    want = oracle_lib.blend(oracle_lib.rescale_fg(oracle_lib.shift_fg(img, -0.2072, -2.6028), 1.2),
                            oracle_lib.rescale_fg(oracle_lib.shift_fg(m3, -0.2072, -2.6028), 1.2), bg2)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("H,W", [(270, 480), (135, 241), (64, 96)])
def test_regionfill_matches_direct_solve(nv, oracle_lib, H, W):
    """Conjugate gradients on the device vs the oracle's sparse direct solve (the reference's own method, region_fill.py:63):
    equal to +-1 grey level after the uint8 truncation, and only where the exact value sits next to an integer."""
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.utils import regionfill
    f = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)
    alpha = (f["alpha_true"].numpy() * 255).astype(np.uint8)
    bg = oracle_lib.get_bg(alpha, f["frame"].numpy())
    mask = oracle_lib.morph(np.where(alpha > 128, 255, 0).astype(np.uint8), 3, 2, "dilate")
    mask[:6, :9] = 255                    # a region touching the image corner (2- and 3-neighbour rows of the matrix)
    mask[-5:, W // 2:W // 2 + 7] = 255
    got, iters = nv.regionfill(dev(bg), dev(mask), want_iters=True)
    got = host(got)
    assert 4 <= iters <= 100                             # multigrid-preconditioned: tens of iterations at any size
    exact = np.stack([oracle_lib.regionfill(bg[..., c], mask) for c in range(3)], -1)
    want = exact.astype(np.uint8)
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1
    near = np.abs(exact - np.round(exact)) < 2e-2
    assert (near | (d == 0)).all()                       # a flip needs the exact value within 0.02 of an integer
    assert np.array_equal(got[mask == 0], bg[mask == 0])
    assert np.array_equal(host(nv.regionfill(dev(bg), dev(mask))), got)          # fixed-order reductions: reproducible
    one = regionfill(bg[..., 1], mask)
    assert one.dtype == np.float64 and np.array_equal(one, got[..., 1])
    # nothing to fill: the image comes back
    z = np.zeros((H, W), np.uint8)
    assert np.array_equal(host(nv.regionfill(dev(bg), dev(z))), bg)


@pytest.mark.parametrize("H,W", [(5, 7), (8, 8), (9, 9), (33, 70), (2, 40), (40, 2), (97, 161)])
def test_regionfill_small_and_ragged_sizes(nv, oracle_lib, H, W):
    """Sizes that exercise the solver's edges: an image of at most 8 x 8 cells is ONE multigrid level handled inside the single-workgroup
    tail (the float64 residual is cast for it), 9 x 9 is two levels, 33 x 70 / 97 x 161 leave partial 32 x 32 tiles on both axes and odd
    coarse sizes, 2 x N / N x 2 have no interior cell at all (one-pixel-high images are outside the reference's domain: its formRightSide
    indexes row 1).  Against the sparse direct solve, +-1 level next to
    integers; a budget that is too small is continued (first_iter > 0) to the same bits as one long enqueue."""
    rng = np.random.default_rng(H * 131 + W)
    bg = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    mask[H // 4:max(H // 4 + 1, 3 * H // 4), W // 5:max(W // 5 + 1, 4 * W // 5)] = 255
    if H > 6 and W > 6:
        mask[0:2, 0:3] = 255                                   # touches a corner
        mask[H - 1, W // 2:] = 255                             # and runs along the bottom edge
    if mask.all():
        mask[0, 0] = 0                                         # (at least one known pixel: the system is singular otherwise)
    got, iters = nv.regionfill(dev(bg), dev(mask), want_iters=True)
    got = host(got)
    exact = np.stack([oracle_lib.regionfill(bg[..., c], mask) for c in range(3)], -1)
    d = np.abs(got.astype(int) - exact.astype(np.uint8).astype(int))
    assert d.max() <= 1, (iters, int(d.max()))
    assert ((np.abs(exact - np.round(exact)) < 2e-2) | (d == 0)).all()
    assert np.array_equal(got[mask == 0], bg[mask == 0])
    # two iterations per call, continued until converged: the same bits as the single enqueue
    out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    status = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = torch.empty(nv.load().usc_regionfill_ws_bytes(H, W), dtype=torch.uint8, device="cuda")
    bgd, md = dev(bg), dev(mask)
    it = 0
    while True:
        rc = nv.load().usc_regionfill_u8(bgd.data_ptr(), md.data_ptr(), H, W, out.data_ptr(), 1e-7, it, 2, status.data_ptr(), ws.data_ptr(), ws.numel(),
                                         torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        it += 2
        used, conv = status.cpu().tolist()
        if conv or it > 400:
            break
    assert conv and used == iters and np.array_equal(host(out), got)


def test_regionfill_full_size(nv, oracle_lib):
    """The inpainting at the real frame size (1920 x 1080, a silhouette-sized hole with a ragged edge): the status block reports
    convergence within the default budget logic, the image outside the hole is untouched, every filled value obeys the discrete
    maximum principle (between the smallest and largest pixel of the hole's rim, per channel), two runs agree bit for bit, the
    sub-sampled solution agrees with the float solution of the SAME system at half the tolerance, and one channel agrees with the
    oracle's sparse direct solve (the reference's method, region_fill.py:63) to +-1 level next to integers."""
    from video_unscreen_amd.synth import SyntheticClip
    H, W = 1080, 1920
    f = SyntheticClip(H, W, 1, config_id=2, device="cpu").frame(0)
    alpha = (f["alpha_true"].numpy() * 255).astype(np.uint8)
    rng = np.random.default_rng(9)
    ragged = alpha.copy()
    ragged[rng.random((H, W)) < 0.002] = 255                       # speckle: many one-pixel holes besides the silhouette
    bg = oracle_lib.get_bg(alpha, f["frame"].numpy())
    mask = oracle_lib.morph(np.where(ragged > 128, 255, 0).astype(np.uint8), 3, 2, "dilate")
    got, iters = nv.regionfill(dev(bg), dev(mask), want_iters=True)
    got = host(got)
    assert 4 <= iters <= 120, iters
    assert np.array_equal(got[mask == 0], bg[mask == 0])
    rim = (oracle_lib.morph(mask, 3, 1, "dilate") > 0) & (mask == 0)
    for c in range(3):
        lo, hi = int(bg[..., c][rim].min()), int(bg[..., c][rim].max())
        inside = got[..., c][mask > 0]
        assert lo <= int(inside.min()) and int(inside.max()) <= hi, (c, lo, hi, int(inside.min()), int(inside.max()))
    again, status = nv.regionfill(dev(bg), dev(mask), want_iters=True, sync=False, budget=iters + 6)
    torch.cuda.synchronize()
    assert np.array_equal(host(again), got) and status.cpu().tolist() == [iters, 1]         # one enqueue, no host round trip: same bits
    tight = host(nv.regionfill(dev(bg), dev(mask), rtol=1e-10))
    assert np.abs(tight.astype(int) - got.astype(int)).max() <= 1
    exact = oracle_lib.regionfill(bg[..., 1], mask)
    d = np.abs(got[..., 1].astype(int) - exact.astype(np.uint8).astype(int))
    assert d.max() <= 1 and ((np.abs(exact - np.round(exact)) < 2e-2) | (d == 0)).all()


@pytest.mark.parametrize("H,W", [(2160, 3840), (1083, 1925)])
def test_regionfill_at_3840x2160_and_an_odd_size_satisfies_its_equation(nv, H, W):
    """BASELINE.json's largest frame (and a size no level of which is even): the oracle's sparse direct solve takes minutes there, so the
    size-independent property instead - the truncated solution satisfies the discrete Laplace equation of region_fill.py:26-63 up to the
    truncation (five values, each cut by less than one level: |4 x - sum of neighbours| < 4 at every interior hole pixel), the image
    outside the hole is untouched, and the solve converges in the iteration count the Galerkin coarse operators give (about 20)."""
    from video_unscreen_amd.synth import SyntheticClip
    f = SyntheticClip(H, W, 1, config_id=2, device="cuda").frame(0)
    alpha = (f["alpha_true"] * 255).to(torch.uint8)
    bg = nv.get_bg(alpha, f["frame"])
    hole = nv.morph((alpha > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
    out, iters = nv.regionfill(bg, hole, want_iters=True)
    assert 4 <= iters <= 60, iters
    m = hole > 0
    assert torch.equal(out[~m], bg[~m])
    x = out.double()
    p = torch.nn.functional.pad(x.permute(2, 0, 1)[None], (1, 1, 1, 1), mode="replicate")[0].permute(1, 2, 0)
    lap = 4 * x - (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:])
    inner = m.clone()
    inner[0, :] = inner[-1, :] = False
    inner[:, 0] = inner[:, -1] = False
    assert float(lap[inner].abs().max()) < 4 and float(lap[inner].abs().mean()) < 0.5
