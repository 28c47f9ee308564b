"""CPU tests: the oracle against the golden fixtures generated from the reference's own code
(tests/golden/gen_golden.py) and against independent restatements (scipy / colorsys) of the primitives."""
import colorsys
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_target_size_matches_reference_table(oracle_lib):
    table = np.load(os.path.join(GOLD, "target_size.npz"))["table"]
    for h, w, L, d, th, tw in table:
        assert oracle_lib.get_target_size(h, w, L, d) == (th, tw), (h, w, L, d)


def test_score_map_matches_reference(oracle_lib):
    z = np.load(os.path.join(GOLD, "score_map.npz"))
    for key in z.files:
        if not key.startswith("score_") or key.startswith("score_c_"):
            continue
        h, w = map(int, key[len("score_"):].split("x"))
        center = [0.5, 0.5] if w > h else [0.6, 0.5]
        got = oracle_lib.score_map(h, w, center)
        assert np.array_equal(got, z[key]), key            # float64, bit for bit
    assert np.array_equal(oracle_lib.score_map(40, 60, (0.3, 0.8)), z["score_c_40x60"])


def test_exist_foreground_matches_reference(oracle_lib):
    z = np.load(os.path.join(GOLD, "score_map.npz"))
    for m, row in zip(z["ef_masks"], z["ef_expected"]):
        for t, e in zip(z["ef_thr"], row):
            assert oracle_lib.exist_foreground(m, float(t)) == bool(e)


def test_morphology_against_scipy(oracle_lib):
    rng = np.random.default_rng(3)
    m = (ndi.gaussian_filter(rng.random((70, 90)), 2) > 0.5).astype(np.uint8) * 255
    cross = ndi.generate_binary_structure(2, 1)
    for it in (1, 2, 5):
        d = ndi.binary_dilation(m > 0, cross, iterations=it)
        e = ndi.binary_erosion(m > 0, cross, iterations=it, border_value=1)
        assert np.array_equal(oracle_lib.morph(m, 3, it, "dilate") > 0, d)
        assert np.array_equal(oracle_lib.morph(m, 3, it, "erode") > 0, e)
    # 4x4 ellipse: rows 0010 / 1111 / 1111 / 1111 around anchor (2, 2); grey-level max
    g = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    fp = np.zeros((5, 5), bool)       # scipy needs an odd, centred footprint: embed the 4x4 element
    se = np.array([[0, 0, 1, 0], [1, 1, 1, 1], [1, 1, 1, 1], [1, 1, 1, 1]], bool)
    fp[0:4, 0:4] = se                 # anchor (2,2) of the 4x4 == centre (2,2) of the 5x5
    # cv2.dilate does NOT reflect the element: dst(p) = max src(p + k - anchor) == scipy's maximum_filter
    ref = ndi.maximum_filter(g, footprint=fp, mode="constant", cval=0)
    assert np.array_equal(oracle_lib.morph(g, 4, 1, "dilate"), ref)
    ref2 = ndi.maximum_filter(ref, footprint=fp, mode="constant", cval=0)
    assert np.array_equal(oracle_lib.morph(g, 4, 2, "dilate"), ref2)


def test_hsv_against_colorsys(oracle_lib):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    img[0, 0] = (0, 0, 0)
    img[0, 1] = (255, 255, 255)
    img[0, 2] = (10, 10, 10)
    hsv = oracle_lib.bgr2hsv(img)
    for y in range(32):
        for x in range(32):
            b, g, r = (int(v) for v in img[y, x])
            h, s, v = colorsys.rgb_to_hsv(r / 255., g / 255., b / 255.)
            H, S, V = (int(c) for c in hsv[y, x])
            assert V == max(b, g, r)
            assert abs(S - s * 255) <= 1
            dh = abs(H - h * 180) % 180
            assert min(dh, 180 - dh) <= 1.01, (b, g, r, H, h * 180)
    back = oracle_lib.hsv2bgr(hsv)
    assert np.abs(back.astype(int) - img.astype(int)).max() <= 6      # 8-bit HSV is lossy, not wrong


def test_gray_identity_on_equal_channels(oracle_lib):
    v = np.arange(256, dtype=np.uint8)
    img = np.stack([v, v, v], -1)[None]
    assert np.array_equal(oracle_lib.bgr2gray(img)[0], v)


def test_resize_properties(oracle_lib):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    assert np.array_equal(oracle_lib.resize_linear(img, 40, 64), img)
    half = oracle_lib.resize_linear(img, 20, 32)            # exact 2x: box average, round half up
    box = (img[0::2, 0::2].astype(int) + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(half, box)
    const = np.full((13, 17), 200, np.uint8)
    for dh, dw in ((26, 34), (5, 9), (40, 11)):
        assert np.all(oracle_lib.resize_linear(const, dh, dw) == 200)
        assert np.all(oracle_lib.resize_nearest(const, dh, dw) == 200)
    up = oracle_lib.resize_nearest(img, 80, 128)
    assert np.array_equal(up[::2, ::2], img) and np.array_equal(up[1::2, 1::2], img)
    f = rng.random((12, 20)).astype(np.float32)
    fu = oracle_lib.resize_linear(f, 24, 40)
    assert fu.min() >= f.min() - 1e-6 and fu.max() <= f.max() + 1e-6
    # 2x float upsample: interior samples are the 0.75/0.25 blends
    exp = (f[3, 5] * np.float32(0.75) + f[3, 6] * np.float32(0.25)) * np.float32(0.75) + \
          (f[4, 5] * np.float32(0.75) + f[4, 6] * np.float32(0.25)) * np.float32(0.25)
    assert fu[7, 11] == np.float32(exp)


def test_trimap_structure(oracle_lib):
    yy, xx = np.mgrid[0:120, 0:200]
    m = ((((yy - 60) / 40.) ** 2 + ((xx - 100) / 50.) ** 2) < 1).astype(np.uint8) * 255
    tri = oracle_lib.trimap(m, long_side=100, ksize=3, iters=5)
    assert set(np.unique(tri)) == {0, 128, 255}
    assert np.all(tri[m == 0][ndi.binary_erosion(m == 0, iterations=14)[m == 0]] == 0)
    assert tri[60, 100] == 255 and tri[0, 0] == 0
    band = tri == 128
    assert band.sum() > 0 and np.all(ndi.binary_dilation(m > 0, iterations=16)[band])
    assert np.all(oracle_lib.trimap(np.zeros((50, 60), np.uint8), 60) == 0)
    assert np.all(oracle_lib.trimap(np.full((50, 60), 255, np.uint8), 60) == 255)


def test_contours_count_and_fill_against_scipy(oracle_lib, cfg):
    """Independent check of the border follower: #contours = #8-connected objects + #4-connected holes, and
    objects kept/removed as whole filled regions."""
    rng = np.random.default_rng(11)
    for trial in range(12):
        H, W = [(60, 80), (90, 50), (128, 128)][trial % 3]
        x = ndi.gaussian_filter(rng.random((H, W)), [1, 2, 3][trial % 3])
        m = (x > np.quantile(x, [0.4, 0.6, 0.8][trial % 3])).astype(np.uint8) * 255
        _, stats = oracle_lib.remove_invalid_objects(cfg, m.copy(), return_stats=True)
        _, n_obj = ndi.label(m > 0, structure=np.ones((3, 3)))
        lab_bg, n_bg = ndi.label(np.pad(m == 0, 1, constant_values=True))       # 4-connected background
        n_holes = n_bg - 1                                                      # minus the outside
        assert len(stats) == n_obj + n_holes


def test_remove_invalid_objects_semantics(oracle_lib, cfg):
    H, W = 200, 300
    a = np.zeros((H, W), np.uint8)
    a[80:130, 120:180] = 255          # big central object: kept
    a[5:9, 5:9] = 255                 # polygon area 9 < 100: dropped
    a[100:110, 140:150] = 0           # hole inside the kept object: stays a hole
    out = oracle_lib.remove_invalid_objects(cfg, a.copy())
    assert out[100, 130] == 255 and out[6, 6] == 0 and out[105, 145] == 0
    assert np.array_equal(out[80:130, 120:180], a[80:130, 120:180])
    # consensus: object absent from the segmentation mask and of modest saliency is dropped
    seg = np.zeros_like(a)
    b = np.zeros((H, W), np.uint8)
    b[10:22, 10:22] = 255             # area 121, near the corner: saliency between thr and 10*thr?
    out2, st = oracle_lib.remove_invalid_objects(cfg, b.copy(), seg, return_stats=True)
    area, sal, cons, kept = st[0]
    assert area == 121 and cons == 0
    assert kept == float(sal > 1e-4)
    # in-place contract (reference maskprocess.py:151-152)
    c = a.copy()
    r = oracle_lib.remove_invalid_objects(cfg, c)
    assert r is c


def test_unet_restatement_matches_reference_golden():
    """The folded conv plan (product module, run on CPU as the network's oracle) against the reference UNet."""
    import sys
    import torch
    sys.path.insert(0, GOLD)
    from gen_golden import unet_input
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    z = np.load(os.path.join(GOLD, "unet_64x96.npz"))
    net = MattingNet(synthetic_state_dict(0), device="cpu")
    y = net(unet_input(64, 96, int(z["seed"]))).numpy()
    assert np.abs(y - z["y"]).max() < 1e-5
    y2 = net(unet_input(96, 160, int(z["seed2"]))).numpy()[:, :, ::4, ::4]
    assert np.abs(y2 - z["y2_sub"]).max() < 1e-5


def test_schp_plan_matches_reference_golden():
    """Folded SCHP plan (BN folded, dead heads dropped) on torch-CPU against the reference network's own output."""
    import zlib
    import torch
    from video_unscreen_amd.binseg.schp_net import SchpNetTorch, schp_checkpoint_layout, schp_synthetic_state_dict
    z = np.load(os.path.join(GOLD, "schp_97x129.npz"))
    lay = schp_checkpoint_layout()
    assert len(lay) == int(z["n_keys"])
    assert sum(zlib.crc32((k + str(tuple(sh))).encode()) for k, sh in lay) == int(z["keys_crc"])    # reference key / shape list
    g = torch.Generator().manual_seed(int(z["seed"]))
    x = torch.randn(1, 3, 97, 129, generator=g)
    y = SchpNetTorch(schp_synthetic_state_dict(0))(x).numpy()
    assert np.abs(y - z["y"]).max() < 2e-5 * np.abs(z["y"]).max()
    # a DataParallel-style checkpoint (module.* keys, agent_human.py:141-148) loads the same way
    sd = {"module." + k: v for k, v in schp_synthetic_state_dict(0).items()}
    y2 = SchpNetTorch(sd)(x).numpy()
    assert np.array_equal(y, y2)


def test_deeplab_plan_and_segagent_geometry_match_reference_golden(oracle_lib):
    """Folded DeepLabv3+/R50 plan on torch-CPU against the reference model's own output (incl. its final upsample), and
    SegAgent.get_target_size / grid_crop locations against the reference's."""
    import zlib
    import torch
    from video_unscreen_amd.binseg.deeplab_net import DeepLabNetTorch, deeplab_checkpoint_layout, deeplab_synthetic_state_dict
    z = np.load(os.path.join(GOLD, "deeplab_97x129.npz"))
    lay = deeplab_checkpoint_layout()
    assert len(lay) == int(z["n_keys"])
    assert sum(zlib.crc32((k + str(tuple(sh))).encode()) for k, sh in lay) == int(z["keys_crc"])
    g = torch.Generator().manual_seed(int(z["seed"]))
    x = torch.randn(1, 3, 97, 129, generator=g)
    net = DeepLabNetTorch(deeplab_synthetic_state_dict(0))
    y = net.full(x).numpy()
    assert np.abs(y - z["y"]).max() < 2e-5 * np.abs(z["y"]).max()
    # the oracle's align_corners=False upsample is the one the model ends with
    low = net(x).numpy()[0].transpose(1, 2, 0)
    up = oracle_lib.upsample_nc(low, 97, 129)
    assert np.abs(up.transpose(2, 0, 1) - z["y"][0]).max() < 2e-5 * np.abs(z["y"]).max()
    for h, w, L, th, tw in z["target_sizes"]:
        assert oracle_lib.segagent_target_size(h, w, L) == (th, tw)
        assert np.array_equal(oracle_lib.grid_crop_locations(th, tw), z[f"loc_{h}x{w}_{L}"])


def test_stm_plan_matches_reference_golden(oracle_lib):
    """Folded STM plan on torch-CPU against the reference model's own memorize() / segment() outputs."""
    import zlib
    import torch
    from video_unscreen_amd.stm.net import StmNetTorch, stm_checkpoint_layout, stm_synthetic_state_dict
    z = np.load(os.path.join(GOLD, "stm_64x96.npz"))
    lay = stm_checkpoint_layout()
    assert len(lay) == int(z["n_keys"])
    assert sum(zlib.crc32((k + str(tuple(sh))).encode()) for k, sh in lay) == int(z["keys_crc"])
    g = torch.Generator().manual_seed(int(z["seed"]))
    f0, f1 = torch.randn(1, 3, 64, 96, generator=g), torch.randn(1, 3, 64, 96, generator=g)
    m = (torch.rand(1, 1, 64, 96, generator=g) > 0.5).float()
    net = StmNetTorch(stm_synthetic_state_dict(0))
    k4, v4 = net.memorize(f0, torch.cat([1 - m, m], 1))
    assert np.abs(k4.numpy() - z["k4"]).max() < 2e-5 * np.abs(z["k4"]).max()
    assert np.abs(v4.numpy() - z["v4"]).max() < 2e-5 * np.abs(z["v4"]).max()
    keys, values = k4.unsqueeze(2), v4.unsqueeze(2)
    l1 = net.segment(f1, keys, values).numpy()
    assert np.abs(l1 - z["logit1"]).max() < 2e-3                   # log-odds of a float32 probability: 1e-4 relative
    keys2, values2 = torch.cat([keys, keys * 0.5 + 0.1], 2), torch.cat([values, values.flip(3)], 2)
    l2 = net.segment(f1, keys2, values2).numpy()
    assert np.abs(l2 - z["logit2"]).max() < 2e-3
    assert ((l1[:, 1] > l1[:, 0]) != (z["logit1"][:, 1] > z["logit1"][:, 0])).mean() < 1e-3
    # the oracle's post-processing reproduces segment() + the agent's softmax from the decoder logits
    p2 = net.decoder_logits(f1, keys, values)[0].permute(1, 2, 0).numpy()
    mask, prob = oracle_lib.stm_post(p2, 64, 96, 64, 96, want_prob=True)
    ref = torch.softmax(torch.from_numpy(z["logit1"]), 1)[0].permute(1, 2, 0).numpy()
    assert np.abs(prob - ref).max() < 1e-4
    assert (mask != (ref[..., 1] > ref[..., 0]) * 255).mean() < 1e-3


def test_humanseg_affine_and_warps(oracle_lib):
    """Whole-frame affine (agent_human.py:91-125, 247-260): uniform scale about the centre, letterboxed."""
    for (w, h) in ((1920, 1080), (1080, 1920), (256, 256), (47, 33)):
        M = oracle_lib.humanseg_affine(w, h)
        a = 472.0 / (max(w, h) - 1)
        assert abs(M[0, 0] - a) < 1e-12 and abs(M[1, 1] - a) < 1e-12 and abs(M[0, 1]) < 1e-12 and abs(M[1, 0]) < 1e-12
        assert abs(M[0, 2] - (236 - a * (w - 1) / 2)) < 1e-9 and abs(M[1, 2] - (236 - a * (h - 1) / 2)) < 1e-9
        Mi = oracle_lib.humanseg_affine(w, h, inv=True)
        full = np.vstack([M, [0, 0, 1]]) @ np.vstack([Mi, [0, 0, 1]])
        assert np.abs(full - np.eye(3)).max() < 1e-9
    # identity warp reproduces the image; a one-pixel shift moves it; outside reads 0
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    I = np.array([[1, 0, 0], [0, 1, 0]], float)
    assert np.array_equal(oracle_lib.warp_affine(img, I, 20, 30), img)
    S = np.array([[1, 0, 1], [0, 1, 0]], float)
    sh = oracle_lib.warp_affine(img, S, 20, 30)
    assert np.array_equal(sh[:, 1:], img[:, :-1]) and np.all(sh[:, 0] == 0)
    f = rng.random((20, 30)).astype(np.float32)
    assert np.array_equal(oracle_lib.warp_affine(f, I, 20, 30), f)
    half = oracle_lib.warp_affine(f, np.array([[1, 0, 0.5], [0, 1, 0]], float), 20, 30)
    assert np.allclose(half[:, 1:], 0.5 * (f[:, :-1] + f[:, 1:]), atol=1e-6)
    # align_corners upsample hits the source samples exactly on the coarse grid
    lg = rng.normal(size=(5, 7, 9)).astype(np.float32)
    up = oracle_lib.upsample_align_corners(lg, 13, 17)
    assert np.array_equal(up[::2, ::2].transpose(2, 0, 1), lg)


def test_green_mode_oracle_pieces(oracle_lib):
    """BGR2Lab against known OpenCV 8-bit values and the CIE formula (fixed-point error <= 3 levels), the mixture tables against
    float64, the adaptive threshold and colour correction against straight NumPy restatements of the reference lines."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from green_ref import green_screen_clip
    px = np.array([[[255, 255, 255], [0, 0, 0], [0, 255, 0], [255, 0, 0], [0, 0, 255]]], np.uint8)
    assert oracle_lib.bgr2lab(px).tolist() == [[[255, 128, 128], [0, 128, 128], [224, 42, 211], [82, 207, 20], [136, 208, 195]]]
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    b, g, r = [img[..., i].astype(np.float64) / 255 for i in range(3)]
    gam = lambda x: np.where(x <= 0.04045, x / 12.92, ((x + 0.055) / 1.055) ** 2.4)  # noqa: E731
    R, G, B = gam(r), gam(g), gam(b)
    X = (0.412453 * R + 0.357580 * G + 0.180423 * B) / 0.950456
    Y = 0.212671 * R + 0.715160 * G + 0.072169 * B
    Z = (0.019334 * R + 0.119193 * G + 0.950227 * B) / 1.088754
    f = lambda t: np.where(t > 0.008856, np.cbrt(t), 7.787 * t + 16 / 116)  # noqa: E731
    ref = np.stack([np.where(Y > 0.008856, 116 * np.cbrt(Y) - 16, 903.3 * Y) * 2.55, 500 * (f(X) - f(Y)) + 128, 200 * (f(Y) - f(Z)) + 128], -1)
    assert np.abs(oracle_lib.bgr2lab(img).astype(np.float64) - ref).max() < 3.0
    # mixture table
    m, s, w = np.array([20., 90., 200.]), np.array([3., 12., 30.]), np.array([0.2, 0.5, 0.3])
    x = np.arange(256)[None, :]
    ref = (w[:, None] / (s[:, None] * np.sqrt(2 * np.pi)) * np.exp(-0.5 * ((x - m[:, None]) / s[:, None]) ** 2)).sum(0)
    assert np.abs(oracle_lib.gmm_lut(m, s, w) - ref).max() < 1e-7
    # likelihood ratio: NumPy float32 restatement of agent.py:232-257
    hsv = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    lb = np.stack([oracle_lib.gmm_lut(m + 5 * i, s, w) for i in range(3)])
    lf = np.stack([oracle_lib.gmm_lut(m[::-1] - 3 * i, s, w) for i in range(3)])
    bgp = np.ones((40, 50), np.float32)
    fgp = np.ones((40, 50), np.float32)
    for i in range(3):
        bgp *= lb[i][hsv[..., i]]
        fgp *= lf[i][hsv[..., i]]
    bgp, fgp = np.cbrt(bgp.astype(np.float64)).astype(np.float32), np.cbrt(fgp.astype(np.float64)).astype(np.float32)
    prob = fgp / (bgp + fgp + np.float32(1e-6))
    alpha, p = oracle_lib.gmm_alpha(hsv, lb, lf, want_prob=True)
    assert np.array_equal(p, prob) and np.array_equal(alpha, np.clip(prob * 255, 0, 255).astype(np.uint8))
    # adaptive threshold (agent.py:275-278)
    mask = (rng.random((40, 50)) > 0.4).astype(np.uint8) * 255
    a = rng.integers(0, 256, (40, 50), dtype=np.uint8)
    want = a.copy()
    thr = a.astype(np.float64)[(a > 128) & (mask > 0)].mean() * 0.8
    want[want < thr] = 0
    want = oracle_lib.morph(oracle_lib.morph(want, 3, 2, "dilate"), 3, 2, "erode")
    want = oracle_lib.morph(oracle_lib.morph(want, 3, 2, "erode"), 3, 2, "dilate")
    assert np.array_equal(oracle_lib.cf_postprocess(a, mask), want)
    # colour correction (imgprocess.py:263-300) in NumPy float32
    frame, coarse = green_screen_clip(120, 200, 3)
    al = np.where(coarse > 0, 180, 0).astype(np.uint8)
    th, tw = oracle_lib.get_target_size(120, 200, 960)
    small = oracle_lib.bgr2lab(oracle_lib.resize_linear(frame, th, tw)).astype(np.float32) / np.float32(255)
    bgl = oracle_lib.bgr2lab(frame[0:1, 0:1]).astype(np.float32) / np.float32(255)
    d = np.sqrt(((small - bgl)[:, :, 1:] ** 2).sum(-1, dtype=np.float32))
    d = (d - d.min()) / (d.max() - d.min())
    ar = oracle_lib.resize_linear(al, th, tw).astype(np.float32)
    n = 0
    while d[(ar > 0) & (d > 0)].mean(dtype=np.float64) < np.float32(0.95):
        d = np.sqrt(d)
        n += 1
    d[ar == 0] = 0
    sy = np.minimum(np.floor(np.arange(120, dtype=np.float32) * np.float32(th / 120)).astype(int), th - 1)
    sx = np.minimum(np.floor(np.arange(200, dtype=np.float32) * np.float32(tw / 200)).astype(int), tw - 1)
    want = (al.astype(np.float32) * d[sy][:, sx]).astype(np.uint8)
    got, it = oracle_lib.color_correct(frame, al, frame[0, 0], want_iters=True)
    assert it == n and np.array_equal(got, want)
    # box filter: constants preserved, 3x3 mean of an impulse = round(255 / 9)
    imp = np.zeros((9, 9), np.uint8)
    imp[4, 4] = 255
    out = oracle_lib.box_filter(imp, 3)
    assert out[4, 4] == 28 and out[3:6, 3:6].min() == 28 and out.sum() == 28 * 9
    assert np.array_equal(oracle_lib.box_filter(np.full((5, 7, 3), 9, np.uint8), 5), np.full((5, 7, 3), 9, np.uint8))
