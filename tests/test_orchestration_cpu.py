"""CPU tests: the oracle's fused stage functions against tests/golden/orchestration.npz - outputs of the REFERENCE's own Python
(TrimapAgent, VMattingAgent incl. its UNet on the CPU, remove_invalid_objects, get_fg / get_bg, color_correct, is_pixel_inrange,
the bg_offline.py:145-176 frame loop) run against a cv2-named shim over the oracle's primitives (tests/golden/gen_orchestration.py).

Pinned here: the reference's order of operations, thresholds, dtype promotions, truncations, in-place behaviour.  NOT pinned:
OpenCV's arithmetic inside a call (both sides use the oracle's restatement of it)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "orchestration.npz")


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def gcfg(z):
    return json.loads(bytes(z["cfg_json"]).decode())


def test_generate_trimap(oracle_lib, z):
    for name in "abcd":
        L, k, it = (int(v) for v in z[f"tri_{name}_params"])
        assert np.array_equal(oracle_lib.trimap(z[f"tri_{name}_mask"], L, k, it), z[f"tri_{name}_out"]), name


def test_generate_trimap_withbg_and_inrange(oracle_lib, z):
    for name in ("lo", "hi"):
        mask, img, bg = z[f"triwb_{name}_mask"], z[f"triwb_{name}_img"], z[f"triwb_{name}_bg"]
        assert np.array_equal(oracle_lib.is_pixel_inrange(img, bg, (10, 100, 180)), z[f"triwb_{name}_inrange"] > 0), name
        got = oracle_lib.trimap_withbg(mask, img, bg, (10, 100, 180), 144, 3, 2)
        assert np.array_equal(got, z[f"triwb_{name}_out"]), name
    # the two cases sit on either side of the 0.1 ratio test (trimap/agent.py:93)
    ratios = []
    for name in ("lo", "hi"):
        m = z[f"triwb_{name}_mask"] > 0
        ratios.append(float((m & (z[f"triwb_{name}_inrange"] > 0)).sum()) / m.sum())
    assert ratios[0] <= 0.1 < ratios[1], ratios
    assert np.array_equal(oracle_lib.is_pixel_inrange(z["inr_img"], z["inr_bgimg"], (20, 20, 120)), z["inr_out"] > 0)


def test_dilate_erode(oracle_lib, z):
    for k, it in ((3, 5), (4, 2), (5, 1)):
        assert np.array_equal(oracle_lib.morph(z["morph_mask"], k, it, "dilate"), z[f"morph_dil_{k}_{it}"])
        assert np.array_equal(oracle_lib.morph(z["morph_mask"], k, it, "erode"), z[f"morph_ero_{k}_{it}"])


def test_remove_invalid_objects(oracle_lib, z, gcfg):
    import copy
    for name in ("land", "port"):
        alpha, seg = z[f"rio_{name}_alpha"], z[f"rio_{name}_seg"]
        a = alpha.copy()
        got = oracle_lib.remove_invalid_objects(gcfg, a)
        assert got is a and np.array_equal(got, z[f"rio_{name}_plain_out"]), name
        c2 = copy.deepcopy(gcfg)
        c2["objectremoval"]["saliency_thr"] = float(z[f"rio_{name}_thr_seg"])
        got = oracle_lib.remove_invalid_objects(c2, alpha.copy(), seg)
        assert np.array_equal(got, z[f"rio_{name}_seg_out"]), name
        assert 0 < (got > 0).sum() < (alpha > 0).sum()             # the case removes some objects and keeps some
    assert [oracle_lib.exist_foreground(z["rio_land_alpha"], t) for t in (0.001, 0.2, 0.9)] == [bool(v) for v in z["ef"]]


def test_get_fg_get_bg_color_correct(oracle_lib, z):
    fg, _ = oracle_lib.get_fg(z["fg_img"], z["fg_alpha"], z["fg_bg"], paste_thr=0)
    assert np.array_equal(fg, z["fg_out"])
    assert np.array_equal(oracle_lib.get_bg(z["fg_alpha"], z["fg_bg"]), z["bg_out"])
    assert np.array_equal(oracle_lib.color_correct(z["cc_img"], z["cc_alpha"], z["cc_bg"], target_long_side=96), z["cc_out"])


def test_harmonization(oracle_lib, z):
    """HarmonizationAgent (harmonization/agent.py:14-113) restated on the oracle's primitives."""
    fg, bg, alpha = z["ha_fg"], z["ha_bg"], z["ha_alpha"]
    h, w = alpha.shape

    def means(img, mask, L):
        th, tw = oracle_lib.get_target_size(h, w, L)
        small = oracle_lib.resize_linear(img, th, tw)
        if mask is None:
            return small.reshape(-1, 3).mean(axis=0)
        return small[oracle_lib.resize_linear(mask.astype(np.uint8) * 255, th, tw) > 0].reshape(-1, 3).mean(axis=0)

    for key, ratio, hi, lo in (("ha_toned", (0.5, 0.05, 0.05), 15, -30), ("ha_toned_strong", (1.0, 1.0, 1.0), 40, -40)):
        fl, bl = oracle_lib.bgr2lab(fg), oracle_lib.bgr2lab(bg)
        fm, bm = means(fl, alpha > 0, 240), means(bl, None, 240)
        f = fl.astype(np.float64)
        for i in range(3):
            f[:, :, i] = f[:, :, i] + max(min(ratio[i] * (bm[i] - fm[i]), hi), lo)
        assert np.array_equal(oracle_lib.lab2bgr(np.clip(f, 0, 255).astype(np.uint8)), z[key]), key
    assert np.allclose(means(bg, alpha > 0, 45), z["ha_means"], rtol=0, atol=1e-12)

    def smooth(img, L):
        th, tw = oracle_lib.get_target_size(h, w, L)
        x = oracle_lib.resize_linear(img, th, tw)
        for _ in range(3):
            x = oracle_lib.box_filter(x, 3)
        return oracle_lib.resize_linear(x, h, w)

    assert np.array_equal(smooth(alpha, 120), z["ha_alpha_smooth"])
    assert np.array_equal(smooth(bg, 45), z["ha_bg_blur"])


def test_lab2bgr_against_float_formula(oracle_lib):
    """The 8-bit integer Lab -> BGR path against the CIE formulas in float64 (independent restatement): within 2 levels."""
    rng = np.random.default_rng(11)
    lab = oracle_lib.bgr2lab(rng.integers(0, 256, (120, 160, 3), dtype=np.uint8))
    got = oracle_lib.lab2bgr(lab).astype(int)
    L = lab[..., 0].astype(float) * 100 / 255
    a, b = lab[..., 1].astype(float) - 128, lab[..., 2].astype(float) - 128
    fy = (L + 16) / 116
    finv = lambda t: np.where(t > 6 / 29, t ** 3, (t - 16 / 116) / 7.787)
    xyz = np.stack([finv(fy + a / 500) * 0.950456, np.where(L > 8, fy ** 3, L / 903.3), finv(fy - b / 200) * 1.088754], -1)
    M = np.array([[3.240479, -1.53715, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])
    rgb = np.clip(xyz @ M.T, 0, 1)
    g = np.where(rgb <= 0.0031308, rgb * 12.92, 1.055 * rgb ** (1 / 2.4) - 0.055)
    want = np.clip(np.round(g * 255), 0, 255)[..., ::-1]
    d = np.abs(got - want)
    assert d.max() <= 2 and (d > 1).mean() < 1e-3, (d.max(), (d > 1).mean())
    # grey axis: a = b = 128 maps back to equal B = G = R
    grey = np.stack([np.arange(256), np.full(256, 128), np.full(256, 128)], -1).astype(np.uint8)[None]
    out = oracle_lib.lab2bgr(grey)[0].astype(int)
    assert np.abs(out - out[:, :1]).max() <= 1 and out[0].tolist() == [0, 0, 0] and out[255].min() >= 254


def test_vmatting_pre_post(oracle_lib, z, gcfg):
    x7 = z["vm_pre_x7"]
    in_h, in_w = oracle_lib.vmat_input_size(*z["vm_trimap"].shape, gcfg["vmatting"]["input_long_side"])
    assert x7.shape == (7, in_h, in_w)
    got = oracle_lib.vmat_pre(z["vm_frame"], z["vm_alpha_pre"], z["vm_trimap"], in_h, in_w)
    assert np.array_equal(got, x7)                                  # float32, bit for bit
    assert np.array_equal(oracle_lib.vmat_post(z["vm_post_pred"], z["vm_trimap"]), z["vm_post_out"])


def test_bg_offline_frame_loop(oracle_lib, z, gcfg):
    """The reference's stage-3 loop (its own UNet on the CPU) against the oracle chain with the folded network on torch-CPU:
    index masks bit-exact, alpha within 1 LSB (two f32 evaluation orders of the same network, then a truncation), foreground
    bit-exact given the same alpha."""
    import __graft_entry__ as ge
    import torch
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict

    class P:                                     # what oracle_frame needs of a pipeline: the CPU network
        net = MattingNet(synthetic_state_dict(0), device="cpu", dtype=torch.float32)

        def cpu_net(self):
            return self.net

    alpha_pre = None
    for t in range(len(z["loop_frames"])):
        want = ge.oracle_frame(oracle_lib, gcfg, P(), z["loop_frames"][t], z["loop_seg"][t], z["loop_bg"][t], z["loop_bg_always"], alpha_pre)
        assert np.array_equal(want["alphaor"], z["loop_alphaor"][t]), t
        assert np.array_equal(want["trimap"], z["loop_trimap"][t]), t
        d = np.abs(want["alpha"].astype(int) - z["loop_alpha"][t].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (t, d.max(), (d > 0).sum())
        fg, _ = oracle_lib.get_fg(z["loop_frames"][t], z["loop_alpha"][t], want["bg"], paste_thr=1)
        assert np.array_equal(fg, z["loop_fg"][t]), t
        alpha_pre = z["loop_alpha"][t]           # both chains continue from the reference's alpha


def test_replace_loop_oracle_chain_matches_reference_python(oracle_lib):
    """tests/golden/replace_loop.npz = the reference's own tools/replace/replace.py (comp_dx_dy :28-41 and the frame loop :59-82, lifted
    with ast) over its own imgprocess.py, run on the cv2 shim (gen_replace.py).  The oracle chain the GPU tests use as their checker -
    get_center of both masks, the float32 shift matrix, shift before zoom, the float64 blend, the background resized once, the
    concatenation - must reproduce it bit for bit: this pins the ORDER and the arguments, not OpenCV's interior."""
    z = np.load(os.path.join(os.path.dirname(GOLD), "replace_loop.npz"))
    n = sum(1 for k in z.files if k.startswith("res_"))
    assert n == 3 and bytes(z["cv2_calls_reached"]).decode().split(",") == ["cvtColor", "moments", "resize", "warpAffine"]
    H, W = z["tgt_fg_0"].shape[:2]
    dxs, dys = [], []
    for t in range(n):
        dc = oracle_lib.get_center(z[f"tgt_mask_{t}"])
        sc = oracle_lib.get_center(oracle_lib.resize_linear(z[f"src_mask_{t}"], H, W))
        dxs.append(sc[0] - dc[0])
        dys.append(sc[1] - dc[1])
    dx, dy = float(np.mean(dxs)), float(np.mean(dys))
    assert [dx, dy] == z["dx_dy_mean"].tolist()
    bg = oracle_lib.resize_linear(z["bg"], H, W)
    M = np.float32([[1, 0, dx], [0, 1, dy]]).astype(np.float64)
    for t in range(n):
        fs = oracle_lib.rescale_fg(oracle_lib.warp_affine(z[f"tgt_fg_{t}"], M, H, W), 1.2)
        ms = oracle_lib.rescale_fg(oracle_lib.warp_affine(z[f"tgt_mask_{t}"], M, H, W), 1.2)
        res = oracle_lib.blend(fs, ms, bg)
        assert np.array_equal(res, z[f"res_{t}"]), t
        assert np.array_equal(np.concatenate((oracle_lib.resize_linear(z[f"src_frame_{t}"], H, W), res), axis=1), z[f"compare_{t}"]), t
