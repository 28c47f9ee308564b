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
