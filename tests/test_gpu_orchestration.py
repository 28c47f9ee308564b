"""GPU tests (-m gpu): the drop-in agents / utils (NumPy API of the reference, arithmetic in libunscreen_hip.so) and the device
pipeline against tests/golden/orchestration.npz - outputs of the REFERENCE's own Python (run against a cv2-named shim over the
oracle's primitives, tests/golden/gen_orchestration.py).  Same calls, same arguments, same results as the reference's classes:
byte stages bit-exact, alpha within 1 LSB where a float network sits in between.  (What the fixture does not pin: OpenCV's
arithmetic inside a call - see test_orchestration_cpu.py.)"""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "orchestration.npz")


@pytest.fixture(scope="module")
def z():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from video_unscreen_amd import _native
    _native.load()
    return np.load(GOLD)


@pytest.fixture(scope="module")
def gcfg(z):
    return json.loads(bytes(z["cfg_json"]).decode())


def test_trimap_agent(z):
    from video_unscreen_amd.trimap import TrimapAgent
    for name in "abcd":
        L, k, it = (int(v) for v in z[f"tri_{name}_params"])
        m = z[f"tri_{name}_mask"].copy()
        got = TrimapAgent(input_long_side=L, kernelsize=k, iters=it).forward(m)
        assert np.array_equal(got, z[f"tri_{name}_out"]), name
        assert np.array_equal(m, z[f"tri_{name}_mask"])                      # argument untouched
    agent = TrimapAgent(input_long_side=144, iters=2)
    for name in ("lo", "hi"):
        got = agent.forward(z[f"triwb_{name}_mask"].copy(), z[f"triwb_{name}_img"].copy(), z[f"triwb_{name}_bg"].copy())
        assert np.array_equal(got, z[f"triwb_{name}_out"]), name
    empty = np.zeros((20, 30), np.uint8)
    assert agent.forward(empty, np.zeros((20, 30, 3), np.uint8), np.array([60, 200, 60], np.uint8)) is empty


def test_utils_functions(z, gcfg):
    from video_unscreen_amd import utils
    for name in ("lo", "hi"):
        got = utils.is_pixel_inrange(z[f"triwb_{name}_img"], z[f"triwb_{name}_bg"], (10, 100, 180))
        assert got.dtype == bool and np.array_equal(got, z[f"triwb_{name}_inrange"] > 0)
    assert np.array_equal(utils.is_pixel_inrange(z["inr_img"], z["inr_bgimg"], (20, 20, 120)), z["inr_out"] > 0)
    for k, it in ((3, 5), (4, 2), (5, 1)):
        assert np.array_equal(utils.dilate_mask(z["morph_mask"].copy(), k, it), z[f"morph_dil_{k}_{it}"])
        assert np.array_equal(utils.erode_mask(z["morph_mask"].copy(), k, it), z[f"morph_ero_{k}_{it}"])
    for name in ("land", "port"):
        a = z[f"rio_{name}_alpha"].copy()
        got = utils.remove_invalid_objects(gcfg, a)
        assert got is a and np.array_equal(got, z[f"rio_{name}_plain_out"]), name   # in place, like the reference
        c2 = copy.deepcopy(gcfg)
        c2["objectremoval"]["saliency_thr"] = float(z[f"rio_{name}_thr_seg"])
        got = utils.remove_invalid_objects(c2, z[f"rio_{name}_alpha"].copy(), z[f"rio_{name}_seg"].copy())
        assert np.array_equal(got, z[f"rio_{name}_seg_out"]), name
    assert [utils.exist_foreground(z["rio_land_alpha"], t) for t in (0.001, 0.2, 0.9)] == [bool(v) for v in z["ef"]]
    assert np.array_equal(utils.get_fg(z["fg_img"].copy(), z["fg_alpha"].copy(), z["fg_bg"].copy()), z["fg_out"])
    assert np.array_equal(utils.get_bg(z["fg_alpha"].copy(), z["fg_bg"].copy()), z["bg_out"])
    assert np.array_equal(utils.color_correct(z["cc_img"].copy(), z["cc_alpha"].copy(), z["cc_bg"].copy(), target_long_side=96), z["cc_out"])


def test_harmonization_agent(z):
    from video_unscreen_amd.harmonization import HarmonizationAgent
    ha = HarmonizationAgent()
    fg, bg, alpha = z["ha_fg"], z["ha_bg"], z["ha_alpha"]
    assert np.array_equal(ha.foreground_toning(fg.copy(), bg.copy(), alpha.copy()), z["ha_toned"])
    got = ha.foreground_toning(fg.copy(), bg.copy(), alpha.copy(), toning_ratio=(1.0, 1.0, 1.0), max_shift=40, min_shift=-40)
    assert np.array_equal(got, z["ha_toned_strong"])
    assert np.allclose(ha.get_means(bg.copy(), alpha > 0, target_long_side=45), z["ha_means"], rtol=0, atol=1e-9)
    assert np.array_equal(ha.alpha_smoothing(alpha.copy(), target_long_side=120), z["ha_alpha_smooth"])
    assert np.array_equal(ha.background_blurring(bg.copy(), target_long_side=45), z["ha_bg_blur"])


def test_vmatting_pre_post_kernels(z, gcfg):
    from video_unscreen_amd import _native as nv
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    x7 = z["vm_pre_x7"]
    in_h, in_w = x7.shape[1:]
    got = nv.vmat_pre(dev(z["vm_frame"]), dev(z["vm_alpha_pre"]), dev(z["vm_trimap"]), in_h, in_w, layout=nv.NCHW)
    assert np.array_equal(got.cpu().numpy()[0], x7)                          # the reference's preprocess tensors, bit for bit
    got = nv.vmat_post(dev(z["vm_post_pred"]), dev(z["vm_trimap"]))
    assert np.array_equal(got.cpu().numpy(), z["vm_post_out"])


def test_bg_offline_frame_loop(z, gcfg):
    """BgStepPipeline.step against the reference's stage-3 loop (tools/unscreen/bg_offline.py:145-176, its own UNet on the CPU)."""
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.pipeline import BgStepPipeline
    frames = z["loop_frames"]
    n, H, W, _ = frames.shape
    pipe = BgStepPipeline(gcfg, H, W, device="cuda:0")
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    bga = dev(z["loop_bg_always"])
    for t in range(n):
        out = pipe.step(dev(frames[t]), dev(z["loop_seg"][t]), dev(z["loop_bg"][t]), bga)
        assert np.array_equal(out["alphaor"].cpu().numpy(), z["loop_alphaor"][t]), t
        assert np.array_equal(out["trimap"].cpu().numpy(), z["loop_trimap"][t]), t
        a = out["alpha"].cpu().numpy()
        d = np.abs(a.astype(int) - z["loop_alpha"][t].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 1e-3, (t, int(d.max()), int((d > 0).sum()))
        ref_a = dev(z["loop_alpha"][t])
        fg = nv.get_fg(dev(frames[t]), ref_a, out["bg"], paste_thr=1)
        assert np.array_equal(fg.cpu().numpy(), z["loop_fg"][t]), t
        pipe.alpha_pre = ref_a.clone()                                       # both chains continue from the reference's alpha
