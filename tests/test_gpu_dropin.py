"""GPU tests of the NumPy-level drop-in helpers (unscreen.utils names the reference's tools / agents import): the HIP kernels of
csrc/usc_dropin.hip against the outputs of the REFERENCE's own functions (tests/golden/dropin.npz) and against the oracle on
larger random inputs - bit-exact (integer / single-rounded float arithmetic)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def z():
    return np.load(os.path.join(GOLD, "dropin.npz"))


def test_get_center_get_mask_vs_reference(z, oracle_lib):
    from unscreen.utils import get_center, get_mask
    for name in ("a", "b"):
        img = z[f"gc_{name}_img"]
        assert get_center(img.copy()) == tuple(z[f"gc_{name}_center"])
        th, binm = get_mask(img.copy())
        assert th.dtype == np.uint8 and np.array_equal(th, z[f"gm_{name}_thresh"])
        assert binm.dtype == np.float64 and np.array_equal(binm, z[f"gm_{name}_binary"])
    assert get_center(np.zeros((8, 9, 3), np.uint8)) == (0, 0)
    img = np.random.default_rng(1).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)        # sums beyond 32 bits
    assert get_center(img) == oracle_lib.get_center(img)
    assert np.array_equal(get_mask(img)[0], oracle_lib.get_mask(img)[0])


def test_pad_resize_vs_reference(z, oracle_lib):
    from unscreen.utils import inv_pad_resize, pad_resize
    for n in sorted(k[3:-4] for k in z.files if k.startswith("pr_") and k.endswith("_img")):
        out, ratio = pad_resize(z[f"pr_{n}_img"].copy(), tuple(z[f"pr_{n}_tgt"]))
        assert np.array_equal(out, z[f"pr_{n}_out"]), n
        assert ratio == float(z[f"pr_{n}_ratio"])
    for n in sorted(k[4:-4] for k in z.files if k.startswith("ipr_") and k.endswith("_img")):
        out = inv_pad_resize(z[f"ipr_{n}_img"].copy(), tuple(z[f"ipr_{n}_ori"]))
        want = z[f"ipr_{n}_out"]
        assert out.dtype == want.dtype and np.array_equal(out, want), n
    rng = np.random.default_rng(2)
    for (h, w), tgt in (((1080, 1920), (544, 960)), ((2160, 3840), (544, 960)), ((256, 256), (960, 960)), ((1920, 1080), (800, 450))):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(pad_resize(img, tgt)[0], oracle_lib.pad_resize(img, tgt)[0]), (h, w)
        m = rng.integers(0, 256, (h, w), dtype=np.uint8)
        assert np.array_equal(pad_resize(m, tgt)[0], oracle_lib.pad_resize(m, tgt)[0])
        p = rng.random(tgt).astype(np.float32)
        assert np.array_equal(inv_pad_resize(p, (h, w)), oracle_lib.inv_pad_resize(p, (h, w)))
        s = rng.random(tgt + (2,))
        assert np.array_equal(inv_pad_resize(s, (h // 2, w // 2)), oracle_lib.inv_pad_resize(s, (h // 2, w // 2)))


def test_imnormalize_vs_reference(z, oracle_lib):
    from unscreen.utils import imnormalize
    arg = z["imn_img"].copy()
    out = imnormalize(arg)
    assert out.dtype == np.float32 and np.array_equal(out, z["imn_out"])
    assert np.array_equal(arg, z["imn_arg_after"])                       # the argument is RGB now (reference imgprocess.py:157)
    arg = z["imn_img"].copy()
    assert np.array_equal(imnormalize(arg, to_rgb=False), z["imn_out_bgr"]) and np.array_equal(arg, z["imn_img"])
    img = np.random.default_rng(3).integers(0, 256, (540, 960, 3), dtype=np.uint8)
    assert np.array_equal(imnormalize(img.copy()), oracle_lib.imnormalize(img.copy()))


def test_score_maps_vs_reference(z, cfg):
    from unscreen.utils import build_score_map, get_score_map
    for name, (h, w) in (("land", (54, 96)), ("port", (96, 54)), ("sq", (50, 50))):
        got = build_score_map(h, w, cfg)
        assert got.dtype == np.float64 and np.array_equal(got, z[f"sm_{name}"])
    assert np.array_equal(get_score_map((40, 70), (0.3, 0.8)), z["sm_custom"])


def test_fgfuncs_options_no_tool_reaches_vs_reference(z):
    """is_pixel_inrange(long_side_input > 0) in both background forms, get_fg_naive, get_fg_with_colorremove (alpha modified in place),
    composite_fgbg(extend=True): outputs of the reference's own functions (fgfuncs.py:9-81, 140-214)."""
    from unscreen.utils import composite_fgbg, get_fg_naive, get_fg_with_colorremove, is_pixel_inrange
    img, bgimg, color = z["ir_img"], z["ir_bgimg"], z["ir_color"]
    for L in (48, 144):
        got = is_pixel_inrange(img.copy(), color.copy(), (10, 100, 180), long_side_input=L)
        assert got.dtype == bool and np.array_equal(got.astype(np.uint8), z[f"ir_color_L{L}"]), L
    assert np.array_equal(is_pixel_inrange(img.copy(), bgimg.copy(), (20, 20, 120), long_side_input=48).astype(np.uint8), z["ir_img_L48"])
    assert np.array_equal(get_fg_naive(img.copy(), z["fgn_alpha"].copy()), z["fgn_out"])
    a = z["fgn_alpha"].copy()
    assert np.array_equal(get_fg_with_colorremove(img.copy(), a, bgimg.copy(), winsize=(10, 100, 120), long_side_input=48), z["fgcr_out"])
    assert np.array_equal(a, z["fgcr_alpha_after"]) and not np.array_equal(a, z["fgn_alpha"])
    for name in ("wide", "tall"):
        bg = z[f"comp_{name}_bg"]
        assert np.array_equal(composite_fgbg(z["comp_fg"].copy(), z["comp_alpha"].copy(), bg.copy()), z[f"comp_{name}_plain"])
        ext = composite_fgbg(z["comp_fg"].copy(), z["comp_alpha"].copy(), bg.copy(), extend=True)
        assert ext.shape == z[f"comp_{name}_extend"].shape and np.array_equal(ext, z[f"comp_{name}_extend"]), name
