"""Goldens for the NumPy-level names of `unscreen.utils` that the reference's tools and agents import but that round 2 only had
in fused form (VERDICT round 2, row b): get_center, get_mask, pad_resize, inv_pad_resize, imnormalize, get_score_map,
build_score_map - outputs of the reference's OWN functions (unscreen/utils/imgprocess.py, maskprocess.py, loaded by path exactly
as gen_orchestration.py does, `cv2` = the oracle-backed shim) - and the list of every name the reference's tools import from
`unscreen.*` (ast over /root/reference/tools/**).  Output: tests/golden/dropin.npz + tests/golden/tool_imports.json (data only).
Usage: python tests/golden/gen_dropin.py"""
import ast
import glob
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_orchestration as go      # noqa: E402

REF = go.REF


def tool_imports():
    rows = set()
    for p in sorted(glob.glob(os.path.join(REF, "tools", "**", "*.py"), recursive=True)):
        for n in ast.walk(ast.parse(open(p).read())):
            if isinstance(n, ast.ImportFrom) and n.module and n.module.split(".")[0] == "unscreen":
                for a in n.names:
                    rows.add((n.module, a.name, os.path.relpath(p, REF)))
    return sorted(rows)


def main():
    utils, _, _, shim = go.load_reference()
    rng = np.random.default_rng(20260007)
    out = {}
    # ---- get_center / get_mask (imgprocess.py:7-30)
    for name, (h, w) in (("a", (40, 64)), ("b", (64, 40))):
        img = np.zeros((h, w, 3), np.uint8)
        m = go.blobs(rng, h, w, 3, 12) > 0
        img[m] = rng.integers(0, 256, (int(m.sum()), 3), dtype=np.uint8)
        out[f"gc_{name}_img"] = img
        out[f"gc_{name}_center"] = np.array(utils.get_center(img.copy()), np.int64)
        th, binm = utils.get_mask(img.copy())
        out[f"gm_{name}_thresh"], out[f"gm_{name}_binary"] = th, binm
    out["gc_zero_center"] = np.array(utils.get_center(np.zeros((8, 9, 3), np.uint8)), np.int64)
    # ---- pad_resize (imgprocess.py:67-97): 3 / 1 channels, both aspect branches, exact 2x decimation, same size
    cases = (("land3", (54, 96, 3), (64, 96)), ("port1", (96, 54), (96, 64)), ("half3", (64, 96, 3), (32, 48)), ("same1", (32, 48), (32, 48)),
             ("up3", (27, 48, 3), (64, 96)), ("tall3", (90, 40, 3), (64, 64)))
    for name, shp, tgt in cases:
        img = rng.integers(0, 256, shp, dtype=np.uint8)
        res, ratio = utils.pad_resize(img.copy(), tgt)
        out[f"pr_{name}_img"], out[f"pr_{name}_tgt"], out[f"pr_{name}_out"], out[f"pr_{name}_ratio"] = img, np.array(tgt), res, np.float64(ratio)
    # ---- inv_pad_resize (imgprocess.py:100-132): uint8, float32 (network alpha / probability), float64 2-channel (SegAgent scores)
    for name, shp, ori, dt in (("u8", (64, 96), (54, 96), np.uint8), ("f32", (64, 96), (135, 240), np.float32), ("f32half", (64, 96), (32, 48), np.float32),
                               ("f64c2", (64, 96, 2), (100, 150), np.float64), ("f32tall", (96, 64), (200, 90), np.float32),
                               ("u8c3", (64, 96, 3), (27, 48), np.uint8), ("f32same", (64, 96), (64, 96), np.float32)):
        img = rng.integers(0, 256, shp, dtype=np.uint8) if dt == np.uint8 else rng.random(shp).astype(dt)
        out[f"ipr_{name}_img"], out[f"ipr_{name}_ori"] = img, np.array(ori)
        out[f"ipr_{name}_out"] = utils.inv_pad_resize(img.copy(), ori)
    # ---- imnormalize (imgprocess.py:135-161): result AND what it leaves in its argument
    img = rng.integers(0, 256, (24, 36, 3), dtype=np.uint8)
    arg = img.copy()
    out["imn_img"], out["imn_out"], out["imn_arg_after"] = img, utils.imnormalize(arg), arg
    arg2 = img.copy()
    out["imn_out_bgr"] = utils.imnormalize(arg2, to_rgb=False)
    assert np.array_equal(arg2, img)
    # ---- get_score_map / build_score_map (maskprocess.py:155-189)
    for name, (h, w) in (("land", (54, 96)), ("port", (96, 54)), ("sq", (50, 50))):
        out[f"sm_{name}"] = utils.build_score_map(h, w, go.CFG)
    out["sm_custom"] = utils.get_score_map((40, 70), (0.3, 0.8))
    # ---- options of fgfuncs.py that no reference tool reaches ("dead-branch options"): is_pixel_inrange with
    # long_side_input > 0 (both background forms), get_fg_naive, get_fg_with_colorremove (modifies its alpha), composite_fgbg(extend)
    h, w = 54, 96
    bg_color = np.array([60, 200, 60], np.uint8)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    near = rng.random((h, w)) < 0.4
    img[near] = np.clip(bg_color.astype(int) + rng.integers(-4, 5, (int(near.sum()), 3)), 0, 255).astype(np.uint8)
    bgimg = np.clip(img.astype(int) + rng.integers(-30, 31, img.shape), 0, 255).astype(np.uint8)
    out["ir_img"], out["ir_bgimg"], out["ir_color"] = img, bgimg, bg_color
    out["ir_color_L48"] = utils.is_pixel_inrange(img.copy(), bg_color.copy(), (10, 100, 180), long_side_input=48).astype(np.uint8)
    out["ir_color_L144"] = utils.is_pixel_inrange(img.copy(), bg_color.copy(), (10, 100, 180), long_side_input=144).astype(np.uint8)
    out["ir_img_L48"] = utils.is_pixel_inrange(img.copy(), bgimg.copy(), (20, 20, 120), long_side_input=48).astype(np.uint8)
    alpha = rng.integers(0, 256, (h, w), dtype=np.uint8)
    out["fgn_alpha"], out["fgn_out"] = alpha, utils.get_fg_naive(img.copy(), alpha.copy())
    a_arg = alpha.copy()
    out["fgcr_out"] = utils.get_fg_with_colorremove(img.copy(), a_arg, bgimg.copy(), winsize=(10, 100, 120), long_side_input=48)
    out["fgcr_alpha_after"] = a_arg
    fg = rng.integers(0, 256, (40, 64, 3), dtype=np.uint8)
    fa = rng.integers(0, 256, (40, 64), dtype=np.uint8)
    for name, bshape in (("wide", (50, 120, 3)), ("tall", (90, 70, 3))):
        cbg = rng.integers(0, 256, bshape, dtype=np.uint8)
        out[f"comp_{name}_bg"] = cbg
        out[f"comp_{name}_plain"] = utils.composite_fgbg(fg.copy(), fa.copy(), cbg.copy())
        out[f"comp_{name}_extend"] = utils.composite_fgbg(fg.copy(), fa.copy(), cbg.copy(), extend=True)
    out["comp_fg"], out["comp_alpha"] = fg, fa
    # ---- index helpers: get_fgbox, remove_blackborder / add_blackborder
    m = np.zeros((40, 60), np.uint8)
    m[12:30, 20:41] = 255
    out["box_mask"], out["box_out"] = m, np.array(utils.get_fgbox(m, 5), np.int64)
    bb = np.zeros((30, 44, 3), np.uint8)
    bb[4:25, 6:40] = rng.integers(1, 256, (21, 34, 3), dtype=np.uint8)
    cropped, loc = utils.remove_blackborder(bb.copy())
    out["bb_img"], out["bb_cropped"], out["bb_loc"] = bb, cropped, np.array(loc, np.int64)
    out["bb_restored"] = utils.add_blackborder(cropped.copy(), (30, 44), loc)
    np.savez_compressed(os.path.join(HERE, "dropin.npz"), **out)
    rows = tool_imports()
    json.dump([list(r) for r in rows], open(os.path.join(HERE, "tool_imports.json"), "w"), indent=0)
    print("dropin.npz:", len(out), "arrays; tool_imports.json:", len(rows), "rows; cv2 calls:", sorted({c[0] for c in shim.calls}))


if __name__ == "__main__":
    main()
