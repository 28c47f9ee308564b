"""Orchestration golden of the replace tool (BASELINE.json configs[4], the compositing half): the reference's OWN
tools/replace/replace.py - comp_dx_dy (:28-41) and the frame loop of run() (:59-82), lifted out with ast and executed as they stand -
over the reference's own unscreen/utils/imgprocess.py (get_center, adaptive_resize, shift_fg, rescale_fg), in the build container only,
against a module named `cv2` that is backed by the C oracle (tests/golden/cv2_shim.py) and an `mmcv` whose imread / imwrite hand arrays
over in memory.  Output: tests/golden/replace_loop.npz (data only).

Pins which array reaches which call in which order: the 3-channel alpha masks through get_center (RGB2GRAY + moments, int truncation),
the float32 shift matrix, shift BEFORE zoom, fx = fy = 1.2 with the crop offsets int((new - old) / 2), the float64 blend and its
truncation, the background resized once to the target's size, the side-by-side concatenation.  It does not pin OpenCV's arithmetic inside a
call (the shim is the oracle's restatement).  Usage: python tests/golden/gen_replace.py"""
import ast
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"


def main():
    import scipy.ndimage as ndi                                     # (before np.bool is aliased below: numpy.ma trips over the alias while importing)
    import gen_orchestration as go
    utils, _, _, shim = go.load_reference()                         # the reference's imgprocess.py over the shim; np.float restored
    rng = np.random.default_rng(20260007)
    H, W, sh, sw, n = 61, 93, 48, 80, 3                             # 61 x 1.2 and 93 x 1.2 are not integers; the source has another size
    tex = lambda h, w: np.clip(ndi.gaussian_filter(rng.normal(128, 90, (h, w, 3)), (1.2, 1.2, 0)), 0, 255).astype(np.uint8)   # noqa: E731

    def silhouette(h, w, cy, cx):
        yy, xx = np.mgrid[0:h, 0:w]
        d = np.hypot((yy - h * cy) / (h * 0.36), (xx - w * cx) / (w * 0.2))
        m = np.clip((1.05 - d) * 6, 0, 1) * 255
        return np.clip(np.stack([m] * 3, -1) + rng.integers(-3, 4, (h, w, 3)), 0, 255).astype(np.uint8)

    files = {}
    for t in range(n):
        m3 = silhouette(H, W, 0.55 + 0.02 * t, 0.40 + 0.03 * t)
        files[f"tgt/alphamask_{t:06d}.jpg"] = m3
        files[f"tgt/fg_{t:06d}.jpg"] = (tex(H, W) * (m3 / 255.)).astype(np.uint8)
        files[f"src/alphamask_{t:06d}.jpg"] = silhouette(sh, sw, 0.50, 0.62 - 0.02 * t)
        files[f"src/frame_{t:06d}.jpg"] = tex(sh, sw)
    files["bg.jpg"] = tex(70, 110)
    saved = {}
    mmcv = types.SimpleNamespace(imread=lambda path: files[path].copy(), imwrite=lambda img, path: saved.__setitem__(path, np.array(img, copy=True)),
                                 frames2video=lambda *a, **k: None)
    args = types.SimpleNamespace(tgt_data_dir="tgt", src_data_dir="src", dst_data_dir="dst", src_bg_image="bg.jpg", src="golden", tgt="golden")
    tree = ast.parse(open(os.path.join(REF, "tools/replace/replace.py")).read())
    fns = {x.name: x for x in tree.body if isinstance(x, ast.FunctionDef)}
    ns = {"np": np, "osp": os.path, "mmcv": mmcv, "tqdm": lambda x: x, "get_center": utils.get_center, "adaptive_resize": utils.adaptive_resize,
          "shift_fg": utils.shift_fg, "rescale_fg": utils.rescale_fg}
    exec(compile(ast.Module(body=[fns["comp_dx_dy"]], type_ignores=[]), "replace.py", "exec"), ns)
    assert 25 <= fns["comp_dx_dy"].lineno <= 30
    dx_mean, dy_mean = ns["comp_dx_dy"](args, n)
    # the frame loop of run(): `for fid in tqdm(range(numframes))` with the blend inside
    loops = [x for x in ast.walk(fns["run"]) if isinstance(x, ast.For) and getattr(x.target, "id", "") == "fid"]
    assert len(loops) == 1 and 55 <= loops[0].lineno <= 62 and 78 <= loops[0].end_lineno <= 84, (loops[0].lineno, loops[0].end_lineno)
    ns.update(args=args, numframes=n, dx_mean=dx_mean, dy_mean=dy_mean, src_bg_image=mmcv.imread(args.src_bg_image))
    exec(compile(ast.Module(body=[loops[0]], type_ignores=[]), "replace.py", "exec"), ns)
    out = {"dx_dy_mean": np.array([dx_mean, dy_mean], np.float64), "bg": files["bg.jpg"]}
    for t in range(n):
        out[f"tgt_fg_{t}"], out[f"tgt_mask_{t}"] = files[f"tgt/fg_{t:06d}.jpg"], files[f"tgt/alphamask_{t:06d}.jpg"]
        out[f"src_frame_{t}"], out[f"src_mask_{t}"] = files[f"src/frame_{t:06d}.jpg"], files[f"src/alphamask_{t:06d}.jpg"]
        out[f"res_{t}"], out[f"compare_{t}"] = saved[f"dst/res_{t:06d}.jpg"], saved[f"dst/compare_{t:06d}.jpg"]
    used = sorted({c[0] for c in shim.calls})
    out["cv2_calls_reached"] = np.frombuffer(",".join(used).encode(), np.uint8)
    np.savez_compressed(os.path.join(HERE, "replace_loop.npz"), **out)
    print("replace_loop.npz:", len(out), "arrays; mean offset", dx_mean, dy_mean, "; cv2 entry points reached:", used)
    print("res != bg on", [float((out[f'res_{t}'] != 0).mean()) for t in range(n)])


if __name__ == "__main__":
    main()
