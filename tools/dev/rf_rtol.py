"""How the regionfill tolerance shows in the u8 output: compare rtol = 1e-4 ... 1e-8 with a 1e-12 solve on two 1080p holes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
H, W = 1080, 1920
for cid, t in ((2, 0), (2, 40), (3, 17)):
    f = SyntheticClip(H, W, 64, config_id=cid, device="cuda").frame(t)
    alpha = (f["alpha_true"] * 255).to(torch.uint8)
    bg = nv.get_bg(alpha, f["frame"])
    hole = nv.morph((alpha > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
    ref, it0 = nv.regionfill(bg, hole, rtol=1e-12, max_iters=400, want_iters=True)
    for rtol in (1e-4, 1e-5, 1e-6, 1e-7, 1e-8):
        out, it = nv.regionfill(bg, hole, rtol=rtol, want_iters=True)
        d = (out.int() - ref.int()).abs()
        print(f"clip {cid} frame {t}: rtol {rtol:g}: iters {it:3d} (ref {it0}), differing values {int((d > 0).sum())}, max diff {int(d.max())}")
