import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
H, W = 1080, 1920
f = SyntheticClip(H, W, 1, config_id=2, device="cuda").frame(0)
alpha = (f["alpha_true"] * 255).to(torch.uint8)
bg = nv.get_bg(alpha, f["frame"])
hole = nv.morph((alpha > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
print("hole fraction", float((hole > 0).float().mean()))
out, it = nv.regionfill(bg, hole, want_iters=True)
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    out, it = nv.regionfill(bg, hole, want_iters=True)
torch.cuda.synchronize()
print("scale", os.environ.get("USC_RF_SCALE", "2"), "iters", it, "ms", (time.time() - t) / 3 * 1e3, "checksum", int(out.long().sum()))
