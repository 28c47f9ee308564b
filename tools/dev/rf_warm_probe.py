"""Would a warm start pay for the online mode's inpainting?  Consecutive frames of a clip: cold solve (x0 = the image) against a solve
whose x0 inside the hole is the previous frame's filled background, at looser tolerances - iterations and difference of the u8 result
from the cold 1e-7 solve.   python tools/dev/rf_warm_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
H, W = 1080, 1920
clip = SyntheticClip(H, W, 8, config_id=2, device="cuda")
prev = None
for t in range(6):
    f = clip.frame(t)
    alpha = (f["alpha_true"] * 255).to(torch.uint8)
    bg = nv.get_bg(alpha, f["frame"])
    hole = nv.morph((alpha > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
    cold, it_cold = nv.regionfill(bg, hole, rtol=1e-7, want_iters=True)
    line = f"frame {t}: cold 1e-7: {it_cold} iterations"
    if prev is not None:
        guess = torch.where((hole > 0).unsqueeze(-1), prev, bg)
        for rtol in (1e-7, 1e-5, 1e-4, 1e-3):
            out, it = nv.regionfill(guess, hole, rtol=rtol, want_iters=True)
            d = (out.int() - cold.int()).abs()
            line += f" | warm rtol {rtol:g}: {it} it, {int((d > 0).sum())} values differ (max {int(d.max())})"
    print(line, flush=True)
    prev = cold
