"""Laplace solver at 3840 x 2160 (and an odd size): iterations, time, and the residual of the discrete Laplace equation of the float
solution's uint8 truncation neighbourhood (max |n x - sum of neighbours| over the hole must be a few grey levels: truncated values)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
for (H, W) in ((2160, 3840), (1083, 1925), (717, 1279)):
    f = SyntheticClip(H, W, 1, config_id=2, device="cuda").frame(0)
    alpha = (f["alpha_true"] * 255).to(torch.uint8)
    bg = nv.get_bg(alpha, f["frame"])
    hole = nv.morph((alpha > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
    out, it = nv.regionfill(bg, hole, want_iters=True)
    torch.cuda.synchronize()
    t = time.time()
    out, it = nv.regionfill(bg, hole, want_iters=True)
    torch.cuda.synchronize()
    dt = time.time() - t
    x = out.double()
    m = hole > 0
    p = torch.nn.functional.pad(x.permute(2, 0, 1)[None], (1, 1, 1, 1), mode="replicate")[0].permute(1, 2, 0)
    nb = p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:]
    res = (4 * x - nb)[m]
    inner = m.clone(); inner[0, :] = inner[-1, :] = False; inner[:, 0] = inner[:, -1] = False
    print("%d x %d: hole %.3f, %d iterations, %.2f ms; |4x - sum nb| over the hole (uint8 values): mean %.3f, max %.0f" % (
        W, H, float(m.float().mean()), it, dt * 1e3, float((4 * x - nb)[inner].abs().mean()), float((4 * x - nb)[inner].abs().max())), flush=True)
