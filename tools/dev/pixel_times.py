"""Standalone durations of the pixel stages of the bg_step loop at one frame size (HIP events, 50 launches each):
python tools/dev/pixel_times.py [H W]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from video_unscreen_amd import _native as nv
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
nv.load()
clip = SyntheticClip(H, W, 2, config_id=2, device="cuda")
f = clip.frame(0)
pipe = BgStepPipeline(default_config(), H, W)
out = pipe.step(f["frame"], f["seg"], f["bg"], clip.bg_always)
slot = pipe.slots[0]
alpha, tri, gate = out["alpha"].clone(), out["trimap"].clone(), slot.gate.clone()
pred = out["pred"].clone().view(pipe.in_h, pipe.in_w)
logits = torch.randn(119, 119, 32, device="cuda") * 5
flt = nv.DeviceObjectFilter()
alphaor = torch.empty_like(gate)


def t(name, fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) * 1e3 / n:8.2f} us")


t("bgdiff_mask", lambda: nv.bgdiff_mask(f["frame"], f["bg"], clip.bg_always, f["seg"], pipe.beta, pipe.thr, out_bg=slot.bg, out_alpha=slot.gate))
t("object filter", lambda: flt(alphaor, None, pipe.center, pipe.sal_thr, pipe.cons_thr, src=gate))
t("trimap", lambda: nv.trimap(gate, pipe.long_side, pipe.ksize, pipe.iters, out=slot.trimap, ws=pipe.ws_trimap))
t("vmat_pre", lambda: nv.vmat_pre(f["frame"], alpha, tri, pipe.in_h, pipe.in_w, layout=pipe._layout, out=pipe.x7))
t("vmat_post", lambda: nv.vmat_post(pred, tri, out_u8=alpha))
t("get_fg", lambda: nv.get_fg(f["frame"], alpha, slot.bg, paste_thr=1, out_fg=out["fg"]))
t("vmat_post_fg (the two as one launch)", lambda: nv.vmat_post_fg(pred, tri, f["frame"], slot.bg, paste_thr=1, out_u8=alpha, out_fg=out["fg"]))
t("humanseg_pre", lambda: nv.humanseg_pre(f["frame"], 473, 473))
seg_out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
t("humanseg_post", lambda: nv.humanseg_post(logits, 20, H, W, out=seg_out))
