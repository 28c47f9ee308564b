"""The hole the bg-online bench's Laplace solver works on (the synthetic clip through the untrained networks: a ragged mask), saved for
the NumPy model of the solver (tools/dev/rf_model.py):  python tools/dev/dump_bgonline_hole.py -> gpurun_out/bgonline_hole.npz"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
from video_unscreen_amd import _native as nv
H, W = 1080, 1920
cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
for k in ("binseg", "vmatting", "stm"):
    cfg[k]["model_path"] = "synthetic"
pipe = BgOnlinePipeline(cfg, H, W, device="cuda:0")
clip = SyntheticClip(H, W, 12, config_id=2, device="cuda:0")
clip.gen.manual_seed(20260000)
out = None
for t in range(12):
    out = pipe.step(clip.frame(t)["frame"])
bg, hole = out["bg_holes"], out["hole"]
res, iters = nv.regionfill(bg, hole, want_iters=True)
print("iterations", iters, "hole fraction", float((hole > 0).float().mean()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "bgonline_hole.npz"), hole=np.packbits(hole.cpu().numpy() > 0), bg=bg.cpu().numpy()[::1, ::1, 1],
                    fill=res.cpu().numpy()[..., 1], shape=np.array([H, W]), iters=iters)
