"""Where the host spends a bg-online frame (cProfile over the bench's loop): python tools/dev/bgonline_host_profile.py"""
import cProfile, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
from video_unscreen_amd.synth import SyntheticClip
H, W, K, Wm = 1080, 1920, 40, 10
cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
for k in ("binseg", "vmatting", "stm"):
    cfg[k]["model_path"] = "synthetic"
pipe = BgOnlinePipeline(cfg, H, W, device="cuda:0")
clip = SyntheticClip(H, W, K + Wm, config_id=2, device="cuda:0")
clip.gen.manual_seed(20260000)
frames = [clip.frame(t)["frame"] for t in range(K + Wm)]
for f in frames[:Wm]:
    pipe.step(f)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for f in frames[Wm:]:
    pipe.step(f)
torch.cuda.synchronize()
pr.disable()
print("ms per frame", (time.perf_counter() - t0) / K * 1e3)
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
