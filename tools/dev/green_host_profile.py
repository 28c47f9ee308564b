"""Where the host spends a green-mode frame (cProfile over the bench's loop): python tools/dev/green_host_profile.py"""
import cProfile, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from video_unscreen_amd.pipeline_green import GreenPipeline
from video_unscreen_amd.synth import SyntheticClip
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
K, Wm = 40, 10
cfg = json.load(open(os.path.join(ROOT, "configs", "green.json")))
for k in ("binseg", "vmatting"):
    cfg[k]["model_path"] = "synthetic"
pipe = GreenPipeline(cfg, H, W, device="cuda:0", cf_random_state=0)
clip = SyntheticClip(H, W, K + Wm, config_id=1, device="cuda:0")
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
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
