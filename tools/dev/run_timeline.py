"""Timeline of BgStepPipeline.run(): when each frame's result is handed back (host clock), per seg_batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.vmatting.net import synthetic_state_dict
H, W = 1080, 1920
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
flt = sys.argv[2] if len(sys.argv) > 2 else "auto"
n = 6 * B
clip = SyntheticClip(H, W, n, config_id=2, device="cuda")
frames = []
for t in range(n):
    f = clip.frame(t)
    frames.append((f["frame"], f["seg"], f["bg"], clip.bg_always))
pipe = BgStepPipeline(default_config(), H, W, state_dict=synthetic_state_dict(0), seg_agent=HumanSegAgent(model_path=None), seg_batch=B, object_filter=flt)
pipe.run(iter(frames[:2 * B]))
torch.cuda.synchronize()
stamps = []
t0 = time.perf_counter()
pipe.run(iter(frames), on_result=lambda k, o: stamps.append(time.perf_counter() - t0))
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"seg_batch {B} filter {flt}: {n / tot:.1f} fps")
prev = 0.
for g in range(0, n, B):
    row = " ".join(f"{(stamps[i] - (stamps[i - 1] if i else 0.)) * 1e3:5.1f}" for i in range(g, g + B))
    print(f"  group {g // B}: done at {stamps[g + B - 1] * 1e3:7.1f} ms | per-frame gaps ms: {row}")
