import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd import _native as nv
from video_unscreen_amd.synth import SyntheticClip
agent = HumanSegAgent(None)
frame = SyntheticClip(1080, 1920, 1, device="cuda").frame(0)["frame"]
def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
print("agent.forward_device 1080p: %.3f ms" % timeit(lambda: agent.forward_device(frame)))
print("  pre: %.3f ms" % timeit(lambda: nv.humanseg_pre(frame, out=agent.net.x4)))
print("  net: %.3f ms  (152.97 GFLOP -> %.1f TF/s)" % ((t := timeit(lambda: agent.net.forward(None))), 152.97 / t))
lg = agent.net.forward(None)
print("  post: %.3f ms" % timeit(lambda: nv.humanseg_post(lg, 20, 1080, 1920, ws=agent._ws)))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(3): agent.net.forward(None)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=60))
