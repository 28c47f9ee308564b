"""Is the frame loop bound by the host thread that issues it?  Device-resident loop of bench.py's default workload: time until
pipe.run() has ISSUED everything vs. time until the GPU has finished; and the host cost of the loop with the GPU side removed
(HIP launches still go out, on a 16 x 16 clip the kernels are empty).   python tools/dev/host_issue_probe.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_native()
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
dev = torch.device("cuda", 0)
H, W, K = 1080, 1920, 240
for sb in (12, 24):
    seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=sb)
    clip = SyntheticClip(H, W, K, config_id=2, device=dev)
    clip.gen.manual_seed(20260002)
    frames = []
    for t in range(K):
        f = clip.frame(t)
        frames.append((f["frame"], None, f["bg"], clip.bg_always))
    for rep in range(3):
        pipe.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); pipe.run(iter(frames)); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"seg_batch {sb}: issued after {(t1 - t0) / K * 1e3:.3f} ms/frame, GPU done after {(t2 - t0) / K * 1e3:.3f} ms/frame ({K / (t2 - t0):.1f} fps)", flush=True)
    del pipe, seg_agent, frames
    torch.cuda.empty_cache()

# where does the host time go?  (cProfile of one run of the loop, seg_batch 12)
import cProfile, pstats, io
seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
clip = SyntheticClip(H, W, K, config_id=2, device=dev)
frames = []
for t in range(K):
    f = clip.frame(t)
    frames.append((f["frame"], None, f["bg"], clip.bg_always))
pipe.run(iter(frames)); torch.cuda.synchronize()
pr = cProfile.Profile()
pipe.reset(); torch.cuda.synchronize()
pr.enable(); pipe.run(iter(frames)); pr.disable(); torch.cuda.synchronize()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(22); print(st.getvalue()[:6000])
