"""Does the hardware queue priority of the pipeline's streams change the frame rate?  Device-resident loop of bench.py's default
workload, one pipeline per (main, front, seg) priority triple (-1 = high, 0 = normal).   python tools/dev/stream_prio_probe.py"""
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
clip = SyntheticClip(H, W, K, config_id=2, device=dev)
clip.gen.manual_seed(20260002)
frames = []
for t in range(K):
    f = clip.frame(t)
    frames.append((f["frame"], None, f["bg"], clip.bg_always))
sd, ssd = synthetic_state_dict(0), schp_synthetic_state_dict(0)
for (pm, pf, ps) in ((0, 0, 0), (0, -1, 0), (0, 0, -1), (-1, 0, 0), (-1, 0, -1), (0, -1, -1), (0, 0, 0)):
    seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=ssd)
    pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=sd, seg_agent=seg_agent, seg_batch=12)
    pipe.front_priority, pipe.seg_priority = pf, ps
    main = torch.cuda.Stream(dev, priority=pm)
    best = 1e9
    with torch.cuda.stream(main):
        for _ in range(3):
            pipe.reset(); torch.cuda.synchronize()
            t0 = time.perf_counter(); pipe.run(iter(frames)); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
    print(f"priorities main {pm:2d} front {pf:2d} seg {ps:2d}: {K / best:7.1f} frames/s", flush=True)
    del pipe, seg_agent
    torch.cuda.empty_cache()
