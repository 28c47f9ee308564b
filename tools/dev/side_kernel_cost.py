"""What do the side-stream stages cost the frame loop?  Device-resident bench loop (same clip and pipeline as bench.py) with
(a) everything, (b) the object filter replaced by a plain copy, (c) also the trimap replaced by a constant map (the matting net then
sees other inputs: timing only), (d) no segmentation network (masks given).   python tools/dev/side_kernel_cost.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build_native()
from video_unscreen_amd import _native as nv
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.binseg import HumanSegAgent
from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
dev = torch.device("cuda", 0)
H, W, K = 1080, 1920, 240
seg_agent = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
pipe = BgStepPipeline(default_config(), H, W, device=dev, state_dict=synthetic_state_dict(0), seg_agent=seg_agent, seg_batch=12)
clip = SyntheticClip(H, W, K, config_id=2, device=dev)
clip.gen.manual_seed(20260002)
frames = []
for t in range(K):
    f = clip.frame(t)
    frames.append((f["frame"], None, f["bg"], clip.bg_always))

def timed(reps=3):
    best = 1e9
    for _ in range(reps):
        pipe.reset(); torch.cuda.synchronize()
        t0 = time.perf_counter(); pipe.run(iter(frames)); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best

def report(name):
    t = timed()
    print(f"{name:58s} {K / t:7.1f} frames/s  {t / K * 1e3:.3f} ms/frame", flush=True)

report("everything")
real_filter = pipe.dfilter
pipe.dfilter = lambda alpha, seg, center, s, c, src=None: alpha.copy_(src) if src is not None else alpha
report("object filter -> plain copy")
real_trimap = nv.trimap
tri_const = torch.full((H, W), 128, dtype=torch.uint8, device=dev)
nv.trimap = lambda a, L, k, it, out=None, ws=None: out.copy_(tri_const) if out is not None else tri_const
report("+ trimap -> constant")
nv.trimap = real_trimap
pipe.dfilter = real_filter
report("everything again")
