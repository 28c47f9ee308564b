"""Time the green and bg-online frame loops at a given size (synthetic clips, synthetic weights)."""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.synth import SyntheticClip

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 12
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, pipe, clip, extra=lambda p: ""):
    frames = [clip.frame(t)["frame"] for t in range(n)]
    torch.cuda.synchronize()
    ts = []
    for f in frames:
        t = time.time()
        out = pipe.step(f)
        torch.cuda.synchronize()
        ts.append(time.time() - t)
    tr = ts[n // 2:]
    print(f"{name} {H}x{W}: first {ts[0]*1e3:.1f} ms, steady {sum(tr)/len(tr)*1e3:.1f} ms/frame ({len(tr)/sum(tr):.1f} fps) "
          f"tracking={pipe.tracking_flag} fg={(out['alpha']>127).float().mean().item():.3f} {extra(pipe)}", flush=True)


cfg = json.load(open(os.path.join(ROOT, "configs", "green.json")))
for k in ("binseg", "vmatting", "stm"):
    cfg[k]["model_path"] = "synthetic"
from video_unscreen_amd.pipeline_green import GreenPipeline
cfg.pop("stm")
run("green", GreenPipeline(cfg, H, W, cf_random_state=0), SyntheticClip(H, W, n, config_id=1, device="cuda"),
    lambda p: f"tracked={p.tracking_count}")

cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
for k in ("binseg", "vmatting", "stm"):
    cfg[k]["model_path"] = "synthetic"
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
run("bg-online", BgOnlinePipeline(cfg, H, W), SyntheticClip(H, W, n, config_id=2, device="cuda"), lambda p: f"fill_iters={p.fill_iters}")
