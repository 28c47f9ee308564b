"""Per-stage timing of one tracked green-mode frame (synchronising between stages)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.pipeline_green import GreenPipeline
from video_unscreen_amd.synth import SyntheticClip
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = json.load(open(os.path.join(ROOT, "configs", "green.json")))
for k in ("binseg", "vmatting", "stm"):
    if k in cfg:
        cfg[k]["model_path"] = "synthetic"
cfg.pop("stm", None)
pipe = GreenPipeline(cfg, H, W, cf_random_state=0)
clip = SyntheticClip(H, W, 8, config_id=1, device="cuda")
frames = [clip.frame(t)["frame"] for t in range(8)]
for f in frames[:5]:
    out = pipe.step(f)
torch.cuda.synchronize()
print("tracking", pipe.tracking_flag, "fg", float((out["alpha"] > 127).float().mean()))


def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:34s} {(time.time()-t)/n*1e3:8.2f} ms", flush=True)
    return r


f = frames[5]
seg = pipe.alpha_pre.clone()
acf, bgimg, _ = timed("colour filter inference (iters=0)", lambda: pipe.cfagent.forward_device(f, seg, iters=0))
bgc = pipe.cfagent.last_bg_color
aor = timed("object filter (device)", lambda: pipe.filter(acf.clone(), None, pipe.center, cfg["objectremoval"]["saliency_thr"], cfg["objectremoval"]["consensus_thr"]))
tri = timed("trimap_withbg", lambda: pipe.trimapagent.forward_device_withbg(aor, f, bgc))
am = timed("matting net", lambda: pipe.vmatagent.forward_device(f, pipe.alpha_pre, tri))
a = timed("color_correct", lambda: nv.color_correct(f, am, bgc))
timed("get_fg", lambda: nv.get_fg(f, a, bgimg.contiguous(), paste_thr=128, want_bg=True))
timed("exist_foreground", lambda: pipe._exist_foreground(a))
timed("whole step", lambda: pipe.step(frames[6]))
