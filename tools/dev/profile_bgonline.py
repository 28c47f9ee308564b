"""Per-stage timing of one tracked bg-online frame at 1080p (synchronising between stages)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.pipeline_bg import BgOnlinePipeline
from video_unscreen_amd.synth import SyntheticClip
H, W = 1080, 1920
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = json.load(open(os.path.join(ROOT, "configs", "bg.json")))
for k in ("binseg", "vmatting", "stm"):
    cfg[k]["model_path"] = "synthetic"
pipe = BgOnlinePipeline(cfg, H, W)
clip = SyntheticClip(H, W, 6, config_id=2, device="cuda")
frames = [clip.frame(t)["frame"] for t in range(6)]
for f in frames[:4]:
    out = pipe.step(f)
torch.cuda.synchronize()

def timed(name, fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.time()-t)/n*1e3:8.2f} ms", flush=True)
    return r

seed = torch.where(pipe.alpha_pre >= 128, torch.full_like(pipe.alpha_pre, 255), pipe.alpha_pre)
seg = timed("stm (memorize+segment)", lambda: pipe.stmagent.forward_device([frames[3], frames[4]], seed)[-1])
aor = timed("object filter (host)", lambda: pipe._object_filter(seg))
tri = timed("trimap", lambda: pipe.trimapagent.forward_device(aor))
a1 = timed("matting net", lambda: pipe.vmatagent.forward_device(frames[4], pipe.alpha_pre, tri))
bg = timed("get_bg", lambda: nv.get_bg(a1, frames[4]))
hole = timed("hole (thr + dilate)", lambda: nv.morph((a1 > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE))
bgimg = timed("regionfill", lambda: nv.regionfill(bg, hole, ws=pipe._ws_fill))
print("   hole fraction %.3f, iterations %d" % (float((hole > 0).float().mean()), nv.regionfill(bg, hole, ws=pipe._ws_fill, want_iters=True)[1]), flush=True)
timed("bgdiff", lambda: nv.bgdiff_mask(frames[4], bgimg, None, a1, 1.0, 25, ws=pipe._ws_diff))
timed("get_fg", lambda: nv.get_fg(frames[4], a1, bgimg, paste_thr=1, want_bg=True))
timed("exist_foreground", lambda: pipe._exist_foreground(a1))
timed("whole step", lambda: pipe.step(frames[5]))
net = pipe.stmagent.model_for(*pipe.stmagent.input_size(H, W))
timed("  stm.memorize graph", lambda: net.mem.run())
timed("  stm.query graph", lambda: net.qry.run())
timed("  stm.read(T=1)", lambda: net._read_for(1).run())
timed("  stm.decoder graph", lambda: net.dec.run())
