import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
cfg = default_config()
H, W, n = 270, 480, 5
clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
frames = [clip.frame(t) for t in range(n)]
bga = clip.bg_always.cuda()
ins = [(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga) for f in frames]
keys = ("gate", "alphaor", "trimap", "alpha", "fg", "pred", "bg")
res = {}
for mode in ("host", "device"):
    pipe = BgStepPipeline(cfg, H, W, object_filter=mode)
    res[mode, "step"] = []
    for i in ins:
        o = pipe.step(*i)
        res[mode, "step"].append(tuple(o[k].clone() for k in keys))
    for rep in range(2):
        pipe = BgStepPipeline(cfg, H, W, object_filter=mode)
        got = []
        pipe.run(iter(ins), on_result=lambda k, o: got.append(tuple(o[key].clone() for key in keys)))
        torch.cuda.synchronize()
        res[mode, "run%d" % rep] = got
ref = res["host", "step"]
for key, got in res.items():
    for t, (a, b) in enumerate(zip(ref, got)):
        d = [int((x != y).sum()) for x, y in zip(a, b)]
        if any(d): print(key, "frame", t, dict(zip(keys, d)), "max |dalpha|", int((a[3].int() - b[3].int()).abs().max()))
print("done")
