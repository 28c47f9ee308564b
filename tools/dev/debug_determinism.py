import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.pipeline import BgStepPipeline, default_config
from video_unscreen_amd.synth import SyntheticClip
cfg = default_config()
H, W, n = 270, 480, 6
clip = SyntheticClip(H, W, n, config_id=2, device="cpu")
frames = [clip.frame(t) for t in range(n)]
bga = clip.bg_always.cuda()
ins = [(f["frame"].cuda(), f["seg"].cuda(), f["bg"].cuda(), bga) for f in frames]
def run_step():
    p = BgStepPipeline(cfg, H, W)
    out = []
    for i in ins:
        o = p.step(*i)
        out.append((o["alpha"].clone(), o["pred"].clone(), o["trimap"].clone()))
    return out
def run_pipe():
    p = BgStepPipeline(cfg, H, W)
    out = []
    p.run(iter(ins), on_result=lambda k, o: out.append((o["alpha"].clone(), o["pred"].clone(), o["trimap"].clone())))
    torch.cuda.synchronize()
    return out
a = run_step(); b = run_step(); c = run_pipe(); d = run_pipe()
def cmp(x, y, name):
    for k, ((a1, p1, t1), (a2, p2, t2)) in enumerate(zip(x, y)):
        print(name, k, "alpha ne", int((a1 != a2).sum()), "maxd", int((a1.int() - a2.int()).abs().max()), "pred maxd", float((p1 - p2).abs().max()), "tri ne", int((t1 != t2).sum()))
cmp(a, b, "step-step"); cmp(a, c, "step-pipe"); cmp(c, d, "pipe-pipe")
# net determinism alone
p = BgStepPipeline(cfg, H, W)
x = torch.randn(1, 7, 160, 288, device="cuda").contiguous(memory_format=torch.channels_last)
y0 = p.net(x).clone()
for i in range(5):
    print("net repeat maxdiff", float((p.net(x) - y0).abs().max()))
