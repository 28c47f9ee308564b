"""A/B of two builds of the library on the networks of the headline loop: python tools/dev/ab_nets.py <lib.so> [<lib.so> ...]
-> us per matting forward (544 x 960), ms per human-parsing forward at 12 and at 4 frames, best of three blocks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import subprocess
if len(sys.argv) > 2:                                   # one process per library (a library is loaded once per process)
    for lib in sys.argv[1:]:
        subprocess.run([sys.executable, os.path.abspath(__file__), lib], check=True)
    sys.exit(0)
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = os.path.abspath(sys.argv[1])
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict


def timed(net, n):
    for _ in range(5):
        net()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            net()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


# clock spin-up
m = HipMattingNet(synthetic_state_dict(0), 544, 960)
m.x8.normal_()
for _ in range(300):
    m()
torch.cuda.synchronize()
tm = timed(m, 60)
ssd = schp_synthetic_state_dict(0)
res = []
for b in (12, 4):
    s = HipSchpNet(ssd, 473, 473, batch=b, plan_batch=b)
    s.x4.normal_()
    res.append(timed(s, 10))
    s.close()
print(f"{os.path.basename(sys.argv[1]):36s} matting {tm * 1e3:7.1f} us   parsing b12 {res[0]:7.3f} ms ({res[0] / 12:.4f} per frame)   b4 {res[1]:7.3f} ms ({res[1] / 4:.4f} per frame)", flush=True)
