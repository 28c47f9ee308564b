import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
sd = synthetic_state_dict(0)
a = HipMattingNet(sd, 544, 960, use_graph=False)
b = HipMattingNet(sd, 544, 960, use_graph=True)
x = torch.randn(544, 960, 8, device="cuda"); x[..., 7] = 0
ya = a(x).clone(); yb = b(x).clone(); torch.cuda.synchronize()
print("graph == sequence:", torch.equal(ya, yb), float((ya - yb).abs().max()))
def timeit(net, it=30):
    for _ in range(5): net()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): net()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / it
for _ in range(3):
    print("sequence %.3f ms   graph %.3f ms" % (timeit(a), timeit(b)))
