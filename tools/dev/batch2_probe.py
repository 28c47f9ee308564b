"""How much of the matting forward is launch granularity?  Two independent images through ONE set of launches (usc_conv_desc.batch = 2, planned
as for one image: bit-identical per-image results; the five four-phase transposed layers and the head still run once per image) against
two single-image forwards.  VERDICT round 3, item 6 asked for this number before any pipeline work:  python tools/dev/batch2_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
H, W = 544, 960
sd = synthetic_state_dict(0)
n1, n2 = HipMattingNet(sd, H, W), HipMattingNet(sd, H, W, batch=2)
g = torch.Generator(device="cuda").manual_seed(1)
xa, xb = (torch.randn(H, W, 8, device="cuda", generator=g) for _ in range(2))
xa[..., 7] = 0; xb[..., 7] = 0
pa = n1(xa).clone(); pb = n1(xb).clone()
p2 = n2(torch.cat([xa, xb], 0)).clone()
print("batch-2 forward == two single forwards, bit for bit:", torch.equal(p2[0], pa[0]) and torch.equal(p2[1], pb[0]))


def timed(net, n=50):
    for _ in range(10): net()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): net()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


t1, t2 = timed(n1), timed(n2)
print("one image per forward: %.1f us (%d launches); two images per forward: %.1f us (%d launches) = %.1f us per image (%.1f %% less)"
      % (t1, n1.num_layers, t2, n2.num_layers, t2 / 2, 100 * (1 - t2 / 2 / t1)))
