"""Stage-entry merge A/B on the matting net: forward time and agreement of the two builds of the network."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
nets = {m: HipMattingNet(synthetic_state_dict(0), 544, 960, merge_stage_entry=m) for m in (False, True)}
x = torch.randn(544, 960, 8, device="cuda"); x[..., 7] = 0
outs = {}
for m, net in nets.items():
    net.x8.copy_(x)
    outs[m] = net().clone()
    for _ in range(10): net()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): net()
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    print("merge_stage_entry", m, "layers", net.num_layers, "forward %.1f us" % best)
d = (outs[True] - outs[False]).abs().max().item()
print("max |pred difference| %.3e (pred in [0, 1])" % d)
