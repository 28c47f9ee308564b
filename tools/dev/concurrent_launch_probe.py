"""What do the barrier-free launches of independent layers (usc_conv_desc.concurrent_with_prev) buy a matting forward?  The product's
launch path (usc_conv_graph_launch, stream mode) against the same layers strictly one after the other (usc_conv_sequence_f32), and the
result of both against each other (must be identical bits).   python tools/dev/concurrent_launch_probe.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet
net = HipMattingNet(synthetic_state_dict(0), 544, 960)
net.x8.copy_(torch.randn_like(net.x8) * 0.5)
n_conc = sum(d.concurrent_with_prev for d in net._descs)


def t(fn, n=60):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n


a = net().clone()
b = net.forward_serial().clone()
print("layers", net.num_layers, "of which launched without the in-order barrier:", n_conc)
print("identical results:", bool(torch.equal(a, b)), "finite:", bool(torch.isfinite(a).all()))
for r in range(3):
    print("round %d: product launch path %.4f ms   strictly serial %.4f ms" % (r, t(lambda: net()), t(lambda: net.forward_serial())))
for r in range(20):                      # the same input must give the same bits on every run, whatever overlaps
    assert torch.equal(net(), a), r
print("20 repeated forwards bit-identical")
