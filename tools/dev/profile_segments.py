"""Time every segment (conv graph / glue callable) of the batched human-parsing net."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
B = int(os.environ.get("B", "4"))
net = HipSchpNet(schp_synthetic_state_dict(0), 473, 473, batch=B, plan_batch=B)
net.x4.normal_(); net.run(); torch.cuda.synchronize()
lib = net.nv.load(); st = torch.cuda.current_stream().cuda_stream
tot = 0
for i, (kind, a, arr) in enumerate(net._segments):
    fn = (lambda: lib.usc_conv_graph_launch(a, st)) if kind == "conv" else a
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 10
    tot += t
    print(f"segment {i:2d} {kind:5s} {('%d layers' % len(arr)) if kind == 'conv' else '':12s} {t*1e3:9.1f} us")
print(f"sum {tot:.3f} ms")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): net.run()
e1.record(); e1.synchronize()
print(f"whole forward {e0.elapsed_time(e1)/10:.3f} ms for {B} frames")
