"""Per-layer timing of the HIP human-parsing net."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
if os.environ.get('USC_LIB'): nv.LIB_PATH = os.path.abspath(os.environ['USC_LIB'])      # (A/B runs against another build of the library)
from video_unscreen_amd.binseg.schp_net import HipSchpNet, schp_synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import ConvDesc
B = int(os.environ.get('B', '1'))
net = HipSchpNet(schp_synthetic_state_dict(0), 473, 473, batch=B, plan_batch=int(os.environ.get('PLAN', '1')))
lib = nv.load(); st = torch.cuda.current_stream().cuda_stream
net.x4.normal_(); net(); torch.cuda.synchronize()
groups = collections.OrderedDict()
tot_t = tot_f = 0
idx = 0
for kind, a, arr in net._segments:
    if kind != "conv": continue
    for d in arr:
        one = (ConvDesc * 1)(d)
        for _ in range(3): lib.usc_conv_sequence_f32(one, 1, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): lib.usc_conv_sequence_f32(one, 1, st)
        e1.record(); e1.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        M = d.OH * d.OW; fl = 2.0 * M * d.Cout * d.ntaps * d.Cin      # d.OH counts all images of the batch
        key = (M, d.Cin, d.Cout, d.ntaps, d.in_stride)
        g = groups.setdefault(key, [0, 0.0, 0.0]); g[0] += 1; g[1] += t; g[2] += fl
        tot_t += t; tot_f += fl; idx += 1
print(f"{'M':>7} {'Cin':>5} {'Cout':>5} taps s  count   us/layer   total ms   GF/layer  TF/s")
for (M, ci, co, nt, s), (c, t, f) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    print(f"{M:7d} {ci:5d} {co:5d} {nt:4d} {s}  {c:5d}  {t/c*1e6:9.1f}  {t*1e3:9.3f}  {f/c/1e9:8.2f}  {f/t/1e12:6.1f}")
print(f"{idx} conv layers, sum {tot_t*1e3:.3f} ms, {tot_f/1e9:.1f} GF, {tot_f/tot_t/1e12:.1f} TF/s")
