"""Time the matting net's head layer (#53) and layer #22 for a given library build: USC_LIB=... python tools/dev/head_ab.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
if os.environ.get("USC_LIB"):
    nv.LIB_PATH = os.path.abspath(os.environ["USC_LIB"])
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet, ConvDesc
net = HipMattingNet(synthetic_state_dict(0), 544, 960)
lib = nv.load(); st = torch.cuda.current_stream().cuda_stream
net.x8.normal_(); ref = net().clone(); torch.cuda.synchronize()
out = []
for i in [int(a) for a in sys.argv[1:]] or [53, 22]:
    arr = (ConvDesc * 1)(net._descs[i])
    best = 1e9
    for rep in range(3):
        for _ in range(5): lib.usc_conv_sequence_f32(arr, 1, st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.usc_conv_sequence_f32(arr, 1, st)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    out.append("#%d %.1f us" % (i, best))
import zlib
print(os.path.basename(nv.LIB_PATH), " ".join(out), "pred crc %08x" % zlib.crc32(ref.cpu().numpy().tobytes()))
# the head layer against torch on its own input (float64), and run-to-run identity
d = net._descs[53]
x = None
for t in net._keep:
    if t.data_ptr() == d.inp:
        x = t
w = None
import torch.nn.functional as F
sd = synthetic_state_dict(0)
arr = (ConvDesc * 1)(d)
lib.usc_conv_sequence_f32(arr, 1, st); torch.cuda.synchronize()
a = net.pred.clone()
lib.usc_conv_sequence_f32(arr, 1, st); torch.cuda.synchronize()
print("run-to-run identical:", torch.equal(a, net.pred), "input found:", x is not None, "finite:", bool(torch.isfinite(a).all()), "mean %.6f" % float(a.mean()))
