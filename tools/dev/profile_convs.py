"""Per-layer timing of the HIP matting net: us, GFLOP, TFLOP/s, grid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet, ConvDesc

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (544, 960)
net = HipMattingNet(synthetic_state_dict(0), H, W)
lib = nv.load()
st = torch.cuda.current_stream().cuda_stream
net.x8.normal_()
net()
torch.cuda.synchronize()
tot_t = tot_f = 0
rows = []
for i, d in enumerate(net._descs):
    arr = (ConvDesc * 1)(d)
    for _ in range(3):
        lib.usc_conv_sequence_f32(arr, 1, st)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.usc_conv_sequence_f32(arr, 1, st)
    e1.record(); e1.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    M = d.OH * d.OW
    fl = 2.0 * M * d.Cout * d.ntaps * d.Cin * max(d.nphase, 1)
    tot_t += t; tot_f += fl
    mb = (M + 127) // 128
    print(f"{i:3d} M={M:7d} Cin={d.Cin:4d} Cout={d.Cout:4d} taps={d.ntaps:2d} s={d.in_stride} os={d.out_stride} "
          f"blocks={mb}x{max(d.Cout // 32, 1)} {t*1e6:8.1f} us {fl/1e9:6.2f} GF {fl/t/1e12:6.1f} TF/s")
print(f"sum of layers {tot_t*1e3:.3f} ms, {tot_f/1e9:.1f} GF, {tot_f/tot_t/1e12:.1f} TF/s")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    net()
e1.record(); e1.synchronize()
print(f"whole forward {e0.elapsed_time(e1)/20:.3f} ms")
