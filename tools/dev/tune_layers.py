"""Time every admissible (tile width, row groups, K groups) plan of chosen layers of the matting net (usc_conv_force_plan, no grid split):
   python tools/dev/tune_layers.py 7 11 15"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting.net_hip import HipMattingNet, ConvDesc
net = HipMattingNet(synthetic_state_dict(0), 544, 960)
lib = nv.load(); st = torch.cuda.current_stream().cuda_stream
net.x8.normal_(); net(); torch.cuda.synchronize()


def t_layer(d, reps=30):
    arr = (ConvDesc * 1)(d)
    for _ in range(3):
        if lib.usc_conv_sequence_f32(arr, 1, st) != 0:
            return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        lib.usc_conv_sequence_f32(arr, 1, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for i in [int(a) for a in sys.argv[1:]]:
    d = net._descs[i]
    lib.usc_conv_force_plan(0, 0, 0, 0)
    base = t_layer(d)
    plan = nv.conv_plans((ConvDesc * 1)(d), 1)[0]["plan"]
    rows = []
    for nt, rg in ((1, 4), (2, 4), (1, 2), (1, 1)):
        if nt == 2 and d.Cout % 64:
            continue
        for kw in (1, 2, 4, 8, 16):
            if (nt == 2 and kw > 2) or (rg == 4 and kw > 4) or (rg < 4 and not 4 <= rg * kw <= 16):
                continue
            lib.usc_conv_force_plan(nt, rg, 1, kw)
            t = t_layer(d)
            if t is not None:
                rows.append((round(t, 1), nt, rg, kw))
    lib.usc_conv_force_plan(0, 0, 0, 0)
    print(f"layer {i}: M={d.OH * d.OW} Cin={d.Cin} Cout={d.Cout} taps={d.ntaps} planner ({plan}) {base:.1f} us; plans (us, NT, RG, KW): {sorted(rows)[:6]}")
