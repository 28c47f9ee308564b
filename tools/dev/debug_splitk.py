import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes as C
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting.net import synthetic_state_dict
from video_unscreen_amd.vmatting import net_hip
lib = nv.load()
net = net_hip.HipMattingNet(synthetic_state_dict(0), 64, 96)
net.x8.normal_()
net(); torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for i, d in enumerate(net._descs):
    if d.kind != 0: continue
    npix = d.OHf * d.OWf
    out = torch.from_dlpack  # noqa
    # run with slab
    arr = (net_hip.ConvDesc * 1)(d)
    lib.usc_conv_sequence_f32(arr, 1, st); torch.cuda.synchronize()
    import numpy as np
    buf = (C.c_float * (npix * d.Cout)).from_address(0) if False else None
    a = torch.empty(npix * d.Cout, dtype=torch.float32, device="cuda")
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(a.data_ptr()), C.c_void_p(d.out), C.c_size_t(npix * d.Cout * 4), 3)
    slab, sb = d.slab, d.slab_bytes
    d.slab, d.slab_bytes = None, 0
    arr = (net_hip.ConvDesc * 1)(d)
    lib.usc_conv_sequence_f32(arr, 1, st); torch.cuda.synchronize()
    b = torch.empty_like(a)
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(b.data_ptr()), C.c_void_p(d.out), C.c_size_t(npix * d.Cout * 4), 3)
    d.slab, d.slab_bytes = slab, sb
    diff = float((a - b).abs().max())
    nbad = int(((a - b).abs() > 1e-4).sum())
    print(f"layer {i:2d} M={d.OH*d.OW:5d} Cin={d.Cin:3d} Cout={d.Cout:3d} taps={d.ntaps} nphase={d.nphase} res1={bool(d.res1)} res2={bool(d.res2)} aff={bool(d.scale)} act={d.act}: maxdiff {diff:.3e} bad {nbad}/{a.numel()}")
    if nbad:
        idx = torch.nonzero((a - b).abs() > 1e-4)[:8, 0].tolist()
        print("   first bad flat idx", idx, "pix", [k // d.Cout for k in idx], "ch", [k % d.Cout for k in idx])
