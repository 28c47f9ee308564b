"""Per-phase shader-clock stamps of conv_wino_kernel's K loop (library built with EXTRA=-DUSC_WINO_TIMING2): for the h = 0 and the
h = 1 waves, when each fragment phase and each MFMA phase of the first steps starts and ends, relative to the wave's own entry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = os.environ["USC_TIMING_LIB"]
from video_unscreen_amd.vmatting import net_hip as m
lib = nv.load()
T3 = os.environ.get("T3", "0") == "1"          # the transposed form
if T3: lib.usc_conv_force_winograd_form(1, -1, 0)
cases = ((64, 64, 136, 240, 1), (32, 32, 272, 480, 1)) if T3 else ((32, 32, 272, 480, 1), (64, 64, 136, 240, 2), (128, 128, 68, 120, 4), (256, 256, 34, 60, 4))
for (cin, cout, h, w, kw) in cases:
    x = torch.randn(h, w, cin, device="cuda")
    ts = torch.zeros(4096 * 8 * 16, dtype=torch.int64, device="cuda")
    d = m.ConvDesc()
    if T3:
        wt = torch.randn(cin, cout, 4, 4) * 0.05
        out = torch.empty(2 * h, 2 * w, cout, device="cuda")
        phases = [m.convt_phase_taps(wt, py, px) for py in (0, 1) for px in (0, 1)]
        wp = torch.stack([m.pack_k_major(k.float()) for _, k in phases], 0).cuda(); wn = m.pack_wino_t(wt).cuda()
        taps = [t for tp, _ in phases for t in tp]
        d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, h, w, 2 * h, 2 * w, cout
        d.in_stride, d.out_stride, d.ntaps, d.nphase, d.wphase_stride = 1, 2, 4, 4, wp[0].numel()
    else:
        wt = torch.randn(cout, cin, 3, 3) * 0.05
        out = torch.empty(h, w, cout, device="cuda")
        taps, wk = m.conv_taps(wt, 1)
        wp = m.pack_k_major(wk.float()).cuda(); wn = m.pack_wino(wt).cuda()
        d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, h, w, h, w, cout
        d.in_stride, d.out_stride, d.ntaps, d.nphase = 1, 1, 9, 1
    d.inp, d.out, d.wpk, d.wino = x.data_ptr(), out.data_ptr(), wp.data_ptr(), wn.data_ptr()
    d.slab, d.slab_bytes = ts.data_ptr(), 0
    for i, (dy, dx) in enumerate(taps): d.dy[i], d.dx[i] = dy, dx
    lib.usc_conv_force_winograd(1); lib.usc_conv_force_plan(0, 0, 0, kw)
    arr = (m.ConvDesc * 1)(d)
    plan = nv.conv_plans(arr, 1)[0]
    for _ in range(3): lib.usc_conv_sequence_f32(arr, 1, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); ts.zero_(); torch.cuda.synchronize()
    lib.usc_conv_sequence_f32(arr, 1, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    lib.usc_conv_force_winograd(-1); lib.usc_conv_force_plan(0, 0, 0, 0)
    t = ts.cpu().view(-1, 16)[: plan["workgroups"] * 8].double()
    t = t[t[:, 15] != 0]
    hh = torch.arange(len(t)) % 8 >= 4
    print(f"cin {cin} cout {cout} {h}x{w} KW {kw}: {plan['workgroups']} WGs; cycles since the wave's entry (mean over waves)")
    names = {0: "entry", 1: "prologue done", 14: "K loop done", 15: "end"}
    for nm, sel in (("h=0", ~hh), ("h=1", hh)):
        r = (t[sel] - t[sel][:, :1]).mean(0)
        prev = 0.0
        line = []
        for k in range(16):
            if t[sel][:, k].min() == 0: continue
            line.append(f"[{k}] {float(r[k]):.0f} (+{float(r[k]) - prev:.0f})")
            prev = float(r[k])
        print("  " + nm + ": " + "  ".join(line))
