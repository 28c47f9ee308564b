"""In-kernel phase timing of conv_wino_kernel (library built with EXTRA=-DUSC_WINO_TIMING into a scratch copy): per wave shader-clock
and 100 MHz real-time stamps at entry / after the prologue / after the K loop / at the end."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = os.environ["USC_TIMING_LIB"]
from video_unscreen_amd.vmatting import net_hip as m
lib = nv.load()
T3 = os.environ.get("T3", "0") == "1"          # the transposed form (ConvTranspose2d k4 s2 p1) on the matting net's decoder shapes
if T3: lib.usc_conv_force_winograd_form(1, -1, 0)
cases = ((128, 128, 68, 120, 1), (64, 64, 136, 240, 1), (32, 32, 272, 480, 1)) if T3 else \
        ((32, 32, 272, 480, 1), (64, 32, 272, 480, 1), (64, 64, 136, 240, 2), (128, 128, 68, 120, 4), (32, 32, 544, 960, 1))
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); lib.usc_conv_sequence_f32(arr, 1, torch.cuda.current_stream().cuda_stream); e1.record(); torch.cuda.synchronize()
    lib.usc_conv_force_winograd(-1); lib.usc_conv_force_plan(0, 0, 0, 0)
    t = ts.cpu().view(-1, 8, 2)[: plan["workgroups"] * 8]          # 8 waves per workgroup
    live = t[:, 6, 0] != 0
    t = t[live].double()
    cyc, rt = t[..., 0], t[..., 1]
    t0 = rt[:, 0].min()
    steps = cin // 8 // kw
    print(f"cin {cin} cout {cout} {h}x{w} KW {kw}: {plan['workgroups']} WGs, event {e0.elapsed_time(e1)*1e3:.1f} us; waves {len(t)}")
    print(f"   shader clock / real time (100 MHz): {((cyc[:,6]-cyc[:,0]) / (rt[:,6]-rt[:,0])).mean()*100:.0f} MHz")
    print(f"   cycles: prologue {float((cyc[:,1]-cyc[:,0]).mean()):.0f}  K loop {float((cyc[:,2]-cyc[:,1]).mean()):.0f} ({float((cyc[:,2]-cyc[:,1]).mean())/steps:.0f} per step of 2 x 32 MFMAs on the SIMD, ideal 4096)  epilogue {float((cyc[:,6]-cyc[:,2]).mean()):.0f}")
    hh = torch.arange(len(cyc)) % 8 >= 4
    for nm, sel in (("h=0 waves", ~hh), ("h=1 waves", hh)):
        c = cyc[sel]
        print(f"   {nm}: K loop {float((c[:,2]-c[:,1]).mean()):.0f}; output transform {float((c[:,3]-c[:,2]).mean()):.0f}; residual requests {float((c[:,4]-c[:,3]).mean()):.0f}; round 0 {float((c[:,5]-c[:,4]).mean()):.0f}; rounds 1-3 {float((c[:,6]-c[:,5]).mean()):.0f}")
    print(f"   wave start spread {float(rt[:,0].max()-t0)/100:.2f} us, first end {float(rt[:,6].min()-t0)/100:.2f} us, last end {float(rt[:,6].max()-t0)/100:.2f} us")
