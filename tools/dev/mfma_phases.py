"""Where a wave of conv_mfma_kernel spends its time (library built with EXTRA=-DUSC_MFMA_TIMING, path in USC_TIMING_LIB): per wave
the shader-clock and 100 MHz real-time stamps at entry / tables done / first chunk staged / K loop done / end, plus its hardware
id (XCC, SE, CU, SIMD).  Shapes: the human-parsing net's layer3 1x1 layers at 12 frames per forward and two matting-net direct
layers.  Prints per shape: the planner's plan, the launch's span, per-phase cycle statistics, how many waves share a SIMD over
time, and what fraction of the span's SIMD-cycles the K loops' MFMAs could have filled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = os.environ["USC_TIMING_LIB"]
from video_unscreen_amd.vmatting import net_hip as m
lib = nv.load()
st = lambda: torch.cuda.current_stream().cuda_stream

# (name, images, h, w, cin, cout, ksize, residual, act)
CASES = [("schp layer3 reduce 1x1 1024->256 (b12)", 12, 30, 30, 1024, 256, 1, False, 1),
         ("schp layer3 expand 1x1 256->1024 + res + relu (b12)", 12, 30, 30, 256, 1024, 1, True, 1),
         ("schp layer2 1x1 128->512 (b12)", 12, 60, 60, 128, 512, 1, True, 1),
         ("schp 1x1 512->2048 (b12)", 12, 30, 30, 512, 2048, 1, True, 1),
         ("matting #16 3x3 512->512 at 17x30", 1, 17, 30, 512, 512, 3, False, 2),
         ("matting #7 3x3 s1-equivalent 64->256 at 68x120", 1, 68, 120, 64, 256, 3, False, 2)]
sel = [int(v) for v in sys.argv[1:]] or range(len(CASES))
for ci in sel:
    name, nb, h, w, cin, cout, ks, has_res, act = CASES[ci]
    x = torch.randn(nb * h, w, cin, device="cuda")
    out = torch.empty(nb * h, w, cout, device="cuda")
    res = torch.randn(nb * h, w, cout, device="cuda") if has_res else None
    wt = torch.randn(cout, cin, ks, ks) * 0.02
    taps, wk = m.conv_taps(wt, ks // 2)
    wp = m.pack_k_major(wk.float()).cuda()
    bias = torch.randn(cout, device="cuda")
    ts = torch.zeros(1 << 22, dtype=torch.int64, device="cuda")
    d = m.ConvDesc()
    d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, nb * h, w, nb * h, w, cout
    d.in_stride, d.out_stride, d.ntaps, d.nphase, d.act = 1, 1, len(taps), 1, act
    d.batch, d.plan_batch = nb, nb
    d.inp, d.out, d.wpk, d.bias = x.data_ptr(), out.data_ptr(), wp.data_ptr(), bias.data_ptr()
    if res is not None:
        d.res1 = res.data_ptr()
    d.slab, d.slab_bytes = ts.data_ptr(), 0
    for i, (dy, dx) in enumerate(taps):
        d.dy[i], d.dx[i] = dy, dx
    lib.usc_conv_force_winograd(0)
    arr = (m.ConvDesc * 1)(d)
    plan = nv.conv_plans(arr, 1)[0]
    for _ in range(5):
        lib.usc_conv_sequence_f32(arr, 1, st())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        lib.usc_conv_sequence_f32(arr, 1, st())
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    ts.zero_(); torch.cuda.synchronize()
    lib.usc_conv_sequence_f32(arr, 1, st()); torch.cuda.synchronize()
    lib.usc_conv_force_winograd(-1)
    t = ts.cpu().numpy().reshape(-1, 16)
    t = t[t[:, 0] != 0]
    clk, rt, hw = t[:, 0:10:2].astype(np.float64), t[:, 1:10:2].astype(np.float64), t[:, 15]
    gflop = 2.0 * nb * h * w * cout * len(taps) * cin / 1e9
    span_rt = (rt[:, 4].max() - rt[:, 0].min()) / 100.0                       # us (100 MHz)
    ghz = np.median((clk[:, 4] - clk[:, 0]) / np.maximum(rt[:, 4] - rt[:, 0], 1)) * 0.1
    print(f"== {name}: plan {plan}, {len(t)} waves, {us:.1f} us per launch (HIP events, instrumented build), {gflop / us * 1e3:.1f} TFLOP/s; "
          f"first entry -> last end {span_rt:.1f} us; shader clock ~{ghz:.2f} GHz")
    ph = np.diff(clk, axis=1)
    for k, nm in enumerate(("tables + barrier", "first loads + stage", "K loop", "epilogue")):
        v = ph[:, k]
        print(f"   {nm:20s} mean {v.mean():8.0f}  p10 {np.percentile(v, 10):8.0f}  p50 {np.percentile(v, 50):8.0f}  p90 {np.percentile(v, 90):8.0f} cycles")
    tot = clk[:, 4] - clk[:, 0]
    print(f"   {'wave lifetime':20s} mean {tot.mean():8.0f}  p10 {np.percentile(tot, 10):8.0f}  p50 {np.percentile(tot, 50):8.0f}  p90 {np.percentile(tot, 90):8.0f} cycles")
    # K steps and ideal MFMA cycles per wave (from the plan): K loop cycles if the wave had the matrix pipe to itself = MFMAs x 64
    kchunks = (len(taps) * cin + 31) // 32
    nt, _rg, _s, kw = (int(v) for v in plan["plan"].split(","))
    mf = kchunks / max(kw, 1) * 16 * nt * 64
    print(f"   ideal MFMA cycles per wave {mf:.0f} (K loop mean / ideal = {ph[:, 2].mean() / mf:.2f}: waves sharing the SIMD's pipe)")
    # occupancy over time per SIMD: sample the real-time axis
    simd = ((hw >> 32) & 0xf) * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 8) & 0xf) * 100 + ((hw >> 4) & 3)
    t0 = rt[:, 0].min()
    grid = np.linspace(0, rt[:, 4].max() - t0, 41)[1:-1]
    occ, busy = [], []
    nsimd = len(np.unique(simd))
    for g in grid:
        live = (rt[:, 0] - t0 <= g) & (rt[:, 4] - t0 > g)
        inloop = (rt[:, 2] - t0 <= g) & (rt[:, 3] - t0 > g)
        occ.append(live.sum() / 1024.0)
        busy.append(len(np.unique(simd[inloop])) / 1024.0)
    print(f"   distinct SIMDs seen: {nsimd}; waves resident per SIMD over the span (5 % steps): " + " ".join(f"{o:.1f}" for o in occ[::2]))
    print(f"   fraction of the 1024 SIMDs with at least one wave inside its K loop:           " + " ".join(f"{b:.2f}" for b in busy[::2]))
    total_mfma_cycles = mf * len(t)
    print(f"   sum of ideal MFMA cycles / (1024 SIMDs x span) = {total_mfma_cycles / (1024 * span_rt * ghz * 1e3):.2f}")
