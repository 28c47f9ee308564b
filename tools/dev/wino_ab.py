"""A/B timing of Winograd layers between library builds on the same box: python tools/dev/wino_ab.py libA.so libB.so ...
Each library is loaded in its own process (one per build); HIP events around 50 launches of each layer shape."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(8, 32, 544, 960, 1), (32, 32, 544, 960, 1), (32, 32, 272, 480, 1), (64, 32, 272, 480, 1), (64, 64, 136, 240, 2), (128, 64, 136, 240, 2),
          (128, 128, 68, 120, 2), (256, 256, 34, 60, 4)]
CHILD = r"""
import os, sys
sys.path.insert(0, %r)
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = sys.argv[1]
from video_unscreen_amd.vmatting import net_hip as m
lib = nv.load()
out_line = []
for (cin, cout, h, w, kw) in %r:
    x = torch.randn(h, w, cin, device="cuda"); wt = torch.randn(cout, cin, 3, 3) * 0.05
    out = torch.empty(h, w, cout, device="cuda")
    taps, wk = m.conv_taps(wt, 1)
    wp = m.pack_k_major(wk.float()).cuda(); wn = m.pack_wino(wt).cuda()
    d = m.ConvDesc()
    d.inp, d.out, d.wpk, d.wino = x.data_ptr(), out.data_ptr(), wp.data_ptr(), wn.data_ptr()
    d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, h, w, h, w, cout
    d.in_stride, d.out_stride, d.ntaps, d.nphase = 1, 1, 9, 1
    for i, (dy, dx) in enumerate(taps): d.dy[i], d.dx[i] = dy, dx
    lib.usc_conv_force_winograd(1); lib.usc_conv_force_plan(0, 0, 0, kw)
    arr = (m.ConvDesc * 1)(d)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10): lib.usc_conv_sequence_f32(arr, 1, st)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.usc_conv_sequence_f32(arr, 1, st)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    out_line.append("%%6.1f" %% best)
print(" ".join(out_line))
"""
print("shapes (cin, cout, h, w, kw):", SHAPES)
for lib in sys.argv[1:]:
    for rep in range(2):
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, SHAPES), os.path.abspath(lib)], capture_output=True, text=True)
        print(f"{os.path.basename(lib):40s} us: " + (r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-500:]), flush=True)
