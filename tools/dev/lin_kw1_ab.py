"""A/B timing of the layers that the planner runs on conv_wino_kernel<KW = 1, *, linear blocks> (tools/dev/plan_dump.py: one layer of the
human-parsing net at 4 frames per forward, seven of the interactive-segmentation net) between library builds on the same box:
   python tools/dev/lin_kw1_ab.py libA.so libB.so
Each library is loaded in its own process; HIP events around 50 launches of each layer; output checked against build A's bits."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(64, 128, 237, 237, 4), (256, 256, 68, 120, 2), (512, 512, 68, 120, 2)]        # cin, cout, h, w, batch
CHILD = r"""
import os, sys, zlib
sys.path.insert(0, %r)
import torch
from video_unscreen_amd import _native as nv
nv.LIB_PATH = sys.argv[1]
from video_unscreen_amd.vmatting import net_hip as m
lib = nv.load()
out_line = []
slab = torch.empty(64 * 1024 * 1024, device="cuda")
for (cin, cout, h, w, b) in %r:
    g = torch.Generator(device="cuda").manual_seed(cin + h)
    x = torch.randn(b * h, w, cin, device="cuda", generator=g); wt = torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(cout)) * 0.05
    out = torch.empty(b * h, w, cout, device="cuda")
    taps, wk = m.conv_taps(wt, 1)
    wp = m.pack_k_major(wk.float()).cuda(); wn = m.pack_wino(wt).cuda()
    d = m.ConvDesc()
    d.inp, d.out, d.wpk, d.wino = x.data_ptr(), out.data_ptr(), wp.data_ptr(), wn.data_ptr()
    d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, b * h, w, b * h, w, cout
    d.batch, d.plan_batch = b, b
    d.slab, d.slab_bytes = slab.data_ptr(), slab.numel() * 4
    d.in_stride, d.out_stride, d.ntaps, d.nphase = 1, 1, 9, 1
    for i, (dy, dx) in enumerate(taps): d.dy[i], d.dx[i] = dy, dx
    arr = (m.ConvDesc * 1)(d)
    plan = nv.conv_plans(arr, 1)[0]["plan"]
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(10): lib.usc_conv_sequence_f32(arr, 1, st)
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.usc_conv_sequence_f32(arr, 1, st)
        e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
    out_line.append("%%s %%6.1f us crc %%08x" %% (plan, best, zlib.crc32(out.cpu().numpy().tobytes())))
print(" | ".join(out_line))
"""
print("shapes (cin, cout, h, w, batch):", SHAPES)
for lib in sys.argv[1:]:
    for rep in range(2):
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, SHAPES), os.path.abspath(lib)], capture_output=True, text=True)
        print(f"{os.path.basename(lib):32s} " + (r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-800:]), flush=True)
