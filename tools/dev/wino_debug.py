"""Where does conv_wino_kernel differ from torch?  Error maxima per block of 8 x 4 tiles, per output phase, per channel block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from video_unscreen_amd import _native as nv
from video_unscreen_amd.vmatting import net_hip as m

lib = nv.load()
for (cin, cout, h, w, kw, mode) in ((8, 32, 17, 30, 1, "rand"), (8, 32, 16, 32, 1, "center"), (32, 64, 16, 32, 1, "rand"), (64, 64, 33, 47, 2, "rand")):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.1
    if mode == "center":
        wt = torch.zeros(cout, cin, 3, 3); wt[:cin, :, 1, 1] = torch.eye(cin)
    xin = x.permute(0, 2, 3, 1).contiguous().view(h, w, cin).cuda()
    out = torch.full((h, w, cout), float("nan"), device="cuda")
    d = m.ConvDesc()
    taps, wk = m.conv_taps(wt, 1)
    wp = m.pack_k_major(wk.float()).cuda(); wn = m.pack_wino(wt).cuda()
    d.inp, d.out, d.wpk, d.wino = xin.data_ptr(), out.data_ptr(), wp.data_ptr(), wn.data_ptr()
    d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = h, w, cin, h, w, h, w, cout
    d.in_stride, d.out_stride, d.ntaps, d.nphase = 1, 1, 9, 1
    for i, (dy, dx) in enumerate(taps): d.dy[i], d.dx[i] = dy, dx
    lib.usc_conv_force_winograd(1); lib.usc_conv_force_plan(0, 0, 0, kw)
    arr = (m.ConvDesc * 1)(d)
    print(nv.conv_plans(arr, 1))
    rc = lib.usc_conv_sequence_f32(arr, 1, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    lib.usc_conv_force_winograd(-1); lib.usc_conv_force_plan(0, 0, 0, 0)
    want = F.conv2d(x.double(), wt.double(), padding=1)[0].permute(1, 2, 0)
    got = out.cpu().double()
    e = (got - want).abs()
    print(f"case cin {cin} cout {cout} {h}x{w} kw {kw} {mode}: rc {rc} max err {e.max():.3e} / scale {want.abs().max():.3e}, nan {int(torch.isnan(got).sum())}")
    for by in range((h + 7) // 8):
        print("  block row", by, ["%.1e" % e[8 * by:8 * by + 8, 16 * bx:16 * bx + 16].max() for bx in range((w + 15) // 16)])
    print("  per phase (i,j):", {(i, j): "%.1e" % e[i::2, j::2].max() for i in (0, 1) for j in (0, 1)})
    print("  per channel block:", ["%.1e" % e[..., n:n + 32].max() for n in range(0, cout, 32)])
    print("  per tile row in block 0:", ["%.1e" % e[2 * t:2 * t + 2, :16].max() for t in range(4)], "per tile col:", ["%.1e" % e[:8, 2 * t:2 * t + 2].max() for t in range(8)])
