"""GPU tests of the hand-written f32-MFMA convolution (csrc/usc_conv.hip) through the C ABI.

Float kernels: compared with a plain PyTorch float32 reference of the same op (torch on the CPU, float64
accumulate where noted); tolerance 2e-5 relative to the output scale for single layers (f32 accumulation in a
different order), 1e-4 absolute on alpha for the whole network (north_star bound: 1e-3)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def hipnet():
    assert torch.cuda.is_available()
    from video_unscreen_amd import _native
    _native.load()
    from video_unscreen_amd.vmatting import net_hip
    return net_hip


class _Layer:
    """One usc_conv_desc built by hand for a test."""

    def __init__(self, hipnet):
        self.m = hipnet
        self.keep = []

    def run(self, descs):
        from video_unscreen_amd import _native as nv
        arr = (self.m.ConvDesc * len(descs))(*descs)
        rc = nv.load().usc_conv_sequence_f32(arr, len(descs), torch.cuda.current_stream().cuda_stream)
        assert rc == 0, nv.load().usc_conv_last_error()
        torch.cuda.synchronize()

    def desc(self, x, out, taps, wk, bias, IH, IW, Cin, OH, OW, Cout, in_stride=1, out_stride=1, py=0, px=0,
             res1=None, res1_shift=0, act=0, scale=None, shift=None, res2=None):
        d = self.m.ConvDesc()
        wp = self.m.pack_k_major(wk.float()).cuda()
        self.keep += [wp]
        d.inp, d.out, d.wpk = x.data_ptr(), out.data_ptr(), wp.data_ptr()
        for name, t in (("bias", bias), ("res1", res1), ("scale", scale), ("shift", shift), ("res2", res2)):
            if t is not None:
                t = t.cuda().float().contiguous()
                self.keep.append(t)
                setattr(d, name, t.data_ptr())
        d.IH, d.IW, d.Cin, d.OH, d.OW = IH, IW, Cin, OH, OW
        d.OHf, d.OWf, d.Cout = OH * out_stride, OW * out_stride, Cout
        d.in_stride, d.out_stride, d.out_py, d.out_px = in_stride, out_stride, py, px
        d.res1_shift, d.act, d.ntaps, d.kind = res1_shift, act, len(taps), 0
        for i, (dy, dx) in enumerate(taps):
            d.dy[i], d.dx[i] = dy, dx
        return d


def nhwc(t):       # (1, C, H, W) -> (H, W, C) contiguous
    return t[0].permute(1, 2, 0).contiguous()


def rel_err(got, want):
    return float((got - want).abs().max() / (want.abs().max() + 1e-12))


@pytest.mark.parametrize("cin,cout,h,w,stride,k", [
    (8, 32, 20, 36, 1, 3), (32, 32, 17, 23, 1, 3), (32, 64, 16, 24, 2, 3), (64, 64, 9, 130, 1, 3),
    (64, 128, 12, 20, 2, 3), (128, 128, 7, 11, 1, 3), (256, 512, 6, 10, 2, 3), (512, 256, 5, 6, 1, 1),
    (64, 32, 8, 16, 1, 1), (32, 32, 1, 1, 1, 3), (32, 32, 3, 200, 1, 3)])
def test_conv_layer_matches_torch(hipnet, cin, cout, h, w, stride, k):
    g = torch.Generator().manual_seed(cin * 1000 + cout + h)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g)
    pad = k // 2
    want = F.conv2d(x.double(), wt.double(), b.double(), stride=stride, padding=pad).float()
    oh, ow = want.shape[2], want.shape[3]
    L = _Layer(hipnet)
    taps, wk = hipnet.conv_taps(wt, pad)
    xin = nhwc(x).cuda()
    out = torch.full((oh, ow, cout), float("nan"), device="cuda")
    L.run([L.desc(xin, out, taps, wk, b, h, w, cin, oh, ow, cout, in_stride=stride)])
    assert rel_err(out.cpu(), nhwc(want)) < 2e-5


def test_conv_epilogue_fusions(hipnet):
    """bias + residual (full and half resolution) + ReLU / LeakyReLU + affine + second residual."""
    g = torch.Generator().manual_seed(7)
    cin, cout, h, w = 32, 64, 10, 14
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 17.
    b = torch.randn(cout, generator=g)
    r1 = torch.randn(1, cout, h, w, generator=g)
    r1h = torch.randn(1, cout, h // 2, w // 2, generator=g)
    r2 = torch.randn(1, cout, h, w, generator=g)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    base = F.conv2d(x.double(), wt.double(), b.double(), padding=1)
    taps, wk = hipnet.conv_taps(wt, 1)
    xin = nhwc(x).cuda()
    cases = [
        (dict(res1=nhwc(r1), act=1), F.relu(base + r1)),
        (dict(res1=nhwc(r1h), res1_shift=1, act=2), F.leaky_relu(base + F.interpolate(r1h.double(), scale_factor=2, mode="nearest"), 0.2)),
        (dict(act=1, scale=sc, shift=sh), F.relu(base) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)),
        (dict(res1=nhwc(r1), act=2, res2=nhwc(r2)), F.leaky_relu(base + r1, 0.2) + r2),
        (dict(), base),
    ]
    for kw, want in cases:
        L = _Layer(hipnet)
        out = torch.full((h, w, cout), float("nan"), device="cuda")
        L.run([L.desc(xin, out, taps, wk, b, h, w, cin, h, w, cout, **kw)])
        assert rel_err(out.cpu(), nhwc(want.float())) < 2e-5, kw.keys()


@pytest.mark.parametrize("cin,cout,h,w", [(32, 32, 9, 13), (64, 64, 6, 10), (512, 512, 3, 5), (128, 128, 17, 30)])
def test_conv_transpose_phases_match_torch(hipnet, cin, cout, h, w):
    g = torch.Generator().manual_seed(cin + h)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    b = torch.randn(cout, generator=g)
    want = F.leaky_relu(F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1), 0.2).float()
    L = _Layer(hipnet)
    xin = nhwc(x).cuda()
    out = torch.full((2 * h, 2 * w, cout), float("nan"), device="cuda")
    descs = []
    for py in (0, 1):
        for px in (0, 1):
            taps, wk = hipnet.convt_phase_taps(wt, py, px)
            descs.append(L.desc(xin, out, taps, wk, b, h, w, cin, h, w, cout, out_stride=2, py=py, px=px, act=2))
    L.run(descs)
    assert rel_err(out.cpu(), nhwc(want)) < 2e-5


def test_avgpool_1x1_as_one_conv(hipnet):
    g = torch.Generator().manual_seed(3)
    cin, cout, h, w = 64, 128, 12, 20
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / 8.
    b = torch.randn(cout, generator=g)
    want = F.conv2d(F.avg_pool2d(x.double(), 2, 2), wt.double(), b.double()).float()
    L = _Layer(hipnet)
    wd = wt[:, :, 0, 0].t() * 0.25
    out = torch.full((h // 2, w // 2, cout), float("nan"), device="cuda")
    L.run([L.desc(nhwc(x).cuda(), out, [(0, 0), (0, 1), (1, 0), (1, 1)], torch.cat([wd] * 4, 0), b, h, w, cin,
                  h // 2, w // 2, cout, in_stride=2)])
    assert rel_err(out.cpu(), nhwc(want)) < 2e-5


@pytest.mark.parametrize("H,W", [(64, 96), (96, 160), (544, 960)])
def test_full_network_matches_torch_plan_and_golden(hipnet, H, W):
    """Whole UNet on the HIP convs vs the folded plan on torch-CPU float32 (itself pinned to the reference UNet by
    tests/golden/unet_64x96.npz).  Tolerance 1e-4 absolute on alpha in [0,1]; north_star allows 1e-3."""
    sys.path.insert(0, GOLD)
    from gen_golden import unet_input
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    sd = synthetic_state_dict(0)
    seed = {(64, 96): 1234, (96, 160): 99}.get((H, W), 5)
    x = unet_input(H, W, seed)
    net = hipnet.HipMattingNet(sd, H, W)
    x8 = torch.cat([x, torch.zeros(1, 1, H, W)], 1)[0].permute(1, 2, 0).contiguous().cuda()
    got = net(x8).cpu()
    torch.cuda.synchronize()
    if (H, W) == (544, 960):
        want = MattingNet(sd, device="cuda").forward(x.cuda()).cpu()        # full size: MIOpen f32 as the comparator
    else:
        want = MattingNet(sd, device="cpu")(x)
    d = float((got - want).abs().max())
    assert d < 1e-4, d
    z = np.load(os.path.join(GOLD, "unet_64x96.npz"))
    if (H, W) == (64, 96):
        assert np.abs(got.numpy() - z["y"]).max() < 1e-4                    # the reference module's own output
    if (H, W) == (96, 160):
        assert np.abs(got.numpy()[:, :, ::4, ::4] - z["y2_sub"]).max() < 1e-4
    # deterministic: same input twice -> bit-identical (no atomics in the accumulation)
    again = net(x8).cpu()
    assert torch.equal(again, got)


def test_full_network_with_every_winograd_form_forced_matches_golden(hipnet):
    """The same 64 x 96 network with the Winograd form forced on every layer that admits one - standard (both block forms), K
    halves on two workgroups, and the transposed F(3x3, 2x2) form on all five upsampling layers (the planner only picks them for
    large layers) - against the reference module's own output (tests/golden/unet_64x96.npz) and the torch plan."""
    from video_unscreen_amd import _native as nv
    sys.path.insert(0, GOLD)
    from gen_golden import unet_input
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    H, W = 64, 96
    sd = synthetic_state_dict(0)
    x = unet_input(H, W, 1234)
    x8 = torch.cat([x, torch.zeros(1, 1, H, W)], 1)[0].permute(1, 2, 0).contiguous().cuda()
    want = MattingNet(sd, device="cpu")(x)
    z = np.load(os.path.join(GOLD, "unet_64x96.npz"))
    lib = nv.load()
    for split in ("1", "2"):
        lib.usc_conv_force_winograd_form(1, -1, int(split))
        lib.usc_conv_force_winograd(1)
        try:
            net = hipnet.HipMattingNet(sd, H, W)                      # plans are made here
            plans = [p["plan"] for p in nv.conv_plans(net._arr, len(net._descs))]
        finally:
            lib.usc_conv_force_winograd(-1)
            lib.usc_conv_force_winograd_form(-1, -1, 0)
        assert sum("winograd-t3" in p for p in plans) == 5, plans
        assert sum(p.startswith("winograd") for p in plans) >= 35, plans
        if split == "2":
            assert any("S=2" in p for p in plans), plans
        got = net(x8).cpu()
        torch.cuda.synchronize()
        assert float((got - want).abs().max()) < 1e-4
        assert np.abs(got.numpy() - z["y"]).max() < 1e-4
        assert torch.equal(net(x8).cpu(), got)
        net.close()


@pytest.mark.parametrize("in_relu", [False, True])
def test_conv_every_plan_gives_the_same_layer(hipnet, in_relu):
    """All kernel variants on one layer: tile width (32 / 64 columns) x pixels per workgroup (128 / 64 / 32) x grid K split x
    K groups per workgroup, forced through the tuning hook.  Every plan must give the reference result (the K partition only changes the summation order), with the
    fused epilogue (bias, residual, ReLU) applied exactly once whether it runs in the conv kernel or in the split-K reduce."""
    from video_unscreen_amd import _native as nv
    lib = nv.load()
    g = torch.Generator().manual_seed(99)
    cin, cout, h, w = 64, 128, 21, 37               # M = 777: a ragged last tile; K = 576: 18 steps
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24.
    b = torch.randn(cout, generator=g)
    r1 = torch.randn(1, cout, h, w, generator=g)
    xin = F.relu(x) if in_relu else x
    want = F.relu(F.conv2d(xin.double(), wt.double(), b.double(), padding=1) + r1.double()).float()
    L = _Layer(hipnet)
    taps, wk = hipnet.conv_taps(wt, 1)
    xd, r1d = nhwc(x).cuda(), nhwc(r1).cuda()
    slab = torch.empty(8 * h * w * cout, device="cuda")
    results = {}
    try:
        for nt, rg in ((1, 4), (2, 4), (1, 2), (1, 1)):
            for s in (1, 2, 3, 6):
                for kw in (1, 2, 4, 8, 16):
                    if (nt == 2 and kw > 2) or (rg == 4 and kw > 4) or (rg < 4 and not 4 <= rg * kw <= 16) or s * kw > 18:
                        continue                                   # not a kernel variant / more slices than K steps
                    lib.usc_conv_force_plan(nt, rg, s, kw)
                    out = torch.full((h, w, cout), float("nan"), device="cuda")
                    d = L.desc(xd, out, taps, wk, b, h, w, cin, h, w, cout, res1=r1d, act=1)          # USC_ACT_RELU
                    d.in_relu = 1 if in_relu else 0
                    d.slab, d.slab_bytes = slab.data_ptr(), slab.numel() * 4
                    L.run([d])
                    assert rel_err(out.cpu(), nhwc(want)) < 2e-5, (nt, rg, s, kw)
                    results[(nt, rg, s, kw)] = out.clone()
    finally:
        lib.usc_conv_force_plan(0, 0, 0, 0)
    # neither the tile width nor the pixels per workgroup touch the summation order: bit-identical for the same (S, KW)
    for (nt, rg, s_, kw), out in results.items():
        base = results.get((1, 4, s_, kw))
        if base is not None:
            assert torch.equal(out, base), (nt, rg, s_, kw)
    assert len(results) >= 30


def test_conv_graphs_are_destroyed_for_real(hipnet):
    """usc_conv_graph_destroy frees the HIP graph objects (after draining the device): the round-1 sequence that faulted -
    networks built, launched on several streams and dropped with work still queued, a dozen times - runs clean, the
    results stay right, and device memory does not grow with the number of networks built."""
    import gc
    from video_unscreen_amd.vmatting.net import MattingNet, synthetic_state_dict
    sd = synthetic_state_dict(0)
    H, W = 64, 96
    x = torch.randn(1, 7, H, W, generator=torch.Generator().manual_seed(3)) * 0.5
    want = MattingNet(sd, device="cpu")(x)
    x8 = torch.zeros(H, W, 8, device="cuda")
    x8[..., :7] = x[0].permute(1, 2, 0).cuda()
    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    used = []
    for it in range(12):
        net = hipnet.HipMattingNet(sd, H, W)
        net.x8.copy_(x8)
        torch.cuda.synchronize()
        for st in [torch.cuda.current_stream()] + side:          # one instantiation per stream
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                for _ in range(3):
                    net()
            torch.cuda.current_stream().wait_stream(st)
        if it % 6 == 0:
            torch.cuda.synchronize()
            assert (net.pred.cpu() - want).abs().max() < 1e-4
            net()
        del net                                                  # dropped with the last launches possibly still queued
        gc.collect()
        used.append(torch.cuda.memory_allocated())
    torch.cuda.synchronize()
    assert used[-1] <= used[2], used


@pytest.mark.parametrize("cin,cout,h,w,batch,kw,dil,lin,split", [
    (8, 32, 17, 30, 1, 1, 1, 0, 1), (32, 32, 20, 36, 1, 1, 1, 0, 1), (64, 64, 33, 47, 1, 2, 1, 0, 1), (128, 128, 12, 20, 2, 4, 1, 0, 1), (32, 64, 9, 7, 3, 2, 1, 0, 1),
    (256, 96, 16, 16, 1, 4, 1, 0, 1), (16, 32, 64, 96, 1, 1, 1, 0, 1), (64, 64, 30, 30, 2, 2, 2, 0, 1), (32, 32, 18, 42, 1, 1, 2, 0, 1), (32, 64, 27, 15, 2, 1, 3, 0, 1),
    # blocks of 32 consecutive tiles (tile rows of >= 11 tiles): runs that start mid-row, wrap over up to four rows, end in a partial block
    (32, 32, 20, 36, 1, 1, 1, 1, 1), (64, 64, 33, 47, 1, 2, 1, 1, 1), (16, 32, 64, 96, 1, 1, 1, 1, 1), (128, 128, 68, 120, 1, 4, 1, 1, 1), (32, 64, 23, 22, 3, 2, 1, 1, 1),
    (32, 32, 18, 44, 1, 1, 2, 1, 1), (8, 32, 31, 45, 2, 1, 1, 1, 1),
    # K halves on two workgroups (partial tiles meet in the slab, the second to arrive finishes)
    (128, 128, 12, 20, 2, 4, 1, 0, 2), (64, 64, 33, 47, 1, 2, 1, 1, 2), (256, 96, 16, 16, 1, 4, 1, 0, 2), (128, 128, 34, 60, 1, 4, 1, 1, 2),
    (64, 64, 30, 30, 2, 2, 2, 0, 2), (32, 64, 9, 7, 3, 2, 1, 0, 2), (256, 256, 34, 60, 2, 4, 1, 1, 2)])
def test_winograd_layer_matches_torch(hipnet, cin, cout, h, w, batch, kw, dil, lin, split):
    """conv_wino_kernel (Winograd F(2x2, 3x3): 3x3 / stride 1 / pad 1) against torch's float64 convolution: odd sizes (partial
    tiles and blocks), images stacked along y, K groups, the full fused epilogue, and dilated layers (dil x dil standard
    convolutions on the pixel-parity sub-images), and K split over two workgroups."""
    from video_unscreen_amd import _native as nv
    g = torch.Generator().manual_seed(cin * 1000 + cout + h)
    slab = torch.empty(8 * 1024 * 1024, device="cuda").fill_(float("nan")) if split == 2 else None
    x = torch.randn(batch, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (1.5 / (3 * cin ** 0.5))
    b = torch.randn(cout, generator=g)
    r1 = torch.randn(batch, cout, h, w, generator=g)
    r2 = torch.randn(batch, cout, h, w, generator=g)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().view(batch * h, w, -1)
    L = _Layer(hipnet)
    taps, wk = hipnet.conv_taps(wt, 1)
    taps = [(dy * dil, dx * dil) for dy, dx in taps]
    lib = nv.load()
    xin = nhwc(x).cuda()
    for fused in (False, True):
        out = torch.full((batch * h, w, cout), float("nan"), device="cuda")
        d = L.desc(xin, out, taps, wk, b, h, w, cin, batch * h, w, cout, act=2 if fused else 0,
                   res1=nhwc(r1) if fused else None, res2=nhwc(r2) if fused else None, scale=sc if fused else None,
                   shift=sh if fused else None)
        d.batch = batch
        wino = hipnet.pack_wino(wt).cuda()
        L.keep.append(wino)
        d.wino = wino.data_ptr()
        if slab is not None:
            d.slab, d.slab_bytes = slab.data_ptr(), slab.numel() * 4
        lib.usc_conv_force_winograd(1)
        lib.usc_conv_force_plan(0, 0, 0, kw)
        lib.usc_conv_force_winograd_form(-1, int(lin), int(split))
        try:
            arr = (hipnet.ConvDesc * 1)(d)
            plan = nv.conv_plans(arr, 1)[0]
            assert plan["winograd"] and ("-lin" in plan["plan"]) == bool(lin) and ("S=2" in plan["plan"]) == (split == 2), plan
            L.run([d])
            if split == 2:                                   # the same bits on every run, whichever workgroup arrives second
                first = out.clone()
                for _ in range(5):
                    out.fill_(float("nan"))
                    L.run([d])
                    assert torch.equal(out, first)
        finally:
            lib.usc_conv_force_winograd_form(-1, -1, 0)
            lib.usc_conv_force_winograd(-1)
            lib.usc_conv_force_plan(0, 0, 0, 0)
        y = F.conv2d(x.double(), wt.double(), b.double(), padding=dil, dilation=dil)
        if fused:
            y = y + r1.double()
            y = torch.where(y > 0, y, y * 0.2)
            y = y * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1) + r2.double()
        got = out.cpu().view(batch, h, w, cout).permute(0, 3, 1, 2).double()
        assert torch.isfinite(got).all()
        err = (got - y).abs().max() / y.abs().max()
        assert err < 2e-5, (fused, float(err))


@pytest.mark.parametrize("cin,cout,h,w", [
    (32, 32, 12, 36), (64, 64, 17, 30), (8, 32, 20, 40), (128, 128, 34, 60), (256, 64, 9, 35), (32, 64, 7, 16),
    (64, 32, 33, 100), (16, 32, 5, 4), (32, 32, 48, 96), (32, 32, 51, 333)])
def test_winograd_transposed_layer_matches_torch(hipnet, cin, cout, h, w):
    """conv_wino_kernel's transposed form (ConvTranspose2d(k 4, s 2, p 1) as Winograd F(3x3, 2x2) per output phase) against
    torch's float64 transposed convolution: sizes that are no multiples of 3, narrow maps (blocks of whole tile rows), rows of
    more than 32 tiles, bias / activation / second residual."""
    from video_unscreen_amd import _native as nv
    g = torch.Generator().manual_seed(cin * 1000 + cout + h)
    x = torch.randn(1, cin, h, w, generator=g)
    wt = torch.randn(cin, cout, 4, 4, generator=g) * (1.5 / (2 * cin ** 0.5))
    b = torch.randn(cout, generator=g)
    r2 = torch.randn(1, cout, 2 * h, 2 * w, generator=g)
    L = _Layer(hipnet)
    lib = nv.load()
    xin = nhwc(x).cuda()
    phases = [hipnet.convt_phase_taps(wt, py, px) for py in (0, 1) for px in (0, 1)]
    packed = torch.stack([hipnet.pack_k_major(wk.float()) for _, wk in phases], 0).cuda()
    wino = hipnet.pack_wino_t(wt).cuda()
    for fused in (False, True):
        out = torch.full((2 * h, 2 * w, cout), float("nan"), device="cuda")
        d = L.desc(xin, out, phases[0][0], phases[0][1], b, h, w, cin, h, w, cout, out_stride=2, act=2 if fused else 0,
                   res2=nhwc(r2) if fused else None)
        d.wpk, d.nphase, d.wphase_stride = packed.data_ptr(), 4, packed[0].numel()
        for i, (dy, dx) in enumerate([t for tp, _ in phases for t in tp]):
            d.dy[i], d.dx[i] = dy, dx
        d.wino = wino.data_ptr()
        lib.usc_conv_force_winograd_form(1, -1, 0)
        try:
            plan = nv.conv_plans((hipnet.ConvDesc * 1)(d), 1)[0]
            assert "winograd-t3" in plan["plan"], plan
            L.run([d])
            first = out.clone()
            L.run([d])
            assert torch.equal(out, first)
            lib.usc_conv_force_winograd_form(0, -1, 0)            # the direct kernel on the same descriptor
            ref = torch.full_like(out, float("nan"))
            d.out = ref.data_ptr()
            assert "winograd" not in nv.conv_plans((hipnet.ConvDesc * 1)(d), 1)[0]["plan"]
            L.run([d])
        finally:
            lib.usc_conv_force_winograd_form(-1, -1, 0)
        y = F.conv_transpose2d(x.double(), wt.double(), b.double(), stride=2, padding=1)
        if fused:
            y = torch.where(y > 0, y, y * 0.2) + r2.double()
        got = out.cpu().permute(2, 0, 1)[None].double()
        assert torch.isfinite(got).all()
        err = (got - y).abs().max() / y.abs().max()
        assert err < 2e-5, (fused, float(err))
        assert (ref.cpu().permute(2, 0, 1)[None].double() - y).abs().max() / y.abs().max() < 2e-5


def test_winograd_is_batch_invariant_and_deterministic(hipnet):
    """Per-image results of a Winograd layer do not depend on the batch it runs in, nor on the run."""
    from video_unscreen_amd import _native as nv
    cin, cout, h, w = 64, 64, 30, 30
    g = torch.Generator().manual_seed(77)
    x = torch.randn(3, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    taps, wk = hipnet.conv_taps(wt, 1)
    L = _Layer(hipnet)
    wino = hipnet.pack_wino(wt).cuda()
    outs = []
    for batch in (1, 3, 3):
        xs = x[:batch].permute(0, 2, 3, 1).contiguous().view(batch * h, w, cin).cuda()
        out = torch.empty((batch * h, w, cout), device="cuda")
        d = L.desc(xs, out, taps, wk, None, h, w, cin, batch * h, w, cout)
        d.batch, d.plan_batch, d.wino = batch, 3, wino.data_ptr()
        nv.load().usc_conv_force_winograd(1)
        try:
            L.run([d])
        finally:
            nv.load().usc_conv_force_winograd(-1)
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1][:h]) and torch.equal(outs[1], outs[2])


def test_two_conv_stacks_on_two_streams_give_the_bits_of_one():
    """The frame loop runs the matting net, the human-parsing net and the pixel stages on three streams at once.  A conv launch must
    not depend on having the GPU to itself: the Winograd kernel hands staged patch images from one wave of a pair to the other, and
    until round 3 the hand-over of a pair's second image rested on its requests landing within a phase and a half (fine on an idle
    GPU; beside another stream's kernels a unit now and then multiplied one K step of stale values - about one launch in 20 000).
    Here: 40 parsing forwards (12 frames each) beside 400 matting forwards, every output compared with the one computed alone."""
    from video_unscreen_amd.binseg import HumanSegAgent
    from video_unscreen_amd.binseg.schp_net import schp_synthetic_state_dict
    from video_unscreen_amd.synth import SyntheticClip
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import HipMattingNet
    dev = torch.device("cuda", 0)
    net = HipMattingNet(synthetic_state_dict(0), 544, 960, device=dev)
    g = torch.Generator(device="cpu")
    g.manual_seed(5)
    x = torch.rand((544, 960, 8), generator=g)
    x[..., 7] = 0
    net.x8.copy_(x.to(dev))
    seg = HumanSegAgent(model_path=None, cuda_device=0, state_dict=schp_synthetic_state_dict(0))
    seg.set_plan_batch(12)
    clip = SyntheticClip(1080, 1920, 12, config_id=2, device=dev)
    frames = [clip.frame(t)["frame"] for t in range(12)]
    masks = [torch.empty((1080, 1920), dtype=torch.uint8, device=dev) for _ in range(12)]
    ref = net().clone()
    seg.forward_device_batch(frames, outs=masks)
    torch.cuda.synchronize()
    mref = torch.stack(masks).clone()
    s2 = torch.cuda.Stream(dev)
    bad, mbad = [], []
    for _ in range(40):
        with torch.cuda.stream(s2):
            seg.forward_device_batch(frames, outs=masks)
            mbad.append((torch.stack(masks) != mref).sum())
        for _ in range(10):
            bad.append((net() != ref).sum())
    torch.cuda.synchronize()
    assert sum(int(b) for b in bad) == 0, [i for i, b in enumerate(bad) if int(b)]
    assert sum(int(b) for b in mbad) == 0, [i for i, b in enumerate(mbad) if int(b)]
    net.close()


def test_runs_mode_is_bit_identical():
    """USC_CONV_LAUNCH_RUNS (opt-in; profiles/r04/runs_mode.md): runs of same-shape Winograd layers of the matting net as ONE launch
    each - items from a counter, per-block arrival counters polled with a bound, write-through stores / sc1 loads across the eight
    L2s.  Every forward must give the bits of the per-layer launches, also beside a second stream that keeps a varying share of the
    CUs busy (uneven load is where a broken hand-over shows), every intermediate tensor included, and no poll may ever run out."""
    import ctypes as C
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import HipMattingNet
    net = HipMattingNet(synthetic_state_dict(0), 544, 960)
    net.x8.normal_()
    nruns, nlayers, bad_polls = C.c_int(), C.c_int(), C.c_uint()
    assert nv.load().usc_conv_graph_run_info(net._graph, C.byref(nruns), C.byref(nlayers), C.byref(bad_polls)) == 0
    assert nruns.value >= 3 and nlayers.value >= 8            # #3-6, #26-27, #45-46 of profiles/r03/unet_layer_table.md
    ref = net().clone()
    acts = [t for t in net._keep if t.dim() == 3]
    acts_ref = [t.clone() for t in acts]
    net.set_launch_mode("runs")
    side = torch.cuda.Stream()
    junk = torch.randn(2048, 2048, device="cuda")
    bad = []
    for i in range(60):
        if i >= 20:
            with torch.cuda.stream(side):
                for _ in range(1 + i % 3):
                    junk = (junk @ junk).clamp_(-1, 1)
        bad.append((net() != ref).sum())
    torch.cuda.synchronize()
    assert sum(int(b) for b in bad) == 0, [i for i, b in enumerate(bad) if int(b)]
    assert all(torch.equal(a, b) for a, b in zip(acts, acts_ref))
    assert nv.load().usc_conv_graph_run_info(net._graph, None, None, C.byref(bad_polls)) == 0 and bad_polls.value == 0
    net.set_launch_mode("stream")
    assert torch.equal(net(), ref)
    net.close()


def test_stage_entry_merge_matches_separate_launches():
    """usc_conv_desc.out2: a ResNet stage's 3x3 / stride-2 conv and its AvgPool + 1x1 shortcut as ONE two-output launch (the shortcut's
    weights on four of the nine taps) against the two launches of rounds 1 - 3: every activation tensor both networks have in common
    agrees to float rounding (the K partition of the merged layer differs), the prediction to 1e-6; the planner refuses the
    combinations the kernel cannot serve (a grid K split, residual tensors, a split inside a 64-channel tile)."""
    import ctypes as C
    from video_unscreen_amd import _native as nv
    from video_unscreen_amd.vmatting.net import synthetic_state_dict
    from video_unscreen_amd.vmatting.net_hip import ConvDesc, HipMattingNet
    sd = synthetic_state_dict(0)
    a, b = HipMattingNet(sd, 160, 224, merge_stage_entry=False), HipMattingNet(sd, 160, 224, merge_stage_entry=True)
    assert a.num_layers == b.num_layers + 3
    x = torch.randn(160, 224, 8, generator=torch.Generator().manual_seed(3)).cuda()
    x[..., 7] = 0
    pa, pb = a(x).clone(), b(x).clone()
    assert float((pa - pb).abs().max()) < 1e-6
    merged = [d for d in b._descs if d.out2]
    assert len(merged) == 3 and all(d.split_c * 2 == d.Cout and d.ntaps == 9 and d.in_stride == 2 for d in merged)
    plans = nv.conv_plans((ConvDesc * 3)(*merged), 3)
    assert all(p["launches"] == 1 for p in plans)                       # never the split-K reduce path
    # the stage outputs (inputs of the next stage's merged layer) agree
    for d_b in merged[1:]:
        t_b = next(t for t in b._keep if t.data_ptr() == d_b.inp)
        t_a = next(t for t in a._keep if t.shape == t_b.shape and float((t - t_b).abs().max()) < 1e-4 * float(t_b.abs().max()))
        assert t_a is not None
    bad = ConvDesc.from_buffer_copy(bytes(merged[0]))
    bad.split_c = 48
    info = nv.ConvPlanInfo()
    assert nv.load().usc_conv_plan_query(C.byref(bad), C.byref(info)) != 0
    bad = ConvDesc.from_buffer_copy(bytes(merged[0]))
    bad.res1 = bad.out
    assert nv.load().usc_conv_plan_query(C.byref(bad), C.byref(info)) != 0
    a.close(); b.close()
