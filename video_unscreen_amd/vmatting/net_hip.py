"""The matting network on the hand-written f32-MFMA convolution kernel (csrc/usc_conv.hip).

Same folded plan as net.py (reference unscreen/vmatting/model.py:435-568, spectral norm and conv->BN folded at
load), but instead of ~230 PyTorch / MIOpen kernel launches per frame the whole forward is ONE C call that
enqueues 54 fused layers: every bias, residual add, ReLU / LeakyReLU, shortcut BN and nearest-x2 upsample lives
in a conv epilogue, ConvTranspose2d(k4, s2, p1) runs as its four 2x2 output phases, AvgPool2d(2)+1x1 as one
2x2-stride-2 conv.  All tensors are NHWC float32 buffers allocated once per frame size.
"""
import ctypes as C
import sys

import torch
import torch.nn.functional as F

from .. import _native as nv
from .net import fold_state_dict

ACT = {"none": 0, "relu": 1, "leaky": 2}


class ConvDesc(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("wpk", C.c_void_p), ("bias", C.c_void_p),
                ("res1", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p), ("res2", C.c_void_p),
                ("IH", C.c_int), ("IW", C.c_int), ("Cin", C.c_int),
                ("OH", C.c_int), ("OW", C.c_int),
                ("OHf", C.c_int), ("OWf", C.c_int), ("Cout", C.c_int),
                ("in_stride", C.c_int),
                ("out_stride", C.c_int), ("out_py", C.c_int), ("out_px", C.c_int),
                ("res1_shift", C.c_int), ("act", C.c_int), ("ntaps", C.c_int), ("kind", C.c_int),
                ("dy", C.c_byte * 16), ("dx", C.c_byte * 16),
                ("nphase", C.c_int), ("wphase_stride", C.c_int), ("ksplit", C.c_int), ("in_relu", C.c_int),
                ("slab", C.c_void_p), ("slab_bytes", C.c_size_t), ("out_ld", C.c_int), ("out_c0", C.c_int),
                ("batch", C.c_int), ("plan_batch", C.c_int), ("wino", C.c_void_p),
                ("out2", C.c_void_p), ("split_c", C.c_int), ("act2", C.c_int)]


def d_out_elems(OH, OW, out_stride, OHf, OWf, Cout):
    return (OHf or OH * out_stride) * (OWf or OW * out_stride) * Cout


def pack_k_major(wk):
    """(K, Cout) rows k = tap*Cin + ci  ->  [Kpad/4][Cout][4] with K padded to a multiple of 32."""
    K, N = wk.shape
    Kp = (K + 31) // 32 * 32
    if Kp != K:
        wk = torch.cat([wk, wk.new_zeros(Kp - K, N)], 0)
    return wk.view(Kp // 4, 4, N).permute(0, 2, 1).contiguous()


_WINO_G = torch.tensor([[1., 0., 0.], [.5, .5, .5], [.5, -.5, .5], [0., 0., 1.]], dtype=torch.float64)


def pack_wino(w):
    """Conv2d(3x3, stride 1) weight (Cout, Cin, 3, 3), Cin % 8 == 0 -> the Winograd F(2x2, 3x3) operand of conv_wino_kernel:
    U[a][b] = G g G^T per (cout, cin) (float64, rounded once to float32), laid out [Cin/8][16 positions][2][Cout][4] with
    cin = 8c + 4h + q - the order the kernel's weight fragments are read in (one 512-byte run per position and half)."""
    co, ci, kh, kw = w.shape
    assert (kh, kw) == (3, 3) and ci % 8 == 0
    U = torch.einsum('ai,ocij,bj->abco', _WINO_G, w.double(), _WINO_G).float()          # (4, 4, Cin, Cout)
    U = U.reshape(16, ci // 8, 2, 4, co)                                                 # [p][c][h][q][n]
    return U.permute(1, 0, 2, 4, 3).contiguous()                                         # [c][p][h][n][q]


_WINO_GT = torch.tensor([[1., 0.], [.5, .5], [.5, -.5], [0., -1.]], dtype=torch.float64)


def pack_wino_t(w):
    """ConvTranspose2d(k 4, s 2, p 1) weight (Cin, Cout, 4, 4), Cin % 8 == 0 -> the Winograd F(3x3, 2x2) operands of
    conv_wino_kernel's transposed form, one per output phase (py, px): along each axis phase 0 is the correlation of
    (in[y - 1], in[y]) with (w[3], w[1]), phase 1 of (in[y], in[y + 1]) with (w[2], w[0]) (convt_phase_taps); U = G g G^T with
    G = [1 0; 1/2 1/2; 1/2 -1/2; 0 -1] - the last row negated, so that the kernel's input transform is F(2x2, 3x3)'s own.
    Layout [phase][Cin/8][16 positions][2][Cout][4] (pack_wino's, per phase)."""
    ci, co, kh, kw = w.shape
    assert (kh, kw) == (4, 4) and ci % 8 == 0
    sel = {0: [3, 1], 1: [2, 0]}
    out = []
    for py in (0, 1):
        for px in (0, 1):
            g = w.double()[:, :, sel[py], :][:, :, :, sel[px]]                          # (Cin, Cout, 2, 2)
            U = torch.einsum('ai,coij,bj->abco', _WINO_GT, g, _WINO_GT).float()        # (4, 4, Cin, Cout)
            U = U.reshape(16, ci // 8, 2, 4, co)
            out.append(U.permute(1, 0, 2, 4, 3).contiguous())
    return torch.stack(out, 0)


def conv_taps(w, pad):
    """Conv2d weight (Cout, Cin, kh, kw) -> (taps [(dy, dx)], (K, Cout) matrix)."""
    co, ci, kh, kw = w.shape
    taps = [(ky - pad, kx - pad) for ky in range(kh) for kx in range(kw)]
    return taps, w.permute(2, 3, 1, 0).reshape(kh * kw * ci, co)


def convt_phase_taps(w, py, px):
    """ConvTranspose2d(k=4, s=2, p=1) weight (Cin, Cout, 4, 4), output phase (py, px):
    out[2y+py, 2x+px] = sum over the two (dy, ky) pairs per axis of in[y+dy, x+dx] * w[:, :, ky, kx]."""
    pairs = {0: [(0, 1), (-1, 3)], 1: [(1, 0), (0, 2)]}
    taps, mats = [], []
    for dy, ky in pairs[py]:
        for dx, kx in pairs[px]:
            taps.append((dy, dx))
            mats.append(w[:, :, ky, kx])                     # (Cin, Cout)
    return taps, torch.cat(mats, 0)


class HipMattingNet:
    def __init__(self, state_dict, in_h, in_w, device="cuda:0", use_graph=True, merge_stage_entry=True, batch=1):
        """merge_stage_entry: a ResNet stage's first 3x3 / stride-2 conv and its AvgPool + 1x1 shortcut as ONE two-output launch (51 instead
        of 54 launches per forward; False: the three shortcut convs as launches of their own, as in rounds 1 - 3).
        batch > 1: that many independent images per forward, stacked along y in x8 / pred (usc_conv_desc.batch, planned as for one image:
        bit-identical per-image results).  The four-phase transposed layers and the head do not take a batch: they are launched once per
        image.  (A measurement aid - tools/dev/batch2_probe.py: within a frame range the matting net cannot be batched, alpha_pre
        chains the frames.)"""
        self.merge_stage_entry = merge_stage_entry
        self.batch = max(1, int(batch))
        if in_h % 32 or in_w % 32:
            raise ValueError("HipMattingNet: input size must be a multiple of 32 (vmatting/agent.py:38-42 guarantees it)")
        self.device = torch.device(device)
        self.H, self.W = in_h, in_w
        nv.load()
        self._keep = []               # every tensor a descriptor points to
        self._descs = []
        self.slab = torch.empty(96 * 1024 * 1024, dtype=torch.float32, device=self.device)   # split-K partial sums (384 MiB)
        self._slab_used = 0
        self._build(fold_state_dict(state_dict))
        self._arr = (ConvDesc * len(self._descs))(*self._descs)
        self._graph = C.c_void_p()
        if use_graph:
            rc = nv.load().usc_conv_graph_create(self._arr, len(self._descs), C.byref(self._graph))
            if rc != 0:
                raise nv.NativeError(f"usc_conv_graph_create failed ({rc}): {nv.load().usc_conv_last_error().decode()}")

    def close(self):
        """Destroy the graph (usc_conv_graph_destroy drains the device first), then release the buffers its nodes point to."""
        if getattr(self, "_graph", None) is not None and self._graph:
            nv.load().usc_conv_graph_destroy(self._graph)
            self._graph = C.c_void_p()
        self._keep = []

    def __del__(self):
        try:
            if sys.is_finalizing():        # the HIP runtime may already be gone: the process frees everything anyway
                return
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _dev(self, t):
        t = t.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(t)
        return t

    def _buf(self, h, w, c):
        t = torch.empty((self.batch * h, w, c), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def _add(self, x, out, taps, wk, bias, IH, IW, Cin, OH, OW, Cout, in_stride=1, out_stride=1, py=0, px=0,
             res1=None, res1_shift=0, act="none", scale=None, shift=None, res2=None, kind=0, OHf=None, OWf=None,
             phases=None, wino=None, wino_t=None, out2=None, split_c=0, act2="none"):
        """phases: optional list of 4 (taps, wk) pairs (transposed conv); then `taps` / `wk` are ignored.
        out2 / split_c / act2: output channels [split_c, Cout) go to the second tensor with their own activation (usc_conv_desc.out2)."""
        d = ConvDesc()
        d.inp, d.out = x.data_ptr(), out.data_ptr()
        if out2 is not None:
            d.out2, d.split_c, d.act2 = out2.data_ptr(), split_c, ACT[act2]
        if phases is not None:
            packed = [pack_k_major(w.float()) for _, w in phases]
            wp = self._dev(torch.stack(packed, 0))
            d.nphase, d.wphase_stride = len(phases), packed[0].numel()
            taps_all = [t for tp, _ in phases for t in tp]
            taps = phases[0][0]
        else:
            wp = self._dev(pack_k_major(wk.float()) if kind == 0 else wk.float().reshape(-1))
            d.nphase, d.wphase_stride = 1, 0
            taps_all = taps
        d.wpk = wp.data_ptr()
        if wino is not None:                                   # 3x3 / stride 1: the Winograd operand of the same weights
            d.wino = self._dev(pack_wino(wino)).data_ptr()
        if wino_t is not None:                                 # transposed conv: the F(3x3, 2x2) operands of the four phases
            d.wino = self._dev(pack_wino_t(wino_t)).data_ptr()
        # every layer gets its own 16-slice region of the split-K scratch, so layers on different graph branches
        # never share partial-sum memory
        tiles = ((OH * OW + 127) // 128) * max(Cout // 32, 1) * (4 if phases is not None else 1)
        nch = (len(taps) * Cin + 31) // 32
        slices = max(1, min(16, -(-768 // tiles), nch // 6))            # the library's own split rule (usc_conv.hip plan_layer)
        need = slices * d_out_elems(OH, OW, out_stride, OHf, OWf, Cout) * 4
        if kind == 0 and tiles < 384 and slices > 1 and self._slab_used + need <= self.slab.numel() * 4:
            d.slab, d.slab_bytes = self.slab.data_ptr() + self._slab_used, need
            self._slab_used += need
        d.bias = self._dev(bias).data_ptr() if bias is not None else None
        d.res1 = res1.data_ptr() if res1 is not None else None
        d.scale = self._dev(scale.reshape(-1)).data_ptr() if scale is not None else None
        d.shift = self._dev(shift.reshape(-1)).data_ptr() if shift is not None else None
        d.res2 = res2.data_ptr() if res2 is not None else None
        d.IH, d.IW, d.Cin = IH, IW, Cin
        d.OH, d.OW = OH, OW
        d.OHf, d.OWf, d.Cout = (OHf or OH * out_stride), (OWf or OW * out_stride), Cout
        d.in_stride, d.out_stride, d.out_py, d.out_px = in_stride, out_stride, py, px
        d.res1_shift, d.act, d.ntaps, d.kind = res1_shift, ACT[act], len(taps), kind
        for i, (dy, dx) in enumerate(taps_all):
            d.dy[i], d.dx[i] = dy, dx
        B = self.batch
        if B == 1:
            self._descs.append(d)
        elif phases is None and kind == 0:                      # the kernels stack the images along y
            d.batch, d.plan_batch = B, 1
            d.OH, d.OHf = B * d.OH, B * d.OHf
            self._descs.append(d)
        else:                                                   # one launch per image, pointers moved to the image
            ohf, owf = d.OHf, d.OWf
            for b in range(B):
                e = ConvDesc.from_buffer_copy(bytes(d))
                e.inp = d.inp + b * IH * IW * Cin * 4
                e.out = d.out + b * ohf * owf * Cout * 4
                if res2 is not None:
                    e.res2 = d.res2 + b * ohf * owf * Cout * 4
                if res1 is not None:
                    e.res1 = d.res1 + b * (ohf >> res1_shift) * (owf >> res1_shift) * Cout * 4
                self._descs.append(e)

    def _conv(self, c, x, IH, IW, out=None, **kw):
        """Plain Conv2d from the folded plan (net.Conv): 3x3 s1/s2 pad 1 or 1x1."""
        w = c.w
        co, ci, kh, _ = w.shape
        if ci % 4:                                             # the 7-channel input is stored with 8 channels
            w = F.pad(w, (0, 0, 0, 0, 0, 4 - ci % 4))
            ci = w.shape[1]
        taps, wk = conv_taps(w, c.pad)
        OH, OW = IH // c.stride, IW // c.stride
        out = out if out is not None else self._buf(OH, OW, co)
        wino = w if (kh == 3 and c.stride == 1 and c.pad == 1 and ci % 8 == 0) else None
        self._add(x, out, taps, wk, c.b, IH, IW, ci, OH, OW, co, in_stride=c.stride, act=kw.pop("act", c.act), wino=wino, **kw)
        return out

    def _convt(self, c, x, IH, IW, res2=None):
        """ConvTranspose2d(k4, s2, p1) as four phase convolutions writing one (2 IH, 2 IW) tensor."""
        ci, co = c.w.shape[0], c.w.shape[1]
        out = self._buf(2 * IH, 2 * IW, co)
        phases = [convt_phase_taps(c.w, py, px) for py in (0, 1) for px in (0, 1)]       # phase index = 2*py + px
        self._add(x, out, None, None, c.b, IH, IW, ci, IH, IW, co, out_stride=2, act=c.act, res2=res2, phases=phases,
                  wino_t=c.w if ci % 8 == 0 else None)
        return out

    # ------------------------------------------------------------------ graph
    def _build(self, p):
        H, W = self.H, self.W
        self.x8 = self._buf(H, W, 8)
        x = self.x8
        # stem (model.py:460-469)
        t = self._conv(p["stem"][0], x, H, W)
        x1 = self._conv(p["stem"][1], t, H // 2, W // 2)
        t = self._conv(p["stem"][2], x1, H // 2, W // 2)
        h, w = H // 4, W // 4
        feats = []
        for blocks in p["enc"]:                                 # BasicBlock x2 per stage (model.py:119-157)
            for blk in blocks:
                s = blk["conv1"].stride
                if blk["down"] is not None and self.merge_stage_entry and s == 2 and blk["conv1"].w.shape[2] == 3:
                    # Stage entry as ONE launch: the 3x3 / stride-2 conv and the shortcut (AvgPool2d(2, 2) + 1x1 conv + BN = a 2x2 /
                    # stride-2 conv on the taps (0, 0), (0, 1), (1, 0), (1, 1) of the same nine) read the same input with the same
                    # geometry: output channels [0, C) = conv1 with its activation -> y, [C, 2C) = the shortcut -> idt (two tensors).
                    c1 = blk["conv1"]
                    wd = blk["down"].w[:, :, 0, 0].t() * 0.25   # (Cin, Cout)
                    taps, wk1 = conv_taps(c1.w, c1.pad)         # wk1: (9 Cin, C), rows tap-major
                    ci, co = wd.shape
                    wk2 = torch.zeros_like(wk1)
                    for ti, tp in enumerate(taps):
                        if tp in ((0, 0), (0, 1), (1, 0), (1, 1)):
                            wk2[ti * ci:(ti + 1) * ci] = wd
                    y, idt = self._buf(h // 2, w // 2, co), self._buf(h // 2, w // 2, co)
                    self._add(t, y, taps, torch.cat([wk1, wk2], 1), torch.cat([c1.b, blk["down"].b]), h, w, ci, h // 2, w // 2, 2 * co,
                              in_stride=2, act=c1.act, out2=idt, split_c=co, act2="none")
                    h, w = h // s, w // s
                    t = self._conv(blk["conv2"], y, h, w, res1=idt, act="relu")
                    continue
                if blk["down"] is not None:                     # AvgPool2d(2, 2) + 1x1 conv + BN as one 2x2/s2 conv
                    wd = blk["down"].w[:, :, 0, 0].t() * 0.25   # (Cin, Cout)
                    taps = [(0, 0), (0, 1), (1, 0), (1, 1)]
                    idt = self._buf(h // 2, w // 2, wd.shape[1])
                    self._add(t, idt, taps, torch.cat([wd] * 4, 0), blk["down"].b, h, w, wd.shape[0], h // 2, w // 2,
                              wd.shape[1], in_stride=2)
                else:
                    idt = t
                y = self._conv(blk["conv1"], t, h, w)
                h, w = h // s, w // s
                t = self._conv(blk["conv2"], y, h, w, res1=idt, act="relu")
            feats.append((t, h, w))
        (x2, h2, w2), (x3, h3, w3), (x4, h4, w4), (out, h5, w5) = feats
        # shortcut branches: conv -> ReLU -> BN twice (model.py:452-456)
        fea = []
        for s, (z, zh, zw) in zip(p["shortcut"], ((x, H, W), (x1, H // 2, W // 2), (x2, h2, w2), (x3, h3, w3), (x4, h4, w4))):
            y = self._conv(s["c1"], z, zh, zw, scale=s["s1"], shift=s["t1"])
            fea.append(self._conv(s["c2"], y, zh, zw, scale=s["s2"], shift=s["t2"]))
        # decoder (model.py:274-324, 505-515)
        t, h, w = out, h5, w5
        for blocks, skip in zip(p["dec"], (fea[4], fea[3], fea[2], fea[1])):
            b0, b1 = blocks
            up = self._conv(b0["up"], t, h, w)                  # 1x1 at low resolution; nearest x2 happens in the read
            y = self._convt(b0["conv1"], t, h, w)
            h, w = 2 * h, 2 * w
            t = self._conv(b0["conv2"], y, h, w, res1=up, res1_shift=1, act="leaky")
            y = self._conv(b1["conv1"], t, h, w)
            t = self._conv(b1["conv2"], y, h, w, res1=t, act="leaky", res2=skip)
        t = self._convt(p["head_up"], t, h, w, res2=fea[0])
        self.pred = torch.empty((self.batch, 1, H, W), dtype=torch.float32, device=self.device)
        hw = p["head"].w                                         # (1, 32, 3, 3)
        taps, wk = conv_taps(hw, 1)
        self._add(t, self.pred, taps, wk, p["head"].b, H, W, hw.shape[1], H, W, 1, kind=1)

    # ------------------------------------------------------------------ run
    @property
    def num_layers(self):
        return len(self._descs)

    def forward(self, x8=None):
        """x8: (H, W, 8) float32 NHWC (channel 7 zero) - or None when the caller wrote self.x8 directly.
        Returns self.pred, (1, 1, H, W)."""
        if x8 is not None and x8.data_ptr() != self.x8.data_ptr():
            self.x8.copy_(x8.view(self.batch * self.H, self.W, 8))
        st = torch.cuda.current_stream().cuda_stream
        if self._graph:
            rc = nv.load().usc_conv_graph_launch(self._graph, st)
        else:
            rc = nv.load().usc_conv_sequence_f32(self._arr, len(self._descs), st)
        if rc != 0:
            raise nv.NativeError(f"conv launch failed ({rc}): {nv.load().usc_conv_last_error().decode()}")
        if getattr(self, "_launch_mode", "stream") == "runs" and self.check_runs:
            self.raise_on_broken_handover()
        return self.pred

    __call__ = forward

    check_runs = True          # "runs" mode only: every forward is followed by a (blocking) look at the hand-over counters

    def raise_on_broken_handover(self):
        """"runs" mode: a workgroup whose bounded poll of a producer's arrival counter ran out has gone on with possibly stale activations
        (it must not hang the GPU) and counted that in the handle - a prediction computed that way must not be used.  Blocks (reads device
        memory), which is why the mode's forward() is synchronous unless check_runs is switched off by a caller that checks itself."""
        import ctypes as C
        bad = C.c_uint(0)
        rc = nv.load().usc_conv_graph_run_info(self._graph, None, None, C.byref(bad))
        if rc != 0:
            raise nv.NativeError(f"usc_conv_graph_run_info failed ({rc}): {nv.load().usc_conv_last_error().decode()}")
        if bad.value:
            raise nv.NativeError(f"USC_CONV_LAUNCH_RUNS: {bad.value} arrival-counter poll(s) of this network have run out - a layer hand-over "
                                 f"inside a multi-layer launch is broken and the prediction may be built on stale activations; use "
                                 f"set_launch_mode('stream')")

    def set_launch_mode(self, mode):
        """"stream" (default: the planned launches one by one on the stream), "graph" (one hipGraphLaunch per forward) or "runs" (as
        stream, runs of same-shape Winograd layers as one launch each: USC_CONV_LAUNCH_RUNS)."""
        rc = nv.load().usc_conv_graph_set_mode(self._graph, {"stream": 0, "graph": 1, "runs": 2}[mode])
        if rc != 0:
            raise nv.NativeError(f"usc_conv_graph_set_mode failed ({rc})")
        self._launch_mode = mode     # ("runs": ONE launch of this network in flight at a time - queue and arrival counters belong to the handle)

    def forward_serial(self):
        """The same layers launched one after the other on the current stream (no graph, no branch overlap): what the per-layer
        profile table and bench.py's roofline time."""
        rc = nv.load().usc_conv_sequence_f32(self._arr, len(self._descs), torch.cuda.current_stream().cuda_stream)
        if rc != 0:
            raise nv.NativeError(f"conv launch failed ({rc}): {nv.load().usc_conv_last_error().decode()}")
        return self.pred
