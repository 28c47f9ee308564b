"""ConvGraphNet: builder / runner shared by the networks that execute on csrc/usc_conv.hip (the f32-MFMA implicit-GEMM
kernel).  A network is a list of segments: runs of fused convolution layers, each captured as one hipGraph
(usc_conv_graph_create), separated by "glue" callables (pooling, bilinear resampling into concat slices) that are
enqueued on the same stream.  All activations are NHWC float32 buffers allocated once."""
import ctypes as C
import sys

import torch
import torch.nn.functional as F


def _out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


class ConvGraphNet:
    ACTS = {"none": 0, "relu": 1, "leaky": 2, "leaky01": 3}

    def __init__(self, device="cuda:0", slab_floats=64 * 1024 * 1024, batch=1, plan_batch=1):
        """batch > 1: every buffer holds `batch` images stacked along y ((batch * h, w, c)); layers carry `batch` to the kernel.
        plan_batch: the batch the kernel plans its tiles / K partition for; networks built with the same
        plan_batch give bit-identical per-image results whatever their batch."""
        from . import _native as nv
        from .vmatting.net_hip import ConvDesc, conv_taps, pack_k_major, pack_wino
        self.nv, self.ConvDesc, self.conv_taps, self.pack, self.pack_wino = nv, ConvDesc, conv_taps, pack_k_major, pack_wino
        if C.sizeof(ConvDesc) != nv.load().usc_conv_desc_bytes():
            raise nv.NativeError("ConvDesc (vmatting/net_hip.py) does not match usc_conv_desc of the built library: rebuild it")
        self.device = torch.device(device)
        self.batch = batch
        self.plan_batch = max(1, int(plan_batch))
        self._keep = []
        self.slab = torch.empty(slab_floats, dtype=torch.float32, device=self.device)     # split-K partial sums
        self._segments = []            # list of ("conv", graph handle, ctypes array) | ("glue", callable, None)
        self._cur = []

    # ---- buffers / descriptors
    def _buf(self, h, w, c):
        t = torch.empty((self.batch * h, w, c), dtype=torch.float32, device=self.device)
        self._keep.append(t)
        return t

    def _dev(self, t):
        t = t.detach().to(self.device, torch.float32).contiguous()
        self._keep.append(t)
        return t

    def nchw(self, buf):
        """(batch * H, W, C) NHWC buffer viewed as a channels-last (batch, C, H, W) tensor (no copy)."""
        bh, w, c = buf.shape
        return buf.view(self.batch, bh // self.batch, w, c).permute(0, 3, 1, 2)

    def image(self, buf, b):
        """Image b of a stacked buffer (a contiguous slice)."""
        h = buf.shape[0] // self.batch
        return buf[b * h:(b + 1) * h]

    def _conv(self, c, x, IH, IW, cin_buf, out=None, res1=None, act=None, cout_pad=None, out_ld=0, out_c0=0, bias_buf=None,
              res2=None, in_relu=False, wpk_buf=None):
        """c: folded conv (w (Cout, Cin, k, k), b, stride, pad, dil, act).  x: (IH, IW, cin_buf) buffer; cin_buf >= Cin (the
        extra channels get zero weights).  bias_buf: a device buffer that replaces c.b (a bias computed on the device).
        res2: added after the activation; in_relu: the layer computes conv(relu(x)); wpk_buf: a device buffer already in the
        kernel's [K/4][Cout][4] layout that replaces the packed c.w (operands produced on the device, see usc_pack_kn_f32)."""
        w = c.w
        co, ci, k, _ = w.shape
        if cin_buf != ci:
            w = F.pad(w, (0, 0, 0, 0, 0, cin_buf - ci))
        b = c.b
        if cout_pad and cout_pad != co:
            w = F.pad(w, (0, 0, 0, 0, 0, 0, 0, cout_pad - co))
            b = F.pad(b, (0, cout_pad - co))
            co = cout_pad
        taps = [((ky - (k // 2)) * c.dil, (kx - (k // 2)) * c.dil) for ky in range(k) for kx in range(k)] if k > 1 else [(0, 0)]
        if k > 1:
            assert c.pad == c.dil * (k // 2)
        wk = w.permute(2, 3, 1, 0).reshape(k * k * cin_buf, co)
        OH, OW = _out_size(IH, k, c.stride, c.pad, c.dil), _out_size(IW, k, c.stride, c.pad, c.dil)
        out = out if out is not None else self._buf(OH, OW, co)
        d = self.ConvDesc()
        d.inp, d.out = x.data_ptr(), out.data_ptr()
        d.wpk = wpk_buf.data_ptr() if wpk_buf is not None else self._dev(self.pack(wk.float())).data_ptr()
        if wpk_buf is None and k == 3 and c.stride == 1 and c.pad == c.dil and cin_buf % 8 == 0 and IH % c.dil == 0 and IW % c.dil == 0:
            d.wino = self._dev(self.pack_wino(w.float())).data_ptr()     # the library may run the layer in the Winograd form
        d.bias = bias_buf.data_ptr() if bias_buf is not None else self._dev(b).data_ptr()
        d.res1 = res1.data_ptr() if res1 is not None else None
        d.res2 = res2.data_ptr() if res2 is not None else None
        d.in_relu = 1 if in_relu else 0
        B = self.batch
        d.IH, d.IW, d.Cin, d.OH, d.OW, d.OHf, d.OWf, d.Cout = IH, IW, cin_buf, B * OH, OW, B * OH, OW, co
        d.batch = B
        d.plan_batch = self.plan_batch
        d.in_stride, d.out_stride = c.stride, 1
        d.act = self.ACTS[act or c.act]
        d.ntaps, d.kind, d.nphase = len(taps), 0, 1
        d.out_ld, d.out_c0 = out_ld, out_c0
        # one shared split-K scratch: the nets are chains, and the graph builder orders layers that share a slab
        d.slab, d.slab_bytes = self.slab.data_ptr(), self.slab.numel() * 4
        for i, (dy, dx) in enumerate(taps):
            d.dy[i], d.dx[i] = dy, dx
        self._cur.append(d)
        return out, OH, OW

    def _flush(self):
        if self._cur:
            arr = (self.ConvDesc * len(self._cur))(*self._cur)
            handle = C.c_void_p()
            rc = self.nv.load().usc_conv_graph_create(arr, len(self._cur), C.byref(handle))
            if rc != 0:
                raise self.nv.NativeError(f"usc_conv_graph_create failed ({rc}): {self.nv.load().usc_conv_last_error().decode()}")
            self._segments.append(("conv", handle, arr))
            self._cur = []

    def _glue(self, fn):
        self._flush()
        self._segments.append(("glue", fn, None))

    def run(self):
        """Enqueue every segment on the current stream."""
        st = torch.cuda.current_stream().cuda_stream
        lib = self.nv.load()
        for kind, a, _ in self._segments:
            if kind == "conv":
                rc = lib.usc_conv_graph_launch(a, st)
                if rc != 0:
                    raise self.nv.NativeError(f"conv launch failed ({rc}): {lib.usc_conv_last_error().decode()}")
            else:
                a()

    def set_launch_mode(self, mode):
        """"stream" (default) or "graph": how every conv segment is enqueued (usc_conv_graph_set_mode)."""
        for kind, a, _ in self._segments:
            if kind == "conv":
                self.nv.load().usc_conv_graph_set_mode(a, {"stream": 0, "graph": 1, "runs": 2}[mode])

    def plan_gflop(self):
        """2 * M * N * K over every conv launch of the plan (GFLOP per run(), padded channels included; a Winograd layer counts its
        algorithmic multiplications, not the 2.25x fewer it issues): the numerator of a roofline figure for this network."""
        total = 0
        for kind, _, arr in self._segments:
            if kind != "conv":
                continue
            for d in arr:
                total += 2 * d.OH * d.OW * max(1, d.nphase) * d.Cout * d.ntaps * d.Cin
        return total / 1e9

    def n_layers(self):
        return sum(len(arr) for kind, _, arr in self._segments if kind == "conv")

    def close(self):
        """Destroy the conv graphs (usc_conv_graph_destroy drains the device first: a launch may still be queued) and only then
        let go of the buffers whose addresses are baked into the graph nodes."""
        segs, self._segments = getattr(self, "_segments", []), []
        for kind, a, _ in segs:
            if kind == "conv" and a:
                self.nv.load().usc_conv_graph_destroy(a)
        self._keep = []

    def __del__(self):
        try:
            if sys.is_finalizing():        # the HIP runtime may already be gone: the process frees everything anyway
                return
            self.close()
        except Exception:
            pass
