"""The space-time memory network on the hand-written f32-MFMA convolution kernel (csrc/usc_conv.hip).

memorize : x8 (R, G, B, object, background) -> im2col(5 ch, K 245 -> 256) -> ResNet-50 trunk -> Key / Value 3x3 -> one
           (hw, 128) key and one (hw, 512) value slot of the device-resident memory bank (the reference bounces every key /
           value through the host each step, stm/agent.py:138-139, 159-160).
segment  : x8 -> im2col(3 ch) -> trunk -> query key / value; memory read = two 1x1 layers of the same MFMA kernel whose
           operands are packed on the device from the bank (affinity = q K^T, row softmax, value read into channels
           [0, 512) of the decoder's 1024-channel input next to the query value); decoder with pre-activation ResBlocks
           (conv(relu(x)) is a flag of the kernel's LDS staging, the residual adds and the bilinear skip are epilogue
           operands) -> (h/4, w/4, 32) logits for usc_stm_post.
"""
import math

import torch
import torch.nn.functional as F

from ..binseg.schp_net import FConv
from ..convnet import ConvGraphNet, _out_size
from .net import KEYDIM, VALDIM, StmPlan


class _Part(ConvGraphNet):
    """A ConvGraphNet that shares the split-K slab of its owner."""

    def __init__(self, owner):
        super().__init__(owner.device, slab_floats=16)
        self.slab = owner.slab


class HipStmNet:
    def __init__(self, state_dict, in_h, in_w, capacity=11, device="cuda:0"):
        if in_h % 16 or in_w % 16:
            raise ValueError("HipStmNet: input size must be a multiple of 16 (stm/agent.py:57 division)")
        from .. import _native as nv
        self.nv = nv
        nv.load()
        self.device = torch.device(device)
        self.in_h, self.in_w = in_h, in_w
        self.plan = StmPlan(state_dict)
        self.slab = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=self.device)
        self.h16, self.w16 = in_h // 16, in_w // 16
        self.hw = self.h16 * self.w16
        self.capacity = capacity
        self.kbank = torch.zeros((capacity * self.hw, KEYDIM), dtype=torch.float32, device=self.device)
        self.vbank = torch.zeros((capacity * self.hw, VALDIM), dtype=torch.float32, device=self.device)
        self.x8 = torch.zeros((in_h, in_w, 8), dtype=torch.float32, device=self.device)
        self._build_memorize()
        self._build_query()
        self._reads = {}
        self._build_decoder()

    # ---- shared pieces
    def _trunk(self, net, stem, layers, cuse, kpad):
        H, W, nv = self.in_h, self.in_w, self.nv
        sh, sw = _out_size(H, 7, 2, 3, 1), _out_size(W, 7, 2, 3, 1)
        cols = net._buf(sh, sw, kpad)
        net._glue(lambda: nv.im2col(self.x8, cuse, 7, 7, 2, 3, cols))
        w7 = stem.w.permute(0, 2, 3, 1).reshape(64, 49 * cuse)                 # k = (ky*7 + kx)*cuse + c
        t, h, w = net._conv(FConv(F.pad(w7, (0, kpad - 49 * cuse)).reshape(64, kpad, 1, 1), stem.b, act="relu"), cols, sh, sw, kpad)
        ph, pw = _out_size(h, 3, 2, 1, 1), _out_size(w, 3, 2, 1, 1)
        pooled = net._buf(ph, pw, 64)
        src = t
        net._glue(lambda: net.nv.maxpool3x3s2(src, pooled, batch=net.batch))
        t, h, w = pooled, ph, pw
        feats = []
        for blocks in layers:
            for blk in blocks:
                cin = t.shape[2]
                idt = net._conv(blk["down"], t, h, w, cin)[0] if blk["down"] is not None else t
                y, _, _ = net._conv(blk["c1"], t, h, w, cin)
                y, h2, w2 = net._conv(blk["c2"], y, h, w, y.shape[2])
                t, h, w = net._conv(blk["c3"], y, h2, w2, y.shape[2], res1=idt)
            feats.append((t, h, w))
        return feats

    def _build_memorize(self):
        p = self.plan
        net = self.mem = _Part(self)
        (_, _, _), (_, _, _), (r4, h, w) = self._trunk(net, p.m_stem, p.m_layers, 5, 256)
        assert (h, w) == (self.h16, self.w16)
        self.mkey, _, _ = net._conv(p.kv_m[0], r4, h, w, 1024)
        self.mval, _, _ = net._conv(p.kv_m[1], r4, h, w, 1024)
        net._flush()

    def _build_query(self):
        p = self.plan
        net = self.qry = _Part(self)
        (self.r2, self.h4, self.w4), (self.r3, self.h8, self.w8), (r4, h, w) = self._trunk(net, p.q_stem, p.q_layers, 3, 160)
        self.qkey, _, _ = net._conv(p.kv_q[0], r4, h, w, 1024)
        self.cat = net._buf(h, w, 2 * VALDIM)                                   # [memory read | query value]
        net._conv(p.kv_q[1], r4, h, w, 1024, out=self.cat, out_ld=2 * VALDIM, out_c0=VALDIM)
        net._flush()

    def _read_for(self, T):
        """The memory read for T stored frames (stm/model.py:137-156) as two device-operand 1x1 layers."""
        if T not in self._reads:
            nv, hw = self.nv, self.hw
            n = T * hw
            npad = (n + 31) // 32 * 32
            net = _Part(self)
            w1 = torch.zeros((KEYDIM // 4, npad, 4), dtype=torch.float32, device=self.device)
            w2 = torch.zeros((npad // 4, VALDIM, 4), dtype=torch.float32, device=self.device)
            aff = net._buf(self.h16, self.w16, npad)
            zero_b1 = torch.zeros(npad, dtype=torch.float32, device=self.device)
            zero_b2 = torch.zeros(VALDIM, dtype=torch.float32, device=self.device)
            net._keep += [w1, w2, zero_b1, zero_b2]
            net._glue(lambda: nv.pack_kn(self.kbank, KEYDIM, n, 1, KEYDIM, w1))             # (k = channel, n = memory position)
            shape1 = FConv(torch.zeros(npad, KEYDIM, 1, 1), torch.zeros(npad))              # shapes only: operands are w1 / zero_b1
            net._conv(shape1, self.qkey, self.h16, self.w16, KEYDIM, out=aff, wpk_buf=w1, bias_buf=zero_b1)
            net._glue(lambda: (nv.row_softmax(aff.view(hw, npad), n, math.sqrt(KEYDIM)),
                               nv.pack_kn(self.vbank, n, VALDIM, VALDIM, 1, w2)))           # (k = memory position, n = channel)
            shape2 = FConv(torch.zeros(VALDIM, npad, 1, 1), torch.zeros(VALDIM))
            net._conv(shape2, aff, self.h16, self.w16, npad, out=self.cat, out_ld=2 * VALDIM, out_c0=0, wpk_buf=w2, bias_buf=zero_b2)
            net._flush()
            self._reads[T] = net
        return self._reads[T]

    def _build_decoder(self):
        p, nv = self.plan, self.nv
        net = self.dec = _Part(self)
        h, w = self.h16, self.w16

        def resblock(pair, x, hh, ww, res2=None, act=None):
            r, _, _ = net._conv(pair[0], x, hh, ww, 256, in_relu=True, act="relu")        # relu(conv1(relu(x))) feeds conv2
            return net._conv(pair[1], r, hh, ww, 256, res1=x, res2=res2, act=act)[0]      # x + conv2(.) [+ skip]

        x, _, _ = net._conv(p.convFM, self.cat, h, w, 2 * VALDIM)
        m4 = resblock(p.ResMM, x, h, w)

        def refine(rf, f, fh, fw, cin, pm, last=False):
            up = net._buf(fh, fw, 256)
            net._glue(lambda: nv.upsample_nc(pm, up, 0))
            s, _, _ = net._conv(rf["convFS"], f, fh, fw, cin)
            m = resblock(rf["ResFS"], s, fh, fw, res2=up)                                  # ResFS(s) + up(pm)
            return resblock(rf["ResMM"], m, fh, fw)
        m3 = refine(p.RF3, self.r3, self.h8, self.w8, 512, m4)
        m2 = refine(p.RF2, self.r2, self.h4, self.w4, 256, m3)
        self.p2, _, _ = net._conv(p.pred2, m2, self.h4, self.w4, 256, in_relu=True, cout_pad=32)
        net._flush()

    # ---- execution
    def memorize(self, slot):
        """Encode self.x8 (frame + masks) into memory slot `slot` of the bank."""
        self.mem.run()
        hw = self.hw
        self.kbank[slot * hw:(slot + 1) * hw].copy_(self.mkey.view(hw, KEYDIM))
        self.vbank[slot * hw:(slot + 1) * hw].copy_(self.mval.view(hw, VALDIM))

    def segment(self, T):
        """Decode self.x8 (frame) against the first T slots of the bank -> (h/4, w/4, 32) logits buffer (classes 0, 1)."""
        self.qry.run()
        self._read_for(T).run()
        self.dec.run()
        return self.p2
