"""The interactive-segmentation network on the HIP kernels: every dense layer on csrc/usc_conv.hip (f32 MFMA implicit GEMM /
Winograd), the four depthwise 3x3 layers of the two separable-conv heads on usc_dwconv3x3_nhwc_f32 (9 multiply-adds per output:
not a contraction, kept off the matrix cores), the input stage (image normalisation, click distance maps, rgb_conv) and the output
stage (upsample, flip mean, sigmoid, inverse resize, threshold) in csrc/usc_iseg.hip.  Plan and reference citations: net.py.

    x4      (B * H, W, 4)      output of usc_iseg_pre: the image through rgb_conv, and its mirror when B = 2
    stem    3 x 3x3 (stride 2, 1, 1), max-pool 3x3 / 2 (torch glue)
    layers  resnet50_v1s, stride on conv2, layer3 / layer4 dilated: output stride 8
    ASPP    1x1 + three dilated 3x3 (12, 24, 36) into slices of one concat buffer; the image-pooling branch is constant over
            the image, so it becomes a per-image bias of the projection layer (two M = 1 layers produce it on the device)
    decoder upsample (align_corners) into [0, 128) of the 160-channel concat, skip projection into [128, 160); two separable
            convs (depthwise kernel + pointwise MFMA layer), 1x1; SepConvHead: two separable convs, 1x1 -> logits32
"""
import torch
import torch.nn.functional as F

from ..binseg.schp_net import FConv
from ..convnet import ConvGraphNet, _out_size
from .net import CH, IsegPlan


class HipIsegNet(ConvGraphNet):
    def __init__(self, state_dict, H, W, device="cuda:0", batch=2):
        """batch = 2: the image and its mirror (the predictor's flip test-time augmentation), stacked along y."""
        super().__init__(device, slab_floats=64 * 1024 * 1024, batch=batch, plan_batch=batch)
        self.H, self.W = H, W
        self.plan = IsegPlan(state_dict)
        self._build()

    def rgb_params(self):
        p = self.plan
        return torch.cat([p.rgb_w1.reshape(-1), p.rgb_b1, p.rgb_w2.reshape(-1), p.rgb_b2]).numpy()

    def _sepconv(self, sc, x, h, w, cin):
        nv = self.nv
        w9c = self._dev(sc.dw.view(cin, 9).t().contiguous())                # (9, C) tap-major
        y = self._buf(h, w, cin)
        self._glue(lambda: nv.dwconv3x3(x, w9c, y, batch=self.batch))
        return self._conv(sc.pw, y, h, w, cin)[0]

    def _build(self):
        p, H, W, nv = self.plan, self.H, self.W, self.nv
        self.x4 = self._buf(H, W, 4)
        t, h, w = self._conv(p.stem[0], self.x4, H, W, 4)
        t, h, w = self._conv(p.stem[1], t, h, w, 64)
        t, h, w = self._conv(p.stem[2], t, h, w, 64)
        ph, pw = _out_size(h, 3, 2, 1, 1), _out_size(w, 3, 2, 1, 1)
        pooled = self._buf(ph, pw, 128)
        src = t
        self._glue(lambda: self.nv.maxpool3x3s2(src, pooled, batch=self.batch))
        t, h, w = pooled, ph, pw
        c1 = None
        for li, blocks in enumerate(p.layers):
            for blk in blocks:
                cin = t.shape[2]
                idt = self._conv(blk["down"], t, h, w, cin)[0] if blk["down"] is not None else t
                y, _, _ = self._conv(blk["c1"], t, h, w, cin)
                y, h2, w2 = self._conv(blk["c2"], y, h, w, y.shape[2])
                t, h, w = self._conv(blk["c3"], y, h2, w2, y.shape[2], res1=idt)
            if li == 0:
                c1, lh, lw = t, h, w
        x5, h5, w5 = t, h, w
        # ---- ASPP (deeplab_v3.py:148-215)
        cat = self._buf(h5, w5, 4 * CH)
        self._conv(p.aspp0, x5, h5, w5, 2048, out=cat, out_ld=4 * CH, out_c0=0)
        for i, c in enumerate(p.aspp_d):
            self._conv(c, x5, h5, w5, 2048, out=cat, out_ld=4 * CH, out_c0=CH * (i + 1))
        gap = self._buf(1, 1, 2048)
        self._glue(lambda: nv.adaptive_avgpool(x5, 0, 2048, gap, batch=self.batch))
        pool_feat, _, _ = self._conv(p.aspp_pool, gap, 1, 1, 2048)
        wp = p.aspp_proj.w                                                  # (128, 640, 1, 1), BN folded
        dyn_bias, _, _ = self._conv(FConv(wp[:, 4 * CH:].contiguous(), p.aspp_proj.b), pool_feat, 1, 1, CH)
        bias_map = self._buf(h5, w5, CH)
        self._glue(lambda: self.nchw(bias_map).copy_(self.nchw(dyn_bias).expand(self.batch, CH, h5, w5)))
        aspp, _, _ = self._conv(FConv(wp[:, :4 * CH].contiguous(), torch.zeros(CH), act="relu"), cat, h5, w5, 4 * CH, res1=bias_map)
        # ---- decoder (deeplab_v3.py:82-95, 111-145): [ASPP upsampled 128 | skip 32]
        dcat = self._buf(lh, lw, CH + 32)
        self._glue(lambda: nv.upsample_ac(aspp, dcat, 0, batch=self.batch))
        self._conv(p.skip, c1, lh, lw, 256, out=dcat, out_ld=CH + 32, out_c0=CH)
        y = self._sepconv(p.dl_head[0], dcat, lh, lw, CH + 32)
        y = self._sepconv(p.dl_head[1], y, lh, lw, CH)
        y, _, _ = self._conv(p.dl_head_out, y, lh, lw, CH)
        # ---- SepConvHead (basic_blocks.py:44-86)
        y = self._sepconv(p.head[0], y, lh, lw, CH)
        y = self._sepconv(p.head[1], y, lh, lw, CH // 2)
        self.logits32, _, _ = self._conv(p.head_out, y, lh, lw, CH // 2, cout_pad=32)
        self._flush()
        self.out_h, self.out_w = lh, lw

    def forward(self, x4=None):
        if x4 is not None and x4.data_ptr() != self.x4.data_ptr():
            self.x4.copy_(x4)
        self.run()
        return self.logits32

    __call__ = forward
