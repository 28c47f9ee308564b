"""DeepLabv3+ / ResNet-50 (output stride 8) as a folded inference plan: reference unscreen/binseg/deeplab_model.py:4-14
-> deeplab/modeling.py:6-33 (`_segm_resnet`: replace_stride_with_dilation [F, T, T], ASPP rates 12/24/36), deeplab/_deeplab.py:29-60
(`DeepLabHeadV3Plus`), :141-204 (`ASPPConv`, `ASPPPooling`, `ASPP`), deeplab/utils.py:8-27 (final bilinear to the input size),
deeplab/backbone/resnet.py:102-271 (`Bottleneck`, `ResNet`, stride on conv2).

* `deeplab_checkpoint_layout()` reproduces the reference state-dict keys (SegAgent loads the file directly,
  binseg/agent.py:58: no wrapper key, no `module.` prefix).
* `DeepLabPlan` folds every BatchNorm into its conv; `DeepLabNetTorch` runs the plan on torch ops (the CPU oracle of the
  network); `HipDeepLabNet` runs every convolution of one 513 x 513 crop on csrc/usc_conv.hip:
    - the 7x7/2 stem as im2col (K = 147 -> 160) + a 1x1 layer,
    - the four parallel ASPP branches write straight into their slices of the concat buffer,
    - the image-pooling branch is constant over the crop, so it is folded into the bias of the ASPP projection
      (two M = 1 layers produce that bias on the device),
    - the decoder concat is [48 -> 64 low-level | 256 upsampled ASPP] = 320 channels (zero weights on the pad).
  Its output is the classifier's (129, 129, 32) logits buffer (classes in channels 0, 1); the model's last x4 upsample is
  fused into the crop merge (csrc/usc_segagent.hip).
"""
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

from ..convnet import ConvGraphNet, _out_size
from .schp_net import FConv, _bn, _fold

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
STRIDES = (1, 2, 1, 1)            # layer3 / layer4 trade their stride for dilation (output stride 8)
ASPP_RATES = (12, 24, 36)
NUM_CLASSES = 2


def _block_dilations():
    """(first-block dilation, other-blocks dilation) per layer (resnet.py:214-240 _make_layer)."""
    out, dil = [], 1
    for li, dilate in enumerate((False, False, True, True)):
        prev = dil
        if dilate:
            dil *= 2
        out.append((prev, dil))
    return out


def deeplab_checkpoint_layout():
    """Ordered (key, shape) list of the reference model's state dict."""
    k = [("backbone.conv1.weight", (64, 3, 7, 7))] + _bn("backbone.bn1", 64)
    inpl = 64
    for li, (n, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(n):
            p = f"backbone.layer{li}.{b}."
            k += [(p + "conv1.weight", (planes, inpl, 1, 1))] + _bn(p + "bn1", planes)
            k += [(p + "conv2.weight", (planes, planes, 3, 3))] + _bn(p + "bn2", planes)
            k += [(p + "conv3.weight", (planes * 4, planes, 1, 1))] + _bn(p + "bn3", planes * 4)
            if b == 0:
                k += [(p + "downsample.0.weight", (planes * 4, inpl, 1, 1))] + _bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    k += [("classifier.project.0.weight", (48, 256, 1, 1))] + _bn("classifier.project.1", 48)
    k += [("classifier.aspp.convs.0.0.weight", (256, 2048, 1, 1))] + _bn("classifier.aspp.convs.0.1", 256)
    for i in (1, 2, 3):
        k += [(f"classifier.aspp.convs.{i}.0.weight", (256, 2048, 3, 3))] + _bn(f"classifier.aspp.convs.{i}.1", 256)
    k += [("classifier.aspp.convs.4.1.weight", (256, 2048, 1, 1))] + _bn("classifier.aspp.convs.4.2", 256)
    k += [("classifier.aspp.project.0.weight", (256, 1280, 1, 1))] + _bn("classifier.aspp.project.1", 256)
    k += [("classifier.classifier.0.weight", (256, 304, 3, 3))] + _bn("classifier.classifier.1", 256)
    k += [("classifier.classifier.3.weight", (NUM_CLASSES, 256, 1, 1)), ("classifier.classifier.3.bias", (NUM_CLASSES,))]
    return k


_SYNTH = {}


def deeplab_synthetic_state_dict(seed=0):
    """Keyed-deterministic weights (no trained checkpoint exists offline); the last BN of every bottleneck gets a small
    gamma so 16 residual blocks of random weights keep the activations bounded, and the class-1 bias is calibrated on a
    seeded probe so that both classes appear."""
    if seed not in _SYNTH:
        sd = OrderedDict()
        for key, shape in deeplab_checkpoint_layout():
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed ^ 0x3d1b) & 0x7FFFFFFF)
            if key.endswith("num_batches_tracked"):
                t = torch.tensor(1000, dtype=torch.long)
            elif key.endswith("running_var"):
                t = 0.6 + 0.8 * torch.rand(shape, generator=g)
            elif key.endswith("running_mean"):
                t = 0.1 * torch.randn(shape, generator=g)
            elif key.endswith("bn3.weight"):
                t = 0.15 + 0.1 * torch.rand(shape, generator=g)
            elif key.endswith(".weight") and len(shape) == 1:
                t = 0.8 + 0.4 * torch.rand(shape, generator=g)
            elif key.endswith(".bias"):
                t = 0.05 * torch.randn(shape, generator=g)
            else:
                fan_in = shape[1] * shape[2] * shape[3]
                t = torch.randn(shape, generator=g) * (1.6 / fan_in) ** 0.5
                if key == "classifier.classifier.3.weight":
                    t = t * 4.0
            sd[key] = t
        from ..synth import SyntheticClip
        fr = SyntheticClip(128, 160, 1, config_id=2, device="cpu").frame(0)["frame"].float()
        probe = fr.permute(2, 0, 1)[None].flip(1) / 255.
        probe = (probe - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        o = DeepLabNetTorch(sd)(probe)
        sd["classifier.classifier.3.bias"][1] -= (o[:, 1] - o[:, 0]).flatten().median()
        _SYNTH[seed] = sd
    return OrderedDict((k, v.clone()) for k, v in _SYNTH[seed].items())


class DeepLabPlan:
    def __init__(self, sd):
        self.stem = _fold(sd, "backbone.conv1.weight", "backbone.bn1", stride=2, pad=3, act="relu")
        self.layers = []
        for li, (n, stride, (d_first, d_rest)) in enumerate(zip(LAYERS, STRIDES, _block_dilations()), start=1):
            blocks = []
            for b in range(n):
                p = f"backbone.layer{li}.{b}."
                dil = d_first if b == 0 else d_rest
                st = stride if b == 0 else 1
                blk = {"c1": _fold(sd, p + "conv1.weight", p + "bn1", act="relu"),
                       "c2": _fold(sd, p + "conv2.weight", p + "bn2", stride=st, pad=dil, dil=dil, act="relu"),
                       "c3": _fold(sd, p + "conv3.weight", p + "bn3", act="relu"),          # relu after the residual add
                       "down": _fold(sd, p + "downsample.0.weight", p + "downsample.1", stride=st) if b == 0 else None}
                blocks.append(blk)
            self.layers.append(blocks)
        c = "classifier."
        self.low = _fold(sd, c + "project.0.weight", c + "project.1", act="relu")
        self.aspp0 = _fold(sd, c + "aspp.convs.0.0.weight", c + "aspp.convs.0.1", act="relu")
        self.aspp_d = [_fold(sd, c + f"aspp.convs.{i}.0.weight", c + f"aspp.convs.{i}.1", pad=r, dil=r, act="relu")
                       for i, r in zip((1, 2, 3), ASPP_RATES)]
        self.aspp_pool = _fold(sd, c + "aspp.convs.4.1.weight", c + "aspp.convs.4.2", act="relu")
        self.aspp_proj = _fold(sd, c + "aspp.project.0.weight", c + "aspp.project.1", act="relu")
        self.cls0 = _fold(sd, c + "classifier.0.weight", c + "classifier.1", pad=1, act="relu")
        self.cls1 = FConv(sd[c + "classifier.3.weight"].float().clone(), sd[c + "classifier.3.bias"].float().clone())


def _interp_nc(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class DeepLabNetTorch:
    """The folded plan on torch ops: forward((N, 3, H, W) normalised RGB) -> (N, 2, ~H/4, ~W/4) classifier logits;
    `full()` adds the model's last bilinear upsample to the input size (deeplab/utils.py:20-21)."""

    def __init__(self, state_dict, device="cpu"):
        self.plan = DeepLabPlan(state_dict)
        self.device = torch.device(device)

    @torch.no_grad()
    def forward(self, x):
        p = self.plan
        x = p.stem.torch_forward(x.to(self.device, torch.float32))
        x = F.max_pool2d(x, 3, 2, 1)
        low = None
        for li, blocks in enumerate(p.layers):
            for blk in blocks:
                idt = blk["down"].torch_forward(x) if blk["down"] is not None else x
                y = blk["c2"].torch_forward(blk["c1"].torch_forward(x))
                x = blk["c3"].torch_forward(y, residual=idt)
            if li == 0:
                low = x
        size = x.shape[2:]
        res = [p.aspp0.torch_forward(x)] + [c.torch_forward(x) for c in p.aspp_d]
        res.append(_interp_nc(p.aspp_pool.torch_forward(F.adaptive_avg_pool2d(x, 1)), size))
        y = p.aspp_proj.torch_forward(torch.cat(res, 1))
        lowp = p.low.torch_forward(low)
        y = _interp_nc(y, lowp.shape[2:])
        return p.cls1.torch_forward(p.cls0.torch_forward(torch.cat([lowp, y], 1)))

    __call__ = forward

    @torch.no_grad()
    def full(self, x):
        return _interp_nc(self.forward(x), x.shape[2:])


class HipDeepLabNet(ConvGraphNet):
    """One crop per call: x4 (H, W, 4) float32 NHWC (R, G, B, 0) -> logits32 (H/4, W/4, 32) buffer."""

    def __init__(self, state_dict, H=513, W=513, device="cuda:0", batch=1, plan_batch=1):
        """batch > 1: that many crops per forward, stacked along y in x4 / logits32 (bit-identical to single-crop forwards)."""
        super().__init__(device, slab_floats=64 * 1024 * 1024, batch=batch, plan_batch=plan_batch)
        self.H, self.W = H, W
        self.plan = DeepLabPlan(state_dict)
        self._build()

    def _build(self):
        p, H, W, nv = self.plan, self.H, self.W, self.nv
        self.x4 = self._buf(H, W, 4)
        # ---- stem: 7x7/2 over (R, G, B) as a [pixels][160] patch matrix times a (160, 64) weight
        sh, sw = _out_size(H, 7, 2, 3, 1), _out_size(W, 7, 2, 3, 1)
        cols = self._buf(sh, sw, 160)
        self._glue(lambda: [nv.im2col(self.image(self.x4, b), 3, 7, 7, 2, 3, self.image(cols, b)) for b in range(self.batch)])
        w7 = p.stem.w.permute(0, 2, 3, 1).reshape(64, 147)                 # k = (ky*7 + kx)*3 + c, as the patch matrix
        stem1x1 = FConv(F.pad(w7, (0, 13)).reshape(64, 160, 1, 1), p.stem.b, act="relu")
        t, h, w = self._conv(stem1x1, cols, sh, sw, 160)
        ph, pw = _out_size(h, 3, 2, 1, 1), _out_size(w, 3, 2, 1, 1)
        pooled = self._buf(ph, pw, 64)
        src = t
        self._glue(lambda: self.nv.maxpool3x3s2(src, pooled, batch=self.batch))
        t, h, w = pooled, ph, pw
        low = None
        for li, blocks in enumerate(p.layers):
            for blk in blocks:
                cin = t.shape[2]
                idt = self._conv(blk["down"], t, h, w, cin)[0] if blk["down"] is not None else t
                y, _, _ = self._conv(blk["c1"], t, h, w, cin)
                y, h2, w2 = self._conv(blk["c2"], y, h, w, y.shape[2])
                t, h, w = self._conv(blk["c3"], y, h2, w2, y.shape[2], res1=idt)
            if li == 0:
                low, lh, lw = t, h, w
        x5, h5, w5 = t, h, w
        # ---- ASPP: four conv branches into one (h5, w5, 1024) buffer; the pooled branch becomes the projection's bias
        cat = self._buf(h5, w5, 1024)
        self._conv(p.aspp0, x5, h5, w5, 2048, out=cat, out_ld=1024, out_c0=0)
        for i, c in enumerate(p.aspp_d):
            self._conv(c, x5, h5, w5, 2048, out=cat, out_ld=1024, out_c0=256 * (i + 1))
        gap = self._buf(1, 1, 2048)
        self._glue(lambda: nv.adaptive_avgpool(x5, 0, 2048, gap, batch=self.batch))
        pool_feat, _, _ = self._conv(p.aspp_pool, gap, 1, 1, 2048)
        wp = p.aspp_proj.w                                                  # (256, 1280, 1, 1), BN folded
        dyn_bias, _, _ = self._conv(FConv(wp[:, 1024:].contiguous(), p.aspp_proj.b), pool_feat, 1, 1, 256)
        # per-crop bias: broadcast over the crop's pixels and added before the ReLU as the layer's first residual operand
        bias_map = self._buf(h5, w5, 256)
        self._glue(lambda: self.nchw(bias_map).copy_(self.nchw(dyn_bias).expand(self.batch, 256, h5, w5)))
        y, _, _ = self._conv(FConv(wp[:, :1024].contiguous(), torch.zeros(256), act="relu"), cat, h5, w5, 1024, res1=bias_map)
        # ---- decoder (deeplab/_deeplab.py:47-58): [low-level 48 | ASPP upsampled x4 256] -> 3x3 -> 1x1.  The ASPP half of the 3x3
        # conv multiplies an upsampled 33 x 33 map: conv(up(a)) = sum_taps shift_tap(up(a W_tap)) (see usc_upsample_taps_accum_f32),
        # so the nine tap matrices act at 33 x 33 (one 1x1 layer, 256 -> 9 x 256) and the 3x3 conv only sees the 48 (+16 zero)
        # low-level channels, with the accumulated ASPP map as its residual: 18 of the decoder's 23 GFLOP per crop less.
        low64 = self._buf(lh, lw, 64)
        self._conv(p.low, low, lh, lw, 256, out=low64, cout_pad=64)                 # 48 real channels + 16 zeros
        aspp_out = y
        w0 = p.cls0.w                                                        # (256, 304, 3, 3): [0:48] low, [48:304] ASPP
        wa = w0[:, 48:]                                                      # (co, 256, 3, 3) -> rows tap * 256 + co
        wq = wa.permute(2, 3, 0, 1).reshape(9 * 256, 256, 1, 1).contiguous()
        q, _, _ = self._conv(FConv(wq, wq.new_zeros(9 * 256), act="none"), aspp_out, h5, w5, 256)
        aspp_map = self._buf(lh, lw, 256)
        self._glue(lambda: nv.upsample_taps_accum(q, aspp_map, batch=self.batch, accumulate=False, align_corners=False))
        w64 = torch.cat([w0[:, :48], w0.new_zeros(256, 16, 3, 3)], 1).contiguous()
        y, _, _ = self._conv(FConv(w64, p.cls0.b, pad=1, act="relu"), low64, lh, lw, 64, res1=aspp_map)
        self.logits32, _, _ = self._conv(p.cls1, y, lh, lw, 256, cout_pad=32)
        self._flush()
        self.out_h, self.out_w = lh, lw

    def forward(self, x4=None):
        if x4 is not None and x4.data_ptr() != self.x4.data_ptr():
            self.x4.copy_(x4)
        self.run()
        return self.logits32

    __call__ = forward
