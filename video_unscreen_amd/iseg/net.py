"""The interactive-segmentation network of ISegAgent's no-BRS predictor as a folded inference plan: reference
unscreen/iseg/agent.py:27-50 (`build_predictor`: DistMapsModel(DeepLabV3Plus(resnet50, ch=128), SepConvHead(1, 128, 64, 2))),
iseg/brs/models/distmap.py:5-48 (`DistMapsModel`: rgb_conv on image + click distance maps, backbone, head, bilinear
align_corners up to the image size), :70-124 (`DistMaps`), iseg/brs/models/deeplab_v3.py:11-215 (`DeepLabV3Plus`, `_ASPP` rates
12 / 24 / 36, `_AsppPooling`, `_SkipProject`, `_DeepLabHead` of two depthwise-separable 3x3 + a 1x1), iseg/brs/models/
resnetv1b.py:61-338, 363-374 (`resnet50_v1s`: deep stem 3 x 3x3 (64, 64, 128), stride on conv2, layer3 / layer4 dilated 2 / 4 with the
first block of each at half that: output stride 8), iseg/brs/models/basic_blocks.py:44-121 (`SepConvHead`, `SeparableConv2d` =
depthwise 3x3 -> pointwise 1x1 -> BN -> ReLU).

* `iseg_checkpoint_layout()` reproduces the reference state-dict keys (the checkpoint is a bare dict loaded with strict=False,
  iseg/agent.py:20,44).
* `IsegPlan` folds every BatchNorm into the conv in front of it (the depthwise convs keep their weights: they have no BN of their
  own); `IsegNetTorch` runs the plan on torch ops (the CPU oracle of the network); `net_hip.HipIsegNet` runs it on the HIP kernels.
"""
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

from ..binseg.schp_net import BN_EPS, FConv, _bn, _fold

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)
STRIDES = (1, 2, 1, 1)
# (first block, other blocks) dilation of conv2 per layer (resnetv1b.py:239-318: dilation 2 -> first block 1, dilation 4 -> first 2)
DILATIONS = ((1, 1), (1, 1), (1, 2), (2, 4))
ASPP_RATES = (12, 24, 36)
CH = 128
NORM_RADIUS = 260.0               # DistMaps(norm_radius=260, spatial_scale=1.0), distmap.py:26


def iseg_checkpoint_layout():
    """Ordered (key, shape) list of the reference DistMapsModel's state dict."""
    k = [("rgb_conv.0.weight", (8, 5, 1, 1)), ("rgb_conv.0.bias", (8,))] + _bn("rgb_conv.2", 8) + \
        [("rgb_conv.3.weight", (3, 8, 1, 1)), ("rgb_conv.3.bias", (3,))]
    bb = "feature_extractor.backbone."
    k += [(bb + "conv1.0.weight", (64, 3, 3, 3))] + _bn(bb + "conv1.1", 64) + [(bb + "conv1.3.weight", (64, 64, 3, 3))] + _bn(bb + "conv1.4", 64) + \
         [(bb + "conv1.6.weight", (128, 64, 3, 3))] + _bn(bb + "bn1", 128)
    inpl = 128
    for li, (n, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(n):
            p = f"{bb}layer{li}.{b}."
            k += [(p + "conv1.weight", (planes, inpl, 1, 1))] + _bn(p + "bn1", planes)
            k += [(p + "conv2.weight", (planes, planes, 3, 3))] + _bn(p + "bn2", planes)
            k += [(p + "conv3.weight", (planes * 4, planes, 1, 1))] + _bn(p + "bn3", planes * 4)
            if b == 0:
                k += [(p + "downsample.0.weight", (planes * 4, inpl, 1, 1))] + _bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    fe = "feature_extractor."
    for i, cin in enumerate((CH + 32, CH)):
        k += [(f"{fe}head.block.{i}.body.0.weight", (cin, 1, 3, 3)), (f"{fe}head.block.{i}.body.1.weight", (CH, cin, 1, 1))] + _bn(f"{fe}head.block.{i}.body.2", CH)
    k += [(fe + "head.block.2.weight", (CH, CH, 1, 1)), (fe + "head.block.2.bias", (CH,))]
    k += [(fe + "skip_project.skip_project.0.weight", (32, 256, 1, 1))] + _bn(fe + "skip_project.skip_project.1", 32)
    k += [(fe + "aspp.concurent.0.0.weight", (CH, 2048, 1, 1))] + _bn(fe + "aspp.concurent.0.1", CH)
    for i in (1, 2, 3):
        k += [(f"{fe}aspp.concurent.{i}.0.weight", (CH, 2048, 3, 3))] + _bn(f"{fe}aspp.concurent.{i}.1", CH)
    k += [(fe + "aspp.concurent.4.gap.1.weight", (CH, 2048, 1, 1))] + _bn(fe + "aspp.concurent.4.gap.2", CH)
    k += [(fe + "aspp.project.0.weight", (CH, 5 * CH, 1, 1))] + _bn(fe + "aspp.project.1", CH)
    for i, (cin, cout) in enumerate(((CH, CH // 2), (CH // 2, CH // 2))):
        k += [(f"head.layers.{i}.body.0.weight", (cin, 1, 3, 3)), (f"head.layers.{i}.body.1.weight", (cout, cin, 1, 1))] + _bn(f"head.layers.{i}.body.2", cout)
    k += [("head.layers.2.weight", (1, CH // 2, 1, 1)), ("head.layers.2.bias", (1,))]
    return k


_SYNTH = {}


def iseg_synthetic_state_dict(seed=0):
    """Keyed-deterministic weights (no trained checkpoint exists offline).  The last BN of every bottleneck gets a small gamma so
    the 16 residual blocks keep the activations bounded; the first rgb_conv layer weighs the two click-distance channels up and
    the output bias is calibrated on a seeded probe, so the predicted mask depends on the image AND on the clicks."""
    if seed not in _SYNTH:
        sd = OrderedDict()
        for key, shape in iseg_checkpoint_layout():
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed ^ 0x15e9) & 0x7FFFFFFF)
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
                if key == "rgb_conv.0.weight":
                    t[:, 3:] = t[:, 3:] * 3.0
                if key == "head.layers.2.weight":
                    t = t * 6.0
            sd[key] = t
        from ..synth import SyntheticClip
        fr = SyntheticClip(96, 128, 1, config_id=2, device="cpu").frame(0)["frame"].float()
        probe = fr.permute(2, 0, 1)[None].flip(1) / 255.
        probe = (probe - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        pts = torch.tensor([[[40., 60.], [-1., -1.], [10., 10.], [-1., -1.]]])
        o = IsegNetTorch(sd)(probe, pts)
        sd["head.layers.2.bias"] -= o.flatten().median()
        _SYNTH[seed] = sd
    return OrderedDict((k, v.clone()) for k, v in _SYNTH[seed].items())


class SepConv:
    """SeparableConv2d (basic_blocks.py:89-121): depthwise 3x3 (pad 1, no bias) -> pointwise 1x1 + folded BN + ReLU."""

    def __init__(self, sd, prefix):
        self.dw = sd[prefix + ".body.0.weight"].float().clone()                    # (C, 1, 3, 3)
        self.pw = _fold(sd, prefix + ".body.1.weight", prefix + ".body.2", act="relu")

    def torch_forward(self, x):
        y = F.conv2d(x, self.dw.to(x.device), None, padding=1, groups=self.dw.shape[0])
        return self.pw.torch_forward(y)


class IsegPlan:
    def __init__(self, sd):
        # rgb_conv (distmap.py:16-22): Conv(5, 8, 1) -> LeakyReLU(0.2) -> BN(8) -> Conv(8, 3, 1).  The BN sits AFTER the activation:
        # it folds into the SECOND conv (w2' = w2 * s, b2' = b2 + w2 @ t with s = gamma / sqrt(var + eps), t = beta - mean * s).
        w1, b1 = sd["rgb_conv.0.weight"].double().view(8, 5), sd["rgb_conv.0.bias"].double()
        g, b = sd["rgb_conv.2.weight"].double(), sd["rgb_conv.2.bias"].double()
        m, v = sd["rgb_conv.2.running_mean"].double(), sd["rgb_conv.2.running_var"].double()
        s = g / torch.sqrt(v + BN_EPS)
        t = b - m * s
        w2, b2 = sd["rgb_conv.3.weight"].double().view(3, 8), sd["rgb_conv.3.bias"].double()
        self.rgb_w1, self.rgb_b1 = w1.float(), b1.float()
        self.rgb_w2, self.rgb_b2 = (w2 * s.view(1, 8)).float(), (b2 + w2 @ t).float()
        bb = "feature_extractor.backbone."
        self.stem = [_fold(sd, bb + "conv1.0.weight", bb + "conv1.1", stride=2, pad=1, act="relu"),
                     _fold(sd, bb + "conv1.3.weight", bb + "conv1.4", pad=1, act="relu"),
                     _fold(sd, bb + "conv1.6.weight", bb + "bn1", pad=1, act="relu")]
        self.layers = []
        for li, (n, stride, (d_first, d_rest)) in enumerate(zip(LAYERS, STRIDES, DILATIONS), start=1):
            blocks = []
            for bi in range(n):
                p = f"{bb}layer{li}.{bi}."
                dil = d_first if bi == 0 else d_rest
                st = stride if bi == 0 else 1
                blocks.append({"c1": _fold(sd, p + "conv1.weight", p + "bn1", act="relu"),
                               "c2": _fold(sd, p + "conv2.weight", p + "bn2", stride=st, pad=dil, dil=dil, act="relu"),
                               "c3": _fold(sd, p + "conv3.weight", p + "bn3", act="relu"),          # relu after the residual add
                               "down": _fold(sd, p + "downsample.0.weight", p + "downsample.1", stride=st) if bi == 0 else None})
            self.layers.append(blocks)
        fe = "feature_extractor."
        self.skip = _fold(sd, fe + "skip_project.skip_project.0.weight", fe + "skip_project.skip_project.1", act="relu")
        self.aspp0 = _fold(sd, fe + "aspp.concurent.0.0.weight", fe + "aspp.concurent.0.1", act="relu")
        self.aspp_d = [_fold(sd, f"{fe}aspp.concurent.{i}.0.weight", f"{fe}aspp.concurent.{i}.1", pad=r, dil=r, act="relu")
                       for i, r in zip((1, 2, 3), ASPP_RATES)]
        self.aspp_pool = _fold(sd, fe + "aspp.concurent.4.gap.1.weight", fe + "aspp.concurent.4.gap.2", act="relu")
        self.aspp_proj = _fold(sd, fe + "aspp.project.0.weight", fe + "aspp.project.1", act="relu")
        self.dl_head = [SepConv(sd, fe + "head.block.0"), SepConv(sd, fe + "head.block.1")]
        self.dl_head_out = FConv(sd[fe + "head.block.2.weight"].float().clone(), sd[fe + "head.block.2.bias"].float().clone())
        self.head = [SepConv(sd, "head.layers.0"), SepConv(sd, "head.layers.1")]
        self.head_out = FConv(sd["head.layers.2.weight"].float().clone(), sd["head.layers.2.bias"].float().clone())


def dist_maps(points, rows, cols):
    """DistMaps.get_coord_features (distmap.py:80-114) for spatial_scale 1: points (B, 2 * P, 2) float (y, x), (-1, -1) = absent,
    first P positive, last P negative -> (B, 2, rows, cols) float32 tanh(2 * sqrt(min_p ((y - py)^2 + (x - px)^2) / 260^2))."""
    B, n2, _ = points.shape
    P = n2 // 2
    pts = points.reshape(-1, 2).float()
    invalid = pts.max(dim=1)[0] < 0
    yy = torch.arange(rows, dtype=torch.float32).view(1, rows, 1)
    xx = torch.arange(cols, dtype=torch.float32).view(1, 1, cols)
    dy = (yy - pts[:, 0].view(-1, 1, 1)) / NORM_RADIUS
    dx = (xx - pts[:, 1].view(-1, 1, 1)) / NORM_RADIUS
    d = dy * dy + dx * dx
    d[invalid] = 1e6
    d = d.view(B * 2, P, rows, cols).min(dim=1)[0].view(B, 2, rows, cols)
    return torch.tanh(torch.sqrt(d) * 2)


class IsegNetTorch:
    """The folded plan on torch ops: forward(image (B, 3, H, W) normalised RGB, points (B, 2P, 2)) -> (B, 1, H, W) logits at the
    image size (DistMapsModel.forward, distmap.py:30-48)."""

    def __init__(self, state_dict, device="cpu"):
        self.plan = IsegPlan(state_dict)
        self.device = torch.device(device)

    @torch.no_grad()
    def rgb_conv(self, image, points):
        p = self.plan
        x = torch.cat([image.float(), dist_maps(points, image.shape[2], image.shape[3])], 1)
        y = F.leaky_relu(F.conv2d(x, p.rgb_w1.view(8, 5, 1, 1), p.rgb_b1), 0.2)
        return F.conv2d(y, p.rgb_w2.view(3, 8, 1, 1), p.rgb_b2)

    @torch.no_grad()
    def features(self, x):
        """Backbone + DeepLabV3+ decoder + SepConvHead on the rgb_conv output: (B, 3, H, W) -> (B, 1, h/4, w/4) logits."""
        p = self.plan
        for c in p.stem:
            x = c.torch_forward(x)
        x = F.max_pool2d(x, 3, 2, 1)
        c1 = None
        for li, blocks in enumerate(p.layers):
            for blk in blocks:
                idt = blk["down"].torch_forward(x) if blk["down"] is not None else x
                y = blk["c2"].torch_forward(blk["c1"].torch_forward(x))
                x = blk["c3"].torch_forward(y, residual=idt)
            if li == 0:
                c1 = x
        size = x.shape[2:]
        res = [p.aspp0.torch_forward(x)] + [c.torch_forward(x) for c in p.aspp_d]
        pool = p.aspp_pool.torch_forward(F.adaptive_avg_pool2d(x, 1))
        res.append(F.interpolate(pool, size=size, mode="bilinear", align_corners=True))
        y = p.aspp_proj.torch_forward(torch.cat(res, 1))
        skip = p.skip.torch_forward(c1)
        y = F.interpolate(y, size=skip.shape[2:], mode="bilinear", align_corners=True)
        y = torch.cat([y, skip], 1)
        for sc in p.dl_head:
            y = sc.torch_forward(y)
        y = p.dl_head_out.torch_forward(y)
        for sc in p.head:
            y = sc.torch_forward(y)
        return p.head_out.torch_forward(y)

    @torch.no_grad()
    def forward(self, image, points):
        low = self.features(self.rgb_conv(image, points))
        return F.interpolate(low, size=image.shape[2:], mode="bilinear", align_corners=True)

    __call__ = forward
