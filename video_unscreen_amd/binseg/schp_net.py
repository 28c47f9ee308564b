"""SCHP / CE2P human-parsing network (deep-stem ResNet-101, PSP context, edge branch, fusion head) as a folded
inference plan: reference unscreen/binseg/human_parse.py:194-265 (`ResNet`), :42-80 (`Bottleneck`), :83-111
(`PSPModule`), :114-156 (`EdgeModule`), :159-191 (`DecoderModule`), :25-39 (`InPlaceABNSync` = BN + LeakyReLU(0.01)).

Only what reaches the returned tensor is kept: `fusion_result` (human_parse.py:265).  The edge maps (edge.conv4/5)
and the parsing head (decoder.conv4) are computed by the reference and thrown away, so they are not built.

* `schp_checkpoint_layout()` reproduces the reference state-dict keys (722 tensors); the checkpoint the reference
  loads carries a `module.` prefix from nn.DataParallel (agent_human.py:141-148) - both spellings are accepted.
* `SchpPlan` folds every BatchNorm into its conv; `SchpNetTorch` runs the plan with torch ops (CPU oracle of the
  network / MIOpen path); `HipSchpNet` runs every convolution on csrc/usc_conv.hip.
"""
import ctypes as C
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

from ..convnet import ConvGraphNet, _out_size

BN_EPS = 1e-5
LAYERS = (3, 4, 23, 3)
PLANES = (64, 128, 256, 512)
STRIDES = (1, 2, 2, 1)
DILATIONS = (1, 1, 1, 2)
PSP_SIZES = (1, 2, 3, 6)
NUM_CLASSES = 20


def _bn(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)), (prefix + ".running_mean", (c,)),
            (prefix + ".running_var", (c,)), (prefix + ".num_batches_tracked", ())]


def schp_checkpoint_layout():
    """Ordered (key, shape) list of the reference network's state dict."""
    k = [("conv1.weight", (64, 3, 3, 3))] + _bn("bn1", 64) + [("conv2.weight", (64, 64, 3, 3))] + _bn("bn2", 64) + \
        [("conv3.weight", (128, 64, 3, 3))] + _bn("bn3", 128)
    inpl = 128
    for li, (n, planes, stride) in enumerate(zip(LAYERS, PLANES, STRIDES), start=1):
        for b in range(n):
            p = f"layer{li}.{b}."
            k += [(p + "conv1.weight", (planes, inpl, 1, 1))] + _bn(p + "bn1", planes)
            k += [(p + "conv2.weight", (planes, planes, 3, 3))] + _bn(p + "bn2", planes)
            k += [(p + "conv3.weight", (planes * 4, planes, 1, 1))] + _bn(p + "bn3", planes * 4)
            if b == 0 and (stride != 1 or inpl != planes * 4):
                k += [(p + "downsample.0.weight", (planes * 4, inpl, 1, 1))] + _bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    for i in range(4):
        k += [(f"context_encoding.stages.{i}.1.weight", (512, 2048, 1, 1))] + _bn(f"context_encoding.stages.{i}.2.bn", 512)
    k += [("context_encoding.bottleneck.0.weight", (512, 4096, 3, 3))] + _bn("context_encoding.bottleneck.1.bn", 512)
    for i, cin in enumerate((256, 512, 1024), start=1):
        k += [(f"edge.conv{i}.0.weight", (256, cin, 1, 1))] + _bn(f"edge.conv{i}.1.bn", 256)
    k += [("edge.conv4.weight", (2, 256, 3, 3)), ("edge.conv4.bias", (2,)), ("edge.conv5.weight", (2, 6, 1, 1)), ("edge.conv5.bias", (2,))]
    k += [("decoder.conv1.0.weight", (256, 512, 1, 1))] + _bn("decoder.conv1.1.bn", 256)
    k += [("decoder.conv2.0.weight", (48, 256, 1, 1))] + _bn("decoder.conv2.1.bn", 48)
    k += [("decoder.conv3.0.weight", (256, 304, 1, 1))] + _bn("decoder.conv3.1.bn", 256)
    k += [("decoder.conv3.2.weight", (256, 256, 1, 1))] + _bn("decoder.conv3.3.bn", 256)
    k += [("decoder.conv4.weight", (NUM_CLASSES, 256, 1, 1)), ("decoder.conv4.bias", (NUM_CLASSES,))]
    k += [("fushion.0.weight", (256, 1024, 1, 1))] + _bn("fushion.1.bn", 256)
    k += [("fushion.3.weight", (NUM_CLASSES, 256, 1, 1)), ("fushion.3.bias", (NUM_CLASSES,))]
    return k


_SYNTH = {}


def schp_synthetic_state_dict(seed=0):
    """Keyed-deterministic weights (no trained checkpoint exists offline).  The last BN of every bottleneck gets a
    small gamma so 33 residual blocks of random weights do not blow the activations up."""
    if seed not in _SYNTH:
        sd = OrderedDict()
        for key, shape in schp_checkpoint_layout():
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed ^ 0x5c4b) & 0x7FFFFFFF)
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
                if key == "fushion.3.weight":
                    t = t * 4.0
            sd[key] = t
        _SYNTH[seed] = sd
        # Random features have a large constant component, so one class would win everywhere.  Calibrate the
        # background bias on a fixed seeded input until background and person classes split the image about evenly:
        # the mask (argmax > 0) then depends on the input, which is what the parity tests need.
        from ..synth import SyntheticClip                     # a smooth frame with a textured blob, like the clips
        fr = SyntheticClip(128, 160, 1, config_id=2, device="cpu").frame(0)["frame"].float()
        probe = fr.permute(2, 0, 1)[None].flip(1) / 255.
        probe = (probe - torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)) / torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        o = SchpNetTorch(sd)(probe)
        margin = (o[:, 0] - o[:, 1:].max(1)[0]).flatten()
        sd["fushion.3.bias"][0] -= margin.median()
    return OrderedDict((k, v.clone()) for k, v in _SYNTH[seed].items())


def strip_module_prefix(sd):
    if all(k.startswith("module.") for k in sd):
        return OrderedDict((k[len("module."):], v) for k, v in sd.items())
    return sd


class FConv:
    """conv + folded BN (+ activation): w (Cout, Cin, k, k), b (Cout,)."""

    def __init__(self, w, b, stride=1, pad=0, dil=1, act="none"):
        self.w, self.b, self.stride, self.pad, self.dil, self.act = w, b, stride, pad, dil, act

    def torch_forward(self, x, residual=None):
        y = F.conv2d(x, self.w.to(x.device), self.b.to(x.device), stride=self.stride, padding=self.pad, dilation=self.dil)
        if residual is not None:
            y = y + residual
        if self.act == "relu":
            y = F.relu(y)
        elif self.act == "leaky01":
            y = F.leaky_relu(y, 0.01)
        return y


def _fold(sd, conv_key, bn_prefix, **kw):
    w = sd[conv_key].double()
    g, b = sd[bn_prefix + ".weight"].double(), sd[bn_prefix + ".bias"].double()
    m, v = sd[bn_prefix + ".running_mean"].double(), sd[bn_prefix + ".running_var"].double()
    s = g / torch.sqrt(v + BN_EPS)
    return FConv((w * s.view(-1, 1, 1, 1)).float(), (b - m * s).float(), **kw)


class SchpPlan:
    def __init__(self, state_dict):
        sd = strip_module_prefix(state_dict)
        self.stem = [_fold(sd, "conv1.weight", "bn1", stride=2, pad=1, act="relu"),
                     _fold(sd, "conv2.weight", "bn2", pad=1, act="relu"),
                     _fold(sd, "conv3.weight", "bn3", pad=1, act="relu")]
        self.layers = []
        for li, (n, stride, dil) in enumerate(zip(LAYERS, STRIDES, DILATIONS), start=1):
            blocks = []
            for b in range(n):
                p = f"layer{li}.{b}."
                blk = {"c1": _fold(sd, p + "conv1.weight", p + "bn1", act="relu"),
                       "c2": _fold(sd, p + "conv2.weight", p + "bn2", stride=stride if b == 0 else 1, pad=dil, dil=dil, act="relu"),
                       "c3": _fold(sd, p + "conv3.weight", p + "bn3", act="relu"),          # relu after the residual add
                       "down": None}
                if (p + "downsample.0.weight") in sd:
                    blk["down"] = _fold(sd, p + "downsample.0.weight", p + "downsample.1", stride=stride)
                blocks.append(blk)
            self.layers.append(blocks)
        self.psp = [_fold(sd, f"context_encoding.stages.{i}.1.weight", f"context_encoding.stages.{i}.2.bn", act="leaky01") for i in range(4)]
        self.psp_bottleneck = _fold(sd, "context_encoding.bottleneck.0.weight", "context_encoding.bottleneck.1.bn", pad=1, act="leaky01")
        self.edge = [_fold(sd, f"edge.conv{i}.0.weight", f"edge.conv{i}.1.bn", act="leaky01") for i in (1, 2, 3)]
        self.dec1 = _fold(sd, "decoder.conv1.0.weight", "decoder.conv1.1.bn", act="leaky01")
        self.dec2 = _fold(sd, "decoder.conv2.0.weight", "decoder.conv2.1.bn", act="leaky01")
        self.dec3a = _fold(sd, "decoder.conv3.0.weight", "decoder.conv3.1.bn", act="leaky01")
        self.dec3b = _fold(sd, "decoder.conv3.2.weight", "decoder.conv3.3.bn", act="leaky01")
        self.fuse1 = _fold(sd, "fushion.0.weight", "fushion.1.bn", act="leaky01")
        self.fuse2 = FConv(sd["fushion.3.weight"].float().clone(), sd["fushion.3.bias"].float().clone())


def _interp(x, size):
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


class SchpNetTorch:
    """The folded plan on torch ops: forward((1, 3, H, W) normalised RGB) -> (1, 20, ~H/4, ~W/4) logits."""

    def __init__(self, state_dict, device="cpu"):
        self.plan = SchpPlan(state_dict)
        self.device = torch.device(device)

    @torch.no_grad()
    def forward(self, x):
        p = self.plan
        x = x.to(self.device, torch.float32)
        for c in p.stem:
            x = c.torch_forward(x)
        x = F.max_pool2d(x, 3, 2, 1)
        feats = []
        for blocks in p.layers:
            for blk in blocks:
                idt = blk["down"].torch_forward(x) if blk["down"] is not None else x
                y = blk["c1"].torch_forward(x)
                y = blk["c2"].torch_forward(y)
                x = blk["c3"].torch_forward(y, residual=idt)
            feats.append(x)
        x2, x3, x4, x5 = feats
        h, w = x5.shape[2:]
        priors = [_interp(c.torch_forward(F.adaptive_avg_pool2d(x5, s)), (h, w)) for c, s in zip(p.psp, PSP_SIZES)] + [x5]
        ctx = p.psp_bottleneck.torch_forward(torch.cat(priors, 1))
        h2, w2 = x2.shape[2:]
        xt = _interp(p.dec1.torch_forward(ctx), (h2, w2))
        xl = p.dec2.torch_forward(x2)
        parsing_fea = p.dec3b.torch_forward(p.dec3a.torch_forward(torch.cat([xt, xl], 1)))
        e1 = p.edge[0].torch_forward(x2)
        e2 = _interp(p.edge[1].torch_forward(x3), (h2, w2))
        e3 = _interp(p.edge[2].torch_forward(x4), (h2, w2))
        fused = p.fuse1.torch_forward(torch.cat([parsing_fea, e1, e2, e3], 1))
        return p.fuse2.torch_forward(fused)

    __call__ = forward


# --------------------------------------------------------------------------------------------
# HIP executor
# --------------------------------------------------------------------------------------------


class HipSchpNet(ConvGraphNet):
    """Every convolution on the f32-MFMA kernel (csrc/usc_conv.hip), NHWC buffers; pooling / bilinear glue between
    conv segments stays on torch ops (a handful of tiny tensors)."""

    def __init__(self, state_dict, H=473, W=473, device="cuda:0", batch=1, plan_batch=1):
        """batch > 1: `batch` frames per forward, stacked along y in x4 / logits32 - same bits per frame as batch 1, half the
        launches per frame (the 30 x 30 layers of this net are launch-bound)."""
        super().__init__(device, slab_floats=64 * 1024 * 1024, batch=batch, plan_batch=plan_batch)
        self.H, self.W = H, W
        self.plan = SchpPlan(state_dict)
        self._build()

    # ---- graph
    def _build(self):
        p, H, W = self.plan, self.H, self.W
        self.x4 = self._buf(H, W, 4)                                   # RGB + one zero channel
        t, h, w = self._conv(p.stem[0], self.x4, H, W, 4)
        t, h, w = self._conv(p.stem[1], t, h, w, 64)
        t, h, w = self._conv(p.stem[2], t, h, w, 64)
        ph, pw = _out_size(h, 3, 2, 1, 1), _out_size(w, 3, 2, 1, 1)
        pooled = self._buf(ph, pw, 128)
        src = t
        self._glue(lambda: self.nv.maxpool3x3s2(src, pooled, batch=self.batch))
        t, h, w = pooled, ph, pw
        feats = []
        nv = self.nv
        for li, blocks in enumerate(p.layers):
            for bi, blk in enumerate(blocks):
                cin = t.shape[2]
                if blk["down"] is not None:
                    idt, _, _ = self._conv(blk["down"], t, h, w, cin)
                else:
                    idt = t
                y, _, _ = self._conv(blk["c1"], t, h, w, cin)
                y, h2, w2 = self._conv(blk["c2"], y, h, w, y.shape[2])
                t, h, w = self._conv(blk["c3"], y, h2, w2, y.shape[2], res1=idt)
            feats.append((t, h, w))
        (x2, h2, w2), (x3, h3, w3), (x4, h4, w4), (x5, h5, w5) = feats
        # PSP (human_parse.py:83-111): adaptive pools (s x s, s = 1, 2, 3, 6) -> 1x1 conv + ABN -> bilinear (align_corners) up,
        # concatenated with x5 and fed to a 3x3 bottleneck with 4096 input channels.  Half of that layer's 34 GFLOP would
        # multiply upsampled s x s maps: a 3x3 conv of an upsampled map is sum_taps shift_tap(up(p W_tap)) (the channel mixing
        # of a tap commutes with the spatial upsampling), so each prior goes through ONE tiny 1x1 layer with the nine tap
        # matrices side by side (512 -> 9 x 512) and usc_upsample_taps_accum_f32 adds the nine upsampled results at the tap
        # offsets into a map that the bottleneck over x5 alone (2048 input channels) takes as its residual.  Same sum, other
        # order; 17 GFLOP and the 4096-channel concat buffer less per frame.
        pools = [self._buf(s, s, 2048) for s in PSP_SIZES]

        def pool_all():
            for s, pb in zip(PSP_SIZES, pools):
                nv.adaptive_avgpool(x5, 0, 2048, pb, batch=self.batch)
        self._glue(pool_all)
        stage_out = [self._conv(c, pb, s, s, 2048)[0] for c, pb, s in zip(p.psp, pools, PSP_SIZES)]
        wb = p.psp_bottleneck.w                                          # (512, 4096, 3, 3), BN folded: priors first, x5 last
        qs = []
        for i, (so, s) in enumerate(zip(stage_out, PSP_SIZES)):
            wi = wb[:, 512 * i:512 * (i + 1)]                            # (co, ci, 3, 3) -> rows tap * 512 + co
            wq = wi.permute(2, 3, 0, 1).reshape(9 * 512, 512, 1, 1).contiguous()
            qs.append(self._conv(FConv(wq, wq.new_zeros(9 * 512), act="none"), so, s, s, 512)[0])
        prior_map = self._buf(h5, w5, 512)

        def priors():
            for i, q in enumerate(qs):
                nv.upsample_taps_accum(q, prior_map, batch=self.batch, accumulate=i > 0)
        self._glue(priors)
        bott = FConv(wb[:, 2048:].contiguous(), p.psp_bottleneck.b, pad=1, act="leaky01")
        ctx, _, _ = self._conv(bott, x5, h5, w5, 2048, res1=prior_map)
        d1, _, _ = self._conv(p.dec1, ctx, h5, w5, 512)
        cat_dec = self._buf(h2, w2, 320)
        cat_fuse = self._buf(h2, w2, 1024)
        self._conv(p.dec2, x2, h2, w2, 256, out=cat_dec, cout_pad=64, out_ld=320, out_c0=256)   # 48 real channels + 16 zeros
        self._conv(p.edge[0], x2, h2, w2, 256, out=cat_fuse, out_ld=1024, out_c0=256)
        e2, _, _ = self._conv(p.edge[1], x3, h3, w3, 512)
        e3, _, _ = self._conv(p.edge[2], x4, h4, w4, 1024)

        def assemble_dec():
            nv.upsample_ac(d1, cat_dec, 0, batch=self.batch)
            nv.upsample_ac(e2, cat_fuse, 512, batch=self.batch)
            nv.upsample_ac(e3, cat_fuse, 768, batch=self.batch)
        self._glue(assemble_dec)
        # decoder.conv3: 304 input channels in the reference = 256 + 48; here 256 + 64 with zero weights for the pad
        w3a = self.plan.dec3a.w
        dec3a = FConv(torch.cat([w3a, w3a.new_zeros(w3a.shape[0], 16, 1, 1)], 1), self.plan.dec3a.b, act="leaky01")
        y, _, _ = self._conv(dec3a, cat_dec, h2, w2, 320)
        self._conv(p.dec3b, y, h2, w2, 256, out=cat_fuse, out_ld=1024, out_c0=0)
        y, _, _ = self._conv(p.fuse1, cat_fuse, h2, w2, 1024)
        self.logits32, _, _ = self._conv(p.fuse2, y, h2, w2, 256, cout_pad=32)
        self._flush()
        self.out_h, self.out_w = h2, w2

    def forward(self, x4=None):
        """x4: (H, W, 4) float32 NHWC normalised RGB + zero channel (or None if the caller wrote self.x4).
        Returns the (out_h, out_w, 32) logits buffer: channels 0..19 are the classes."""
        if x4 is not None and x4.data_ptr() != self.x4.data_ptr():
            self.x4.copy_(x4)
        self.run()
        return self.logits32

    __call__ = forward
