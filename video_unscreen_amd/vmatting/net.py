"""Matting network (ResShortCut UNet-18) as a flat, load-time-folded conv plan.

Mirrors the *function* of the reference network
(reference unscreen/vmatting/model.py:552-568 `UNet`, :435-486 encoder,
:492-515 decoder, :45-113 `SpectralNorm`) without sharing its module tree:

* a reference-format checkpoint (426 tensors, keys such as
  ``backbone.encoder.conv1.module.weight_bar/_u/_v``) is consumed once by
  :func:`fold_state_dict`;
* spectral norm is resolved at load time (``w = w_bar / (u . (W v))``, the
  eval-mode rule of model.py:72-79) instead of 53 mat-vecs per frame;
* an inference BatchNorm that directly follows a conv is folded into the conv's
  weight and bias; the shortcut branches are conv -> ReLU -> BN (model.py:452-456)
  so those BNs stay as per-channel affines;
* tensors are kept channels-last (NHWC) so the pixel kernels' 7-channel packed
  output feeds the first conv without a transpose.

``MattingNet`` below evaluates the folded plan with torch ops: on the CPU it is the network's f32 oracle for the
tests, on the GPU (``net_backend="miopen"``) a comparison baseline.  The product path is ``net_hip.HipMattingNet``:
the same plan on the hand-written f32-MFMA kernel of csrc/usc_conv.hip.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5

ENC_STAGES = ((64, 1), (128, 2), (256, 2), (512, 2))     # (planes, stride) layer1..4
DEC_STAGES = (256, 128, 64, 32)                           # planes layer1..4
SHORTCUT_IO = ((7, 32), (32, 32), (64, 64), (128, 128), (256, 256))


# --------------------------------------------------------------------------
# checkpoint layout (reference key names) and keyed-deterministic weights
# --------------------------------------------------------------------------

def _sn_conv_keys(prefix, cout, cin, k, transposed=False):
    """(key, shape) for a spectrally-normalised conv: u, v, bar in that order
    (model.py:89-107 registers u, v, then bar)."""
    if transposed:
        wshape = (cin, cout, k, k)
        height, width = cin, cout * k * k
    else:
        wshape = (cout, cin, k, k)
        height, width = cout, cin * k * k
    return [(prefix + ".module.weight_u", (height,)),
            (prefix + ".module.weight_v", (width,)),
            (prefix + ".module.weight_bar", wshape)]


def _bn_keys(prefix, c):
    return [(prefix + ".weight", (c,)), (prefix + ".bias", (c,)),
            (prefix + ".running_mean", (c,)), (prefix + ".running_var", (c,)),
            (prefix + ".num_batches_tracked", ())]


def checkpoint_layout():
    """Ordered (key, shape) list of the reference UNet state dict (426 entries)."""
    keys = []
    e = "backbone.encoder."
    keys += _sn_conv_keys(e + "conv1", 32, 7, 3)
    keys += _sn_conv_keys(e + "conv2", 32, 32, 3)
    keys += _sn_conv_keys(e + "conv3", 64, 32, 3)
    keys += _bn_keys(e + "bn1", 32) + _bn_keys(e + "bn2", 32) + _bn_keys(e + "bn3", 64)
    cin = 64
    for li, (planes, stride) in enumerate(ENC_STAGES, start=1):
        for b in range(2):
            p = f"{e}layer{li}.{b}."
            keys += _sn_conv_keys(p + "conv1", planes, cin if b == 0 else planes, 3)
            keys += _bn_keys(p + "bn1", planes)
            keys += _sn_conv_keys(p + "conv2", planes, planes, 3)
            keys += _bn_keys(p + "bn2", planes)
            if b == 0 and stride != 1:
                keys += _sn_conv_keys(p + "downsample.1", planes, cin, 1)
                keys += _bn_keys(p + "downsample.2", planes)
        cin = planes
    for si, (ci, co) in enumerate(SHORTCUT_IO):
        p = f"{e}shortcut.{si}."
        keys += _sn_conv_keys(p + "0", co, ci, 3)
        keys += _bn_keys(p + "2", co)
        keys += _sn_conv_keys(p + "3", co, co, 3)
        keys += _bn_keys(p + "5", co)
    d = "backbone.decoder."
    keys += _sn_conv_keys(d + "conv1", 32, 32, 4, transposed=True)
    keys += _bn_keys(d + "bn1", 32)
    keys += [(d + "conv2.weight", (1, 32, 3, 3)), (d + "conv2.bias", (1,))]
    cin = 512
    for li, planes in enumerate(DEC_STAGES, start=1):
        p = f"{d}layer{li}.0."
        keys += _sn_conv_keys(p + "conv1", cin, cin, 4, transposed=True)
        keys += _bn_keys(p + "bn1", cin)
        keys += _sn_conv_keys(p + "conv2", planes, cin, 3)
        keys += _bn_keys(p + "bn2", planes)
        keys += _sn_conv_keys(p + "upsample.1", planes, cin, 1)
        keys += _bn_keys(p + "upsample.2", planes)
        p = f"{d}layer{li}.1."
        keys += _sn_conv_keys(p + "conv1", planes, planes, 3)
        keys += _bn_keys(p + "bn1", planes)
        keys += _sn_conv_keys(p + "conv2", planes, planes, 3)
        keys += _bn_keys(p + "bn2", planes)
        cin = planes
    return keys


_SYNTH_CACHE: Dict[int, Dict[str, torch.Tensor]] = {}


def synthetic_state_dict(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Keyed-deterministic weights in the reference checkpoint format (cached per seed; returns clones)."""
    if seed not in _SYNTH_CACHE:
        _SYNTH_CACHE[seed] = _synthetic_state_dict(seed)
    return {k: v.clone() for k, v in _SYNTH_CACHE[seed].items()}


def _synthetic_state_dict(seed: int) -> Dict[str, torch.Tensor]:
    """Keyed-deterministic weights in the reference checkpoint format.

    No trained checkpoint exists offline, so tests, the bench and
    the golden fixtures all use these: every tensor is drawn from a generator
    seeded by crc32(key) ^ seed, scaled so activations stay O(1) and the raw
    alpha stays in about [-2, 2] (default init saturates tanh to {0, 1}).
    """
    sd = {}
    for key, shape in checkpoint_layout():
        g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed) & 0x7FFFFFFF)
        if key.endswith("num_batches_tracked"):
            t = torch.tensor(1000, dtype=torch.long)
        elif key.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif key.endswith("running_mean"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif key.endswith(".weight") and len(shape) == 1:      # BN gamma
            t = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif key.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif key.endswith("weight_u") or key.endswith("weight_v"):
            continue                                            # derived from weight_bar below
        elif key.endswith("decoder.conv2.weight"):
            t = torch.randn(shape, generator=g) * 0.05
        else:                                                   # weight_bar
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        sd[key] = t
    # u, v: what training leaves behind - the leading singular pair of W (power iteration,
    # model.py:56-66), so sigma = u.(W v) is the spectral norm rather than a random tiny number.
    for key, shape in checkpoint_layout():
        if key.endswith("weight_bar"):
            w = sd[key].reshape(shape[0], -1).double()
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed ^ 0x5bd1) & 0x7FFFFFFF)
            u = torch.randn(w.shape[0], generator=g, dtype=torch.double)
            for _ in range(30):
                v = torch.mv(w.t(), u)
                v = v / (v.norm() + 1e-12)
                u = torch.mv(w, v)
                u = u / (u.norm() + 1e-12)
            sd[key[:-3] + "u"] = u.float()
            sd[key[:-3] + "v"] = v.float()
    return {k: sd[k] for k, _ in checkpoint_layout()}


# --------------------------------------------------------------------------
# folding
# --------------------------------------------------------------------------

def _sn_weight(sd, prefix):
    """Eval-mode spectral norm (model.py:72-79): w_bar / (u . (W v)), W = w_bar.view(h, -1)."""
    u = sd[prefix + ".module.weight_u"].double()
    v = sd[prefix + ".module.weight_v"].double()
    w = sd[prefix + ".module.weight_bar"]
    sigma = torch.dot(u, w.double().reshape(w.shape[0], -1).mv(v))
    return (w.double() / sigma)


def _bn_affine(sd, prefix):
    g = sd[prefix + ".weight"].double()
    b = sd[prefix + ".bias"].double()
    m = sd[prefix + ".running_mean"].double()
    var = sd[prefix + ".running_var"].double()
    scale = g / torch.sqrt(var + BN_EPS)
    return scale, b - m * scale


@dataclass
class Conv:
    """One fused step: conv (or transposed conv) + bias [+ residual] [+ activation]."""
    w: torch.Tensor
    b: Optional[torch.Tensor]
    stride: int = 1
    pad: int = 1
    transposed: bool = False
    act: str = "none"           # none | relu | leaky

    def to(self, device, dtype):
        self.w = self.w.to(device=device, dtype=dtype)
        if self.w.dim() == 4:
            self.w = self.w.contiguous(memory_format=torch.channels_last)
        if self.b is not None:
            self.b = self.b.to(device=device, dtype=dtype)
        return self


def _fused(sd, conv_prefix, bn_prefix, stride=1, pad=1, transposed=False, act="none"):
    w = _sn_weight(sd, conv_prefix)
    scale, shift = _bn_affine(sd, bn_prefix)
    if transposed:                       # weight is (in, out, k, k): scale the out axis
        w = w * scale.view(1, -1, 1, 1)
    else:
        w = w * scale.view(-1, 1, 1, 1)
    return Conv(w.float(), shift.float(), stride, pad, transposed, act)


def fold_state_dict(sd: Dict[str, torch.Tensor]):
    """Resolve spectral norm + fold conv->BN pairs. Returns the plan dict used by MattingNet."""
    e, d = "backbone.encoder.", "backbone.decoder."
    plan = {}
    plan["stem"] = [_fused(sd, e + "conv1", e + "bn1", stride=2, act="relu"),
                    _fused(sd, e + "conv2", e + "bn2", stride=1, act="relu"),
                    _fused(sd, e + "conv3", e + "bn3", stride=2, act="relu")]
    enc = []
    for li, (planes, stride) in enumerate(ENC_STAGES, start=1):
        blocks = []
        for b in range(2):
            p = f"{e}layer{li}.{b}."
            blk = {"conv1": _fused(sd, p + "conv1", p + "bn1", stride=stride if b == 0 else 1, act="relu"),
                   "conv2": _fused(sd, p + "conv2", p + "bn2"),
                   "down": None}
            if b == 0 and stride != 1:
                blk["down"] = _fused(sd, p + "downsample.1", p + "downsample.2", pad=0)
            blocks.append(blk)
        enc.append(blocks)
    plan["enc"] = enc
    sc = []
    for si in range(5):
        p = f"{e}shortcut.{si}."
        s1, t1 = _bn_affine(sd, p + "2")
        s2, t2 = _bn_affine(sd, p + "5")
        sc.append({"c1": Conv(_sn_weight(sd, p + "0").float(), None, act="relu"),
                   "s1": s1.float(), "t1": t1.float(),
                   "c2": Conv(_sn_weight(sd, p + "3").float(), None, act="relu"),
                   "s2": s2.float(), "t2": t2.float()})
    plan["shortcut"] = sc
    dec = []
    for li in range(1, 5):
        p0, p1 = f"{d}layer{li}.0.", f"{d}layer{li}.1."
        dec.append([
            {"conv1": _fused(sd, p0 + "conv1", p0 + "bn1", stride=2, pad=1, transposed=True, act="leaky"),
             "conv2": _fused(sd, p0 + "conv2", p0 + "bn2"),
             "up": _fused(sd, p0 + "upsample.1", p0 + "upsample.2", pad=0)},
            {"conv1": _fused(sd, p1 + "conv1", p1 + "bn1", act="leaky"),
             "conv2": _fused(sd, p1 + "conv2", p1 + "bn2"),
             "up": None}])
    plan["dec"] = dec
    plan["head_up"] = _fused(sd, d + "conv1", d + "bn1", stride=2, pad=1, transposed=True, act="leaky")
    plan["head"] = Conv(sd[d + "conv2.weight"].float().clone(), sd[d + "conv2.bias"].float().clone())
    return plan


# --------------------------------------------------------------------------
# execution
# --------------------------------------------------------------------------

def _run(c: Conv, x, residual=None):
    if c.transposed:
        y = F.conv_transpose2d(x, c.w, c.b, stride=c.stride, padding=c.pad)
    else:
        y = F.conv2d(x, c.w, c.b, stride=c.stride, padding=c.pad)
    if residual is not None:
        y = y.add_(residual)
    if c.act == "relu":
        y = F.relu_(y)
    elif c.act == "leaky":
        y = F.leaky_relu_(y, 0.2)
    return y


class MattingNet:
    """Inference-only executor of the folded plan.

    ``forward(x7)`` takes the packed 7-channel input (N,7,H,W) - img RGB normalised,
    alpha_pre/255, one-hot trimap - i.e. the ``torch.cat`` of model.py:564 already
    done by the pre-processing kernel, and returns (tanh(raw)+1)/2 as (N,1,H,W) f32.
    """

    def __init__(self, state_dict, device="cpu", dtype=torch.float32):
        self.device = torch.device(device)
        self.dtype = dtype
        self.plan = fold_state_dict(state_dict)
        self._move()

    def _move(self):
        dev, dt = self.device, self.dtype
        for c in self.plan["stem"]:
            c.to(dev, dt)
        for blocks in self.plan["enc"]:
            for blk in blocks:
                for k in ("conv1", "conv2", "down"):
                    if blk[k] is not None:
                        blk[k].to(dev, dt)
        for s in self.plan["shortcut"]:
            s["c1"].to(dev, dt)
            s["c2"].to(dev, dt)
            for k in ("s1", "t1", "s2", "t2"):
                s[k] = s[k].to(device=dev, dtype=dt).view(1, -1, 1, 1)
        for blocks in self.plan["dec"]:
            for blk in blocks:
                for k in ("conv1", "conv2", "up"):
                    if blk[k] is not None:
                        blk[k].to(dev, dt)
        self.plan["head_up"].to(dev, dt)
        self.plan["head"].to(dev, dt)

    def _shortcut(self, s, x):
        y = _run(s["c1"], x)
        y = y.mul_(s["s1"]).add_(s["t1"])
        y = _run(s["c2"], y)
        return y.mul_(s["s2"]).add_(s["t2"])

    @torch.no_grad()
    def forward(self, x7: torch.Tensor) -> torch.Tensor:
        p = self.plan
        x = x7.to(self.dtype)
        if x.dim() == 4 and not x.is_contiguous(memory_format=torch.channels_last):
            x = x.contiguous(memory_format=torch.channels_last)
        t = _run(p["stem"][0], x)
        x1 = _run(p["stem"][1], t)
        t = _run(p["stem"][2], x1)
        feats = []
        for blocks in p["enc"]:
            for blk in blocks:
                if blk["down"] is not None:
                    idt = _run(blk["down"], F.avg_pool2d(t, 2, 2))
                else:
                    idt = t
                y = _run(blk["conv1"], t)
                y = _run(blk["conv2"], y, residual=idt)
                t = F.relu_(y)
            feats.append(t)
        x2, x3, x4, out = feats
        fea = [self._shortcut(s, z) for s, z in zip(p["shortcut"], (x, x1, x2, x3, x4))]
        t = out
        for blocks, skip in zip(p["dec"], (fea[4], fea[3], fea[2], fea[1])):
            for blk in blocks:
                if blk["up"] is not None:
                    idt = _run(blk["up"], F.interpolate(t, scale_factor=2, mode="nearest"))
                else:
                    idt = t
                y = _run(blk["conv1"], t)
                y = _run(blk["conv2"], y, residual=idt)
                t = F.leaky_relu_(y, 0.2)
            t = t.add_(skip)
        t = _run(p["head_up"], t).add_(fea[0])
        raw = _run(p["head"], t)
        return ((torch.tanh(raw.float()) + 1.0) / 2.0).contiguous()

    __call__ = forward
