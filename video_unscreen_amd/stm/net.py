"""Space-Time Memory network (reference unscreen/stm/model.py:9-225) as a folded inference plan.

  memorize(frame, masks) : Encoder_M (ResNet-50 trunk to layer3, 7x7 stems for the frame, the object mask and the background
                           mask summed - one 7x7 over 5 channels here) -> KV_M_r4 -> key (128) / value (512) at 1/16
  segment(frame, K, V)   : Encoder_Q -> KV_Q_r4 -> Memory (softmax(K^T q / sqrt(128)) over all memory positions, value read)
                           -> Decoder (convFM, pre-activation ResBlocks, two Refine stages, pred2) -> x4 bilinear ->
                           soft aggregation (model.py:195-202) -> 2-channel logit

* `stm_checkpoint_layout()` reproduces the reference state-dict keys (STMAgent loads the file directly, stm/agent.py:53).
* `StmPlan` folds the trunks' BatchNorms; `StmNetTorch` runs the plan on torch ops (the CPU oracle of the network).
"""
import math
import zlib
from collections import OrderedDict

import torch
import torch.nn.functional as F

from ..binseg.schp_net import FConv, _bn, _fold

TRUNK_LAYERS = (3, 4, 6)
TRUNK_PLANES = (64, 128, 256)
TRUNK_STRIDES = (1, 2, 2)
KEYDIM, VALDIM, MDIM = 128, 512, 256


def _trunk_layout(prefix):
    k = [(prefix + "conv1.weight", (64, 3, 7, 7))] + _bn(prefix + "bn1", 64)
    inpl = 64
    for name, n, planes in zip(("res2", "res3", "res4"), TRUNK_LAYERS, TRUNK_PLANES):
        for b in range(n):
            p = f"{prefix}{name}.{b}."
            k += [(p + "conv1.weight", (planes, inpl, 1, 1))] + _bn(p + "bn1", planes)
            k += [(p + "conv2.weight", (planes, planes, 3, 3))] + _bn(p + "bn2", planes)
            k += [(p + "conv3.weight", (planes * 4, planes, 1, 1))] + _bn(p + "bn3", planes * 4)
            if b == 0:
                k += [(p + "downsample.0.weight", (planes * 4, inpl, 1, 1))] + _bn(p + "downsample.1", planes * 4)
            inpl = planes * 4
    return k


def _conv_wb(prefix, cout, cin, k=3):
    return [(prefix + ".weight", (cout, cin, k, k)), (prefix + ".bias", (cout,))]


def _resblock(prefix, dim):
    return _conv_wb(prefix + ".conv1", dim, dim) + _conv_wb(prefix + ".conv2", dim, dim)


def stm_checkpoint_layout():
    """Ordered (key, shape) list of the reference STM's state dict."""
    k = [("Encoder_M.conv1_m.weight", (64, 1, 7, 7)), ("Encoder_M.conv1_o.weight", (64, 1, 7, 7))] + _trunk_layout("Encoder_M.")
    k += _trunk_layout("Encoder_Q.")
    for kv in ("KV_M_r4", "KV_Q_r4"):
        k += _conv_wb(kv + ".Key", KEYDIM, 1024) + _conv_wb(kv + ".Value", VALDIM, 1024)
    k += _conv_wb("Decoder.convFM", MDIM, 1024) + _resblock("Decoder.ResMM", MDIM)
    for rf, cin in (("RF3", 512), ("RF2", 256)):
        k += _conv_wb(f"Decoder.{rf}.convFS", MDIM, cin) + _resblock(f"Decoder.{rf}.ResFS", MDIM) + _resblock(f"Decoder.{rf}.ResMM", MDIM)
    k += _conv_wb("Decoder.pred2", 2, MDIM)
    return k


_SYNTH = {}


def stm_synthetic_state_dict(seed=0):
    """Keyed-deterministic weights (no trained checkpoint exists offline)."""
    if seed not in _SYNTH:
        sd = OrderedDict()
        for key, shape in stm_checkpoint_layout():
            g = torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ seed ^ 0x57a1) & 0x7FFFFFFF)
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
                gain = 0.5 if ".Res" in key else 1.6          # residual branches of the decoder stay small
                t = torch.randn(shape, generator=g) * (gain / fan_in) ** 0.5
                if key in ("Encoder_M.conv1_m.weight", "Encoder_M.conv1_o.weight"):
                    t = t * 3.0                                # let the mask steer the memory
            sd[key] = t
        # Random decoders saturate the 2-class logit.  Rescale / recentre pred2 on a seeded probe (two frames of a synthetic
        # clip, the first one's coarse mask as memory) so that the margin has unit-order spread around zero: the mask then
        # depends on the input, which is what the parity tests need.
        from ..synth import SyntheticClip
        clip = SyntheticClip(128, 160, 2, config_id=2, device="cpu")
        mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
        fr = [clip.frame(t) for t in range(2)]
        x = [((f["frame"].float().permute(2, 0, 1)[None].flip(1) / 255.) - mean) / std for f in fr]
        m = (fr[0]["seg"] > 127).float()[None, None]
        net = StmNetTorch(sd)
        k4, v4 = net.memorize(x[0], torch.cat([1 - m, m], 1))
        p2 = net.decoder_logits(x[1], k4.unsqueeze(2), v4.unsqueeze(2))
        margin = (p2[:, 1] - p2[:, 0]).flatten()
        scale = 2.0 / float(margin.std())
        sd["Decoder.pred2.weight"] *= scale
        sd["Decoder.pred2.bias"] *= scale
        sd["Decoder.pred2.bias"][1] -= scale * float(margin.median())
        _SYNTH[seed] = sd
    return OrderedDict((k, v.clone()) for k, v in _SYNTH[seed].items())


def _plain(sd, prefix, pad=1, act="none"):
    return FConv(sd[prefix + ".weight"].float().clone(), sd[prefix + ".bias"].float().clone(), pad=pad, act=act)


class StmPlan:
    def __init__(self, sd):
        def trunk(prefix):
            stem = _fold(sd, prefix + "conv1.weight", prefix + "bn1", stride=2, pad=3, act="relu")
            layers = []
            for name, n, stride in zip(("res2", "res3", "res4"), TRUNK_LAYERS, TRUNK_STRIDES):
                blocks = []
                for b in range(n):
                    p = f"{prefix}{name}.{b}."
                    st = stride if b == 0 else 1
                    blocks.append({"c1": _fold(sd, p + "conv1.weight", p + "bn1", act="relu"),
                                   "c2": _fold(sd, p + "conv2.weight", p + "bn2", stride=st, pad=1, act="relu"),
                                   "c3": _fold(sd, p + "conv3.weight", p + "bn3", act="relu"),
                                   "down": _fold(sd, p + "downsample.0.weight", p + "downsample.1", stride=st) if b == 0 else None})
                layers.append(blocks)
            return stem, layers
        self.q_stem, self.q_layers = trunk("Encoder_Q.")
        m_stem, self.m_layers = trunk("Encoder_M.")
        # conv1(frame) + conv1_m(object mask) + conv1_o(background mask), then bn1 (model.py:55-57): one conv over (R, G, B, m, o)
        g = sd["Encoder_M.bn1.weight"].double() / torch.sqrt(sd["Encoder_M.bn1.running_var"].double() + 1e-5)
        w5 = torch.cat([sd["Encoder_M.conv1.weight"], sd["Encoder_M.conv1_m.weight"], sd["Encoder_M.conv1_o.weight"]], 1).double()
        self.m_stem = FConv((w5 * g.view(-1, 1, 1, 1)).float(), m_stem.b, stride=2, pad=3, act="relu")
        self.kv_m = (_plain(sd, "KV_M_r4.Key"), _plain(sd, "KV_M_r4.Value"))
        self.kv_q = (_plain(sd, "KV_Q_r4.Key"), _plain(sd, "KV_Q_r4.Value"))
        self.convFM = _plain(sd, "Decoder.convFM")
        res = lambda p: (_plain(sd, p + ".conv1"), _plain(sd, p + ".conv2"))          # noqa: E731
        self.ResMM = res("Decoder.ResMM")
        self.RF3 = {"convFS": _plain(sd, "Decoder.RF3.convFS"), "ResFS": res("Decoder.RF3.ResFS"), "ResMM": res("Decoder.RF3.ResMM")}
        self.RF2 = {"convFS": _plain(sd, "Decoder.RF2.convFS"), "ResFS": res("Decoder.RF2.ResFS"), "ResMM": res("Decoder.RF2.ResMM")}
        self.pred2 = _plain(sd, "Decoder.pred2")


def soft_aggregation(ps):
    """model.py:195-202 for one object: ps (1, H, W) -> (1, 2, H, W) logit."""
    em = torch.stack([1 - ps[0], ps[0]], 0)[None]
    em = torch.clamp(em, 1e-7, 1 - 1e-7)
    return torch.log(em / (1 - em))


class StmNetTorch:
    def __init__(self, state_dict, device="cpu"):
        self.plan = StmPlan(state_dict)
        self.device = torch.device(device)

    @staticmethod
    def _trunk(stem, layers, x):
        x = F.max_pool2d(stem.torch_forward(x), 3, 2, 1)
        feats = []
        for blocks in layers:
            for blk in blocks:
                idt = blk["down"].torch_forward(x) if blk["down"] is not None else x
                y = blk["c2"].torch_forward(blk["c1"].torch_forward(x))
                x = blk["c3"].torch_forward(y, residual=idt)
            feats.append(x)
        return feats                                          # r2, r3, r4

    @staticmethod
    def _res(pair, x):
        r = pair[0].torch_forward(F.relu(x))
        r = pair[1].torch_forward(F.relu(r))
        return x + r

    @torch.no_grad()
    def memorize(self, frame, masks):
        """frame (1, 3, h, w), masks (1, 2, h, w) [bg, fg] -> key (1, 128, h/16, w/16), value (1, 512, h/16, w/16)."""
        p = self.plan
        x5 = torch.cat([frame, masks[:, 1:2].float(), masks[:, 0:1].float()], 1)
        r4 = self._trunk(p.m_stem, p.m_layers, x5)[2]
        return p.kv_m[0].torch_forward(r4), p.kv_m[1].torch_forward(r4)

    @torch.no_grad()
    def decoder_logits(self, frame, keys, values):
        """The decoder's 1/4-resolution logits p2 (model.py:131-137) - the tensor the HIP executor hands to usc_stm_post."""
        p = self.plan
        r2, r3, r4 = self._trunk(p.q_stem, p.q_layers, frame)
        k4, v4 = p.kv_q[0].torch_forward(r4), p.kv_q[1].torch_forward(r4)
        B, D_e, T, H, W = keys.shape
        mi = keys.reshape(B, D_e, T * H * W).transpose(1, 2)
        pm = torch.bmm(mi, k4.reshape(B, D_e, H * W)) / math.sqrt(D_e)
        pm = F.softmax(pm, dim=1)
        mem = torch.bmm(values.reshape(B, VALDIM, T * H * W), pm).view(B, VALDIM, H, W)
        m4 = self._res(p.ResMM, p.convFM.torch_forward(torch.cat([mem, v4], 1)))

        def refine(rf, f, pm_):
            s = self._res(rf["ResFS"], rf["convFS"].torch_forward(f))
            m = s + F.interpolate(pm_, scale_factor=2, mode="bilinear", align_corners=False)
            return self._res(rf["ResMM"], m)
        m3 = refine(p.RF3, r3, m4)
        m2 = refine(p.RF2, r2, m3)
        return p.pred2.torch_forward(F.relu(m2))

    @torch.no_grad()
    def segment(self, frame, keys, values):
        """-> (1, 2, h, w) soft-aggregated logit (model.py:204-219)."""
        p2 = self.decoder_logits(frame, keys, values)
        logits = F.interpolate(p2, scale_factor=4, mode="bilinear", align_corners=False)
        ps = F.softmax(logits, dim=1)[:, 1]
        return soft_aggregation(ps)
