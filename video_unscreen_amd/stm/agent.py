"""STMAgent: same constructor and forward() as the reference (unscreen/stm/agent.py:10-221).

forward(framelist, mask0) propagates the first frame's mask through the list: per step memorize(previous frame, previous
prediction) -> segment(current frame) -> softmax; masks come back at the original size.  Pre / post processing are HIP
kernels (csrc/usc_stm.hip), the network runs on the MFMA conv kernel (net_hip.HipStmNet) and the memory bank stays in HBM.

The reference's bank update reads `if t - 1 % self.memory_step == 0` (agent.py:166), i.e. `t - (1 % step) == 0`: for
memory_step > 1 the bank is refreshed only at t = 1, so from t = 2 on every step sees exactly [memory of frame 0, memory of
frame t-1].  That behaviour is reproduced (`fix_memory_step=False`); `fix_memory_step=True` applies the evident intent
`(t - 1) % memory_step == 0`.
"""
import numpy as np
import torch

from .. import _native as nv
from ..utils._dev import device, to_dev, to_host
from .net import stm_synthetic_state_dict
from .net_hip import HipStmNet


class STMAgent():

    def __init__(self, model_path, memory_step=5, memory_capacity=10, input_long_side=960, cuda_device=0, fix_memory_step=False, state_dict=None):
        self.cuda_device = cuda_device if cuda_device is not None and cuda_device >= 0 else 0
        self.device = device(self.cuda_device)
        if state_dict is not None:                     # weights handed over in memory (broadcast from rank 0)
            sd = state_dict
        elif model_path is None or model_path == "synthetic":
            sd = stm_synthetic_state_dict(0)          # no trained checkpoint exists offline
        else:
            sd = torch.load(model_path, map_location="cpu")
        self.state_dict = sd
        self.division = 16
        self.input_long_side = input_long_side
        self.memory_step = memory_step
        self.memory_capacity = memory_capacity
        self.fix_memory_step = fix_memory_step
        self._nets = {}

    def input_size(self, h, w):
        return nv.get_target_size(h, w, self.input_long_side, self.division)

    def model_for(self, in_h, in_w):
        key = (in_h, in_w)
        if key not in self._nets:
            self._nets[key] = HipStmNet(self.state_dict, in_h, in_w, capacity=self.memory_capacity + 1, device=self.device)
        return self._nets[key]

    def forward_device(self, framelist, mask0, return_debug=False):
        """framelist: list of (H, W, 3) uint8 CUDA tensors, mask0 (H, W) uint8 CUDA -> list of (H, W) uint8 masks (the first is
        mask0 > 127 re-derived exactly as the reference does: argmax of the pad-resized one-hot mask)."""
        H, W = mask0.shape
        in_h, in_w = self.input_size(H, W)
        net = self.model_for(in_h, in_w)
        hw = net.hw
        masks, debug = [], []
        # frame 0: preds = one-hot(mask0 > 127) at the network size; its output mask is inv_pad_resize + argmax of that
        nv.stm_pre(framelist[0], mask0, in_h, in_w, out=net.x8)
        prob = torch.stack([net.x8[..., 4], net.x8[..., 3]], -1).contiguous()              # (bg, fg)
        masks.append(nv.stm_mask_from_prob(prob, H, W))
        bank = []                          # slots of the persistent bank ("keys" / "values" of the reference loop)
        for t in range(1, len(framelist)):
            # memorize (previous frame, previous prediction): x8 channels 3 / 4 = object / background probability
            if t > 1:
                nv.stm_pre(framelist[t - 1], None, in_h, in_w, out=net.x8)
                net.x8[..., 3] = prob[..., 1]
                net.x8[..., 4] = prob[..., 0]
            # the candidate memory = persistent bank + the fresh slot, oldest dropped beyond capacity (agent.py:149-156)
            if t > 1 and not bank:
                raise RuntimeError("STMAgent: the memory bank was never set (the reference raises NameError here: with "
                                   "memory_step == 1 its update condition `t - 1 % step == 0` is never true)")
            slots = list(bank)
            free = [s for s in range(net.capacity) if s not in slots]
            fresh = free[0]
            net.memorize(fresh)
            slots.append(fresh)
            if t - 1 > 0 and len(slots) > self.memory_capacity:
                slots = slots[1:]
            # the MFMA memory read wants the T live slots contiguous from 0: compact when they are not
            T = len(slots)
            if slots != list(range(T)):
                k = torch.cat([net.kbank[s * hw:(s + 1) * hw] for s in slots]).clone()
                v = torch.cat([net.vbank[s * hw:(s + 1) * hw] for s in slots]).clone()
                net.kbank[:T * hw].copy_(k)
                net.vbank[:T * hw].copy_(v)
                slots = list(range(T))
            nv.stm_pre(framelist[t], None, in_h, in_w, out=net.x8)
            p2 = net.segment(T)
            mask, prob = nv.stm_post(p2, in_h, in_w, H, W)
            masks.append(mask)
            if return_debug:
                debug.append({"p2": p2.clone(), "prob": prob.clone(), "T": T})
            update = ((t - 1) % self.memory_step == 0) if self.fix_memory_step else (t - (1 % self.memory_step) == 0)
            if update:
                bank = slots
        return (masks, debug) if return_debug else masks

    def forward(self, framelist, mask0):
        """NumPy API of the reference (agent.py:203-221): list of frames (h, w, 3) uint8 BGR + first mask -> list of masks."""
        with torch.cuda.device(self.device):
            out = self.forward_device([to_dev(np.ascontiguousarray(f), self.device) for f in framelist],
                                      to_dev(np.ascontiguousarray(mask0), self.device))
            return [to_host(m) for m in out]

