"""SegAgent: same constructor and forward() as the reference (unscreen/binseg/agent.py:14-260).

forward(img) = resize / reflect-pad to the target, normalise, overlapping 513 x 513 crops (+ mirrored copies),
DeepLabv3+/ResNet-50 per crop, softmax, float64 merge, inverse pad-resize, argmax.  Pre- and post-processing are HIP
kernels (csrc/usc_segagent.hip); every convolution runs on the f32-MFMA kernel (deeplab_net.HipDeepLabNet), `crop_batch`
crops per network forward (4: 3.4 ms per crop at 1080p against 5.2 ms one by one).  The conv kernels plan their K
partitions for `plan_batch` crops whatever the size of a forward, so the result does not depend on `crop_batch` (full and
tail groups, or crop-by-crop execution, give the same bits).
"""
import numpy as np
import torch

from .. import _native as nv
from ..utils._dev import device, to_dev, to_host
from .deeplab_net import HipDeepLabNet, deeplab_synthetic_state_dict


class SegAgent():

    def __init__(self, model_path=None, input_long_side=912, crop_h=513, crop_w=513, stride_ratio=1 / 2., flip=True,
                 cuda_device=0, crop_batch=4, plan_batch=4, state_dict=None):
        self.cuda_device = cuda_device if cuda_device is not None and cuda_device >= 0 else 0
        self.device = device(self.cuda_device)
        if state_dict is not None:                     # weights handed over in memory (broadcast from rank 0)
            sd = state_dict
        elif model_path is None or model_path == "synthetic":
            sd = deeplab_synthetic_state_dict(0)       # no trained checkpoint exists offline
        else:
            sd = torch.load(model_path, map_location="cpu")        # a bare state dict (reference agent.py:58)
        self.state_dict = sd
        self.division = 1
        self.crop_h, self.crop_w = crop_h, crop_w
        self.flip = flip
        self.input_long_side = input_long_side
        self.stride_ratio = stride_ratio
        self.crop_batch = max(1, int(crop_batch))       # crops per network forward (results do not depend on it)
        self.plan_batch = max(1, int(plan_batch))       # batch the conv kernels plan their K partitions for (results do)
        self._models = {}
        self.model = self.model_for(1)
        self._logits = None
        self._ws = None

    def model_for(self, batch):
        if batch not in self._models:
            self._models[batch] = HipDeepLabNet(self.state_dict, self.crop_h, self.crop_w, device=self.device, batch=batch,
                                                 plan_batch=self.plan_batch)   # one K partition for full and tail groups
        return self._models[batch]

    def get_target_size(self, h, w):
        """reference agent.py:70-100."""
        return nv.segagent_target_size(h, w, self.input_long_side, self.division, self.crop_h, self.crop_w)

    def grid_crop_locations(self, h, w, stride_ratio=None):
        """Locations of reference grid_crop (agent.py:102-141) on an (h, w) image; the crops themselves are gathered on the
        device by usc_segagent_pre."""
        return nv.grid_crop_locations(h, w, self.crop_h, self.crop_w, self.stride_ratio if stride_ratio is None else stride_ratio,
                                      self.flip)

    def forward_device(self, img, out=None, return_logits=False):
        """img: (H, W, 3) uint8 BGR CUDA tensor -> (H, W) uint8 mask {0, 255} on the device."""
        H, W, _ = img.shape
        th, tw = self.get_target_size(H, W)
        locs = self.grid_crop_locations(th, tw)
        n = len(locs)
        oh, ow = self.model.out_h, self.model.out_w
        if self._logits is None or self._logits.shape[0] != n:
            self._logits = torch.empty((n, oh, ow, 32), dtype=torch.float32, device=self.device)
        for i0 in range(0, n, self.crop_batch):
            nb = min(self.crop_batch, n - i0)
            net = self.model_for(nb)
            for b in range(nb):
                nv.segagent_pre(img, th, tw, locs[i0 + b], out=net.image(net.x4, b))
            self._logits[i0:i0 + nb].copy_(net.forward(None).view(nb, oh, ow, 32))
        need = nv.load().usc_segagent_post_ws_bytes(th, tw)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        mask = nv.segagent_post(self._logits, locs, self.crop_h, self.crop_w, th, tw, H, W, out=out, ws=self._ws)
        return (mask, self._logits, locs) if return_logits else mask

    def forward(self, img):
        """img (h, w, 3) uint8 BGR -> (h, w) uint8, 255 = foreground (reference agent.py:239-260)."""
        with torch.cuda.device(self.device):
            return to_host(self.forward_device(to_dev(np.ascontiguousarray(img), self.device)))
