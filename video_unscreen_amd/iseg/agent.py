"""ISegAgent: same constructor and forward() as the reference (unscreen/iseg/agent.py:10-119), the no-BRS predictor only.

forward(img, click_history) = pad_resize to the target size, imnormalize, clicks scaled by the resize ratio (agent.py:92-104),
the model DistMapsModel(DeepLabV3Plus(resnet50_v1s), SepConvHead) on the image and - with_flip - on its mirror with mirrored
clicks, the pair averaged on the logits, sigmoid (brs_predictor.py:35-106, transforms.py), inv_pad_resize, > prob_thresh
(agent.py:116-119).  Input and output stages are HIP kernels (csrc/usc_iseg.hip), the network runs on the HIP conv kernels
(net_hip.HipIsegNet).

Two things about the reference, stated rather than copied:
* with_brs=True (f-BRS: L-BFGS over scale / bias maps inside the network, brs_predictor.py:108-255, brs_optimizer.py) is an
  interactive optimiser, not per-frame data-parallel work: SURVEY.md section 2 row 16 puts it out of scope, and it raises here.
* In the reference, with_brs=False builds a BasePredictor whose get_prediction() calls the abstract _get_clicks_maps()
  (brs_predictor.py:37, 104-105) and so raises NotImplementedError before the network runs (pinned by tests/golden/gen_iseg.py).
  The click MAPS are not an input of that predictor's network call (:53-56 uses the click POINTS); this agent computes what
  the code describes with the maps left out.
"""
import numpy as np
import torch

from .. import _native as nv
from ..utils._dev import device, to_dev, to_host
from .net import iseg_synthetic_state_dict
from .net_hip import HipIsegNet

NET_CLICKS_LIMIT = 20             # predictor_params_ = {'net_clicks_limit': 20} (agent.py:83)


class ISegAgent():

    def __init__(self, model_path=None, with_brs=False, input_long_side=800, prob_thresh=0.5, with_flip=True, cuda_device=0,
                 state_dict=None):
        if with_brs:
            raise NotImplementedError("ISegAgent(with_brs=True): the f-BRS back-propagating refinement is an interactive L-BFGS optimiser "
                                      "(iseg/brs/brs_predictor.py:108-255) and is not built (SURVEY.md section 2 row 16)")
        self.cuda_device = cuda_device if cuda_device is not None and cuda_device >= 0 else 0
        self.device = device(self.cuda_device)
        if state_dict is not None:
            sd = state_dict
        elif model_path is None or model_path == "synthetic":
            sd = iseg_synthetic_state_dict(0)          # no trained checkpoint exists offline
        else:
            sd = torch.load(model_path, map_location="cpu")          # a bare state dict (reference agent.py:20)
        self.state_dict = sd
        self.input_long_side = input_long_side
        self.prob_thresh = prob_thresh
        self.with_flip = with_flip
        self._nets = {}

    MAX_NETS = 2          # network instances kept (one per input size): each owns its activations and a split-K slab

    def model_for(self, in_h, in_w):
        """The network instance for this input size; the least recently used one is closed when a third size comes along (interactive use
        over clips of many sizes would otherwise grow device memory without bound)."""
        key = (in_h, in_w, bool(self.with_flip))
        net = self._nets.pop(key, None)
        if net is None:
            while len(self._nets) >= self.MAX_NETS:
                old_key = next(iter(self._nets))
                self._nets.pop(old_key).close()
            net = HipIsegNet(self.state_dict, in_h, in_w, device=self.device, batch=2 if self.with_flip else 1)
        self._nets[key] = net                            # (re-inserted: most recently used last)
        return net

    def points_nd(self, click_history, ratio, in_w=None):
        """The (2 P, 2) point table of brs_predictor.py:64-99 (get_points_nd) for the un-mirrored image: P = max(#positive,
        #negative) of the first net_clicks_limit clicks, at least 1, at most net_clicks_limit; positive slots first, (-1, -1) =
        empty; coordinates int(y * ratio), int(x * ratio) (agent.py:98-99)."""
        clicks = [(bool(r[0]), int(r[1] * ratio), int(r[2] * ratio)) for r in click_history]
        npos = sum(c[0] for c in clicks)
        P = max(1, min(NET_CLICKS_LIMIT, max(npos, len(clicks) - npos)))
        clicks = clicks[:NET_CLICKS_LIMIT]
        pos = [(y, x) for p, y, x in clicks if p]
        neg = [(y, x) for p, y, x in clicks if not p]
        rows = pos + [(-1, -1)] * (P - len(pos)) + neg + [(-1, -1)] * (P - len(neg))
        return np.asarray(rows, np.float32).reshape(2 * P, 2)

    def forward_device(self, img, click_history, return_debug=False):
        """img (H, W, 3) uint8 BGR CUDA tensor -> (H, W) uint8 mask {0, 255} on the device."""
        H, W, _ = img.shape
        in_h, in_w = nv.get_target_size(H, W, self.input_long_side, 1)
        ratio = float(in_h) / H if float(H) / W > float(in_h) / in_w else float(in_w) / W          # pad_resize's ratio (imgprocess.py:82-92)
        pts = self.points_nd(click_history, ratio)
        net = self.model_for(in_h, in_w)
        nv.iseg_pre(img, in_h, in_w, pts, net.rgb_params(), flip=self.with_flip, out=net.x4)
        logits = net.forward(None)
        prob, mask = nv.iseg_post(logits, in_h, in_w, self.with_flip, H, W, self.prob_thresh)
        if return_debug:
            return mask, {"prob": prob, "logits": logits, "points": pts, "x4": net.x4, "in_size": (in_h, in_w)}
        return mask

    def forward(self, img, click_history):
        """img (h, w, 3) uint8 BGR, click_history [(is_positive, y, x), ...] -> (h, w) uint8, 255 = object (agent.py:106-119)."""
        with torch.cuda.device(self.device):
            return to_host(self.forward_device(to_dev(np.ascontiguousarray(img), self.device), click_history))
