"""Device-resident green-screen frame loop: tools/unscreen/green.py:70-138 of the reference.

Per frame (every image stays in HBM; host work only where the reference's algorithm is sequential or a library fit):

    segmask   = alpha_pre (tracking)  |  SegAgent (DeepLabv3+ over 12-18 crops)      green.py:74-80
    exist_foreground(segmask)                                                          :82
    alphacf, bg = ColorFilteringAgent(frame, segmask, iters in {0, train_iters})       :88-100   (EM fit on the host)
    alphaor   = remove_invalid_objects(alphacf[, segmask])   (device, label based)     :106-109
    trimap    = generate_trimap_withbg(alphaor, frame, bg_color)                       :114
    alpha     = matting net                                                            :115
    alpha     = color_correct(frame, alpha, bg_color)                                  :120
    fg        = get_fg(frame, alpha, bg pasted where alpha < 128)                      :125-126
    tracking  = exist_foreground(alpha);  alpha_pre = alpha                            :137-138
"""
import numpy as np
import torch

from . import _native as nv
from .binseg import SegAgent
from .colorfiltering import ColorFilteringAgent
from .trimap import TrimapAgent
from .vmatting import VMattingAgent


class GreenPipeline:
    def __init__(self, cfg, H, W, device="cuda:0", segagent=None, vmatagent=None, cf_random_state=None):
        self.cfg = cfg
        self.H, self.W = H, W
        self.device = torch.device(device)
        idx = self.device.index or 0
        self.segagent = segagent if segagent is not None else SegAgent(**dict(cfg["binseg"], cuda_device=idx))
        self.trimapagent = TrimapAgent(**cfg["trimap"])
        self.vmatagent = vmatagent if vmatagent is not None else VMattingAgent(**dict(cfg["vmatting"], cuda_device=idx))
        self.cfagent = ColorFilteringAgent(**cfg["colorfiltering"], random_state=cf_random_state)
        centers = cfg["objectremoval"]["score_map_center"]
        self.center = centers["landscape"] if W > H else centers["portrait"]
        self.filter = nv.DeviceObjectFilter()
        self._cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.reset()

    def reset(self):
        self.cfagent.reset_gmms()
        self.tracking_flag = False
        self.alpha_pre = torch.zeros((self.H, self.W), dtype=torch.uint8, device=self.device)
        self.fid = 0
        self.tracking_count = 0

    def _exist_foreground(self, mask):
        """count(mask >= 128) > thr * h * w (reference utils/maskprocess.py:56-60)."""
        return int(nv.count_ge(mask, 128, out=self._cnt).item()) > self.cfg["fg_exist_thr"] * self.H * self.W

    def step(self, frame):
        """frame (H, W, 3) uint8 BGR CUDA -> dict(alpha, fg, bg, + the intermediates of every stage)."""
        cfg = self.cfg
        tracking = self.tracking_flag
        if tracking:
            segmask = self.alpha_pre.clone()
            self.tracking_count += 1
        else:
            segmask = self.segagent.forward_device(frame)
        out = {"segmask": segmask, "tracking": tracking}
        # tracking: segmask is the previous alpha, whose exist_foreground() set the flag at the end of the last step
        if not (tracking or self._exist_foreground(segmask)):
            alpha, fg, bgimg = torch.zeros_like(segmask), torch.zeros_like(frame), frame
            out.update(cf_iters=None)
        else:
            if self.fid % cfg["colorfiltering_update_duration"] == 0 or not tracking or not self.cfagent.is_trained():
                cf_iters = cfg["colorfiltering_train_iters"]
            else:
                cf_iters = 0
            alphacf, bgimg, _ = self.cfagent.forward_device(frame, segmask, iters=cf_iters)
            # the fitted background colour is known on the host; only the agent's early-outs hand back an image of their own
            bg_color = self.cfagent.last_bg_color if self.cfagent.last_bg_color is not None else bgimg[0, 0].cpu().numpy()
            # object filter on the device (labels, no border following): consensus against segmask unless tracking
            alphaor = self.filter(torch.empty_like(alphacf), None if tracking else segmask, self.center, cfg["objectremoval"]["saliency_thr"],
                                  cfg["objectremoval"]["consensus_thr"], src=alphacf)
            # (an empty mask comes back as an all-zero trimap - what generate_trimap_withbg's early return hands on - without
            # asking the host: the kernels decide on device counters)
            trimap = self.trimapagent.forward_device_withbg(alphaor, frame, bg_color)
            alpha_mat = self.vmatagent.forward_device(frame, self.alpha_pre, trimap)
            alpha = nv.color_correct(frame, alpha_mat, bg_color)
            fg, bgimg = nv.get_fg(frame, alpha, bgimg.contiguous(), paste_thr=128, want_bg=True)
            out.update(cf_iters=cf_iters, alphacf=alphacf, bg_color=bg_color, alphaor=alphaor, trimap=trimap, alpha_mat=alpha_mat)
        self.tracking_flag = self._exist_foreground(alpha)
        self.alpha_pre = alpha
        self.fid += 1
        out.update(alpha=alpha, fg=fg, bg=bgimg)
        return out
