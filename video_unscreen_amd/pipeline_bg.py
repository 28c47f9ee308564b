"""Device-resident bg online mode: tools/unscreen/bg.py:45-104 of the reference.

Per frame:
    segmask   = STMAgent([prev frame, frame], binarised alpha_pre)[-1]  (tracking)  |  HumanSegAgent(frame)      bg.py:49-56
    exist_foreground(segmask)                                                                                    :60
    alphaor   = remove_invalid_objects(segmask) -> trimap -> matting net            (pass 1)             :67-70
    bg        = get_bg(alpha, frame);  hole = dilate3x3x2(alpha > 128);  bgimg = regionfill(bg, hole) per channel :71-81
    alphabg   = dilate4x4x2(gray(|frame - bgimg|) > thr);  alpha_ensm = alpha * (alphabg // 255)                  :85-92
    alphaor   = remove_invalid_objects(alpha_ensm) -> trimap -> matting net          (pass 2)             :93-95
    bgimg[alpha == 0] = frame[alpha == 0];  fg = get_fg(frame, alpha, bgimg)                                      :99-100
    alpha_pre = alpha;  tracking = exist_foreground(alpha)                                                        :103-104
"""
import torch

from . import _native as nv
from .binseg import HumanSegAgent
from .stm import STMAgent
from .trimap import TrimapAgent
from .vmatting import VMattingAgent


class BgOnlinePipeline:
    def __init__(self, cfg, H, W, device="cuda:0", segagent=None, stmagent=None, vmatagent=None, fill_rtol=1e-7):
        self.cfg = cfg
        self.H, self.W = H, W
        self.device = torch.device(device)
        idx = self.device.index or 0
        self.segagent = segagent if segagent is not None else HumanSegAgent(**dict(cfg["binseg"], cuda_device=idx))
        self.stmagent = stmagent if stmagent is not None else STMAgent(**dict(cfg["stm"], cuda_device=idx))
        self.trimapagent = TrimapAgent(**cfg["trimap"])
        self.vmatagent = vmatagent if vmatagent is not None else VMattingAgent(**dict(cfg["vmatting"], cuda_device=idx))
        centers = cfg["objectremoval"]["score_map_center"]
        self.center = centers["landscape"] if W > H else centers["portrait"]
        self.filter = nv.DeviceObjectFilter()
        self.fill_rtol = fill_rtol
        self._cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._ws_fill = torch.empty(nv.load().usc_regionfill_ws_bytes(H, W), dtype=torch.uint8, device=self.device)
        self._ws_diff = torch.empty(nv.load().usc_bgdiff_ws_bytes(H, W), dtype=torch.uint8, device=self.device)
        self._fill_status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._fill_pending = None
        self.reset()

    def reset(self):
        self.tracking_flag = False
        self.alpha_pre = torch.zeros((self.H, self.W), dtype=torch.uint8, device=self.device)
        self.prev_frame = None
        self.fill_iters = 0

    def _fill(self, bg, hole):
        """Inpaint the hole without a host synchronisation of its own: the iteration budget follows the previous frames (their
        count + a margin; 48 to start with), the convergence test runs on the device, and the 8-byte status is read at the
        step's next natural synchronisation point (exist_foreground) - if a budget ever turns out too small the solve is
        continued there, before anything downstream has been consumed."""
        # (every iteration past convergence is 13 launches that return at their first instruction, ~30 us: the margin is kept small;
        # the iteration count of consecutive frames barely moves, and a short budget only costs one continuation below)
        budget = 48 if self.fill_iters <= 0 else min(400, self.fill_iters + max(4, self.fill_iters // 6))
        bgimg = nv.regionfill(bg, hole, rtol=self.fill_rtol, ws=self._ws_fill, budget=budget, sync=False, status=self._fill_status)
        self._fill_pending = (bg, hole, bgimg, budget)
        return bgimg

    def _fill_check(self):
        """Called right after a host synchronisation: look at the inpainting status of this frame; True if the fill had to be
        continued (the caller then recomputes what it derived from the background)."""
        if self._fill_pending is None:
            return False
        bg, hole, bgimg, budget = self._fill_pending
        self._fill_pending = None
        used, conv = (int(v) for v in self._fill_status.cpu())
        redo = not conv
        it = budget
        while not conv and it < 2000:
            nv.load().usc_regionfill_u8(bg.data_ptr(), hole.data_ptr(), self.H, self.W, bgimg.data_ptr(), float(self.fill_rtol), it, 16,
                                        self._fill_status.data_ptr(), self._ws_fill.data_ptr(), self._ws_fill.numel(),
                                        torch.cuda.current_stream().cuda_stream)
            it += 16
            used, conv = (int(v) for v in self._fill_status.cpu())
        self.fill_iters = used
        return redo

    def _exist_foreground(self, mask):
        return int(nv.count_ge(mask, 128, out=self._cnt).item()) > self.cfg["fg_exist_thr"] * self.H * self.W

    def _object_filter(self, mask):
        return self.filter(torch.empty_like(mask), None, self.center, self.cfg["objectremoval"]["saliency_thr"],
                           self.cfg["objectremoval"]["consensus_thr"], src=mask)        # filtered copy; the copy rides on the first scan

    def step(self, frame, passes=2):
        """frame (H, W, 3) uint8 BGR CUDA -> dict(segmask, alpha, fg, bg, + the intermediates of every stage).
        passes=1 stops after the per-frame background: stage 1 of tools/unscreen/bg_offline.py:51-97 (segmask, one matting
        pass, bgimg; alpha_pre / tracking follow the pass-1 alpha) - no bg difference, second pass or foreground."""
        cfg = self.cfg
        tracking = self.tracking_flag
        if tracking:
            seed = torch.where(self.alpha_pre >= 128, torch.full_like(self.alpha_pre, 255), self.alpha_pre)
            segmask = self.stmagent.forward_device([self.prev_frame, frame], seed)[-1]
        else:
            segmask = self.segagent.forward_device(frame)
        out = {"segmask": segmask, "tracking": tracking}
        if not self._exist_foreground(segmask):
            alpha, fg = torch.zeros_like(segmask), torch.zeros_like(frame)
        else:
            alphaor1 = self._object_filter(segmask)
            trimap1 = self.trimapagent.forward_device(alphaor1)
            alpha1 = self.vmatagent.forward_device(frame, self.alpha_pre, trimap1)
            bg = nv.get_bg(alpha1, frame)
            hole = nv.morph((alpha1 > 128).to(torch.uint8) * 255, 3, 2, nv.DILATE)
            bgimg = self._fill(bg, hole)
            if passes == 1:
                self.alpha_pre = alpha1
                self.tracking_flag = self._exist_foreground(alpha1)
                self._fill_check()                           # (continues the fill in place if its budget was too small)
                self.prev_frame = frame
                out.update(alphaor1=alphaor1, trimap1=trimap1, alpha1=alpha1, alpha=alpha1, bg_holes=bg, hole=hole, bgimg=bgimg)
                return out

            def second_pass():
                _, alpha_ensm = nv.bgdiff_mask(frame, bgimg, None, alpha1, 1.0, cfg["bg_mask"]["thr"], ws=self._ws_diff)
                alphaor2 = self._object_filter(alpha_ensm)
                trimap2 = self.trimapagent.forward_device(alphaor2)
                alpha = self.vmatagent.forward_device(frame, self.alpha_pre, trimap2)
                fg, bgp = nv.get_fg(frame, alpha, bgimg, paste_thr=1, want_bg=True)
                out.update(alphaor1=alphaor1, trimap1=trimap1, alpha1=alpha1, bg_holes=bg, hole=hole, bgimg=bgimg, alpha_ensm=alpha_ensm,
                           alphaor2=alphaor2, trimap2=trimap2, bg=bgp)
                return alpha, fg

            alpha, fg = second_pass()
            tracking_next = self._exist_foreground(alpha)      # host synchronisation: the fill's status is known now
            if self._fill_check():                            # budget too small (rare): the fill was continued, redo what used it
                alpha, fg = second_pass()
                tracking_next = self._exist_foreground(alpha)
            self.alpha_pre = alpha
            self.tracking_flag = tracking_next
            self.prev_frame = frame
            out.update(alpha=alpha, fg=fg)
            return out
        self.alpha_pre = alpha
        self.tracking_flag = self._exist_foreground(alpha)
        self.prev_frame = frame
        out.update(alpha=alpha, fg=fg)
        return out
