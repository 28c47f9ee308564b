"""Overall background of a clip: stage 2 of the reference's tools/unscreen/bg_offline.py:99-130 (switched off there with
`if 0:`; the shipped loop reads its result, always_bg.jpg, from disk).

Every frame adds its pixels outside the (3x3 x2 dilated) segmentation mask to a running sum and a count; the mean over the
frames in which a pixel was visible is the background, pixels seen in at most 10 frames are a hole to be inpainted.  With
frames sharded over ranks the two accumulators are per-rank partial sums of integers: ONE all-reduce (RCCL over xGMI on the
GPU box) of a single (H*W*4) float32 bucket - 33 MB at 1080p - makes them global, exactly (float32 holds the integer sums).
The hole: the reference calls cv2.inpaint(INPAINT_TELEA), a fast-marching method; this build fills it with the same membrane
interpolation it uses per frame (usc_regionfill_u8) - a stated deviation (SURVEY.md 8f-2 allows the substitution).
"""
import torch
import torch.distributed as dist

from . import _native as nv


class OverallBackground:
    def __init__(self, H, W, device="cuda:0"):
        self.H, self.W = H, W
        self.device = torch.device(device)
        self.acc = torch.zeros(H * W * 4, dtype=torch.float32, device=self.device)       # one bucket: raw (H, W, 3) | cont (H, W)
        self.raw = self.acc[:H * W * 3].view(H, W, 3)
        self.cont = self.acc[H * W * 3:].view(H, W)
        self.frames = 0

    def add(self, frame, segmask):
        """frame (H, W, 3), segmask (H, W) uint8 CUDA (already at the destination size: every frame of a clip has it)."""
        nv.bg_accumulate(frame, nv.morph(segmask, 3, 2, nv.DILATE), self.raw, self.cont)
        self.frames += 1

    def reduce(self):
        """Sum the partial accumulators of all ranks (no-op for a single process)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.acc)

    def finalize(self, fill_rtol=1e-7):
        """-> dict(bg (H, W, 3) uint8 inpainted, bg_always (before inpainting), hole (H, W) uint8)."""
        bg_always, mask_always = nv.bg_finalize(self.raw, self.cont)
        hole = nv.morph(mask_always, 3, 2, nv.DILATE)
        return {"bg": nv.regionfill(bg_always, hole, rtol=fill_rtol), "bg_always": bg_always, "hole": hole}
