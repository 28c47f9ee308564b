"""TrimapAgent: same constructor and forward() as the reference (unscreen/trimap/agent.py:25-33, 103-128);
the arithmetic is one call into libunscreen_hip.so (usc_trimap_u8)."""
import torch

from .. import _native as nv
from ..utils._dev import to_dev, to_host


class TrimapAgent():

    def __init__(self, input_long_side=960, kernelsize=3, iters=5, color_winsize=(10, 100, 180)):
        self.kernelsize = kernelsize
        self.iters = iters
        self.input_long_side = input_long_side
        self.color_winsize = color_winsize
        self._ws = None

    # ---- device-resident path (torch CUDA uint8 (H, W) in, same out) -------------------------------
    def forward_device(self, mask: torch.Tensor, out=None) -> torch.Tensor:
        need = nv.load().usc_trimap_ws_bytes(mask.shape[0], mask.shape[1], int(self.input_long_side))
        if self._ws is None or self._ws.numel() < need or self._ws.device != mask.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=mask.device)
        return nv.trimap(mask, self.input_long_side, self.kernelsize, self.iters, out=out, ws=self._ws)

    # ---- reference API: NumPy uint8 in / out --------------------------------------------------------
    def generate_trimap(self, mask):
        """mask (h, w) uint8 -> trimap with values {0, 128, 255} (reference agent.py:35-61)."""
        return to_host(self.forward_device(to_dev(mask)))

    def forward_device_withbg(self, mask, img, bg, out=None):
        """mask (H, W), img (H, W, 3) CUDA uint8; bg: CUDA (H, W, 3) image or a 3-value BGR colour."""
        return nv.trimap_withbg(mask, img, bg, self.color_winsize, self.input_long_side, self.kernelsize, self.iters, out=out)

    def generate_trimap_withbg(self, mask, img, bgimg):
        """Background-aware trimap (reference agent.py:63-101).  Like the reference, an empty mask is returned as
        the very same object."""
        if (mask > 0).sum() == 0:
            return mask
        bg = to_dev(bgimg) if getattr(bgimg, "ndim", 1) == 3 else bgimg
        return to_host(self.forward_device_withbg(to_dev(mask), to_dev(img), bg))

    def forward(self, *args, **kwargs):
        if len(args) > 2:
            return self.generate_trimap_withbg(*args, **kwargs)
        return self.generate_trimap(*args, **kwargs)
