"""HarmonizationAgent helpers (reference unscreen/harmonization/agent.py:7-113) on the HIP kernels: region means at a
reduced size, Lab mean-shift toning of the foreground (8-bit BGR -> Lab -> BGR), iterated 3x3 box smoothing of the alpha map /
the background."""
import numpy as np
import torch

from .. import _native as nv
from ..utils._dev import to_dev, to_host


class HarmonizationAgent():

    def __init__(self):
        pass

    def get_means(self, img, mask=None, target_long_side=240):
        """Per-channel mean of the bilinear-downsized image, optionally inside the (downsized > 0) mask (agent.py:14-37)."""
        d = img if torch.is_tensor(img) else to_dev(img)
        th, tw = nv.get_target_size(img.shape[0], img.shape[1], target_long_side)
        small = nv.resize(d, th, tw, 1)
        if mask is None:
            samples = small.reshape(-1, 3)
        else:
            m = nv.resize(to_dev(np.asarray(mask).astype(np.uint8) * 255), th, tw, 1) > 0
            samples = small[m].reshape(-1, 3)
        # integer sums are exact; the division is the float64 one numpy's mean ends with
        return (samples.to(torch.int64).sum(dim=0).cpu().numpy().astype(np.float64) / samples.shape[0])

    def foreground_toning(self, fg, bg, alpha, toning_ratio=(0.5, 0.05, 0.05), max_shift=15, min_shift=-30):
        """Shift the foreground's Lab channel means towards the background's (agent.py:39-76): both images to 8-bit Lab, region
        means at long side 240, per channel shift = clamp(ratio * (bg_mean - fg_mean), min_shift, max_shift) added in float64,
        clip to [0, 255], truncate, Lab -> BGR."""
        fg_lab = nv.cvt_color(to_dev(fg), nv.BGR2LAB)
        bg_lab = nv.cvt_color(to_dev(bg), nv.BGR2LAB)
        fg_means = self.get_means(fg_lab, np.asarray(alpha) > 0)
        bg_means = self.get_means(bg_lab)
        shifts = [max(min(toning_ratio[i] * (bg_means[i] - fg_means[i]), max_shift), min_shift) for i in range(3)]
        shift = torch.tensor(shifts, dtype=torch.float64, device=fg_lab.device)
        toned = torch.clamp(fg_lab.to(torch.float64) + shift, 0, 255).to(torch.uint8)      # np.clip(...).astype(np.uint8): truncation
        return to_host(nv.cvt_color(toned, nv.LAB2BGR))

    def _smooth(self, img, iters, ksize, target_long_side):
        d = to_dev(img)
        h, w = img.shape[:2]
        th, tw = nv.get_target_size(h, w, target_long_side)
        d = nv.resize(d, th, tw, 1)
        for _ in range(iters):
            d = nv.box_filter(d, ksize)
        return to_host(nv.resize(d, h, w, 1))

    def alpha_smoothing(self, alpha, iters=3, ksize=3, target_long_side=1920):
        """Resize, `iters` box filters, resize back (agent.py:76-94)."""
        return self._smooth(alpha, iters, ksize, target_long_side)

    def background_blurring(self, bg, iters=3, ksize=3, target_long_side=480):
        """agent.py:96-113."""
        return self._smooth(bg, iters, ksize, target_long_side)
