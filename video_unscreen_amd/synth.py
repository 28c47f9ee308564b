"""Seeded synthetic clips of the shape BASELINE.json names (there is no dataset offline).

A clip is a static smooth background B (bilinear blend of four random corner colours + N(0, 2^2) noise), one
or two textured ellipses ("person", ~15 % of the frame) moving on a Lissajous path with a 6-pixel feathered
edge, sensor noise N(0, 1), and the inputs the bg_step stage-3 loop reads besides the frame
(tools/unscreen/bg_offline.py:134-141): a coarse segmentation mask, a per-frame background estimate and the
overall background.  Everything is generated with torch ops so the same code fills HBM directly on the GPU
box (bench) or runs on the CPU (tests); the stream is seeded with 20260000 + config_id.
"""
import math

import torch


class SyntheticClip:
    def __init__(self, H, W, nframes, config_id=2, device="cpu", n_objects=2):
        self.H, self.W, self.nframes = H, W, nframes
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device).manual_seed(20260000 + config_id)
        g = self.gen
        corners = torch.randint(40, 216, (2, 2, 3), generator=g, device=self.device).float()
        if config_id == 1:                       # green screen: (60+-10, 200+-20, 60+-10) BGR
            base = torch.tensor([60., 200., 60.], device=self.device)
            spread = torch.tensor([10., 20., 10.], device=self.device)
            corners = base + (torch.rand(2, 2, 3, generator=g, device=self.device) * 2 - 1) * spread
        ys = torch.linspace(0, 1, H, device=self.device).view(H, 1, 1)
        xs = torch.linspace(0, 1, W, device=self.device).view(1, W, 1)
        top = corners[0, 0] * (1 - xs) + corners[0, 1] * xs
        bot = corners[1, 0] * (1 - xs) + corners[1, 1] * xs
        self.B = top * (1 - ys) + bot * ys                                    # (H, W, 3) float
        noise = torch.randn(H, W, 3, generator=g, device=self.device) * 2.0
        self.bg_always = (self.B + noise).round().clamp(0, 255).to(torch.uint8)
        self.yy = torch.arange(H, device=self.device, dtype=torch.float32).view(H, 1)
        self.xx = torch.arange(W, device=self.device, dtype=torch.float32).view(1, W)
        area = 0.15 * H * W / n_objects
        self.objs = []
        for k in range(n_objects):
            ratio = 1.6 + 0.5 * k                                             # tall ellipses
            b = math.sqrt(area / (math.pi * ratio))                           # horizontal semi-axis
            a = b * ratio                                                     # vertical semi-axis
            c0 = (H * (0.5 + 0.05 * k), W * (0.35 + 0.3 * k))
            amp = (H * 0.08, W * 0.10)
            self.objs.append((a, b, c0, amp, k))

    def _alpha_true(self, t):
        alpha = torch.zeros(self.H, self.W, device=self.device)
        for (a, b, c0, amp, k) in self.objs:
            cy = c0[0] + amp[0] * math.sin(2 * math.pi * (t + 40 * k) / 150.)
            cx = c0[1] + amp[1] * math.cos(2 * math.pi * (t + 25 * k) / 100.)
            r = torch.sqrt(((self.yy - cy) / a) ** 2 + ((self.xx - cx) / b) ** 2)
            d = (r - 1.0) * min(a, b)                                         # ~ signed distance in pixels
            alpha = torch.maximum(alpha, (0.5 - d / 6.0).clamp(0, 1))
        return alpha

    def frame(self, t):
        """Returns dict(frame (H,W,3) u8, seg (H,W) u8 {0,255}, bg (H,W,3) u8, alpha_true (H,W) float)."""
        g = self.gen
        alpha = self._alpha_true(t)
        tex = 128 + 30 * torch.randn(self.H, self.W, 3, generator=g, device=self.device)
        a3 = alpha.unsqueeze(-1)
        img = a3 * tex + (1 - a3) * self.B + torch.randn(self.H, self.W, 3, generator=g, device=self.device)
        frame = img.round().clamp(0, 255).to(torch.uint8)
        shift = float(torch.randint(-3, 4, (1,), generator=g, device=self.device).item())
        seg = ((alpha - shift / 12.0) > 0.5).to(torch.uint8) * 255            # edge moved by `shift`/2 pixels
        bg = (self.B + torch.randn(self.H, self.W, 3, generator=g, device=self.device)).round().clamp(0, 255).to(torch.uint8)
        return {"frame": frame, "seg": seg, "bg": bg, "alpha_true": alpha}

    def frames(self, start=0, stop=None):
        for t in range(start, self.nframes if stop is None else stop):
            yield self.frame(t)
