"""Foreground / background extraction with the reference's names (reference unscreen/utils/fgfuncs.py)."""
import numpy as np
import torch

from .. import _native as nv
from ._dev import to_dev, to_host


def get_fg(img, alpha, bg):
    """HSV-space foreground estimate (reference fgfuncs.py:84-110)."""
    return to_host(nv.get_fg(to_dev(img), to_dev(alpha), to_dev(bg), paste_thr=0))


def get_bg(alpha, bg):
    """(1 - alpha) * bg in HSV (reference fgfuncs.py:113-137)."""
    return to_host(nv.get_bg(to_dev(alpha), to_dev(bg)))


def is_pixel_inrange(img, bgimg, winsize=(20, 20, 120), long_side_input=-1):
    """Pixels within (bg - winsize//2, bg + winsize//2) in 8-bit HSV (reference fgfuncs.py:9-65); bool mask.  long_side_input > 0:
    the test runs on cv2.resize()d copies (bilinear) and its mask comes back through `cv2.resize(mask, (w, h), cv2.INTER_NEAREST)` -
    the flag lands in `dst`, so that is the default bilinear too - and `> 0`; the mask that is resized holds 0 / 1 in the
    background-image branch (:47-48) and 0 / 255 in the background-colour branch (:60-61), as in the reference."""
    assert bgimg.ndim == 3 or bgimg.ndim == 1
    h, w = img.shape[:2]
    d_img = to_dev(img)
    bg = to_dev(bgimg) if bgimg.ndim == 3 else bgimg
    if long_side_input > 0:
        ih, iw = nv.get_target_size(h, w, long_side_input)
        d_img = nv.resize(d_img, ih, iw, 1)
        if bgimg.ndim == 3:
            bg = nv.resize(bg, ih, iw, 1)
    m = nv.is_pixel_inrange(d_img, bg, winsize)                       # 0 / 1
    if long_side_input > 0:
        if bgimg.ndim != 3:
            m = m * 255                                                 # cv2.inRange's 0 / 255 is what gets resized in this branch
        m = nv.resize(m.contiguous(), h, w, 1)
    return to_host(m) > 0


def get_fg_naive(img, alpha):
    """uint8(img * alpha / 255) in float64 (reference fgfuncs.py:68-81)."""
    d = to_dev(img)
    return to_host(nv.blend(d, to_dev(alpha), torch.zeros_like(d)))      # img * m + 0 * (1 - m): the same float64 products, plus an exact zero


def get_fg_with_colorremove(img, alpha, bg, winsize=(10, 100, 120), long_side_input=960):
    """alpha[near-background pixels] = 0 IN PLACE on the argument, then get_fg (reference fgfuncs.py:140-169)."""
    bg_mask = is_pixel_inrange(img, bg, winsize, long_side_input)
    alpha[bg_mask] = 0
    return get_fg(img, alpha, bg)


def composite_fgbg(fg, alpha, bg, extend=False):
    """fg + resized bg * (1 - alpha) (reference fgfuncs.py:172-214); extend=True returns the whole resized background with the
    composition pasted at its centre (:210-213)."""
    d_fg, d_bg = to_dev(fg), to_dev(bg)
    comp = nv.composite_fgbg(d_fg, to_dev(alpha), d_bg)
    if not extend:
        return to_host(comp)
    fg_h, fg_w = fg.shape[:2]
    bg_h, bg_w = bg.shape[:2]
    if float(fg_h) / fg_w > float(bg_h) / bg_w:
        new_bg_h = fg_h
        new_bg_w = int(float(bg_w) * new_bg_h / bg_h)
    else:
        new_bg_w = fg_w
        new_bg_h = int(float(bg_h) * new_bg_w / bg_w)
    left, top = max(new_bg_w // 2 - fg_w // 2, 0), max(new_bg_h // 2 - fg_h // 2, 0)
    out = nv.resize(d_bg, new_bg_h, new_bg_w, 1)
    out[top:top + fg_h, left:left + fg_w] = comp
    return to_host(out)


def fuse_fgbg(fg, bg, mask):
    """alpha-blend two images with a 0..255 mask in float64, truncated (reference utils/visualize.py:7-24)."""
    return to_host(nv.blend(to_dev(fg), to_dev(mask), to_dev(bg)))
