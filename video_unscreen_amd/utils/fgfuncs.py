"""Foreground / background extraction with the reference's names (reference unscreen/utils/fgfuncs.py)."""
from .. import _native as nv
from ._dev import to_dev, to_host


def get_fg(img, alpha, bg):
    """HSV-space foreground estimate (reference fgfuncs.py:84-110)."""
    return to_host(nv.get_fg(to_dev(img), to_dev(alpha), to_dev(bg), paste_thr=0))


def get_bg(alpha, bg):
    """(1 - alpha) * bg in HSV (reference fgfuncs.py:113-137)."""
    return to_host(nv.get_bg(to_dev(alpha), to_dev(bg)))


def is_pixel_inrange(img, bgimg, winsize=(20, 20, 120), long_side_input=-1):
    """Pixels within (bg - winsize//2, bg + winsize//2) in 8-bit HSV (reference fgfuncs.py:9-65); bool mask."""
    if long_side_input > 0:
        raise NotImplementedError("is_pixel_inrange with long_side_input > 0 (only get_fg_with_colorremove uses it) is not built")
    assert bgimg.ndim == 3 or bgimg.ndim == 1
    bg = to_dev(bgimg) if bgimg.ndim == 3 else bgimg
    return to_host(nv.is_pixel_inrange(to_dev(img), bg, winsize)) > 0


def composite_fgbg(fg, alpha, bg, extend=False):
    """fg + resized bg * (1 - alpha) (reference fgfuncs.py:172-214)."""
    if extend:
        raise NotImplementedError("composite_fgbg(extend=True) is not built (no caller in the reference tools)")
    return to_host(nv.composite_fgbg(to_dev(fg), to_dev(alpha), to_dev(bg)))


def fuse_fgbg(fg, bg, mask):
    """alpha-blend two images with a 0..255 mask in float64, truncated (reference utils/visualize.py:7-24)."""
    return to_host(nv.blend(to_dev(fg), to_dev(mask), to_dev(bg)))
