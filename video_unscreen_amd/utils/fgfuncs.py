"""Foreground / background extraction with the reference's names (reference unscreen/utils/fgfuncs.py)."""
from .. import _native as nv
from ._dev import to_dev, to_host


def get_fg(img, alpha, bg):
    """HSV-space foreground estimate (reference fgfuncs.py:84-110)."""
    return to_host(nv.get_fg(to_dev(img), to_dev(alpha), to_dev(bg), paste_thr=0))


def get_bg(alpha, bg):
    """(1 - alpha) * bg in HSV (reference fgfuncs.py:113-137)."""
    return to_host(nv.get_bg(to_dev(alpha), to_dev(bg)))
