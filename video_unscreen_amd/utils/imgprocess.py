"""Geometry helpers of the path (reference unscreen/utils/imgprocess.py:67-192), computed by the C ABI's
pure host functions so that the Python side and the kernels can never disagree on a size."""
from .. import _native as nv


def get_target_size(h, w, target_long_side, division=1):
    """Long side -> target_long_side, short side scaled and rounded UP to a multiple of `division`
    (reference imgprocess.py:164-192)."""
    return nv.get_target_size(h, w, target_long_side, division)


def pad_resize_size(h, w, target_size):
    """Size of the aspect-preserving resize inside `target_size` (reference pad_resize, imgprocess.py:67-97)."""
    return nv.pad_resize_geometry(h, w, target_size[0], target_size[1])


def inv_pad_resize_size(h, w, ori_size):
    """Size of the un-padded region (reference inv_pad_resize, imgprocess.py:100-132)."""
    return nv.inv_pad_geometry(h, w, ori_size[0], ori_size[1])
