"""Geometry helpers of the path (reference unscreen/utils/imgprocess.py:67-192), computed by the C ABI's
pure host functions so that the Python side and the kernels can never disagree on a size."""
import numpy as np

from .. import _native as nv
from ._dev import to_dev, to_host


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


def get_center(img):
    """Centre of mass of the gray image, (cX, cY) (reference imgprocess.py:7-14: COLOR_RGB2GRAY, cv2.moments, int())."""
    _, mom = nv.gray_moments(to_dev(img), rgb_order=True)
    m00, m10, m01 = (float(v) for v in mom.tolist())            # exact integer sums, as cv2.moments reports them for 8-bit data
    return (int(m10 / (m00 + 1e-6)), int(m01 / (m00 + 1e-6)))


def get_mask(img):
    """(mask (h, w, 1) uint8 0 / 255, binary mask (h, w, 1) float64 0 / 1): BGR2GRAY > 25 (reference imgprocess.py:17-30)."""
    assert img.shape[2] == 3
    thresh = to_host(nv.gray_moments(to_dev(img), rgb_order=False, thr=25, want_moments=False)[0])
    binary_mask = thresh / 255
    return thresh[..., np.newaxis], binary_mask[..., np.newaxis]


def pad_resize(img, target_size):
    """Aspect-preserving cv2.resize + bottom / right BORDER_REFLECT pad to target_size -> (image, ratio)
    (reference imgprocess.py:67-97; uint8 images, as every caller passes)."""
    target_h, target_w = target_size
    h, w = img.shape[:2]
    ratio = float(target_h) / h if float(h) / w > float(target_h) / target_w else float(target_w) / w
    if img.dtype != np.uint8:
        raise TypeError("pad_resize: uint8 images only (the reference's callers pass frames, trimaps and masks)")
    return to_host(nv.pad_resize(to_dev(img), int(target_h), int(target_w))), ratio


def inv_pad_resize(img, ori_size):
    """Inverse of pad_resize: cut the pad, cv2.resize to ori_size (reference imgprocess.py:100-132); uint8, float32 or float64,
    (h, w) or (h, w, c)."""
    if img.dtype not in (np.uint8, np.float32, np.float64):
        raise TypeError(f"inv_pad_resize: dtype {img.dtype}")
    return to_host(nv.inv_pad_resize(to_dev(img), int(ori_size[0]), int(ori_size[1])))


def imnormalize(img, mean=np.array([0.485, 0.456, 0.406]), std=np.array([0.229, 0.224, 0.225]), to_rgb=True):
    """((img / 255) - mean) / std in float32; like the reference (imgprocess.py:135-161, :157) the ARGUMENT is converted from BGR
    to RGB in place when to_rgb."""
    d = to_dev(img)
    out = nv.imnormalize(d, mean, std, to_rgb)
    if to_rgb:
        img[...] = to_host(d)
    return to_host(out)


def adaptive_resize(img, img_target):
    """cv2.resize(img, size of img_target), default bilinear (reference imgprocess.py:33-37)."""
    return to_host(nv.resize(to_dev(img), img_target.shape[0], img_target.shape[1], 1))


def rescale_fg(img, scale_factor=1.1):
    """Bicubic zoom + centre crop to the original size (reference imgprocess.py:40-52)."""
    return to_host(nv.rescale_fg(to_dev(img), scale_factor))


def shift_fg(img, dx=0, dy=0):
    """Translate with cv2.warpAffine; the matrix is float32 like the reference's (imgprocess.py:55-64)."""
    M = np.float32([[1, 0, dx], [0, 1, dy]]).astype(np.float64)
    return to_host(nv.warp_affine(to_dev(img), M, img.shape[0], img.shape[1]))


def remove_blackborder(img, location=None):
    """Crop all-zero border rows / columns (reference imgprocess.py:195-232: every all-zero row or column COUNTS, wherever it is);
    without a location -> (image, (top, left, bottom, right)), with one -> the cropped image.  Index bookkeeping only."""
    h, w = img.shape[:2]
    if location is not None:
        t, left, b, r = location
        return img[t:b, left:r]
    zero_rows = int((img.reshape(h, -1).sum(axis=1) == 0).sum())
    zero_cols = int((np.moveaxis(img, 1, 0).reshape(w, -1).sum(axis=1) == 0).sum())
    t, left, b, r = zero_rows, zero_cols, h - zero_rows, w - zero_cols
    if left >= r or t >= b:
        return img, (0, 0, h, w)
    return img[t:b, left:r], (t, left, b, r)


def add_blackborder(img, ori_size, location):
    """Inverse of remove_blackborder (reference imgprocess.py:235-260, including its `t == ori_w` comparison)."""
    t, left, b, r = location
    ori_h, ori_w = ori_size
    if t == 0 and left == 0 and b == ori_h and t == ori_w:
        return img
    out = np.zeros((ori_h, ori_w, img.shape[-1]) if img.ndim == 3 else (ori_h, ori_w), np.uint8)
    out[t:b, left:r] = img
    return out


def color_correct(img, alpha, bg_color, target_long_side=960, mean_exp=0.95):
    """Scale alpha by the (min-max normalised, repeatedly square-rooted) Lab a/b distance to the screen colour
    (reference imgprocess.py:263-300): three kernels, the data-dependent sqrt loop resolved on the device."""
    return to_host(nv.color_correct(to_dev(img), to_dev(alpha), np.asarray(bg_color), target_long_side, mean_exp))


def regionfill(I, mask, factor=1.0):
    """Membrane (Laplace) inpainting of one channel inside mask != 0 (reference utils/region_fill.py:7-18, factor = 1 as every
    caller passes).  Returns float64 like the reference; the values are already truncated to whole grey levels, which is what
    the callers do next (tools/unscreen/bg.py:81)."""
    if factor != 1.0:
        raise NotImplementedError("regionfill: only factor = 1.0 (the reference's only use)")
    m = (np.asarray(mask) != 0).astype(np.uint8)
    if not m.any():
        return np.asarray(I).copy()
    img3 = np.repeat(np.asarray(I, np.uint8)[..., None], 3, axis=2)
    return to_host(nv.regionfill(to_dev(img3), to_dev(m)))[..., 0].astype(np.float64)
