"""Mask helpers with the reference's names and NumPy calling convention
(reference unscreen/utils/maskprocess.py); the arithmetic runs in libunscreen_hip.so."""
import numpy as np

from .. import _native as nv
from ._dev import to_dev, to_host

_object_filter = nv.DeviceObjectFilter()


def dilate_mask(mask, kernelsize=5, iters=10):
    """cv2.dilate with an ellipse element (reference maskprocess.py:7-19)."""
    return to_host(nv.morph(to_dev(mask), kernelsize, iters, nv.DILATE))


def erode_mask(mask, kernelsize=5, iters=10):
    """cv2.erode with an ellipse element (reference maskprocess.py:22-34)."""
    return to_host(nv.morph(to_dev(mask), kernelsize, iters, nv.ERODE))


def get_fgbox(fgmask, padsize=5):
    """Padded bounding box of mask > 0, in the reference's naming (left / right index ROWS, top / bottom index COLUMNS:
    reference maskprocess.py:37-53)."""
    h, w = fgmask.shape
    x, y = np.where(fgmask > 0)
    left, right, top, bottom = np.min(x), np.max(x), np.min(y), np.max(y)
    left, right = max(left - padsize, 0), min(right + padsize, h)
    top, bottom = max(top - padsize, 0), min(bottom + padsize, w)
    return left, right, top, bottom


def exist_foreground(mask, fg_exist_thr):
    """count(mask >= 128) > thr*h*w (reference maskprocess.py:56-60)."""
    h, w = mask.shape
    return int(nv.count_ge(to_dev(mask), 128).item()) > fg_exist_thr * h * w


def get_outer_boundary(mask, kernelsize=7, iters=10):
    """reference maskprocess.py:63-74 (uint8 wrap-around subtraction, then clip)."""
    dilated = dilate_mask(mask, kernelsize, iters)
    return np.clip(dilated - mask, 0, 255)


def get_score_map(map_size, center):
    """Normalised inverted distance to `center` (ratios), float64 (reference maskprocess.py:155-178)."""
    from ._dev import device
    h, w = map_size
    return to_host(nv.score_map(h, w, center, device()))


def build_score_map(h, w, config):
    """Score map for the object filter: landscape / portrait centre from the config (reference maskprocess.py:181-189)."""
    return get_score_map((h, w), score_map_center(h, w, config))


def score_map_center(h, w, config):
    centers = config['objectremoval']['score_map_center']
    return centers['landscape'] if w > h else centers['portrait']


def remove_invalid_objects(cfg, alpha, segmask=None, saliency_thr=0.001, consensus_thr=0.5,
                           score_map=None, score_map_center=(3. / 5, 1. / 2)):
    """Contour-based object filter (reference maskprocess.py:77-152).  Like the reference, the thresholds come
    from cfg (the keyword values are ignored, :113-114), the centre from build_score_map (:119, :181-189), and
    `alpha` is modified in place and returned.  Runs on the device from connected-component labels (csrc/usc_objects.hip)."""
    h, w = alpha.shape
    centers = cfg['objectremoval']['score_map_center']
    center = centers['landscape'] if w > h else centers['portrait']
    d = to_dev(alpha)
    _object_filter(d, to_dev(segmask) if segmask is not None else None, center, cfg['objectremoval']['saliency_thr'],
                   cfg['objectremoval']['consensus_thr'])
    alpha[...] = to_host(d)
    return alpha
