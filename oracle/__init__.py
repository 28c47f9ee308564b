"""ctypes binding of the CPU oracle (oracle/usc_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg
of bench.py as the checker.  Nothing under video_unscreen_amd/ may import this package.
Parity status of the arithmetic behind it: see the header of usc_oracle.c ("parity unpinned" at
the OpenCV boundary; the reference has no tests or fixtures).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libusc_oracle.so")
_lib = None

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def build(force=False):
    """Compile the oracle with gcc (seconds).  Building the checker is not using it."""
    src = os.path.join(_HERE, "usc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.orc_count_ge.restype = C.c_longlong
        _lib.orc_count_ge.argtypes = [u8p, C.c_size_t, C.c_int]
        _lib.orc_version.restype = C.c_char_p
        _lib.orc_remove_invalid_objects.restype = C.c_int
        _lib.orc_remove_invalid_objects.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_double, C.c_double,
                                                    C.c_double, C.c_double, f64p, C.c_int]
        _lib.orc_bgdiff_mask.argtypes = [u8p, u8p, u8p, u8p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, u8p, u8p]
        _lib.orc_get_fg.argtypes = [u8p, u8p, u8p, C.c_int, C.c_size_t, u8p, u8p]
        _lib.orc_get_bg.argtypes = [u8p, u8p, C.c_size_t, u8p]
        _lib.orc_blend.argtypes = [u8p, u8p, C.c_int, u8p, C.c_size_t, u8p]
        _lib.orc_score_map.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, f64p]
        _lib.orc_humanseg_post.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_int,
                                           C.c_int, u8p]
        _lib.orc_upsample_ac_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, f32p, C.c_int, C.c_int]
        _lib.orc_inrange_hsv.argtypes = [u8p, u8p, C.c_int, C.POINTER(C.c_int), C.c_size_t, u8p]
        _lib.orc_trimap_withbg.argtypes = [u8p, u8p, u8p, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, u8p]
        _lib.orc_grid_crop_locations.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int)]
        _lib.orc_upsample_nc_f32.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_size_t, f32p, C.c_int, C.c_int]
        _lib.orc_segagent_post.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, u8p, f64p]
        _lib.orc_stm_pre.argtypes = [u8p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p]
        _lib.orc_stm_post.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, f32p]
        _lib.orc_gmm_lut.argtypes = [f64p, f64p, f64p, C.c_int, f32p]
        _lib.orc_gmm_alpha.argtypes = [u8p, C.c_size_t, f32p, f32p, u8p, f32p]
        _lib.orc_cf_threshold.argtypes = [u8p, u8p, C.c_size_t, C.c_double]
        _lib.orc_color_correct.argtypes = [u8p, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_double, u8p, C.POINTER(C.c_int)]
        _lib.orc_box_filter_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p]
        _lib.orc_pad_resize_u8.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        _lib.orc_inv_pad_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        _lib.orc_resize_linear_f32c.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_size_t, f32p, C.c_int, C.c_int]
        _lib.orc_resize_linear_f64c.argtypes = [f64p, C.c_int, C.c_int, C.c_int, C.c_size_t, f64p, C.c_int, C.c_int]
        _lib.orc_imnormalize.argtypes = [u8p, C.c_size_t, f64p, f64p, C.c_int, f32p]
        _lib.orc_moments_u8.argtypes = [u8p, C.c_int, C.c_int, f64p]
        _lib.orc_get_center.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        _lib.orc_get_mask.argtypes = [u8p, C.c_size_t, C.c_int, u8p]
        _lib.orc_iseg_pre.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, f32p, C.c_int, f32p, C.c_int, f32p]
        _lib.orc_iseg_post.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, f32p, u8p]
        for name in ("orc_bgr2gray_u8", "orc_bgr2hsv_u8", "orc_hsv2bgr_u8", "orc_bgr2lab_u8", "orc_lab2bgr_u8"):
            getattr(_lib, name).argtypes = [u8p, u8p, C.c_size_t]
    return _lib


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(u8p)


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def get_target_size(h, w, long_side, division=1):
    th, tw = C.c_int(), C.c_int()
    lib().orc_get_target_size(int(h), int(w), int(long_side), int(division), C.byref(th), C.byref(tw))
    return th.value, tw.value


def pad_resize_geometry(h, w, target_h, target_w):
    a, b = C.c_int(), C.c_int()
    lib().orc_pad_resize_geometry(int(h), int(w), int(target_h), int(target_w), C.byref(a), C.byref(b))
    return a.value, b.value


def resize_nearest(img, dh, dw):
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_resize_nearest_u8(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(u8p), dh, dw)
    return out


def resize_linear(img, dh, dw):
    """cv2.resize(img, (dw, dh)) with the default INTER_LINEAR: uint8 (fixed point), float32 / float64 (float weights), 1..n
    interleaved channels."""
    cn = 1 if img.ndim == 2 else img.shape[2]
    shape = (dh, dw) if img.ndim == 2 else (dh, dw, cn)
    if img.dtype == np.uint8:
        img, p = _u8(img)
        out = np.empty(shape, np.uint8)
        lib().orc_resize_linear_u8(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(u8p), dh, dw)
        return out
    if img.dtype == np.float64:
        img = np.ascontiguousarray(img)
        out = np.empty(shape, np.float64)
        lib().orc_resize_linear_f64c(img.ctypes.data_as(f64p), img.shape[0], img.shape[1], cn, img.shape[1] * cn, out.ctypes.data_as(f64p), dh, dw)
        return out
    img, p = _f32(img)
    out = np.empty(shape, np.float32)
    lib().orc_resize_linear_f32c(p, img.shape[0], img.shape[1], cn, img.shape[1] * cn, out.ctypes.data_as(f32p), dh, dw)
    return out


def pad_resize(img, target_size):
    """imgprocess.py:67-97 on uint8 -> (image, ratio)."""
    img, p = _u8(img)
    th, tw = int(target_size[0]), int(target_size[1])
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((th, tw) if img.ndim == 2 else (th, tw, cn), np.uint8)
    lib().orc_pad_resize_u8(p, h, w, cn, out.ctypes.data_as(u8p), th, tw)
    ratio = float(th) / h if float(h) / w > float(th) / tw else float(tw) / w
    return out, ratio


def inv_pad_resize(img, ori_size):
    """imgprocess.py:100-132 on uint8 / float32 / float64."""
    img = np.ascontiguousarray(img)
    elem = {np.dtype(np.uint8): 0, np.dtype(np.float32): 1, np.dtype(np.float64): 2}[img.dtype]
    oh, ow = int(ori_size[0]), int(ori_size[1])
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((oh, ow) if img.ndim == 2 else (oh, ow, cn), img.dtype)
    lib().orc_inv_pad_resize(img.ctypes.data, img.shape[0], img.shape[1], cn, elem, out.ctypes.data, oh, ow)
    return out


def imnormalize(img, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), to_rgb=True):
    """imgprocess.py:135-161: converts `img` BGR -> RGB in place when to_rgb, returns the float32 result."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous and img.shape[-1] == 3
    mean, std = np.ascontiguousarray(mean, np.float64), np.ascontiguousarray(std, np.float64)
    out = np.empty(img.shape, np.float32)
    lib().orc_imnormalize(img.ctypes.data_as(u8p), img.size // 3, mean.ctypes.data_as(f64p), std.ctypes.data_as(f64p), int(bool(to_rgb)),
                          out.ctypes.data_as(f32p))
    return out


def moments(gray):
    """cv2.moments of a uint8 image: {'m00', 'm10', 'm01'}."""
    g, p = _u8(gray)
    m = np.empty(3, np.float64)
    lib().orc_moments_u8(p, g.shape[0], g.shape[1], m.ctypes.data_as(f64p))
    return {"m00": float(m[0]), "m10": float(m[1]), "m01": float(m[2])}


def get_center(img):
    """imgprocess.py:7-14."""
    img, p = _u8(img)
    cx, cy = C.c_int(), C.c_int()
    lib().orc_get_center(p, img.shape[0], img.shape[1], C.byref(cx), C.byref(cy))
    return cx.value, cy.value


def get_mask(img):
    """imgprocess.py:17-30 -> (thresh (h, w, 1) uint8, binary (h, w, 1) float64)."""
    img, p = _u8(img)
    out = np.empty(img.shape[:2], np.uint8)
    lib().orc_get_mask(p, out.size, 25, out.ctypes.data_as(u8p))
    return out[..., None], (out / 255)[..., None]


def resize_cubic(img, dh, dw):
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_resize_cubic_u8(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(u8p), dh, dw)
    return out


def resize_cubic_fxfy(img, fx, fy):
    """cv2.resize(img, None, fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC) on uint8: dsize = round(size * f), coordinates mapped with 1 / f."""
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    dh, dw = int(round(img.shape[0] * fy)), int(round(img.shape[1] * fx))
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_resize_cubic_fxfy_u8(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(u8p), dh, dw, C.c_double(fx), C.c_double(fy))
    return out


def rescale_fg(img, scale_factor=1.1):
    """imgprocess.py:40-52."""
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    lib().orc_rescale_fg(p, img.shape[0], img.shape[1], cn, C.c_double(scale_factor), out.ctypes.data_as(u8p))
    return out


def shift_fg(img, dx=0, dy=0):
    """imgprocess.py:55-64: warpAffine with the float32 matrix [[1, 0, dx], [0, 1, dy]]."""
    M = np.float32([[1, 0, dx], [0, 1, dy]]).astype(np.float64)
    return warp_affine(img, M, img.shape[0], img.shape[1])


def composite_fgbg(fg, alpha, bg):
    """fgfuncs.py:172-214 with extend=False."""
    fg, pf = _u8(fg)
    alpha, pa = _u8(alpha)
    bg, pb = _u8(bg)
    out = np.empty_like(fg)
    lib().orc_composite_fgbg(pf, pa, fg.shape[0], fg.shape[1], pb, bg.shape[0], bg.shape[1], out.ctypes.data_as(u8p))
    return out


def pad_reflect(img, dh, dw):
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().orc_pad_reflect_u8(p, img.shape[0], img.shape[1], cn, out.ctypes.data_as(u8p), dh, dw)
    return out


def morph(mask, ksize, iters, op):
    """op: 'dilate' or 'erode' (maskprocess.py:7-34)."""
    mask, p = _u8(mask)
    out = np.empty_like(mask)
    lib().orc_morph_u8(p, out.ctypes.data_as(u8p), mask.shape[0], mask.shape[1], int(ksize), int(iters),
                       0 if op == "dilate" else 1)
    return out


def bgr2gray(img):
    img, p = _u8(img)
    out = np.empty(img.shape[:2], np.uint8)
    lib().orc_bgr2gray_u8(p, out.ctypes.data_as(u8p), out.size)
    return out


def bgr2hsv(img):
    img, p = _u8(img)
    out = np.empty_like(img)
    lib().orc_bgr2hsv_u8(p, out.ctypes.data_as(u8p), img.size // 3)
    return out


def hsv2bgr(img):
    img, p = _u8(img)
    out = np.empty_like(img)
    lib().orc_hsv2bgr_u8(p, out.ctypes.data_as(u8p), img.size // 3)
    return out


def bgr2lab(img):
    """cv2.cvtColor(img, cv2.COLOR_BGR2Lab) on uint8."""
    img, p = _u8(img)
    out = np.empty_like(img)
    lib().orc_bgr2lab_u8(p, out.ctypes.data_as(u8p), img.size // 3)
    return out


def lab2bgr(img):
    """cv2.cvtColor(img, cv2.COLOR_LAB2BGR) on uint8 (harmonization/agent.py:73)."""
    img, p = _u8(img)
    out = np.empty_like(img)
    lib().orc_lab2bgr_u8(p, out.ctypes.data_as(u8p), img.size // 3)
    return out


def gmm_lut(means, stds, weights):
    """colorfiltering/agent.py:201-230 evaluated at the 256 uint8 sample values -> float32 (256,)."""
    m = np.ascontiguousarray(np.atleast_1d(means), np.float64)
    sd = np.ascontiguousarray(np.atleast_1d(stds), np.float64)
    w = np.ascontiguousarray(np.atleast_1d(weights), np.float64)
    out = np.empty(256, np.float32)
    lib().orc_gmm_lut(m.ctypes.data_as(f64p), sd.ctypes.data_as(f64p), w.ctypes.data_as(f64p), len(m), out.ctypes.data_as(f32p))
    return out


def gmm_alpha(img_hsv, lut_bg, lut_fg, want_prob=False):
    """colorfiltering/agent.py:232-257; lut_* float32 (3, 256)."""
    img_hsv, p = _u8(img_hsv)
    lb, lbp = _f32(lut_bg)
    lf, lfp = _f32(lut_fg)
    h, w = img_hsv.shape[:2]
    alpha = np.empty((h, w), np.uint8)
    prob = np.empty((h, w), np.float32)
    lib().orc_gmm_alpha(p, h * w, lbp, lfp, alpha.ctypes.data_as(u8p), prob.ctypes.data_as(f32p))
    return (alpha, prob) if want_prob else alpha


def cf_postprocess(alpha, mask, thr_ratio=0.8):
    """colorfiltering/agent.py:259-283: adaptive threshold, then close and open with the 3x3 cross, 2 iterations each."""
    alpha = np.array(alpha, dtype=np.uint8, order="C")
    mask, mp = _u8(mask)
    lib().orc_cf_threshold(alpha.ctypes.data_as(u8p), mp, alpha.size, float(thr_ratio))
    alpha = morph(morph(alpha, 3, 2, "dilate"), 3, 2, "erode")
    alpha = morph(morph(alpha, 3, 2, "erode"), 3, 2, "dilate")
    return alpha


def color_correct(img, alpha, bg_color, target_long_side=960, mean_exp=0.95, want_iters=False):
    """utils/imgprocess.py:263-300."""
    img, ip = _u8(img)
    alpha, ap = _u8(alpha)
    bg, bp = _u8(np.asarray(bg_color).reshape(3))
    out = np.empty_like(alpha)
    it = C.c_int(0)
    lib().orc_color_correct(ip, ap, img.shape[0], img.shape[1], bp, target_long_side, float(mean_exp), out.ctypes.data_as(u8p),
                            C.byref(it))
    return (out, it.value) if want_iters else out


def box_filter(img, ksize=3):
    """cv2.boxFilter(img, -1, (k, k)) on uint8."""
    img, p = _u8(img)
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty_like(img)
    lib().orc_box_filter_u8(p, img.shape[0], img.shape[1], cn, ksize, out.ctypes.data_as(u8p))
    return out


def count_ge(mask, thr=128):
    mask, p = _u8(mask)
    return int(lib().orc_count_ge(p, mask.size, int(thr)))


def exist_foreground(mask, fg_exist_thr):
    """maskprocess.py:56-60."""
    h, w = mask.shape
    return count_ge(mask, 128) > fg_exist_thr * h * w


def trimap(mask, long_side=960, ksize=3, iters=5):
    mask, p = _u8(mask)
    out = np.empty_like(mask)
    lib().orc_trimap(p, mask.shape[0], mask.shape[1], int(long_side), int(ksize), int(iters), out.ctypes.data_as(u8p))
    return out


def bgdiff_mask(frame, bg_i, bg_always, seg, beta, thr):
    frame, pf = _u8(frame)
    bg_i, pi = _u8(bg_i)
    if bg_always is None:                  # tools/unscreen/bg.py:85: a single background image = fusion weight 1 (b*1 + 0*b = b)
        assert float(beta) == 1.0
        bg_always = bg_i
    bg_always, pa = _u8(bg_always)
    seg, ps = _u8(seg)
    H, W = frame.shape[:2]
    out_bg = np.empty_like(frame)
    out_alpha = np.empty((H, W), np.uint8)
    lib().orc_bgdiff_mask(pf, pi, pa, ps, 3 if seg.ndim == 3 else 1, float(beta), int(thr), H, W,
                          out_bg.ctypes.data_as(u8p), out_alpha.ctypes.data_as(u8p))
    return out_bg, out_alpha


def vmat_input_size(H, W, long_side=960, division=32):
    """vmatting/agent.py:38-42 + :93."""
    if long_side % division != 0:
        long_side = (long_side // division + 1) * division
    return get_target_size(H, W, long_side, division)


def vmat_pre(frame, alpha_pre, trimap_, inH, inW, layout="nchw"):
    frame, pf = _u8(frame)
    alpha_pre, pa = _u8(alpha_pre)
    trimap_, pt = _u8(trimap_)
    H, W = frame.shape[:2]
    out = np.empty((7, inH, inW) if layout == "nchw" else (inH, inW, 7), np.float32)
    lib().orc_vmat_pre(pf, pa, pt, H, W, inH, inW, out.ctypes.data_as(f32p), 0 if layout == "nchw" else 1)
    return out


def vmat_post(pred, trimap_, want_float=False):
    pred, pp = _f32(pred)
    trimap_, pt = _u8(trimap_)
    H, W = trimap_.shape
    a8 = np.empty((H, W), np.uint8)
    af = np.empty((H, W), np.float32) if want_float else None
    lib().orc_vmat_post(pp, pred.shape[0], pred.shape[1], pt, H, W, a8.ctypes.data_as(u8p),
                        af.ctypes.data_as(f32p) if want_float else None)
    return (a8, af) if want_float else a8


def get_fg(frame, alpha, bg, paste_thr=0):
    """fgfuncs.py:84-110; paste_thr=1 additionally applies bg[alpha==0]=frame[alpha==0] first
    (bg_offline.py:171), paste_thr=128 the green-mode variant (green.py:125).  Returns (fg, bg_pasted)."""
    frame, pf = _u8(frame)
    alpha, pa = _u8(alpha)
    bg, pb = _u8(bg)
    fg = np.empty_like(frame)
    bgp = np.empty_like(frame)
    lib().orc_get_fg(pf, pa, pb, int(paste_thr), alpha.size, fg.ctypes.data_as(u8p), bgp.ctypes.data_as(u8p))
    return fg, bgp


def get_bg(alpha, bg):
    alpha, pa = _u8(alpha)
    bg, pb = _u8(bg)
    out = np.empty_like(bg)
    lib().orc_get_bg(pa, pb, alpha.size, out.ctypes.data_as(u8p))
    return out


def blend(fg, mask, bg):
    fg, pf = _u8(fg)
    mask, pm = _u8(mask)
    bg, pb = _u8(bg)
    out = np.empty_like(fg)
    lib().orc_blend(pf, pm, 3 if mask.ndim == 3 else 1, pb, fg.size // 3, out.ctypes.data_as(u8p))
    return out


def is_pixel_inrange(img, bg, winsize=(20, 20, 120)):
    """fgfuncs.py:9-65 with long_side_input <= 0.  bg: (3,) colour or (h, w, 3) image.  Returns a bool mask."""
    img, pi = _u8(img)
    bg, pb = _u8(bg)
    win = (C.c_int * 3)(*[int(v) for v in winsize])
    out = np.empty(img.shape[:2], np.uint8)
    lib().orc_inrange_hsv(pi, pb, 1 if bg.ndim == 3 else 0, win, out.size, out.ctypes.data_as(u8p))
    return out > 0


def trimap_withbg(mask, img, bg, winsize=(10, 100, 180), long_side=960, ksize=3, iters=5):
    """trimap/agent.py:63-101."""
    mask, pm = _u8(mask)
    img, pi = _u8(img)
    bg, pb = _u8(bg)
    win = (C.c_int * 3)(*[int(v) for v in winsize])
    out = np.empty_like(mask)
    lib().orc_trimap_withbg(pm, pi, pb, 1 if bg.ndim == 3 else 0, win, mask.shape[0], mask.shape[1], int(long_side),
                            int(ksize), int(iters), out.ctypes.data_as(u8p))
    return out


def humanseg_affine(w, h, in_w=473, in_h=473, inv=False):
    """agent_human.py:91-125 + :247-260 for the whole-frame box; returns the 2x3 float64 matrix."""
    M = np.empty(6, np.float64)
    lib().orc_humanseg_affine(int(w), int(h), int(in_w), int(in_h), int(bool(inv)), M.ctypes.data_as(f64p))
    return M.reshape(2, 3)


def warp_affine(img, M, dh, dw):
    """cv2.warpAffine(img, M, (dw, dh), INTER_LINEAR, BORDER_CONSTANT 0) for uint8 / float32 HWC or HW arrays."""
    M = np.ascontiguousarray(M, np.float64).reshape(6)
    cn = 1 if img.ndim == 2 else img.shape[2]
    if img.dtype == np.uint8:
        img, p = _u8(img)
        out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.uint8)
        lib().orc_warp_affine_u8(p, img.shape[0], img.shape[1], cn, M.ctypes.data_as(f64p), out.ctypes.data_as(u8p), dh, dw)
        return out
    img, p = _f32(img)
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, cn), np.float32)
    lib().orc_warp_affine_f32(p, img.shape[0], img.shape[1], cn, M.ctypes.data_as(f64p), out.ctypes.data_as(f32p), dh, dw)
    return out


def upsample_align_corners(chw, oh, ow):
    """torch Upsample(bilinear, align_corners=True): (C, h, w) float32 -> (oh, ow, C)."""
    chw, p = _f32(chw)
    Cn, ih, iw = chw.shape
    out = np.empty((oh, ow, Cn), np.float32)
    lib().orc_upsample_ac_f32(p, Cn, ih, iw, C.c_size_t(ih * iw), C.c_size_t(1), out.ctypes.data_as(f32p), oh, ow)
    return out


def humanseg_pre(frame, in_h=473, in_w=473):
    """agent_human.py:277-294 -> (3, in_h, in_w) float32 (normalised RGB)."""
    frame, p = _u8(frame)
    out = np.empty((3, in_h, in_w), np.float32)
    lib().orc_humanseg_pre(p, frame.shape[0], frame.shape[1], in_h, in_w, out.ctypes.data_as(f32p))
    return out


def humanseg_post(logits_chw, H, W, in_h=473, in_w=473):
    """agent_human.py:296-311: (C, lh, lw) float32 logits -> (H, W) uint8 mask {0, 255}."""
    logits_chw, p = _f32(logits_chw)
    Cn, lh, lw = logits_chw.shape
    out = np.empty((H, W), np.uint8)
    lib().orc_humanseg_post(p, Cn, lh, lw, C.c_size_t(lh * lw), C.c_size_t(1), in_h, in_w, H, W, out.ctypes.data_as(u8p))
    return out


def segagent_target_size(h, w, long_side, division=1, crop_h=513, crop_w=513):
    """binseg/agent.py:70-100."""
    a, b = C.c_int(), C.c_int()
    lib().orc_segagent_target_size(int(h), int(w), int(long_side), int(division), int(crop_h), int(crop_w), C.byref(a), C.byref(b))
    return a.value, b.value


def grid_crop_locations(h, w, crop_h=513, crop_w=513, stride_ratio=0.5, flip=True):
    """binseg/agent.py:102-141 -> (N, 4) int32 (s_h, e_h, s_w, e_w); mirrored crops have s_w > e_w."""
    n = lib().orc_grid_crop_locations(int(h), int(w), int(crop_h), int(crop_w), float(stride_ratio), int(bool(flip)), None)
    locs = np.empty((n, 4), np.int32)
    lib().orc_grid_crop_locations(int(h), int(w), int(crop_h), int(crop_w), float(stride_ratio), int(bool(flip)),
                                  locs.ctypes.data_as(C.POINTER(C.c_int)))
    return locs


def segagent_pre(img, th, tw, loc):
    """binseg/agent.py:176-198 for one crop -> (crop_h, crop_w, 4) float32 NHWC (R, G, B, 0)."""
    img, p = _u8(img)
    loc = np.ascontiguousarray(np.asarray(loc, np.int32).reshape(4))
    out = np.empty((int(loc[1] - loc[0]), int(abs(loc[3] - loc[2])), 4), np.float32)
    lib().orc_segagent_pre(p, img.shape[0], img.shape[1], int(th), int(tw), loc.ctypes.data_as(C.POINTER(C.c_int)),
                           out.ctypes.data_as(f32p))
    return out


def upsample_nc(hwc, oh, ow):
    """F.interpolate(bilinear, align_corners=False) of an (h, w, C) float32 array -> (oh, ow, C)."""
    a, p = _f32(hwc)
    h, w, c = a.shape
    out = np.empty((oh, ow, c), np.float32)
    lib().orc_upsample_nc_f32(p, c, h, w, 1, c, out.ctypes.data_as(f32p), oh, ow)
    return out


def segagent_post(logits, locs, crop_h, crop_w, th, tw, H, W, want_score=False):
    """binseg/agent.py:200-237; logits (ncrops, lh, lw, cbuf) float32 NHWC with the classes in channels 0, 1."""
    a, p = _f32(logits)
    n, lh, lw, cbuf = a.shape
    locs = np.ascontiguousarray(np.asarray(locs, np.int32).reshape(n, 4))
    out = np.empty((H, W), np.uint8)
    score = np.empty((H, W, 2), np.float64)
    lib().orc_segagent_post(p, n, lh, lw, cbuf, locs.ctypes.data_as(C.POINTER(C.c_int)), int(crop_h), int(crop_w), int(th), int(tw),
                            int(H), int(W), out.ctypes.data_as(u8p), score.ctypes.data_as(f64p))
    return (out, score) if want_score else out


def stm_pre(frame, mask, in_h, in_w):
    """stm/agent.py:62-98 for one frame -> (in_h, in_w, 8) float32 NHWC (R, G, B, object, background, 0, 0, 0); mask may be None."""
    frame, fp = _u8(frame)
    mp = None
    if mask is not None:
        mask, mp = _u8(mask)
    out = np.empty((in_h, in_w, 8), np.float32)
    lib().orc_stm_pre(fp, mp, frame.shape[0], frame.shape[1], int(in_h), int(in_w), out.ctypes.data_as(f32p))
    return out


def stm_post(p2, in_h, in_w, H, W, want_prob=False):
    """Decoder logits (lh, lw, cbuf) -> frame mask (stm/model.py:135-136, 195-219; stm/agent.py:164, 172-201)."""
    a, p = _f32(p2)
    lh, lw, cbuf = a.shape
    out = np.empty((H, W), np.uint8)
    prob = np.empty((in_h, in_w, 2), np.float32)
    lib().orc_stm_post(p, lh, lw, cbuf, int(in_h), int(in_w), int(H), int(W), out.ctypes.data_as(u8p), prob.ctypes.data_as(f32p))
    return (out, prob) if want_prob else out


def regionfill(I, mask):
    """utils/region_fill.py:7-63 (factor = 1): inside mask != 0 solve  n(p) x(p) - sum_{q in N4(p) & mask} x(q) =
    sum_{q in N4(p) \\ mask} I(q)  with scipy's sparse direct solver, like the reference; float64 result, I elsewhere."""
    import scipy.sparse as sp
    import scipy.sparse.linalg as spl
    I = np.asarray(I)
    m = np.asarray(mask) != 0
    out = I.astype(np.float64).copy()
    n = int(m.sum())
    if n == 0:
        return out
    H, W = m.shape
    idx = -np.ones((H, W), np.int64)
    idx[m] = np.arange(n)
    nn = np.full((H, W), 4.0)
    nn[0, :] = 3
    nn[-1, :] = 3
    nn[:, 0] = 3
    nn[:, -1] = 3
    nn[0, 0] = nn[0, -1] = nn[-1, 0] = nn[-1, -1] = 2
    ys, xs = np.nonzero(m)
    rows, cols, vals = [np.arange(n)], [np.arange(n)], [nn[m]]
    b = np.zeros(n)
    If = I.astype(np.float64)
    for dy, dx in ((-1, 0), (0, 1), (1, 0), (0, -1)):
        yy, xx = ys + dy, xs + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        inm = np.zeros(n, bool)
        inm[ok] = m[yy[ok], xx[ok]]
        rows.append(np.nonzero(inm)[0])
        cols.append(idx[yy[inm], xx[inm]])
        vals.append(-np.ones(int(inm.sum())))
        bd = ok & ~inm
        b[bd] += If[yy[bd], xx[bd]]
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    out[m] = spl.spsolve(A, b)
    return out


def score_map(h, w, center):
    out = np.empty((h, w), np.float64)
    lib().orc_score_map(h, w, float(center[0]), float(center[1]), out.ctypes.data_as(f64p))
    return out


def remove_invalid_objects(cfg, alpha, segmask=None, return_stats=False):
    """maskprocess.py:77-152 (+ build_score_map :181-189).  Mutates and returns `alpha` like the reference."""
    assert alpha.dtype == np.uint8 and alpha.flags.c_contiguous
    h, w = alpha.shape
    centers = cfg["objectremoval"]["score_map_center"]
    center = centers["landscape"] if w > h else centers["portrait"]
    seg_p = None
    if segmask is not None:
        segmask, seg_p = _u8(segmask)
    stats = np.zeros((4096, 4), np.float64)
    n = lib().orc_remove_invalid_objects(alpha.ctypes.data_as(u8p), seg_p, h, w, float(center[0]), float(center[1]),
                                         float(cfg["objectremoval"]["saliency_thr"]),
                                         float(cfg["objectremoval"]["consensus_thr"]),
                                         stats.ctypes.data_as(f64p), stats.shape[0])
    if return_stats:
        return alpha, stats[:min(n, stats.shape[0])]
    return alpha


def find_contours(image):
    """cv2.findContours(image, RETR_LIST, CHAIN_APPROX_SIMPLE) of maskprocess.py:123: a list of (n, 2) int32 (x, y) arrays in the
    order the raster scan meets the borders (every border pixel is kept; dropping collinear ones changes neither area nor fill)."""
    image, p = _u8(image)
    h, w = image.shape
    cap = 8 * h * w + 64
    pts = np.empty((cap, 2), np.int32)
    lens = np.empty(h * w + 1, np.int32)
    ip = C.POINTER(C.c_int)
    fn = lib().orc_find_contours
    fn.argtypes = [u8p, C.c_int, C.c_int, ip, C.c_longlong, ip, C.c_int]
    n = fn(p, h, w, pts.ctypes.data_as(ip), cap, lens.ctypes.data_as(ip), lens.size)
    if n < 0:
        raise RuntimeError("orc_find_contours: capacity exceeded")
    out, o = [], 0
    for i in range(n):
        out.append(pts[o:o + lens[i]].copy())
        o += lens[i]
    return out


def contour_area(pts):
    """cv2.contourArea (maskprocess.py:133)."""
    pts = np.ascontiguousarray(pts, np.int32)
    fn = lib().orc_contour_area
    fn.restype = C.c_double
    fn.argtypes = [C.POINTER(C.c_int), C.c_int]
    return fn(pts.ctypes.data_as(C.POINTER(C.c_int)), len(pts))


def draw_contour_filled(img, pts, value=255):
    """cv2.drawContours(img, [contour], 0, value, cv2.FILLED) in place on a one-channel uint8 image (maskprocess.py:137,148)."""
    assert img.dtype == np.uint8 and img.ndim == 2 and img.flags.c_contiguous
    pts = np.ascontiguousarray(pts, np.int32)
    fn = lib().orc_draw_contour_filled
    fn.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, u8p, C.c_int]
    fn(pts.ctypes.data_as(C.POINTER(C.c_int)), len(pts), img.shape[0], img.shape[1], img.ctypes.data_as(u8p), int(value))
    return img


def iseg_pre(frame, in_h, in_w, points_yx, rgb_params, flip=True):
    """ISegAgent.preprocess + DistMaps + rgb_conv -> ((1 + flip) * in_h, in_w, 4) float32 (see orc_iseg_pre)."""
    frame, p = _u8(frame)
    pts, pp = _f32(np.asarray(points_yx, np.float32).reshape(-1, 2))
    prm, pr = _f32(np.asarray(rgb_params, np.float32).reshape(-1))
    assert prm.size == 75
    out = np.empty(((2 if flip else 1) * in_h, in_w, 4), np.float32)
    lib().orc_iseg_pre(p, frame.shape[0], frame.shape[1], int(in_h), int(in_w), pp, pts.shape[0] // 2, pr, int(bool(flip)), out.ctypes.data_as(f32p))
    return out


def iseg_post(logits, in_h, in_w, flip, H, W, prob_thresh=0.5):
    """logits ((1 + flip) * lh, lw, cbuf) float32 -> (prob (in_h, in_w) float32, mask (H, W) uint8) (see orc_iseg_post)."""
    lg, p = _f32(logits)
    nimg = 2 if flip else 1
    prob, mask = np.empty((in_h, in_w), np.float32), np.empty((H, W), np.uint8)
    lib().orc_iseg_post(p, lg.shape[0] // nimg, lg.shape[1], lg.shape[2], int(in_h), int(in_w), int(bool(flip)), int(H), int(W), float(prob_thresh),
                        prob.ctypes.data_as(f32p), mask.ctypes.data_as(u8p))
    return prob, mask


def version():
    return lib().orc_version().decode()


def overall_bg_accumulate(frame, segmask, raw, cont):
    """One frame of tools/unscreen/bg_offline.py:111-117 (stage 2, disabled with `if 0:` in the reference): frames and masks already
    have the destination size; src_mask = dilate3x3x2(mask); raw += float32(frame * float32(1 - mask // 255)); cont += (mask < 250).
    raw / cont: float64 (H, W, 3) accumulators, updated in place."""
    m = morph(np.ascontiguousarray(segmask), 3, 2, "dilate")
    keep = (np.ones_like(m) - m // 255).astype(np.float32)
    raw += (frame * keep[..., None]).astype(np.float32)
    cont += (m < 250).astype(np.float32)[..., None]


def overall_bg_finalize(raw, cont):
    """bg_offline.py:118-127 up to the inpainting: mask_always = cont <= 10; bg_always = uint8(clip(raw / max(cont, 1))) with the
    always-masked pixels zeroed; the hole mask = dilate3x3x2(gray(mask_always)).  Returns (bg_always uint8 (H, W, 3), hole uint8)."""
    mask_always = ((cont <= 10) * 255).astype(np.uint8)
    c = cont.copy()
    c[cont == 0] = 1
    bg_always = np.clip(raw / c, 0, 255).astype(np.uint8)
    bg_always[mask_always == 255] = 0
    hole = morph(bgr2gray(mask_always), 3, 2, "dilate")
    return bg_always, hole
