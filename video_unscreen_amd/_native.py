"""ctypes binding of libunscreen_hip.so (C ABI: include/unscreen_hip.h).

There is NO CPU fallback: if the shared library is missing, or a device function is handed a
non-CUDA tensor, this module raises.  Build the library with ``python -c "import __graft_entry__ as g;
g.build()"`` or ``make -C video_unscreen_amd/csrc``.

Device functions take ``torch`` CUDA tensors only to borrow ``data_ptr()`` and the current HIP stream;
no torch types cross the ABI.
"""
import ctypes as C
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libunscreen_hip.so")

DILATE, ERODE = 0, 1
NCHW, NHWC, NHWC8 = 0, 1, 2

_vp = C.c_void_p
_SIGNATURES = {
    "usc_abi_version": (C.c_int, []),
    "usc_last_error": (C.c_char_p, []),
    "usc_get_target_size": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "usc_pad_resize_geometry": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "usc_inv_pad_geometry": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "usc_pad_resize_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "usc_inv_pad_resize": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "usc_imnormalize_u8": (C.c_int, [_vp, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _vp, _vp]),
    "usc_gray_moments_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_gray_moments_pair_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "usc_iseg_pre": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float), C.c_int, _vp, _vp]),
    "usc_dwconv3x3_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_iseg_post": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp, _vp]),
    "usc_morph_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "usc_morph_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_size_t, _vp]),
    "usc_trimap_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "usc_trimap_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "usc_inrange_hsv_u8": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.c_int, _vp, _vp]),
    "usc_trimap_withbg_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "usc_trimap_withbg_u8": (C.c_int, [_vp, _vp, _vp, _vp, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       _vp, _vp, C.c_size_t, _vp]),
    "usc_bgdiff_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "usc_bgdiff_mask_u8": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                     _vp, _vp, _vp, C.c_size_t, _vp]),
    "usc_vmat_pre": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "usc_vmat_post": (C.c_int, [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_get_fg_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_vmat_post_fg": (C.c_int, [_vp, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "usc_get_bg_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "usc_blend_u8": (C.c_int, [_vp, _vp, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "usc_count_ge_u8": (C.c_int, [_vp, C.c_size_t, C.c_int, _vp, _vp]),
    "usc_resize_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "usc_rescale_fg_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_double, _vp, _vp]),
    "usc_replace_compose_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.c_double, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_warp_affine_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), _vp, C.c_int, C.c_int, _vp]),
    "usc_composite_fgbg_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp, _vp]),
    "usc_lab_tables_bytes": (C.c_size_t, []),
    "usc_lab_tables_host": (None, [_vp]),
    "usc_lab_inv_tables_bytes": (C.c_size_t, []),
    "usc_lab_inv_tables_host": (None, [_vp]),
    "usc_cvt_color_u8": (C.c_int, [_vp, C.c_size_t, C.c_int, _vp, _vp, _vp]),
    "usc_gmm_lut_host": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, _vp]),
    "usc_gmm_alpha_u8": (C.c_int, [_vp, C.c_size_t, _vp, _vp, _vp, _vp]),
    "usc_cf_threshold_ws_bytes": (C.c_size_t, []),
    "usc_cf_threshold_u8": (C.c_int, [_vp, _vp, C.c_size_t, C.c_double, _vp, C.c_size_t, _vp]),
    "usc_color_correct_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "usc_color_correct_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_int, C.c_double, _vp, _vp, _vp, _vp, C.c_size_t, _vp]),
    "usc_box_filter_u8": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "usc_segagent_target_size": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "usc_grid_crop_locations": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _vp, C.c_int]),
    "usc_segagent_pre": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_im2col_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int,
                                      C.c_int, C.c_int, _vp]),
    "usc_upsample_nc_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "usc_adaptive_avgpool_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "usc_segagent_post_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "usc_segagent_post": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    _vp, _vp, _vp, C.c_size_t, _vp]),
    "usc_stm_pre": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "usc_pack_kn_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_longlong, C.c_longlong, C.c_int, C.c_int, _vp, _vp]),
    "usc_row_softmax_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "usc_stm_post": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_stm_mask_from_prob": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "usc_regionfill_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "usc_regionfill_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp, C.c_size_t, _vp]),
    "usc_score_map_f64": (C.c_int, [C.c_int, C.c_int, C.c_double, C.c_double, _vp, _vp, _vp]),
    "usc_remove_invalid_objects_ws_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "usc_remove_invalid_objects_u8": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, _vp, C.c_size_t, _vp]),
    "usc_upsample_taps_accum_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "usc_conv_force_plan": (None, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "usc_conv_force_winograd": (None, [C.c_int]),
    "usc_conv_force_winograd_form": (None, [C.c_int, C.c_int, C.c_int]),
    "usc_bg_accumulate_f32": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_bg_finalize_u8": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "usc_remove_invalid_objects_host": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                  C.c_double, _vp, C.POINTER(C.c_int)]),
    "usc_humanseg_affine": (None, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "usc_humanseg_pre": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp]),
    "usc_humanseg_post_ws_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "usc_humanseg_post": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp,
                                    C.c_size_t, _vp]),
    "usc_upsample_ac_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "usc_upsample_ac_nhwc_batch_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "usc_adaptive_avgpool_nhwc_batch_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, C.c_int, _vp]),
    "usc_maxpool3x3s2_nhwc_f32": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp]),
    "usc_seg_last_error": (C.c_char_p, []),
    "usc_conv_desc_bytes": (C.c_size_t, []),
    "usc_conv_plan_query": (C.c_int, [_vp, _vp]),
    "usc_conv_sequence_f32": (C.c_int, [_vp, C.c_int, _vp]),
    "usc_conv_last_error": (C.c_char_p, []),
    "usc_conv_graph_create": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_void_p)]),
    "usc_conv_graph_launch": (C.c_int, [_vp, _vp]),
    "usc_conv_graph_run_info": (C.c_int, [_vp, _vp, _vp, _vp]),
    "usc_conv_graph_set_mode": (C.c_int, [_vp, C.c_int]),
    "usc_conv_graph_destroy": (None, [_vp]),
}
EXPORTS = tuple(_SIGNATURES)

_lib = None


def load():
    """Load the HIP library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(run `make -C video_unscreen_amd/csrc`). There is no CPU fallback.")
        from . import respect_cpu_share
        respect_cpu_share()                      # torch's CPU pool <= the CPUs this process may use (network set-up runs on it)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class NativeError(RuntimeError):
    pass


class ConvPlanInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("nt", "rowgroups", "ksplit", "kgroups", "workgroups", "threads", "launches", "winograd")]


def conv_plans(desc_array, n):
    """The library's launch plan of each of the n layers of a ctypes array of usc_conv_desc (usc_conv_plan_query)."""
    out = []
    size = load().usc_conv_desc_bytes()
    base = C.addressof(desc_array)
    for i in range(n):
        info = ConvPlanInfo()
        rc = load().usc_conv_plan_query(base + i * size, C.byref(info))
        if rc != 0:
            raise NativeError(f"usc_conv_plan_query failed ({rc}): {load().usc_conv_last_error().decode()}")
        out.append({"plan": (f"winograd{('', '', '-lin', '-t3')[info.winograd]},KW={info.kgroups}{',S=2' if info.ksplit == 2 else ''}") if info.winograd else f"{info.nt},{info.rowgroups},{info.ksplit},{info.kgroups}",
                    "winograd": bool(info.winograd), "workgroups": info.workgroups,
                    "threads": info.threads, "launches": info.launches})
    return out


def _check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed ({rc}): {load().usc_last_error().decode()}")


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev_u8(t, name, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise NativeError(f"{name}: expected a CUDA tensor (the HIP path has no CPU fallback)")
    if t.dtype != torch.uint8 or not t.is_contiguous():
        raise NativeError(f"{name}: expected a contiguous uint8 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise NativeError(f"{name}: shape {tuple(t.shape)} != {tuple(shape)}")
    return t.data_ptr()


# ---------------------------------------------------------------- geometry (host)
def get_target_size(h, w, long_side, division=1):
    a, b = C.c_int(), C.c_int()
    load().usc_get_target_size(int(h), int(w), int(long_side), int(division), C.byref(a), C.byref(b))
    return a.value, b.value


def pad_resize_geometry(h, w, th, tw):
    a, b = C.c_int(), C.c_int()
    load().usc_pad_resize_geometry(int(h), int(w), int(th), int(tw), C.byref(a), C.byref(b))
    return a.value, b.value


def inv_pad_geometry(h, w, ori_h, ori_w):
    a, b = C.c_int(), C.c_int()
    load().usc_inv_pad_geometry(int(h), int(w), int(ori_h), int(ori_w), C.byref(a), C.byref(b))
    return a.value, b.value


# ---------------------------------------------------------------- device stages
def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def morph(src, ksize, iters, op, out=None):
    H, W = src.shape
    ps = _dev_u8(src, "morph.src")
    out = torch.empty_like(src) if out is None else out
    po = _dev_u8(out, "morph.out", (H, W))
    _check(load().usc_morph_u8(ps, po, H, W, int(ksize), int(iters), int(op), None, 0, _stream()), "usc_morph_u8")
    return out


def trimap(mask, long_side=960, ksize=3, iters=5, out=None, ws=None):
    H, W = mask.shape
    pm = _dev_u8(mask, "trimap.mask")
    out = torch.empty_like(mask) if out is None else out
    po = _dev_u8(out, "trimap.out", (H, W))
    need = load().usc_trimap_ws_bytes(H, W, int(long_side))
    if ws is None or ws.numel() < need:
        ws = _ws(need, mask.device)
    _check(load().usc_trimap_u8(pm, H, W, int(long_side), int(ksize), int(iters), po, ws.data_ptr(), ws.numel(), _stream()),
           "usc_trimap_u8")
    return out


def _bg_args(bg, H, W):
    """bg: CUDA (H, W, 3) uint8 tensor, or a 3-element colour (NumPy / list / CPU tensor) passed by host pointer."""
    if isinstance(bg, torch.Tensor) and bg.is_cuda:
        return _dev_u8(bg, "bg", (H, W, 3)), None, None
    col = np.ascontiguousarray(np.asarray(bg.cpu() if isinstance(bg, torch.Tensor) else bg, dtype=np.uint8).reshape(3))
    return None, col.ctypes.data, col


def is_pixel_inrange(img, bg, winsize=(20, 20, 120), out=None):
    """Returns a (H, W) uint8 CUDA tensor: 1 where the pixel is within the HSV window of the background."""
    H, W, _ = img.shape
    pi = _dev_u8(img, "inrange.img", (H, W, 3))
    bg_dev, bg_host, keep = _bg_args(bg, H, W)
    win = (C.c_int * 3)(*[int(v) for v in winsize])
    out = torch.empty((H, W), dtype=torch.uint8, device=img.device) if out is None else out
    _check(load().usc_inrange_hsv_u8(pi, bg_dev, bg_host, win, H, W, _dev_u8(out, "inrange.out", (H, W)), _stream()),
           "usc_inrange_hsv_u8")
    return out


def trimap_withbg(mask, img, bg, winsize=(10, 100, 180), long_side=960, ksize=3, iters=5, out=None, ws=None):
    H, W = mask.shape
    pm = _dev_u8(mask, "trimap_withbg.mask")
    pi = _dev_u8(img, "trimap_withbg.img", (H, W, 3))
    bg_dev, bg_host, keep = _bg_args(bg, H, W)
    win = (C.c_int * 3)(*[int(v) for v in winsize])
    out = torch.empty_like(mask) if out is None else out
    need = load().usc_trimap_withbg_ws_bytes(H, W, int(long_side))
    if ws is None or ws.numel() < need:
        ws = _ws(need, mask.device)
    _check(load().usc_trimap_withbg_u8(pm, pi, bg_dev, bg_host, win, H, W, int(long_side), int(ksize), int(iters),
                                       _dev_u8(out, "trimap_withbg.out", (H, W)), ws.data_ptr(), ws.numel(), _stream()),
           "usc_trimap_withbg_u8")
    return out


def bgdiff_mask(frame, bg_i, bg_always, seg, beta, thr, out_bg=None, out_alpha=None, ws=None):
    H, W, _ = frame.shape
    pf = _dev_u8(frame, "bgdiff.frame", (H, W, 3))
    pi = _dev_u8(bg_i, "bgdiff.bg_i", (H, W, 3))
    pa = _dev_u8(bg_always, "bgdiff.bg_always", (H, W, 3)) if bg_always is not None else None
    seg_cn = 3 if seg.dim() == 3 else 1
    ps = _dev_u8(seg, "bgdiff.seg", (H, W, 3) if seg_cn == 3 else (H, W))
    out_bg = torch.empty_like(frame) if out_bg is None else out_bg
    out_alpha = torch.empty((H, W), dtype=torch.uint8, device=frame.device) if out_alpha is None else out_alpha
    need = load().usc_bgdiff_ws_bytes(H, W)                # 0 since the two kernels became one; `ws` stays accepted
    if need and (ws is None or ws.numel() < need):
        ws = _ws(need, frame.device)
    _check(load().usc_bgdiff_mask_u8(pf, pi, pa, ps, seg_cn, float(beta), int(thr), H, W,
                                     _dev_u8(out_bg, "bgdiff.out_bg", (H, W, 3)), _dev_u8(out_alpha, "bgdiff.out_alpha", (H, W)),
                                     ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream()), "usc_bgdiff_mask_u8")
    return out_bg, out_alpha


def vmat_pre(frame, alpha_pre, trimap_, inH, inW, layout=NHWC, out=None):
    """Returns a (1, 7, inH, inW) float32 tensor (channels-last memory if layout == NHWC)."""
    H, W, _ = frame.shape
    pf = _dev_u8(frame, "vmat_pre.frame", (H, W, 3))
    pa = _dev_u8(alpha_pre, "vmat_pre.alpha_pre", (H, W))
    pt = _dev_u8(trimap_, "vmat_pre.trimap", (H, W))
    if out is None:
        if layout == NHWC8:
            out = torch.empty((inH, inW, 8), dtype=torch.float32, device=frame.device)
        else:
            out = torch.empty((1, 7, inH, inW), dtype=torch.float32, device=frame.device,
                              memory_format=torch.channels_last if layout == NHWC else torch.contiguous_format)
    _check(load().usc_vmat_pre(pf, pa, pt, H, W, int(inH), int(inW), out.data_ptr(), int(layout), _stream()), "usc_vmat_pre")
    return out


def vmat_post(pred, trimap_, out_u8=None, out_f32=None, want_float=False):
    H, W = trimap_.shape
    if not (pred.is_cuda and pred.dtype == torch.float32 and pred.is_contiguous()):
        raise NativeError("vmat_post.pred: expected a contiguous float32 CUDA tensor")
    inH, inW = pred.shape[-2], pred.shape[-1]
    pt = _dev_u8(trimap_, "vmat_post.trimap")
    out_u8 = torch.empty((H, W), dtype=torch.uint8, device=pred.device) if out_u8 is None else out_u8
    if want_float and out_f32 is None:
        out_f32 = torch.empty((H, W), dtype=torch.float32, device=pred.device)
    _check(load().usc_vmat_post(pred.data_ptr(), inH, inW, pt, H, W, _dev_u8(out_u8, "vmat_post.out", (H, W)),
                                out_f32.data_ptr() if out_f32 is not None else None, _stream()), "usc_vmat_post")
    return (out_u8, out_f32) if want_float else out_u8


def get_fg(frame, alpha, bg, paste_thr=0, out_fg=None, out_bg=None, want_bg=False):
    H, W, _ = frame.shape
    pf = _dev_u8(frame, "get_fg.frame", (H, W, 3))
    pa = _dev_u8(alpha, "get_fg.alpha", (H, W))
    pb = _dev_u8(bg, "get_fg.bg", (H, W, 3))
    out_fg = torch.empty_like(frame) if out_fg is None else out_fg
    if want_bg and out_bg is None:
        out_bg = torch.empty_like(frame)
    _check(load().usc_get_fg_u8(pf, pa, pb, int(paste_thr), H, W, _dev_u8(out_fg, "get_fg.out", (H, W, 3)),
                                _dev_u8(out_bg, "get_fg.out_bg", (H, W, 3)) if out_bg is not None else None, _stream()),
           "usc_get_fg_u8")
    return (out_fg, out_bg) if (want_bg or out_bg is not None) else out_fg


def vmat_post_fg(pred, trimap_, frame, bg, paste_thr=0, out_u8=None, out_f32=None, out_fg=None, want_float=False):
    """vmat_post followed by get_fg(paste_thr) as one launch -> (alpha u8, alpha f32 | None, fg)."""
    H, W = trimap_.shape
    if not (pred.is_cuda and pred.dtype == torch.float32 and pred.is_contiguous()):
        raise NativeError("vmat_post_fg.pred: expected a contiguous float32 CUDA tensor")
    inH, inW = pred.shape[-2], pred.shape[-1]
    out_u8 = torch.empty((H, W), dtype=torch.uint8, device=pred.device) if out_u8 is None else out_u8
    out_fg = torch.empty((H, W, 3), dtype=torch.uint8, device=pred.device) if out_fg is None else out_fg
    if want_float and out_f32 is None:
        out_f32 = torch.empty((H, W), dtype=torch.float32, device=pred.device)
    _check(load().usc_vmat_post_fg(pred.data_ptr(), inH, inW, _dev_u8(trimap_, "vmat_post_fg.trimap"),
                                   _dev_u8(frame, "vmat_post_fg.frame", (H, W, 3)), _dev_u8(bg, "vmat_post_fg.bg", (H, W, 3)), int(paste_thr), H, W,
                                   _dev_u8(out_u8, "vmat_post_fg.alpha", (H, W)), out_f32.data_ptr() if out_f32 is not None else None,
                                   _dev_u8(out_fg, "vmat_post_fg.fg", (H, W, 3)), _stream()), "usc_vmat_post_fg")
    return out_u8, out_f32, out_fg


def get_bg(alpha, bg, out=None):
    H, W, _ = bg.shape
    out = torch.empty_like(bg) if out is None else out
    _check(load().usc_get_bg_u8(_dev_u8(alpha, "get_bg.alpha", (H, W)), _dev_u8(bg, "get_bg.bg", (H, W, 3)), H, W,
                                _dev_u8(out, "get_bg.out", (H, W, 3)), _stream()), "usc_get_bg_u8")
    return out


def blend(fg, mask, bg, out=None):
    H, W, _ = fg.shape
    mask_cn = 3 if mask.dim() == 3 else 1
    out = torch.empty_like(fg) if out is None else out
    _check(load().usc_blend_u8(_dev_u8(fg, "blend.fg", (H, W, 3)), _dev_u8(mask, "blend.mask"), mask_cn,
                               _dev_u8(bg, "blend.bg", (H, W, 3)), H, W, _dev_u8(out, "blend.out", (H, W, 3)), _stream()),
           "usc_blend_u8")
    return out


def _img_dims(t):
    return (t.shape[0], t.shape[1], 1) if t.dim() == 2 else (t.shape[0], t.shape[1], t.shape[2])


def resize(img, dh, dw, interp=1, out=None):
    """cv2.resize on a uint8 CUDA image (H, W) or (H, W, 3); interp 0 nearest / 1 linear / 2 cubic."""
    sh, sw, cn = _img_dims(img)
    out = torch.empty((dh, dw) if img.dim() == 2 else (dh, dw, cn), dtype=torch.uint8, device=img.device) if out is None else out
    _check(load().usc_resize_u8(_dev_u8(img, "resize.img"), sh, sw, cn, _dev_u8(out, "resize.out"), int(dh), int(dw), int(interp), _stream()),
           "usc_resize_u8")
    return out


_ELEM = {torch.uint8: 0, torch.float32: 1, torch.float64: 2}


def _dev_any(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.is_contiguous()):
        raise NativeError(f"{name}: expected a contiguous CUDA tensor (the HIP path has no CPU fallback)")
    return t.data_ptr()


def pad_resize(img, target_h, target_w, out=None):
    """imgprocess.py:67-97 on a uint8 CUDA image (H, W) or (H, W, 3): resize + reflect pad to (target_h, target_w), one kernel."""
    h, w, cn = _img_dims(img)
    shape = (target_h, target_w) if img.dim() == 2 else (target_h, target_w, cn)
    out = torch.empty(shape, dtype=torch.uint8, device=img.device) if out is None else out
    _check(load().usc_pad_resize_u8(_dev_u8(img, "pad_resize.img"), h, w, cn, _dev_u8(out, "pad_resize.out", shape), int(target_h), int(target_w),
                                    _stream()), "usc_pad_resize_u8")
    return out


def inv_pad_resize(img, ori_h, ori_w, out=None):
    """imgprocess.py:100-132 on a uint8 / float32 / float64 CUDA image (H, W) or (H, W, C <= 4)."""
    if img.dtype not in _ELEM:
        raise NativeError(f"inv_pad_resize: dtype {img.dtype} (uint8, float32 or float64)")
    h, w = img.shape[:2]
    cn = 1 if img.dim() == 2 else img.shape[2]
    shape = (ori_h, ori_w) if img.dim() == 2 else (ori_h, ori_w, cn)
    out = torch.empty(shape, dtype=img.dtype, device=img.device) if out is None else out
    if out.dtype != img.dtype or tuple(out.shape) != tuple(shape):
        raise NativeError("inv_pad_resize: out has the wrong dtype / shape")
    _check(load().usc_inv_pad_resize(_dev_any(img, "inv_pad_resize.img"), h, w, cn, _ELEM[img.dtype], _dev_any(out, "inv_pad_resize.out"),
                                     int(ori_h), int(ori_w), _stream()), "usc_inv_pad_resize")
    return out


def imnormalize(img, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), to_rgb=True, out=None):
    """imgprocess.py:135-161 on a (H, W, 3) uint8 CUDA image: img becomes RGB IN PLACE when to_rgb; returns float32 (H, W, 3)."""
    H, W, _ = img.shape
    out = torch.empty((H, W, 3), dtype=torch.float32, device=img.device) if out is None else out
    m = (C.c_double * 3)(*[float(v) for v in np.asarray(mean).reshape(-1)])
    sd = (C.c_double * 3)(*[float(v) for v in np.asarray(std).reshape(-1)])
    _check(load().usc_imnormalize_u8(_dev_u8(img, "imnormalize.img", (H, W, 3)), H * W, m, sd, int(bool(to_rgb)), _dev_any(out, "imnormalize.out"),
                                     _stream()), "usc_imnormalize_u8")
    return out


def gray_moments(img, rgb_order=False, thr=None, want_moments=True, out=None):
    """Gray of a (H, W, 3) uint8 CUDA image (rgb_order: cv2.COLOR_RGB2GRAY): -> (mask | None, moments | None); mask = 255 where
    gray > thr (thr given), moments = int64 CUDA tensor (m00, m10, m01)."""
    H, W, _ = img.shape
    mask = torch.empty((H, W), dtype=torch.uint8, device=img.device) if thr is not None else None
    mom = (out if out is not None else torch.empty(3, dtype=torch.int64, device=img.device)) if want_moments else None      # out: 3 int64, e.g. a row of a table
    _check(load().usc_gray_moments_u8(_dev_u8(img, "gray_moments.img", (H, W, 3)), H, W, int(bool(rgb_order)), int(thr if thr is not None else 0),
                                      mask.data_ptr() if mask is not None else None, mom.data_ptr() if mom is not None else None, _stream()),
           "usc_gray_moments_u8")
    return mask, mom


def gray_moments_pair(img_a, img_b, out, rgb_order=True):
    """Moments (m00, m10, m01) of the gray images of two (H, W, 3) uint8 CUDA images in one launch -> out: 6 int64 (a's, then b's)."""
    H, W, _ = img_a.shape
    if not (out.dtype == torch.int64 and out.is_cuda and out.is_contiguous() and out.numel() == 6):
        raise NativeError("gray_moments_pair.out: expected 6 contiguous int64 on the device")
    _check(load().usc_gray_moments_pair_u8(_dev_u8(img_a, "gray_moments_pair.a", (H, W, 3)), _dev_u8(img_b, "gray_moments_pair.b", (H, W, 3)), H, W,
                                           int(bool(rgb_order)), out.data_ptr(), _stream()), "usc_gray_moments_pair_u8")
    return out


def score_map(H, W, center, device):
    """maskprocess.py:155-178 get_score_map as a float64 (H, W) CUDA table."""
    out = torch.empty((H, W), dtype=torch.float64, device=device)
    ws8 = torch.zeros(8, dtype=torch.uint8, device=device)
    _check(load().usc_score_map_f64(int(H), int(W), float(center[0]), float(center[1]), out.data_ptr(), ws8.data_ptr(), _stream()),
           "usc_score_map_f64")
    return out


ISEG_MAX_CLICKS = 20


def iseg_pre(frame, in_h, in_w, points_yx, rgb_params, flip=True, out=None):
    """ISegAgent.preprocess + DistMaps + rgb_conv: frame (H, W, 3) uint8 CUDA -> ((1 + flip) * in_h, in_w, 4) float32 NHWC.
    points_yx: (2 P, 2) float array in network-input coordinates (positive slots, then negative; (-1, -1) = empty);
    rgb_params: 75 floats (w1 8x5, b1 8, w2 3x8, b2 3; BatchNorm folded into the second layer)."""
    H, W, _ = frame.shape
    pts = np.ascontiguousarray(points_yx, np.float32).reshape(-1, 2)
    P = pts.shape[0] // 2
    prm = np.ascontiguousarray(rgb_params, np.float32).reshape(-1)
    if prm.size != 75 or pts.shape[0] != 2 * P:
        raise NativeError("iseg_pre: rgb_params must hold 75 floats and points an even number of (y, x) rows")
    nimg = 2 if flip else 1
    out = torch.empty((nimg * in_h, in_w, 4), dtype=torch.float32, device=frame.device) if out is None else out
    _check(load().usc_iseg_pre(_dev_u8(frame, "iseg_pre.frame", (H, W, 3)), H, W, int(in_h), int(in_w), pts.ctypes.data_as(C.POINTER(C.c_float)), P,
                               prm.ctypes.data_as(C.POINTER(C.c_float)), int(bool(flip)), _dev_any(out, "iseg_pre.out"), _stream()), "usc_iseg_pre")
    return out


def dwconv3x3(src, w9c, dst, batch=1):
    """Depthwise 3x3 (padding 1): src (batch * h, w, C) float32 NHWC CUDA, w9c (9, C) -> dst like src."""
    bh, w, Cn = src.shape
    _check(load().usc_dwconv3x3_nhwc_f32(_dev_any(src, "dwconv3x3.src"), bh // batch, w, Cn, int(batch), _dev_any(w9c, "dwconv3x3.w"),
                                         _dev_any(dst, "dwconv3x3.dst"), _stream()), "usc_dwconv3x3_nhwc_f32")
    return dst


def iseg_post(logits, in_h, in_w, flip, H=None, W=None, prob_thresh=0.5, prob=None, mask=None, want_mask=True):
    """logits ((1 + flip) * lh, lw, cbuf) float32 CUDA (channel 0) -> (prob (in_h, in_w) float32, mask (H, W) uint8 | None)."""
    nimg = 2 if flip else 1
    bh, lw, cbuf = logits.shape
    prob = torch.empty((in_h, in_w), dtype=torch.float32, device=logits.device) if prob is None else prob
    if want_mask and mask is None:
        mask = torch.empty((H, W), dtype=torch.uint8, device=logits.device)
    _check(load().usc_iseg_post(_dev_any(logits, "iseg_post.logits"), bh // nimg, lw, cbuf, int(in_h), int(in_w), int(bool(flip)), int(H or 0), int(W or 0),
                                float(prob_thresh), _dev_any(prob, "iseg_post.prob"), _dev_u8(mask, "iseg_post.mask", (H, W)) if want_mask else None,
                                _stream()), "usc_iseg_post")
    return prob, (mask if want_mask else None)


def rescale_fg(img, scale_factor=1.1, out=None):
    h, w, cn = _img_dims(img)
    out = torch.empty_like(img) if out is None else out
    _check(load().usc_rescale_fg_u8(_dev_u8(img, "rescale_fg.img"), h, w, cn, float(scale_factor), _dev_u8(out, "rescale_fg.out"), _stream()),
           "usc_rescale_fg_u8")
    return out


def warp_affine(img, M, dh, dw, out=None):
    sh, sw, cn = _img_dims(img)
    Mc = (C.c_double * 6)(*[float(v) for v in np.asarray(M, np.float64).reshape(6)])
    out = torch.empty((dh, dw) if img.dim() == 2 else (dh, dw, cn), dtype=torch.uint8, device=img.device) if out is None else out
    rc = load().usc_warp_affine_u8(_dev_u8(img, "warp.img"), sh, sw, cn, Mc, _dev_u8(out, "warp.out"), int(dh), int(dw), _stream())
    if rc != 0:
        raise NativeError(f"usc_warp_affine_u8 failed ({rc}): {load().usc_seg_last_error().decode()}")
    return out


def replace_compose(fg, mask, M, bg, src=None, scale=1.2, res=None, compare=None, want_res=True):
    """tools/replace/replace.py:59-82 per frame, one launch: (res, compare) with res = u8(fg' m' + bg (1 - m')), fg' / m' =
    rescale_fg(shift_fg(., M), scale), compare = [resize(src) | res] (None without src).  Bit-identical to the stand-alone chain."""
    H, W, _ = fg.shape
    mask_cn = 3 if mask.dim() == 3 else 1
    if res is None and want_res:
        res = torch.empty_like(fg)
    if compare is None and src is not None:
        compare = torch.empty((H, 2 * W, 3), dtype=torch.uint8, device=fg.device)
    Mc = (C.c_double * 6)(*[float(v) for v in np.asarray(M, np.float64).reshape(6)])
    sh, sw = (src.shape[0], src.shape[1]) if src is not None else (0, 0)
    _check(load().usc_replace_compose_u8(_dev_u8(fg, "replace.fg", (H, W, 3)), _dev_u8(mask, "replace.mask", (H, W, 3) if mask_cn == 3 else (H, W)), mask_cn, H, W,
                                         Mc, float(scale), _dev_u8(bg, "replace.bg", (H, W, 3)),
                                         _dev_u8(src, "replace.src", (sh, sw, 3)) if src is not None else None, sh, sw,
                                         _dev_u8(res, "replace.res", (H, W, 3)) if res is not None else None,
                                         _dev_u8(compare, "replace.compare", (H, 2 * W, 3)) if compare is not None else None, _stream()),
           "usc_replace_compose_u8")
    return res, compare


def composite_fgbg(fg, alpha, bg, out=None):
    fh, fw, _ = fg.shape
    bh, bw, _ = bg.shape
    out = torch.empty_like(fg) if out is None else out
    _check(load().usc_composite_fgbg_u8(_dev_u8(fg, "composite.fg", (fh, fw, 3)), _dev_u8(alpha, "composite.alpha", (fh, fw)), fh, fw,
                                        _dev_u8(bg, "composite.bg", (bh, bw, 3)), bh, bw, _dev_u8(out, "composite.out", (fh, fw, 3)), _stream()),
           "usc_composite_fgbg_u8")
    return out


# ---------------------------------------------------------------- green mode
BGR2HSV, HSV2BGR, BGR2GRAY, BGR2LAB, LAB2BGR = 0, 1, 2, 3, 4
_lab_tables = {}
_lab_inv_tables = {}


def lab_tables(device):
    """Device copy of the 8-bit BGR->Lab lookup tables (built by the library on the host, uploaded once per device)."""
    key = str(device)
    if key not in _lab_tables:
        blob = np.empty(load().usc_lab_tables_bytes(), np.uint8)
        load().usc_lab_tables_host(blob.ctypes.data)
        _lab_tables[key] = torch.from_numpy(blob).to(device)
    return _lab_tables[key]


def lab_inv_tables(device):
    """Device copy of the 8-bit Lab->BGR tables (L -> (y, f(y)), inverse sRGB gamma, XYZ->BGR coefficients)."""
    key = str(device)
    if key not in _lab_inv_tables:
        blob = np.empty(load().usc_lab_inv_tables_bytes(), np.uint8)
        load().usc_lab_inv_tables_host(blob.ctypes.data)
        _lab_inv_tables[key] = torch.from_numpy(blob).to(device)
    return _lab_inv_tables[key]


def cvt_color(img, code, out=None):
    """cv2.cvtColor on a (H, W, 3) uint8 CUDA tensor; code BGR2HSV / HSV2BGR / BGR2GRAY / BGR2LAB / LAB2BGR."""
    H, W, _ = img.shape
    pi = _dev_u8(img, "cvt_color.img", (H, W, 3))
    shape = (H, W) if code == BGR2GRAY else (H, W, 3)
    out = torch.empty(shape, dtype=torch.uint8, device=img.device) if out is None else out
    tab = lab_tables(img.device).data_ptr() if code == BGR2LAB else lab_inv_tables(img.device).data_ptr() if code == LAB2BGR else None
    _check(load().usc_cvt_color_u8(pi, H * W, int(code), _dev_u8(out, "cvt_color.out", shape), tab, _stream()), "usc_cvt_color_u8")
    return out


def gmm_lut(means, stds, weights):
    """Likelihood of one fitted 1-D mixture at the 256 uint8 values (host, float32) -> np.float32 (256,)."""
    m = np.ascontiguousarray(np.atleast_1d(means), np.float64)
    sd = np.ascontiguousarray(np.atleast_1d(stds), np.float64)
    w = np.ascontiguousarray(np.atleast_1d(weights), np.float64)
    if not (m.shape == sd.shape == w.shape and m.ndim == 1):
        raise NativeError("gmm_lut: means / stds / weights must be 1-D and of equal length")
    out = np.empty(256, np.float32)
    dp = C.POINTER(C.c_double)
    _check(load().usc_gmm_lut_host(m.ctypes.data_as(dp), sd.ctypes.data_as(dp), w.ctypes.data_as(dp), len(m), out.ctypes.data),
           "usc_gmm_lut_host")
    return out


def gmm_alpha(img_hsv, luts, out=None, want_prob=False):
    """img_hsv (h, w, 3) uint8 CUDA; luts (6, 256) float32 CUDA (bg h, s, v then fg h, s, v) -> alpha (h, w) uint8."""
    h, w, _ = img_hsv.shape
    pi = _dev_u8(img_hsv, "gmm_alpha.img_hsv", (h, w, 3))
    if not (luts.is_cuda and luts.dtype == torch.float32 and luts.is_contiguous() and luts.numel() == 6 * 256):
        raise NativeError("gmm_alpha.luts: expected a contiguous (6, 256) float32 CUDA tensor")
    out = torch.empty((h, w), dtype=torch.uint8, device=img_hsv.device) if out is None else out
    prob = torch.empty((h, w), dtype=torch.float32, device=img_hsv.device) if want_prob else None
    _check(load().usc_gmm_alpha_u8(pi, h * w, luts.data_ptr(), _dev_u8(out, "gmm_alpha.out", (h, w)),
                                   prob.data_ptr() if want_prob else None, _stream()), "usc_gmm_alpha_u8")
    return (out, prob) if want_prob else out


def cf_threshold(alpha, mask, thr_ratio=0.8, ws=None):
    """In place: alpha[alpha < mean(alpha[(alpha > 128) & (mask > 0)]) * thr_ratio] = 0."""
    h, w = alpha.shape
    need = load().usc_cf_threshold_ws_bytes()
    if ws is None or ws.numel() < need:
        ws = _ws(need, alpha.device)
    _check(load().usc_cf_threshold_u8(_dev_u8(alpha, "cf_threshold.alpha"), _dev_u8(mask, "cf_threshold.mask", (h, w)), h * w,
                                      float(thr_ratio), ws.data_ptr(), ws.numel(), _stream()), "usc_cf_threshold_u8")
    return alpha


def color_correct(img, alpha, bg_color, target_long_side=960, mean_exp=0.95, out=None, ws=None, want_iters=False):
    H, W, _ = img.shape
    pi = _dev_u8(img, "color_correct.img", (H, W, 3))
    pa = _dev_u8(alpha, "color_correct.alpha", (H, W))
    col = np.ascontiguousarray(np.asarray(bg_color.cpu() if isinstance(bg_color, torch.Tensor) else bg_color, dtype=np.uint8).reshape(3))
    out = torch.empty_like(alpha) if out is None else out
    need = load().usc_color_correct_ws_bytes(H, W, int(target_long_side))
    if ws is None or ws.numel() < need:
        ws = _ws(need, img.device)
    it = torch.zeros(1, dtype=torch.int32, device=img.device) if want_iters else None
    _check(load().usc_color_correct_u8(pi, pa, H, W, col.ctypes.data, int(target_long_side), float(mean_exp),
                                       lab_tables(img.device).data_ptr(), _dev_u8(out, "color_correct.out", (H, W)),
                                       it.data_ptr() if want_iters else None, ws.data_ptr(), ws.numel(), _stream()),
           "usc_color_correct_u8")
    return (out, int(it.item())) if want_iters else out


def box_filter(img, ksize=3, out=None):
    H, W, cn = _img_dims(img)
    out = torch.empty_like(img) if out is None else out
    _check(load().usc_box_filter_u8(_dev_u8(img, "box_filter.img"), H, W, cn, int(ksize), _dev_u8(out, "box_filter.out", tuple(img.shape)),
                                    _stream()), "usc_box_filter_u8")
    return out


def count_ge(mask, thr=128, out=None):
    """Asynchronous: returns a 1-element int64 CUDA tensor holding #(mask >= thr)."""
    pm = _dev_u8(mask, "count_ge.mask")
    out = torch.empty(1, dtype=torch.int64, device=mask.device) if out is None else out
    _check(load().usc_count_ge_u8(pm, mask.numel(), int(thr), out.data_ptr(), _stream()), "usc_count_ge_u8")
    return out


def upsample_ac(src, dst, c0=0, batch=1):
    """src (batch*sh, sw, C) -> channels [c0, c0+C) of dst (batch*dh, dw, ld), both contiguous float32 CUDA NHWC buffers."""
    sh, sw, Cn = src.shape
    dh, dw, ld = dst.shape
    rc = load().usc_upsample_ac_nhwc_batch_f32(src.data_ptr(), sh // batch, sw, Cn, dst.data_ptr(), dh // batch, dw, ld, int(c0), int(batch),
                                               _stream())
    if rc != 0:
        raise NativeError(f"usc_upsample_ac_nhwc_f32 failed ({rc}): {load().usc_seg_last_error().decode()}")
    return dst


def upsample_taps_accum(q, dst, batch=1, accumulate=False, align_corners=True):
    """dst (batch*dh, dw, C) (+)= sum over the 9 taps of the bilinear upsampling of q (batch*sh, sw, 9*C) channels
    [tap*C, (tap+1)*C) sampled at the tap's neighbour: conv3x3(upsample(p)) from q = p [W_0 .. W_8]."""
    sh, sw, c9 = q.shape
    dh, dw, Cn = dst.shape
    if c9 != 9 * Cn or sh % batch or dh % batch:
        raise NativeError("upsample_taps_accum: q must be (batch*sh, sw, 9*C) for dst (batch*dh, dw, C)")
    rc = load().usc_upsample_taps_accum_f32(q.data_ptr(), sh // batch, sw, Cn, dst.data_ptr(), dh // batch, dw, int(batch),
                                            1 if accumulate else 0, 1 if align_corners else 0, _stream())
    if rc != 0:
        raise NativeError(f"usc_upsample_taps_accum_f32 failed ({rc}): {load().usc_seg_last_error().decode()}")
    return dst


def upsample_nc(src, dst, c0=0):
    """F.interpolate(bilinear, align_corners=False): src (sh, sw, C) f32 NHWC -> channels [c0, c0 + C) of dst (dh, dw, ld)."""
    sh, sw, Cc = src.shape
    dh, dw, ld = dst.shape
    _check(load().usc_upsample_nc_nhwc_f32(src.data_ptr(), sh, sw, Cc, dst.data_ptr(), dh, dw, ld, int(c0), _stream()),
           "usc_upsample_nc_nhwc_f32")
    return dst


def maxpool3x3s2(src, dst, batch=1):
    """nn.MaxPool2d(3, 2, 1) of src (batch*H, W, C) f32 NHWC -> dst (batch*OH, OW, C), OH = (H - 1) // 2 + 1 (the ResNet stems)."""
    BH, W, Cn = src.shape
    H = BH // batch
    if BH % batch or tuple(dst.shape) != (batch * ((H - 1) // 2 + 1), (W - 1) // 2 + 1, Cn) or not (src.is_contiguous() and dst.is_contiguous()):
        raise NativeError("maxpool3x3s2: dst must be the contiguous (batch*OH, OW, C) tensor of a contiguous (batch*H, W, C) src")
    _check(load().usc_maxpool3x3s2_nhwc_f32(src.data_ptr(), H, W, Cn, dst.data_ptr(), int(batch), _stream()), "usc_maxpool3x3s2_nhwc_f32")
    return dst


def adaptive_avgpool(src, c0, Cn, dst, batch=1):
    """nn.AdaptiveAvgPool2d(s) of channels [c0, c0 + Cn) of src (batch*H, W, ld) f32 NHWC -> dst (batch*s, s, Cn)."""
    BH, W, ld = src.shape
    bs, s, cd = dst.shape
    if bs != batch * s or cd != Cn or BH % batch:
        raise NativeError("adaptive_avgpool: dst must be (batch*s, s, Cn)")
    _check(load().usc_adaptive_avgpool_nhwc_batch_f32(src.data_ptr(), BH // batch, W, ld, int(c0), int(Cn), dst.data_ptr(), s, int(batch),
                                                      _stream()), "usc_adaptive_avgpool_nhwc_f32")
    return dst


def im2col(src, cuse, kh, kw, stride, pad, out):
    """src (IH, IW, cpitch) f32 NHWC -> out (OH, OW, Kpad) patch matrix of a kh x kw / stride convolution over `cuse` channels."""
    IH, IW, cp = src.shape
    OH, OW, Kp = out.shape
    _check(load().usc_im2col_nhwc_f32(src.data_ptr(), IH, IW, cp, int(cuse), int(kh), int(kw), int(stride), int(pad), out.data_ptr(),
                                      OH, OW, Kp, _stream()), "usc_im2col_nhwc_f32")
    return out


def segagent_target_size(h, w, long_side, division=1, crop_h=513, crop_w=513):
    a, b = C.c_int(), C.c_int()
    load().usc_segagent_target_size(int(h), int(w), int(long_side), int(division), int(crop_h), int(crop_w), C.byref(a), C.byref(b))
    return a.value, b.value


def grid_crop_locations(h, w, crop_h=513, crop_w=513, stride_ratio=0.5, flip=True):
    """(N, 4) int32 array of (s_h, e_h, s_w, e_w); mirrored crops have s_w > e_w (reference agent.py:102-141)."""
    n = load().usc_grid_crop_locations(int(h), int(w), int(crop_h), int(crop_w), float(stride_ratio), int(bool(flip)), None, 0)
    if n < 0:
        _check(n, "usc_grid_crop_locations")
    locs = np.empty((n, 4), np.int32)
    _check(min(load().usc_grid_crop_locations(int(h), int(w), int(crop_h), int(crop_w), float(stride_ratio), int(bool(flip)),
                                              locs.ctypes.data, n), 0), "usc_grid_crop_locations")
    return locs


def segagent_pre(img, th, tw, loc, out=None):
    """One normalised crop (crop_h, crop_w, 4) f32 NHWC of the pad-resized frame; loc = (s_h, e_h, s_w, e_w)."""
    H, W, _ = img.shape
    loc = np.ascontiguousarray(np.asarray(loc, np.int32).reshape(4))
    ch, cw = int(loc[1] - loc[0]), int(abs(loc[3] - loc[2]))
    out = torch.empty((ch, cw, 4), dtype=torch.float32, device=img.device) if out is None else out
    _check(load().usc_segagent_pre(_dev_u8(img, "segagent_pre.img", (H, W, 3)), H, W, int(th), int(tw), loc.ctypes.data, out.data_ptr(),
                                   _stream()), "usc_segagent_pre")
    return out


def segagent_post(logits, locs, crop_h, crop_w, th, tw, H, W, out=None, ws=None, want_score=False):
    """logits (ncrops, lh, lw, cbuf) f32 CUDA; locs (ncrops, 4) host ints -> (H, W) uint8 mask (0 / 255)."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 4):
        raise NativeError("segagent_post.logits: expected a contiguous (ncrops, lh, lw, cbuf) float32 CUDA tensor")
    n, lh, lw, cbuf = logits.shape
    locs = np.ascontiguousarray(np.asarray(locs, np.int32).reshape(n, 4))
    out = torch.empty((H, W), dtype=torch.uint8, device=logits.device) if out is None else out
    need = load().usc_segagent_post_ws_bytes(int(th), int(tw))
    if ws is None or ws.numel() < need:
        ws = _ws(need, logits.device)
    score = torch.empty((H, W, 2), dtype=torch.float64, device=logits.device) if want_score else None
    _check(load().usc_segagent_post(logits.data_ptr(), n, lh, lw, cbuf, locs.ctypes.data, int(crop_h), int(crop_w), int(th), int(tw),
                                    int(H), int(W), _dev_u8(out, "segagent_post.out", (H, W)), score.data_ptr() if want_score else None,
                                    ws.data_ptr(), ws.numel(), _stream()), "usc_segagent_post")
    return (out, score) if want_score else out


def stm_pre(frame, mask, in_h, in_w, out=None):
    """frame (H, W, 3) u8 [+ mask (H, W) u8 or None] -> (in_h, in_w, 8) float32 NHWC (R, G, B, object, background, 0, 0, 0)."""
    H, W, _ = frame.shape
    out = torch.empty((in_h, in_w, 8), dtype=torch.float32, device=frame.device) if out is None else out
    _check(load().usc_stm_pre(_dev_u8(frame, "stm_pre.frame", (H, W, 3)), _dev_u8(mask, "stm_pre.mask", (H, W)) if mask is not None else None,
                              H, W, int(in_h), int(in_w), out.data_ptr(), _stream()), "usc_stm_pre")
    return out


def pack_kn(src, K, N, sk, sn, dst):
    """dst (Kpad/4, Npad, 4) f32 <- src[k*sk + n*sn], zero padded (the weight layout of the conv kernel)."""
    k4, npad, four = dst.shape
    _check(load().usc_pack_kn_f32(src.data_ptr(), int(K), int(N), int(sk), int(sn), k4 * 4, npad, dst.data_ptr(), _stream()),
           "usc_pack_kn_f32")
    return dst


def row_softmax(x, cols, div=1.0):
    """In place softmax(x / div) over the first `cols` columns of every row of a 2-D float32 CUDA tensor."""
    rows, ld = x.shape
    _check(load().usc_row_softmax_f32(x.data_ptr(), rows, int(cols), ld, float(div), _stream()), "usc_row_softmax_f32")
    return x


def stm_post(p2, in_h, in_w, H, W, out=None, prob=None):
    """p2 (lh, lw, cbuf) f32 CUDA decoder logits -> (mask (H, W) u8, prob (in_h, in_w, 2) f32)."""
    lh, lw, cbuf = p2.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=p2.device) if out is None else out
    prob = torch.empty((in_h, in_w, 2), dtype=torch.float32, device=p2.device) if prob is None else prob
    _check(load().usc_stm_post(p2.data_ptr(), lh, lw, cbuf, int(in_h), int(in_w), int(H), int(W), _dev_u8(out, "stm_post.out", (H, W)),
                               prob.data_ptr(), _stream()), "usc_stm_post")
    return out, prob


def stm_mask_from_prob(prob, H, W, out=None):
    """(in_h, in_w, 2) float32 CUDA probabilities (bg, fg) -> inv_pad_resize + argmax * 255 -> (H, W) uint8."""
    in_h, in_w, _ = prob.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=prob.device) if out is None else out
    _check(load().usc_stm_mask_from_prob(prob.data_ptr(), in_h, in_w, int(H), int(W), _dev_u8(out, "stm_mask.out", (H, W)), _stream()),
           "usc_stm_mask_from_prob")
    return out


def regionfill(img, mask, rtol=1e-7, max_iters=2000, out=None, ws=None, want_iters=False, budget=48, sync=True, status=None):
    """Laplace inpainting of the three channels of img (H, W, 3) inside mask > 0 (reference utils/region_fill.py).
    The C entry only enqueues work: `budget` iterations per call, the convergence test runs on the device and turns what is left
    of the budget into no-ops.  sync=True (the NumPy-API default): after each budget the 8-byte status {iterations, converged} is
    read back and the solve continued until it has converged (or max_iters).  sync=False: one budget, no host synchronisation -
    the caller looks at `status` (device int32[2]) whenever it synchronises anyway (BgOnlinePipeline does).
    want_iters: also return the iteration count (an int with sync=True, else the device status tensor)."""
    H, W, _ = img.shape
    out = torch.empty_like(img) if out is None else out
    need = load().usc_regionfill_ws_bytes(H, W)
    if ws is None or ws.numel() < need:
        ws = _ws(need, img.device)
    status = torch.zeros(2, dtype=torch.int32, device=img.device) if status is None else status
    pi, pm, po = _dev_u8(img, "regionfill.img", (H, W, 3)), _dev_u8(mask, "regionfill.mask", (H, W)), _dev_u8(out, "regionfill.out", (H, W, 3))
    it = 0
    while True:
        n = max(1, min(int(budget), int(max_iters) - it))
        _check(load().usc_regionfill_u8(pi, pm, H, W, po, float(rtol), it, n, status.data_ptr(), ws.data_ptr(), ws.numel(), _stream()),
               "usc_regionfill_u8")
        it += n
        if not sync:
            return (out, status) if want_iters else out
        used, conv = (int(v) for v in status.cpu())
        if conv or it >= max_iters:
            return (out, used) if want_iters else out


def bg_accumulate(frame, mask_dilated, raw, cont):
    """raw (H, W, 3) / cont (H, W) float32 CUDA accumulators += one frame outside its (dilated) mask (bg_offline.py:111-117)."""
    H, W, _ = frame.shape
    _check(load().usc_bg_accumulate_f32(_dev_u8(frame, "bg_accumulate.frame", (H, W, 3)), _dev_u8(mask_dilated, "bg_accumulate.mask", (H, W)),
                                        H, W, raw.data_ptr(), cont.data_ptr(), _stream()), "usc_bg_accumulate_f32")


def bg_finalize(raw, cont):
    """-> (bg_always (H, W, 3) uint8, mask_always (H, W) uint8 255 / 0)  (bg_offline.py:118-123)."""
    H, W = cont.shape
    bg = torch.empty((H, W, 3), dtype=torch.uint8, device=cont.device)
    ma = torch.empty((H, W), dtype=torch.uint8, device=cont.device)
    _check(load().usc_bg_finalize_u8(raw.data_ptr(), cont.data_ptr(), H, W, bg.data_ptr(), ma.data_ptr(), _stream()), "usc_bg_finalize_u8")
    return bg, ma


def humanseg_affine(w, h, in_w=473, in_h=473, inv=False):
    M = (C.c_double * 6)()
    load().usc_humanseg_affine(int(w), int(h), int(in_w), int(in_h), int(bool(inv)), M)
    return np.array(M[:], np.float64).reshape(2, 3)


def humanseg_pre(frame, in_h=473, in_w=473, out=None):
    """frame (H, W, 3) uint8 CUDA -> (in_h, in_w, 4) float32 NHWC (normalised R, G, B, 0)."""
    H, W, _ = frame.shape
    pf = _dev_u8(frame, "humanseg_pre.frame", (H, W, 3))
    out = torch.empty((in_h, in_w, 4), dtype=torch.float32, device=frame.device) if out is None else out
    rc = load().usc_humanseg_pre(pf, H, W, int(in_h), int(in_w), out.data_ptr(), _stream())
    if rc != 0:
        raise NativeError(f"usc_humanseg_pre failed ({rc}): {load().usc_seg_last_error().decode()}")
    return out


def humanseg_post(logits, C_classes, H, W, in_h=473, in_w=473, out=None, ws=None):
    """logits: (lh, lw, Cbuf) float32 NHWC CUDA buffer whose first C_classes channels are the classes -> (H, W) uint8."""
    if not (logits.is_cuda and logits.dtype == torch.float32 and logits.is_contiguous() and logits.dim() == 3):
        raise NativeError("humanseg_post.logits: expected a contiguous (lh, lw, C) float32 CUDA tensor")
    lh, lw, cbuf = logits.shape
    out = torch.empty((H, W), dtype=torch.uint8, device=logits.device) if out is None else out
    need = load().usc_humanseg_post_ws_bytes(int(C_classes), int(in_h), int(in_w))      # 0 since round 3; `ws` stays accepted
    if need and (ws is None or ws.numel() < need):
        ws = _ws(need, logits.device)
    rc = load().usc_humanseg_post(logits.data_ptr(), int(C_classes), cbuf, lh, lw, int(in_h), int(in_w), H, W,
                                  _dev_u8(out, "humanseg_post.out", (H, W)), ws.data_ptr() if ws is not None else None,
                                  ws.numel() if ws is not None else 0, _stream())
    if rc != 0:
        raise NativeError(f"usc_humanseg_post failed ({rc}): {load().usc_seg_last_error().decode()}")
    return out


# ---------------------------------------------------------------- object filter on the device
class DeviceObjectFilter:
    """remove_invalid_objects without leaving the GPU (workspace kept per instance; score_map() builds the float64 table the
    kernels evaluate in closed form - tests compare it with the reference's)."""

    def __init__(self):
        self._score = {}
        self._ws = None

    def score_map(self, H, W, center, device):
        key = (H, W, float(center[0]), float(center[1]), str(device))
        if key not in self._score:
            out = torch.empty((H, W), dtype=torch.float64, device=device)
            ws8 = torch.zeros(8, dtype=torch.uint8, device=device)
            _check(load().usc_score_map_f64(H, W, float(center[0]), float(center[1]), out.data_ptr(), ws8.data_ptr(), _stream()),
                   "usc_score_map_f64")
            self._score[key] = out
        return self._score[key]

    def __call__(self, alpha, seg, center, saliency_thr, consensus_thr, src=None):
        """alpha (H, W) uint8 CUDA, modified in place and returned; seg (H, W) uint8 CUDA or None.  With src (same shape), alpha
        becomes the filtered copy of src and src stays as it is."""
        H, W = alpha.shape
        pa = _dev_u8(alpha, "remove_invalid_objects.alpha")
        psrc = _dev_u8(src, "remove_invalid_objects.src", (H, W)) if src is not None else None
        ps = _dev_u8(seg, "remove_invalid_objects.seg", (H, W)) if seg is not None else None
        need = load().usc_remove_invalid_objects_ws_bytes(H, W)
        if self._ws is None or self._ws.numel() < need or self._ws.device != alpha.device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=alpha.device)
        _check(load().usc_remove_invalid_objects_u8(psrc, pa, ps, H, W, float(center[0]), float(center[1]), float(saliency_thr), float(consensus_thr),
                                                    self._ws.data_ptr(), self._ws.numel(), _stream()), "usc_remove_invalid_objects_u8")
        return alpha


# ---------------------------------------------------------------- host stage
class ObjectFilter:
    """remove_invalid_objects on host memory with a per-geometry cached score map."""

    def __init__(self):
        self._score = {}

    def __call__(self, alpha, seg, center, saliency_thr, consensus_thr):
        if not (isinstance(alpha, np.ndarray) and alpha.dtype == np.uint8 and alpha.flags.c_contiguous and alpha.ndim == 2):
            raise NativeError("remove_invalid_objects: alpha must be a contiguous 2-D uint8 ndarray")
        H, W = alpha.shape
        key = (H, W, float(center[0]), float(center[1]))
        if key not in self._score:
            self._score[key] = [np.empty((H, W), np.float64), C.c_int(0)]
        score, flag = self._score[key]
        seg_p = None
        if seg is not None:
            seg = np.ascontiguousarray(seg, dtype=np.uint8)
            seg_p = seg.ctypes.data
        _check(load().usc_remove_invalid_objects_host(alpha.ctypes.data, seg_p, H, W, float(center[0]), float(center[1]),
                                                      float(saliency_thr), float(consensus_thr), score.ctypes.data,
                                                      C.byref(flag)), "usc_remove_invalid_objects_host")
        return alpha
