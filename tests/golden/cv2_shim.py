"""A module with cv2's names, backed by the C oracle (oracle/usc_oracle.c) - TEST INFRASTRUCTURE for gen_orchestration.py.

The reference's Python (unscreen/utils/*.py, unscreen/trimap/agent.py, unscreen/vmatting/agent.py, the frame loop of
tools/unscreen/bg_offline.py) is run in the build container with `cv2` resolving to this module.  What that pins is the
reference's OWN code: order of operations, thresholds, dtype promotions, truncations, in-place / aliasing behaviour, which
arguments reach which OpenCV call (e.g. the interpolation flag passed positionally as `dst`, trimap/agent.py:59).  What it
does NOT pin is OpenCV's arithmetic inside each call: that is the oracle's restatement on both sides of the comparison
(OpenCV itself is not available: SURVEY.md section 8c).

Only the calls those files make are provided, with OpenCV's argument order and defaults."""
import numpy as np

import oracle

INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2
BORDER_CONSTANT, BORDER_REFLECT = 0, 2
COLOR_BGR2RGB, COLOR_BGR2GRAY, COLOR_RGB2GRAY = 4, 6, 7
COLOR_BGR2HSV, COLOR_BGR2Lab, COLOR_HSV2BGR, COLOR_Lab2BGR = 40, 44, 54, 56
COLOR_BGR2LAB, COLOR_LAB2BGR = COLOR_BGR2Lab, COLOR_Lab2BGR
MORPH_ELLIPSE = 2
RETR_LIST, CHAIN_APPROX_SIMPLE, FILLED = 1, 2, -1
calls = []          # (name, detail) log: lets the generator record which OpenCV entry points a reference function reached


def _log(name, detail=""):
    calls.append((name, detail))


def resize(src, dsize, dst=None, fx=0, fy=0, interpolation=INTER_LINEAR):
    """cv2.resize(src, dsize[, dst[, fx[, fy[, interpolation]]]]): a third POSITIONAL argument is `dst`, not the interpolation."""
    if dsize is None or tuple(dsize) == (0, 0):
        # cv::resize maps coordinates with fx / fy themselves in this form (and derives dsize from them); the one call of the reference
        # that reaches this branch on uint8 is rescale_fg (imgprocess.py:47, INTER_CUBIC)
        assert src.dtype == np.uint8 and interpolation == INTER_CUBIC, "resize(fx, fy): only the uint8 cubic form of rescale_fg"
        _log("resize", f"{src.dtype} {src.shape[:2]} fx={fx} fy={fy} interp={interpolation}")
        return oracle.resize_cubic_fxfy(np.ascontiguousarray(src), float(fx), float(fy))
    dw, dh = int(dsize[0]), int(dsize[1])
    _log("resize", f"{src.dtype} {src.shape[:2]} -> {(dh, dw)} interp={interpolation}")
    if src.dtype == np.uint8:
        fn = {INTER_NEAREST: oracle.resize_nearest, INTER_LINEAR: oracle.resize_linear, INTER_CUBIC: oracle.resize_cubic}[interpolation]
        return fn(np.ascontiguousarray(src), dh, dw)
    if src.dtype in (np.float32, np.float64) and interpolation == INTER_LINEAR:
        return oracle.resize_linear(np.ascontiguousarray(src), dh, dw)
    raise NotImplementedError(f"cv2_shim.resize: {src.dtype} {src.shape} interpolation {interpolation}")


def warpAffine(src, M, dsize, dst=None, flags=INTER_LINEAR, borderMode=0, borderValue=0):
    """cv2.warpAffine(img, M, (w, h)): INTER_LINEAR, BORDER_CONSTANT 0 - the form shift_fg uses (imgprocess.py:63); M as the caller built it
    (float32 there), read as double by OpenCV."""
    assert flags == INTER_LINEAR and borderMode == 0 and borderValue == 0 and src.dtype == np.uint8
    _log("warpAffine", f"{src.shape[:2]} -> {(dsize[1], dsize[0])}")
    return oracle.warp_affine(np.ascontiguousarray(src), np.asarray(M, np.float64), int(dsize[1]), int(dsize[0]))


def copyMakeBorder(src, top, bottom, left, right, borderType, dst=None, value=None):
    assert top == 0 and left == 0 and borderType == BORDER_REFLECT, "only the bottom / right BORDER_REFLECT of pad_resize"
    _log("copyMakeBorder", f"{src.shape[:2]} +{bottom} +{right}")
    return oracle.pad_reflect(np.ascontiguousarray(src), src.shape[0] + bottom, src.shape[1] + right)


def cvtColor(src, code, dst=None, dstCn=0):
    _log("cvtColor", str(code))
    if code == COLOR_BGR2RGB:
        out = np.ascontiguousarray(src[..., ::-1])
        if dst is not None:            # cv2.cvtColor(img, cv2.COLOR_BGR2RGB, img): in place (imgprocess.py:157)
            dst[...] = out
            return dst
        return out
    src = np.ascontiguousarray(src)
    shape = src.shape
    flat = src.reshape(-1, 1, 3)
    if code == COLOR_BGR2GRAY:
        return oracle.bgr2gray(flat).reshape(shape[:-1])
    if code == COLOR_RGB2GRAY:
        return oracle.bgr2gray(np.ascontiguousarray(flat[..., ::-1])).reshape(shape[:-1])
    fn = {COLOR_BGR2HSV: oracle.bgr2hsv, COLOR_HSV2BGR: oracle.hsv2bgr, COLOR_BGR2Lab: oracle.bgr2lab, COLOR_Lab2BGR: oracle.lab2bgr}[code]
    return fn(flat).reshape(shape)


THRESH_BINARY = 0


def threshold(src, thresh, maxval, type):
    assert type == THRESH_BINARY and src.dtype == np.uint8
    _log("threshold")
    return float(thresh), np.where(src > thresh, maxval, 0).astype(np.uint8)


def moments(array, binaryImage=False):
    assert array.dtype == np.uint8 and array.ndim == 2 and not binaryImage
    _log("moments")
    return oracle.moments(array)


def boxFilter(src, ddepth, ksize, dst=None, anchor=None, normalize=True, borderType=None):
    assert ddepth == -1 and ksize[0] == ksize[1] and normalize
    _log("boxFilter", str(ksize))
    return oracle.box_filter(np.ascontiguousarray(src), int(ksize[0]))


def _scalar_like(a, b):
    """OpenCV's array-op-scalar rule for a float32 array: the (double) scalar is converted to float32 and the operation runs in
    float32, one scalar entry per channel."""
    b = np.asarray(b, np.float64).reshape(-1)
    assert a.dtype == np.float32 and b.size == a.shape[-1]
    return b.astype(np.float32)


def subtract(src1, src2, dst=None):
    _log("subtract")
    out = src1 - _scalar_like(src1, src2)
    if dst is not None:
        dst[...] = out
        return dst
    return out


def multiply(src1, src2, dst=None):
    _log("multiply")
    out = src1 * _scalar_like(src1, src2)
    if dst is not None:
        dst[...] = out
        return dst
    return out


def getStructuringElement(shape, ksize):
    assert shape == MORPH_ELLIPSE and ksize[0] == ksize[1]
    return ("ellipse", int(ksize[0]))


def dilate(src, kernel, dst=None, anchor=None, iterations=1):
    _log("dilate", f"k={kernel[1]} iters={iterations}")
    return oracle.morph(np.ascontiguousarray(src), kernel[1], iterations, "dilate")


def erode(src, kernel, dst=None, anchor=None, iterations=1):
    _log("erode", f"k={kernel[1]} iters={iterations}")
    return oracle.morph(np.ascontiguousarray(src), kernel[1], iterations, "erode")


def inRange(src, lowerb, upperb):
    _log("inRange")
    lo, hi = np.asarray(lowerb, np.float64).reshape(-1), np.asarray(upperb, np.float64).reshape(-1)
    ok = np.logical_and(src >= lo, src <= hi).all(axis=-1)
    return np.where(ok, 255, 0).astype(np.uint8)


def findContours(image, mode, method):
    assert mode == RETR_LIST and method == CHAIN_APPROX_SIMPLE and image.dtype == np.uint8 and image.ndim == 2
    _log("findContours")
    contours = oracle.find_contours(np.ascontiguousarray(image))
    return [c.reshape(-1, 1, 2) for c in contours[::-1]], None       # OpenCV 4: (contours, hierarchy), last found first


def contourArea(contour):
    return oracle.contour_area(np.ascontiguousarray(contour.reshape(-1, 2)))


def drawContours(image, contours, contourIdx, color, thickness=1):
    assert thickness == FILLED and image.ndim == 2
    oracle.draw_contour_filled(image, np.ascontiguousarray(contours[contourIdx].reshape(-1, 2)), int(color[0]))
    return image
