/*
 * usc_oracle.c - CPU restatement of the reference's per-frame unscreen path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under video_unscreen_amd/ may import, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and only as the checker.  Plain scalar C, one thread, no SIMD: it is the spec the HIP
 * kernels are compared with bit for bit, not a fast CPU implementation.
 *
 * PARITY STATUS: the reference (AnyiRao/video_unscreen) has no tests, fixtures or golden
 * vectors, and all of its non-CNN arithmetic is delegated to OpenCV (cv2), which is neither
 * vendored nor pinned nor installed here.  The OpenCV primitives below (resize, morphology,
 * cvtColor, copyMakeBorder, findContours/contourArea/drawContours) are therefore restated
 * from the published algorithms of OpenCV 4.x's generic (non-SIMD, non-IPP) C++ paths:
 * PARITY UNPINNED at that boundary.  What IS pinned: the order of operations, thresholds,
 * dtype promotions and truncations of the reference's own Python, cited per function as
 * reference file:line (paths relative to the reference repo root).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: every float op rounds once, as the
 * NumPy / torch-CPU elementwise ops of the reference do).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static inline int orc_round_i(double v) { return (int)lrint(v); }          /* cvRound: half-to-even */
static inline int orc_floor_i(double v) { int i = (int)v; return i - (i > v); }
static inline int orc_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline uint8_t orc_sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
static inline short orc_sat_s16(int v) { return (short)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

/* ------------------------------------------------------------------------------------------
 * geometry helpers
 * ---------------------------------------------------------------------------------------- */

/* unscreen/utils/imgprocess.py:164-192 get_target_size.  Python floats are doubles; int() truncates. */
ORC_API void orc_get_target_size(int h, int w, int long_side, int division, int* th, int* tw)
{
    if (h > w) {
        *th = long_side;
        int t = (int)((double)long_side * w / h);
        if (t % division != 0) t = (t / division + 1) * division;
        *tw = t;
    } else {
        *tw = long_side;
        int t = (int)((double)long_side * h / w);
        if (t % division != 0) t = (t / division + 1) * division;
        *th = t;
    }
}

/* unscreen/utils/imgprocess.py:67-97 pad_resize: size of the aspect-preserving resize inside
 * a (target_h, target_w) canvas; the remainder is reflect-padded at the bottom / right. */
ORC_API void orc_pad_resize_geometry(int h, int w, int target_h, int target_w, int* new_h, int* new_w)
{
    if ((double)h / w > (double)target_h / target_w) {
        *new_h = target_h;
        *new_w = (int)((double)target_h * w / h);
    } else {
        *new_w = target_w;
        *new_h = (int)((double)target_w * h / w);
    }
}

/* unscreen/utils/imgprocess.py:100-132 inv_pad_resize: the un-padded region of an (h, w) canvas
 * for an original of (ori_h, ori_w). */
ORC_API void orc_inv_pad_geometry(int h, int w, int ori_h, int ori_w, int* resized_h, int* resized_w)
{
    if ((double)ori_h / ori_w > (double)h / w) {
        *resized_h = h;
        *resized_w = (int)((double)h * ori_w / ori_h);
    } else {
        *resized_w = w;
        *resized_h = (int)((double)w * ori_h / ori_w);
    }
}

/* ------------------------------------------------------------------------------------------
 * cv2.resize
 * ---------------------------------------------------------------------------------------- */

/* cv2.resize(..., interpolation=cv2.INTER_NEAREST)  (call site: unscreen/trimap/agent.py:52-53).
 * OpenCV resizeNN: sx = min(floor(x * (1/fx)), sw-1) with fx = dw/sw as double. */
ORC_API void orc_resize_nearest_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw)
{
    double fx = (double)dw / sw, fy = (double)dh / sh;
    double ifx = 1. / fx, ify = 1. / fy;
    for (int y = 0; y < dh; y++) {
        int sy = orc_floor_i(y * ify);
        if (sy > sh - 1) sy = sh - 1;
        for (int x = 0; x < dw; x++) {
            int sx = orc_floor_i(x * ifx);
            if (sx > sw - 1) sx = sw - 1;
            for (int c = 0; c < cn; c++)
                dst[((size_t)y * dw + x) * cn + c] = src[((size_t)sy * sw + sx) * cn + c];
        }
    }
}

typedef struct { int ofs; short a0, a1; float f0, f1; } orc_tap;

/* Source tap + weights of one destination coordinate for INTER_LINEAR (OpenCV resize():
 * f = (d + 0.5) * scale - 0.5 computed in double, cast to float; s = floor(f); f -= s;
 * left edge: s < 0 -> s = 0, f = 0; right edge: s + 1 >= n -> s = n - 1, f = 0.
 * 8-bit images use 11-bit fixed point weights (INTER_RESIZE_COEF_SCALE = 2048). */
static void orc_linear_taps(int dn, int sn, orc_tap* t, int clamp_edges)
{
    double inv_scale = (double)dn / sn;
    double scale = 1. / inv_scale;
    for (int d = 0; d < dn; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = orc_floor_i(f);
        f -= s;
        if (clamp_edges) {
            if (s < 0) { s = 0; f = 0.f; }
            if (s + 1 >= sn) { s = sn - 1; f = 0.f; }
        }
        t[d].ofs = s;
        t[d].f0 = 1.f - f;
        t[d].f1 = f;
        t[d].a0 = orc_sat_s16(orc_round_i((1.f - f) * 2048.f));
        t[d].a1 = orc_sat_s16(orc_round_i(f * 2048.f));
    }
}

static int orc_is_area_fast_2x(int sh, int sw, int dh, int dw)
{
    double scale_x = 1. / ((double)dw / sw), scale_y = 1. / ((double)dh / sh);
    int ix = orc_round_i(scale_x), iy = orc_round_i(scale_y);
    return fabs(scale_x - ix) < 2.220446049250313e-16 && fabs(scale_y - iy) < 2.220446049250313e-16 && ix == 2 && iy == 2;
}

/* cv2.resize(img, (dw, dh)) with the default INTER_LINEAR on uint8 data
 * (call sites: unscreen/utils/imgprocess.py:95 pad_resize, :36 adaptive_resize;
 * unscreen/trimap/agent.py:59 where the flag lands in `dst` so the default applies).
 * - exact 2x decimation is rerouted by OpenCV to the 2x2 INTER_AREA box average (s+2)>>2;
 * - otherwise horizontal pass keeps S0*a0 + S1*a1 (int), vertical pass is
 *   (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2; rows are clamped, columns zero-weighted. */
ORC_API void orc_resize_linear_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw)
{
    if (sh == dh && sw == dw) { memcpy(dst, src, (size_t)sh * sw * cn); return; }
    if (orc_is_area_fast_2x(sh, sw, dh, dw)) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const uint8_t* p = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    int s = p[0] + p[cn] + p[(size_t)sw * cn] + p[(size_t)sw * cn + cn];
                    dst[((size_t)y * dw + x) * cn + c] = (uint8_t)((s + 2) >> 2);
                }
        return;
    }
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * dw);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * dh);
    orc_linear_taps(dw, sw, tx, 1);
    orc_linear_taps(dh, sh, ty, 0);
    for (int y = 0; y < dh; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, sh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, sh - 1);
        int b0 = ty[y].a0, b1 = ty[y].a1;
        for (int x = 0; x < dw; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < sw ? x0 + 1 : x0;
            int a0 = tx[x].a0, a1 = tx[x].a1;
            for (int c = 0; c < cn; c++) {
                int r0 = src[((size_t)y0 * sw + x0) * cn + c] * a0 + src[((size_t)y0 * sw + x1) * cn + c] * a1;
                int r1 = src[((size_t)y1 * sw + x0) * cn + c] * a0 + src[((size_t)y1 * sw + x1) * cn + c] * a1;
                int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
                dst[((size_t)y * dw + x) * cn + c] = (uint8_t)v;
            }
        }
    }
    free(tx); free(ty);
}

/* cv2.resize on a float32 single-channel image (unscreen/utils/imgprocess.py:131 inv_pad_resize on
 * the network's alpha, reached from unscreen/vmatting/agent.py:127).  Float weights, two passes:
 * row = S[x0]*a0 + S[x1]*a1, then out = row0*b0 + row1*b1, each product and sum rounded to f32. */
ORC_API void orc_resize_linear_f32(const float* src, int sh, int sw, int sstride, float* dst, int dh, int dw)
{
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * dw);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * dh);
    orc_linear_taps(dw, sw, tx, 1);
    orc_linear_taps(dh, sh, ty, 0);
    for (int y = 0; y < dh; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, sh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, sh - 1);
        float b0 = ty[y].f0, b1 = ty[y].f1;
        for (int x = 0; x < dw; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < sw ? x0 + 1 : x0;
            float a0 = tx[x].f0, a1 = tx[x].f1;
            float r0 = src[(size_t)y0 * sstride + x0] * a0 + src[(size_t)y0 * sstride + x1] * a1;
            float r1 = src[(size_t)y1 * sstride + x0] * a0 + src[(size_t)y1 * sstride + x1] * a1;
            dst[(size_t)y * dw + x] = r0 * b0 + r1 * b1;
        }
    }
    free(tx); free(ty);
}

/* cv2.resize(..., interpolation=cv2.INTER_CUBIC) on uint8 (unscreen/utils/imgprocess.py:48 rescale_fg):
 * Keys cubic with A = -0.75, float coefficients rounded to 11-bit fixed point (no sum correction), taps clamped to
 * the image (replicate) in both directions, horizontal pass kept as int, vertical pass
 * (sum + 2^21) >> 22 with saturation (FixedPtCast<int, uchar, 22>). */
static void orc_cubic_coeffs(float x, short* c)
{
    const float A = -0.75f;
    float f[4];
    f[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    f[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    f[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    f[3] = 1.f - f[0] - f[1] - f[2];
    for (int k = 0; k < 4; k++) c[k] = orc_sat_s16(orc_round_i(f[k] * 2048.f));
}

/* inv_x / inv_y: the fx / fy of a cv2.resize(src, None, fx=, fy=) call (cv::resize keeps the caller's factors for the coordinate
 * map and only derives dsize from them; with a dsize it derives the factors from the two sizes), 0 = derive from the sizes. */
static void orc_resize_cubic_scaled(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, double inv_x, double inv_y)
{
    double scale_x = 1. / (inv_x > 0 ? inv_x : (double)dw / sw), scale_y = 1. / (inv_y > 0 ? inv_y : (double)dh / sh);
    for (int y = 0; y < dh; y++) {
        float fy = (float)((y + 0.5) * scale_y - 0.5);
        int sy = orc_floor_i(fy);
        fy -= sy;
        short cy[4];
        orc_cubic_coeffs(fy, cy);
        for (int x = 0; x < dw; x++) {
            float fx = (float)((x + 0.5) * scale_x - 0.5);
            int sx = orc_floor_i(fx);
            fx -= sx;
            short cx[4];
            orc_cubic_coeffs(fx, cx);
            for (int c = 0; c < cn; c++) {
                int acc = 0;
                for (int j = 0; j < 4; j++) {
                    int yy = orc_clampi(sy - 1 + j, 0, sh - 1);
                    int row = 0;
                    for (int i = 0; i < 4; i++) {
                        int xx = orc_clampi(sx - 1 + i, 0, sw - 1);
                        row += src[((size_t)yy * sw + xx) * cn + c] * cx[i];
                    }
                    acc += row * cy[j];
                }
                dst[((size_t)y * dw + x) * cn + c] = orc_sat_u8((acc + (1 << 21)) >> 22);
            }
        }
    }
}

ORC_API void orc_resize_cubic_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw)
{
    orc_resize_cubic_scaled(src, sh, sw, cn, dst, dh, dw, 0., 0.);
}

/* cv2.resize(src, None, fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC): dsize = (round(sw fx), round(sh fy)) chosen by the caller, the
 * coordinate map uses fx / fy themselves. */
ORC_API void orc_resize_cubic_fxfy_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, double fx, double fy)
{
    orc_resize_cubic_scaled(src, sh, sw, cn, dst, dh, dw, fx, fy);
}

/* unscreen/utils/imgprocess.py:40-52 rescale_fg: cv2.resize(img, None, fx=s, fy=s, INTER_CUBIC) (dsize = round(size*s); the
 * coordinate map uses 1 / s itself, which differs from old / new whenever size * s is not an integer),
 * then the centre crop back to the original size (offsets int((new - old) / 2)). */
ORC_API void orc_rescale_fg(const uint8_t* src, int h, int w, int cn, double scale, uint8_t* dst)
{
    int nh = orc_round_i(h * scale), nw = orc_round_i(w * scale);
    uint8_t* big = (uint8_t*)malloc((size_t)nh * nw * cn);
    orc_resize_cubic_scaled(src, h, w, cn, big, nh, nw, scale, scale);
    int h_off = (int)((nh - h) / 2.0), w_off = (int)((nw - w) / 2.0);
    for (int y = 0; y < h; y++)
        memcpy(dst + (size_t)y * w * cn, big + ((size_t)(y + h_off) * nw + w_off) * cn, (size_t)w * cn);
    free(big);
}

/* unscreen/utils/fgfuncs.py:172-214 composite_fgbg (extend = False): bg bilinear-resized to cover the fg, centre ROI,
 * alpha = a/255 (float64) with alpha > 0.9 -> 1, out = clip(fg + bg_roi * (1 - alpha), 0, 255) truncated. */
ORC_API void orc_composite_fgbg(const uint8_t* fg, const uint8_t* alpha, int fh, int fw, const uint8_t* bg, int bh, int bw, uint8_t* out)
{
    int nbw, nbh;
    if ((double)fh / fw > (double)bh / bw) { nbh = fh; nbw = (int)((double)bw * nbh / bh); }
    else { nbw = fw; nbh = (int)((double)bh * nbw / bw); }
    uint8_t* rb = (uint8_t*)malloc((size_t)nbh * nbw * 3);
    orc_resize_linear_u8(bg, bh, bw, 3, rb, nbh, nbw);
    int left = nbw / 2 - fw / 2; if (left < 0) left = 0;
    int top = nbh / 2 - fh / 2; if (top < 0) top = 0;
    for (int y = 0; y < fh; y++)
        for (int x = 0; x < fw; x++) {
            double a = (double)alpha[(size_t)y * fw + x] / 255.;
            if (a > 0.9) a = 1;
            for (int c = 0; c < 3; c++) {
                double v = (double)fg[((size_t)y * fw + x) * 3 + c] + (double)rb[((size_t)(y + top) * nbw + x + left) * 3 + c] * (1 - a);
                v = v < 0 ? 0 : (v > 255 ? 255 : v);
                out[((size_t)y * fw + x) * 3 + c] = (uint8_t)v;
            }
        }
    free(rb);
}

/* cv2.copyMakeBorder(img, 0, pad_h, 0, pad_w, cv2.BORDER_REFLECT)  (unscreen/utils/imgprocess.py:96):
 * fedcba|abcdefgh|hgfedcb - the edge sample is repeated. */
static inline int orc_reflect_past_end(int p, int n)          /* cv::borderInterpolate(BORDER_REFLECT): bounces until inside */
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

ORC_API void orc_pad_reflect_u8(const uint8_t* src, int h, int w, int cn, uint8_t* dst, int dh, int dw)
{
    for (int y = 0; y < dh; y++) {
        int sy = orc_reflect_past_end(y, h);
        for (int x = 0; x < dw; x++) {
            int sx = orc_reflect_past_end(x, w);
            memcpy(dst + ((size_t)y * dw + x) * cn, src + ((size_t)sy * w + sx) * cn, cn);
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * morphology
 * ---------------------------------------------------------------------------------------- */

/* cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (k, k)): row i spans [c-dx, c+dx] with
 * dx = round(c * sqrt((r^2 - dy^2) / r^2)), r = c = k/2.  k=3 -> plus; k=4 -> 0010/1111/1111/1111. */
static void orc_ellipse(int k, uint8_t* se)
{
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1. / ((double)r * r) : 0;
    memset(se, 0, (size_t)k * k);
    for (int i = 0; i < k; i++) {
        int dy = i - r, j1 = 0, j2 = 0;
        if (abs(dy) <= r) {
            int dx = orc_round_i(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < k ? c + dx + 1 : k;
        }
        for (int j = j1; j < j2; j++) se[i * k + j] = 1;
    }
}

/* unscreen/utils/maskprocess.py:7-34 dilate_mask / erode_mask: cv2.dilate / cv2.erode with the
 * ellipse element, anchor at (k/2, k/2), `iters` successive passes, and the default border that
 * never wins (out-of-image taps are ignored).  op: 0 = dilate (max), 1 = erode (min). */
ORC_API void orc_morph_u8(const uint8_t* src, uint8_t* dst, int h, int w, int ksize, int iters, int op)
{
    uint8_t se[64 * 64];
    if (ksize > 64) ksize = 64;
    orc_ellipse(ksize, se);
    int a = ksize / 2;
    uint8_t* cur = (uint8_t*)malloc((size_t)h * w);
    uint8_t* nxt = (uint8_t*)malloc((size_t)h * w);
    memcpy(cur, src, (size_t)h * w);
    for (int it = 0; it < iters; it++) {
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                int best = op ? 255 : 0;
                for (int i = 0; i < ksize; i++) {
                    int yy = y + i - a;
                    if (yy < 0 || yy >= h) continue;
                    for (int j = 0; j < ksize; j++) {
                        if (!se[i * ksize + j]) continue;
                        int xx = x + j - a;
                        if (xx < 0 || xx >= w) continue;
                        int v = cur[(size_t)yy * w + xx];
                        if (op ? v < best : v > best) best = v;
                    }
                }
                nxt[(size_t)y * w + x] = (uint8_t)best;
            }
        uint8_t* t = cur; cur = nxt; nxt = t;
    }
    memcpy(dst, cur, (size_t)h * w);
    free(cur); free(nxt);
}

/* ------------------------------------------------------------------------------------------
 * cv2.cvtColor (8-bit)
 * ---------------------------------------------------------------------------------------- */

/* COLOR_BGR2GRAY, 8-bit: 15-bit fixed point (OpenCV 4.x: B 3735, G 19235, R 9798, +16384 >> 15). */
static inline uint8_t orc_gray(int b, int g, int r) { return (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15); }

ORC_API void orc_bgr2gray_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    for (size_t i = 0; i < n; i++) dst[i] = orc_gray(src[3 * i], src[3 * i + 1], src[3 * i + 2]);
}

static int orc_sdiv[256], orc_hdiv[256], orc_tabs_ready = 0;
static void orc_hsv_tables(void)
{
    if (orc_tabs_ready) return;
    orc_sdiv[0] = orc_hdiv[0] = 0;
    for (int i = 1; i < 256; i++) {
        orc_sdiv[i] = orc_round_i((255 << 12) / (1. * i));
        orc_hdiv[i] = orc_round_i((180 << 12) / (6. * i));
    }
    orc_tabs_ready = 1;
}

/* COLOR_BGR2HSV, 8-bit, H in [0,180): OpenCV RGB2HSV_b integer path with hsv_shift = 12 tables. */
static inline void orc_bgr2hsv_px(int b, int g, int r, uint8_t* out)
{
    int v = b, vmin = b;
    if (g > v) v = g;
    if (r > v) v = r;
    if (g < vmin) vmin = g;
    if (r < vmin) vmin = r;
    int diff = v - vmin;
    int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    int s = (diff * orc_sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * orc_hdiv[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    out[0] = orc_sat_u8(hh); out[1] = (uint8_t)s; out[2] = (uint8_t)v;
}

ORC_API void orc_bgr2hsv_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    orc_hsv_tables();
    for (size_t i = 0; i < n; i++) orc_bgr2hsv_px(src[3 * i], src[3 * i + 1], src[3 * i + 2], dst + 3 * i);
}

/* COLOR_HSV2BGR, 8-bit: h, s/255, v/255 as float, the six-sector float formula, *255 and
 * round-to-nearest-even with saturation (OpenCV HSV2RGB_b scalar path, hscale = 6/180). */
static inline void orc_hsv2bgr_px(const uint8_t* in, uint8_t* out)
{
    static const int sector_data[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    float h = in[0], s = in[1] * (1.f / 255.f), v = in[2] * (1.f / 255.f);
    float b, g, r;
    if (s == 0) b = g = r = v;
    else {
        float tab[4];
        h *= (6.f / 180.f);
        if (h < 0) do h += 6; while (h < 0);
        else if (h >= 6) do h -= 6; while (h >= 6);
        int sector = orc_floor_i(h);
        h -= sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        tab[0] = v;
        tab[1] = v * (1.f - s);
        tab[2] = v * (1.f - s * h);
        tab[3] = v * (1.f - s * (1.f - h));
        b = tab[sector_data[sector][0]];
        g = tab[sector_data[sector][1]];
        r = tab[sector_data[sector][2]];
    }
    out[0] = orc_sat_u8((int)lrintf(b * 255.f));
    out[1] = orc_sat_u8((int)lrintf(g * 255.f));
    out[2] = orc_sat_u8((int)lrintf(r * 255.f));
}

ORC_API void orc_hsv2bgr_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    for (size_t i = 0; i < n; i++) orc_hsv2bgr_px(src + 3 * i, dst + 3 * i);
}

/* ------------------------------------------------------------------------------------------
 * stage restatements
 * ---------------------------------------------------------------------------------------- */

/* unscreen/utils/maskprocess.py:56-60 exist_foreground: number of pixels >= 128 (the caller compares
 * it with fg_exist_thr * h * w in double). */
ORC_API long long orc_count_ge(const uint8_t* mask, size_t n, int thr)
{
    long long c = 0;
    for (size_t i = 0; i < n; i++) c += mask[i] >= thr;
    return c;
}

/* unscreen/trimap/agent.py:35-61 TrimapAgent.generate_trimap:
 *  :50-53 NEAREST resize to get_target_size(h, w, long_side);  :55-56 dilate / erode (k, iters);
 *  :54,57-58 trimap = 128, [eroded > 127] = 255, [dilated < 128] = 0 (in that order);
 *  :59 resize back with the DEFAULT bilinear (the INTER_NEAREST constant is passed as `dst`);
 *  :60 every value strictly between 0 and 255 becomes 128. */
ORC_API void orc_trimap(const uint8_t* mask, int H, int W, int long_side, int ksize, int iters, uint8_t* out)
{
    int h, w;
    orc_get_target_size(H, W, long_side, 1, &h, &w);
    size_t n = (size_t)h * w;
    uint8_t* small = (uint8_t*)malloc(n);
    uint8_t* dil = (uint8_t*)malloc(n);
    uint8_t* ero = (uint8_t*)malloc(n);
    uint8_t* tri = (uint8_t*)malloc(n);
    orc_resize_nearest_u8(mask, H, W, 1, small, h, w);
    orc_morph_u8(small, dil, h, w, ksize, iters, 0);
    orc_morph_u8(small, ero, h, w, ksize, iters, 1);
    for (size_t i = 0; i < n; i++) {
        uint8_t t = 128;
        if (ero[i] > 127) t = 255;
        if (dil[i] < 128) t = 0;
        tri[i] = t;
    }
    orc_resize_linear_u8(tri, h, w, 1, out, H, W);
    for (size_t i = 0; i < (size_t)H * W; i++)
        if (out[i] > 0 && out[i] < 255) out[i] = 128;
    free(small); free(dil); free(ero); free(tri);
}

/* tools/unscreen/bg_offline.py:150-160 (the same arithmetic as tools/unscreen/bg.py:85-92):
 *  :150-151 bg = uint8(f32(bg_i) * beta + (1 - beta) * f32(bg_always))   (f32 products, f32 sum, truncation;
 *           `1 - beta` is evaluated in double first, then used as an f32 scalar)
 *  :152     alpha = BGR2GRAY(segmask, 3 channels)
 *  :154     d = uint8(|f32(frame) - f32(bg)|)
 *  :155-157 g = BGR2GRAY(d); g[g > thr] = 255
 *  :158     dilate_mask(g, 4, 2)
 *  :160     alpha = alpha * (g // 255)    (uint8 product with 0 / 1)
 * seg_cn is 3 (the reference reads the mask JPEGs as 3-channel) or 1 (already grey). */
ORC_API void orc_bgdiff_mask(const uint8_t* frame, const uint8_t* bg_i, const uint8_t* bg_always,
                             const uint8_t* seg, int seg_cn, double beta, int thr, int H, int W,
                             uint8_t* out_bg, uint8_t* out_alpha)
{
    size_t n = (size_t)H * W;
    float fb = (float)beta, fb1 = (float)(1 - beta);
    uint8_t* g = (uint8_t*)malloc(n);
    uint8_t* gd = (uint8_t*)malloc(n);
    for (size_t i = 0; i < n; i++) {
        int d[3];
        for (int c = 0; c < 3; c++) {
            float p0 = (float)bg_i[3 * i + c] * fb;
            float p1 = fb1 * (float)bg_always[3 * i + c];
            float f = p0 + p1;
            uint8_t b = (uint8_t)f;
            out_bg[3 * i + c] = b;
            float df = fabsf((float)frame[3 * i + c] - (float)b);
            d[c] = (uint8_t)df;
        }
        uint8_t gv = orc_gray(d[0], d[1], d[2]);
        g[i] = gv > thr ? 255 : gv;
    }
    orc_morph_u8(g, gd, H, W, 4, 2, 0);
    for (size_t i = 0; i < n; i++) {
        uint8_t a = seg_cn == 3 ? orc_gray(seg[3 * i], seg[3 * i + 1], seg[3 * i + 2]) : seg[i];
        out_alpha[i] = (uint8_t)(a * (gd[i] / 255));
    }
    free(g); free(gd);
}

/* unscreen/vmatting/agent.py:76-106 preprocess + :44-74 to_tensor + unscreen/utils/imgprocess.py:135-161
 * imnormalize, producing the network's 7-channel input in NCHW order (model.py:564 torch.cat):
 *  ch0-2  RGB (the BGR frame is swapped, imgprocess.py:157): (f32(u8) / 255) - f32(mean) then * f32(1/std)
 *         (cv2.subtract / cv2.multiply with float64 scalars run in f32 on an f32 image)
 *  ch3    alpha_pre / 255 evaluated in double, rounded to f32 (agent.py:100 + .float())
 *  ch4-6  one-hot of the bilinearly resized trimap: 0 -> ch4, strictly between -> ch5, 255 -> ch6
 * Each u8 input is resized with pad_resize (bilinear + bottom/right BORDER_REFLECT).
 * layout: 0 = NCHW planes, 1 = NHWC (7 floats per pixel). */
ORC_API void orc_vmat_pre(const uint8_t* frame, const uint8_t* alpha_pre, const uint8_t* trimap, int H, int W,
                          int inH, int inW, float* out, int layout)
{
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    int nh, nw;
    orc_pad_resize_geometry(H, W, inH, inW, &nh, &nw);
    size_t n = (size_t)inH * inW;
    uint8_t* img_r = (uint8_t*)malloc((size_t)nh * nw * 3);
    uint8_t* img_p = (uint8_t*)malloc(n * 3);
    uint8_t* tmp_r = (uint8_t*)malloc((size_t)nh * nw);
    uint8_t* tri_p = (uint8_t*)malloc(n);
    uint8_t* ap_p = (uint8_t*)malloc(n);
    orc_resize_linear_u8(frame, H, W, 3, img_r, nh, nw);
    orc_pad_reflect_u8(img_r, nh, nw, 3, img_p, inH, inW);
    orc_resize_linear_u8(trimap, H, W, 1, tmp_r, nh, nw);
    orc_pad_reflect_u8(tmp_r, nh, nw, 1, tri_p, inH, inW);
    orc_resize_linear_u8(alpha_pre, H, W, 1, tmp_r, nh, nw);
    orc_pad_reflect_u8(tmp_r, nh, nw, 1, ap_p, inH, inW);
    float meanf[3], stdinvf[3];
    for (int c = 0; c < 3; c++) { meanf[c] = (float)mean[c]; stdinvf[c] = (float)(1 / std[c]); }
    for (size_t i = 0; i < n; i++) {
        float v[7];
        for (int c = 0; c < 3; c++) {
            float x = (float)img_p[3 * i + (2 - c)] / 255.f;
            x = x - meanf[c];
            v[c] = x * stdinvf[c];
        }
        v[3] = (float)((double)ap_p[i] / 255.);
        uint8_t t = tri_p[i];
        v[4] = t == 0 ? 1.f : 0.f;
        v[5] = (t > 0 && t < 255) ? 1.f : 0.f;
        v[6] = t == 255 ? 1.f : 0.f;
        for (int c = 0; c < 7; c++) {
            if (layout == 0) out[(size_t)c * n + i] = v[c];
            else out[i * 7 + c] = v[c];
        }
    }
    free(img_r); free(img_p); free(tmp_r); free(tri_p); free(ap_p);
}

/* unscreen/vmatting/agent.py:108-132 postprocess: inv_pad_resize (crop the pad, float bilinear back to H x W),
 * pred[trimap == 0] = 0, pred[trimap == 255] = 1, (pred * 255).astype(uint8) (f32 product, truncation).
 * alpha_f32 (optional) receives the clamped float alpha before the *255 truncation. */
ORC_API void orc_vmat_post(const float* pred, int inH, int inW, const uint8_t* trimap, int H, int W,
                           uint8_t* alpha_u8, float* alpha_f32)
{
    int rh, rw;
    orc_inv_pad_geometry(inH, inW, H, W, &rh, &rw);
    float* up = (float*)malloc(sizeof(float) * (size_t)H * W);
    orc_resize_linear_f32(pred, rh, rw, inW, up, H, W);
    for (size_t i = 0; i < (size_t)H * W; i++) {
        float a = up[i];
        if (trimap[i] == 0) a = 0.f;
        if (trimap[i] == 255) a = 1.f;
        if (alpha_f32) alpha_f32[i] = a;
        float s = a * 255.f;
        alpha_u8[i] = (uint8_t)s;
    }
    free(up);
}

/* tools/unscreen/bg_offline.py:171 paste + unscreen/utils/fgfuncs.py:84-110 get_fg:
 *  bg[alpha == 0] = frame[alpha == 0]   (paste != 0; bg_offline.py:171, bg.py:99, green.py:125 uses alpha < 128)
 *  both to 8-bit HSV; a = f32(alpha) / 255; fg = img_hsv - (1 - a) * bg_hsv on all three channels,
 *  clamp to [0, 255], truncate to uint8, HSV -> BGR.  Every float op rounds to f32 (torch CPU). */
ORC_API void orc_get_fg(const uint8_t* frame, const uint8_t* alpha, const uint8_t* bg, int paste_thr, size_t n,
                        uint8_t* out_fg, uint8_t* out_bg_pasted)
{
    orc_hsv_tables();
    for (size_t i = 0; i < n; i++) {
        uint8_t bpx[3], ih[3], bh[3], fh[3];
        /* paste_thr: 0 -> no paste; 1 -> alpha == 0 (alpha < 1); 128 -> alpha < 128 */
        int paste = alpha[i] < paste_thr;
        for (int c = 0; c < 3; c++) bpx[c] = paste ? frame[3 * i + c] : bg[3 * i + c];
        if (out_bg_pasted) memcpy(out_bg_pasted + 3 * i, bpx, 3);
        orc_bgr2hsv_px(frame[3 * i], frame[3 * i + 1], frame[3 * i + 2], ih);
        orc_bgr2hsv_px(bpx[0], bpx[1], bpx[2], bh);
        float a = (float)alpha[i] / 255.f;
        float w = 1.f - a;
        for (int c = 0; c < 3; c++) {
            float t = w * (float)bh[c];
            float f = (float)ih[c] - t;
            f = f < 0.f ? 0.f : (f > 255.f ? 255.f : f);
            fh[c] = (uint8_t)f;
        }
        orc_hsv2bgr_px(fh, out_fg + 3 * i);
    }
}

/* unscreen/utils/fgfuncs.py:113-137 get_bg: (1 - a) * bg_hsv, clamp, truncate, HSV -> BGR. */
ORC_API void orc_get_bg(const uint8_t* alpha, const uint8_t* bg, size_t n, uint8_t* out)
{
    orc_hsv_tables();
    for (size_t i = 0; i < n; i++) {
        uint8_t bh[3], fh[3];
        orc_bgr2hsv_px(bg[3 * i], bg[3 * i + 1], bg[3 * i + 2], bh);
        float a = (float)alpha[i] / 255.f;
        float w = 1.f - a;
        for (int c = 0; c < 3; c++) {
            float f = w * (float)bh[c];
            f = f < 0.f ? 0.f : (f > 255.f ? 255.f : f);
            fh[c] = (uint8_t)f;
        }
        orc_hsv2bgr_px(fh, out + 3 * i);
    }
}

/* tools/replace/replace.py:74-76: res = fg * m + bg * (1 - m) in float64 with m = mask / 255 per channel,
 * truncated to uint8.  mask_cn: 1 or 3. */
ORC_API void orc_blend(const uint8_t* fg, const uint8_t* mask, int mask_cn, const uint8_t* bg, size_t n, uint8_t* out)
{
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            double m = (double)(mask_cn == 3 ? mask[3 * i + c] : mask[i]) / 255.;
            double r = (double)fg[3 * i + c] * m + (double)bg[3 * i + c] * (1 - m);
            out[3 * i + c] = (uint8_t)r;
        }
}

/* unscreen/utils/fgfuncs.py:9-65 is_pixel_inrange with long_side_input <= 0: the pixel is "background coloured" when
 * every HSV channel lies in [clip(bg - win//2, 10, 255), clip(bg + win//2, 10, 255)] (both ends inclusive:
 * cv2.inRange for a colour, the torch ge/ge pair for a background image).  bg_is_image: bg is H*W*3, else 3 bytes.
 * out: 1 where in range, else 0. */
ORC_API void orc_inrange_hsv(const uint8_t* img, const uint8_t* bg, int bg_is_image, const int* winsize, size_t n, uint8_t* out)
{
    orc_hsv_tables();
    int half[3] = {winsize[0] / 2, winsize[1] / 2, winsize[2] / 2};
    uint8_t bh[3];
    if (!bg_is_image) orc_bgr2hsv_px(bg[0], bg[1], bg[2], bh);
    for (size_t i = 0; i < n; i++) {
        uint8_t ih[3];
        orc_bgr2hsv_px(img[3 * i], img[3 * i + 1], img[3 * i + 2], ih);
        if (bg_is_image) orc_bgr2hsv_px(bg[3 * i], bg[3 * i + 1], bg[3 * i + 2], bh);
        int ok = 1;
        for (int c = 0; c < 3; c++) {
            int lo = orc_clampi((int)bh[c] - half[c], 10, 255), hi = orc_clampi((int)bh[c] + half[c], 10, 255);
            ok &= (ih[c] >= lo) && (ih[c] <= hi);
        }
        out[i] = (uint8_t)ok;
    }
}

/* unscreen/trimap/agent.py:63-101 TrimapAgent.generate_trimap_withbg:
 *  :88-89  empty mask -> returned unchanged (all zeros)
 *  :90-91  bgmask = is_pixel_inrange(img, bg); fuzzy = (mask > 0) & bgmask
 *  :94     if fuzzy.sum() / (mask > 0).sum() > 0.1 (double division): plain trimap of the mask
 *  :97-100 else zero the fuzzy pixels, trimap, then force the fuzzy pixels to 128 */
ORC_API void orc_trimap_withbg(const uint8_t* mask, const uint8_t* img, const uint8_t* bg, int bg_is_image, const int* winsize,
                               int H, int W, int long_side, int ksize, int iters, uint8_t* out)
{
    size_t n = (size_t)H * W;
    long long nm = 0, nf = 0;
    uint8_t* bgmask = (uint8_t*)malloc(n);
    orc_inrange_hsv(img, bg, bg_is_image, winsize, n, bgmask);
    for (size_t i = 0; i < n; i++) { nm += mask[i] > 0; nf += (mask[i] > 0) && bgmask[i]; }
    if (nm == 0) { memcpy(out, mask, n); free(bgmask); return; }
    if ((double)nf / (double)nm > 0.1) {
        orc_trimap(mask, H, W, long_side, ksize, iters, out);
    } else {
        uint8_t* ens = (uint8_t*)malloc(n);
        for (size_t i = 0; i < n; i++) ens[i] = (mask[i] > 0 && bgmask[i]) ? 0 : mask[i];
        orc_trimap(ens, H, W, long_side, ksize, iters, out);
        for (size_t i = 0; i < n; i++) if (mask[i] > 0 && bgmask[i]) out[i] = 128;
        free(ens);
    }
    free(bgmask);
}

/* ------------------------------------------------------------------------------------------
 * remove_invalid_objects (unscreen/utils/maskprocess.py:77-189)
 * ---------------------------------------------------------------------------------------- */

/* numpy.linspace(start, stop, n)[i]: i * ((stop - start) / (n - 1)) + start, last element = stop. */
static double orc_linspace(double start, double stop, int n, int i)
{
    if (n == 1) return start;
    if (i == n - 1) return stop;
    double step = (stop - start) / (double)(n - 1);
    return (double)i * step + start;
}

/* unscreen/utils/maskprocess.py:155-178 get_score_map, float64:
 * s = sqrt(colterm(x)^2 + rowterm(y)^2) with linspace ramps around (y, x) = (int(h*cy), int(w*cx));
 * score = (max(s) - s) / max(s). */
ORC_API void orc_score_map(int h, int w, double cy, double cx, double* out)
{
    int y0 = (int)(h * cy), x0 = (int)(w * cx);
    double* col = (double*)malloc(sizeof(double) * w);
    double* row = (double*)malloc(sizeof(double) * h);
    for (int x = 0; x < w; x++) {
        double v = x >= x0 ? orc_linspace(0, 1, w - x0, x - x0) : orc_linspace(1, 0, x0, x);
        col[x] = v * v;
    }
    for (int y = 0; y < h; y++) {
        double v = y >= y0 ? orc_linspace(0, 1, h - y0, y - y0) : orc_linspace(1, 0, y0, y);
        row[y] = v * v;
    }
    double mx = -1;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            double s = sqrt(col[x] + row[y]);
            out[(size_t)y * w + x] = s;
            if (s > mx) mx = s;
        }
    for (size_t i = 0; i < (size_t)h * w; i++) out[i] = (mx - out[i]) / mx;
    free(col); free(row);
}

/* numpy's pairwise summation of a contiguous float64 vector (what ndarray.sum() does on the
 * boolean-indexed score values, maskprocess.py:134). */
static double orc_pairwise_sum(const double* a, size_t n)
{
    if (n < 8) {
        double r = 0.;
        for (size_t i = 0; i < n; i++) r += a[i];
        return r;
    } else if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        size_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return orc_pairwise_sum(a, n2) + orc_pairwise_sum(a + n2, n - n2);
    }
}

typedef struct { int x, y; } orc_pt;
typedef struct { orc_pt* p; size_t n, cap; } orc_chain;

static void orc_chain_push(orc_chain* c, int x, int y)
{
    if (c->n == c->cap) { c->cap = c->cap ? c->cap * 2 : 256; c->p = (orc_pt*)realloc(c->p, c->cap * sizeof(orc_pt)); }
    c->p[c->n].x = x; c->p[c->n].y = y; c->n++;
}

/* Border following of cv2.findContours(RETR_LIST) (Suzuki-Abe as implemented by OpenCV's
 * icvFetchContour): directions 0..7 = E, NE, N, NW, W, SW, S, SE; border pixels are re-labelled
 * 2, or -126 when the pixel to the east was examined and is zero.  Every visited pixel is pushed
 * (CHAIN_APPROX_SIMPLE only drops collinear ones, which changes neither area nor fill). */
static void orc_follow(int8_t* img, int step, int x, int y, int is_hole, orc_chain* ch)
{
    static const int dxs[8] = {1, 1, 0, -1, -1, -1, 0, 1}, dys[8] = {0, -1, -1, -1, 0, 1, 1, 1};
    int delta[16];
    for (int k = 0; k < 8; k++) delta[k] = delta[k + 8] = dys[k] * step + dxs[k];
    int8_t* i0 = img + (size_t)y * step + x;
    int8_t *i1, *i3, *i4 = 0;
    int s_end, s;
    s_end = s = is_hole ? 0 : 4;
    do {
        s = (s - 1) & 7;
        i1 = i0 + delta[s];
    } while (*i1 == 0 && s != s_end);
    ch->n = 0;
    if (s == s_end) {                      /* isolated pixel */
        *i0 = (int8_t)(2 | -128);
        orc_chain_push(ch, x, y);
        return;
    }
    i3 = i0;
    int px = x, py = y;
    for (;;) {
        s_end = s;
        for (;;) {
            i4 = i3 + delta[++s];
            if (*i4 != 0) break;
        }
        s &= 7;
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (int8_t)(2 | -128);
        else if (*i3 == 1) *i3 = 2;
        orc_chain_push(ch, px, py);
        px += dxs[s]; py += dys[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* cv2.contourArea: |sum(x_prev * y - y_prev * x)| / 2 over the closed polygon. */
static double orc_chain_area(const orc_chain* c)
{
    double a = 0;
    if (c->n == 0) return 0;
    orc_pt prev = c->p[c->n - 1];
    for (size_t i = 0; i < c->n; i++) {
        a += (double)prev.x * c->p[i].y - (double)prev.y * c->p[i].x;
        prev = c->p[i];
    }
    return fabs(a * 0.5);
}

/* cv2.drawContours(zeros, contours, i, 255, cv2.FILLED): the polygon's own pixels plus every pixel
 * whose centre is inside it by the even-odd rule (edges are horizontal, vertical or 45 degrees, so
 * scanline intersections are integers).  obj is H x W, set to 1 inside. */
static void orc_fill(const orc_chain* c, int H, int W, uint8_t* obj)
{
    uint8_t* par = (uint8_t*)calloc((size_t)H * W, 1);
    for (size_t i = 0; i < c->n; i++) {
        orc_pt a = c->p[i], b = c->p[(i + 1) % c->n];
        obj[(size_t)a.y * W + a.x] = 1;
        if (a.y != b.y) {                  /* half-open in y: toggle on the upper row of the move */
            orc_pt top = a.y < b.y ? a : b;
            par[(size_t)top.y * W + top.x] ^= 1;
        }
    }
    for (int y = 0; y < H; y++) {
        int inside = 0;
        for (int x = 0; x < W; x++) {
            if (par[(size_t)y * W + x]) inside ^= 1;
            if (inside) obj[(size_t)y * W + x] = 1;
        }
    }
    free(par);
}

/* unscreen/utils/maskprocess.py:77-152 remove_invalid_objects, with build_score_map (:181-189) choosing the
 * centre by orientation before the call.  alpha is modified in place (and is what the caller gets back).
 *  :123-124 findContours(alpha, RETR_LIST, CHAIN_APPROX_SIMPLE): non-zero = object; outer AND hole borders
 *  :130-132 skip contourArea < 100
 *  :134-135 saliency = sum(score_map[filled]) / (h*w)
 *  :137-138 consensus = mean(segmask[filled]) / 255   (segmask defaults to alpha itself, :115-116)
 *  :142-146 keep if (sal > sal_thr and cons > cons_thr) or sal > 10 * sal_thr
 *  :151     alpha[valid == 0] = 0
 * Returns the number of contours found; stats (optional, 4 doubles per contour, up to max_stats):
 * area, saliency, consensus, kept. */
ORC_API int orc_remove_invalid_objects(uint8_t* alpha, const uint8_t* segmask, int H, int W, double cy, double cx,
                                       double sal_thr, double cons_thr, double* stats, int max_stats)
{
    size_t n = (size_t)H * W;
    const uint8_t* seg = segmask ? segmask : alpha;
    uint8_t* seg_copy = NULL;
    if (!segmask) { seg_copy = (uint8_t*)malloc(n); memcpy(seg_copy, alpha, n); seg = seg_copy; }
    double* score = (double*)malloc(sizeof(double) * n);
    orc_score_map(H, W, cy, cx, score);
    int step = W + 2;
    int8_t* img = (int8_t*)calloc((size_t)(H + 2) * step, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) img[(size_t)(y + 1) * step + x + 1] = alpha[(size_t)y * W + x] != 0;
    uint8_t* valid = (uint8_t*)calloc(n, 1);
    uint8_t* obj = (uint8_t*)malloc(n);
    double* vals = (double*)malloc(sizeof(double) * n);
    orc_chain ch = {0, 0, 0};
    int count = 0;
    /* OpenCV scans rows 1..H and columns 1..W of the zero-padded image (the last padded column is
     * never a scan position, so an object touching the right edge cannot open a "hole"). */
    for (int y = 1; y <= H; y++) {
        int prev = 0;
        for (int x = 1; x <= W; x++) {
            int p = img[(size_t)y * step + x];
            if (p != prev) {
                int is_hole = 0, start = 0;
                if (prev == 0 && p == 1) start = 1;
                else if (p == 0 && prev >= 1) { is_hole = 1; start = 1; }
                if (start) {
                    orc_follow(img, step, x - is_hole, y, is_hole, &ch);
                    for (size_t i = 0; i < ch.n; i++) { ch.p[i].x -= 1; ch.p[i].y -= 1; }
                    double area = orc_chain_area(&ch);
                    double sal = 0, cons = 0; int kept = 0;
                    if (area >= 100) {
                        memset(obj, 0, n);
                        orc_fill(&ch, H, W, obj);
                        size_t m = 0; double ssum = 0;
                        for (size_t i = 0; i < n; i++)
                            if (obj[i]) { vals[m++] = score[i]; ssum += seg[i]; }
                        sal = orc_pairwise_sum(vals, m) / (double)(H * W);
                        cons = (ssum / (double)m) / 255.;
                        if ((sal > sal_thr && cons > cons_thr) || (sal > sal_thr * 10)) {
                            kept = 1;
                            for (size_t i = 0; i < n; i++) if (obj[i]) valid[i] = 255;
                        }
                    }
                    if (stats && count < max_stats) {
                        stats[4 * count] = area; stats[4 * count + 1] = sal;
                        stats[4 * count + 2] = cons; stats[4 * count + 3] = kept;
                    }
                    count++;
                    p = img[(size_t)y * step + x];
                }
            }
            prev = p;
        }
    }
    for (size_t i = 0; i < n; i++) if (!valid[i]) alpha[i] = 0;
    free(ch.p); free(vals); free(obj); free(valid); free(img); free(score); free(seg_copy);
    return count;
}

/* The three OpenCV contour calls of maskprocess.py:123-149 as separate entry points (same border follower, area and fill as
 * orc_remove_invalid_objects above), so that the reference's own Python can be run against them (tests/golden/cv2_shim.py).
 * orc_find_contours: image != 0 is object; every contour's points (x, y pairs, every visited border pixel) are appended to
 * pts (capacity max_pts points), lens[i] = number of points of contour i, in the order the raster scan meets them.
 * Returns the number of contours, or -1 when a capacity is exceeded. */
ORC_API int orc_find_contours(const uint8_t* image, int H, int W, int* pts, long long max_pts, int* lens, int max_contours)
{
    int step = W + 2;
    int8_t* img = (int8_t*)calloc((size_t)(H + 2) * step, 1);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) img[(size_t)(y + 1) * step + x + 1] = image[(size_t)y * W + x] != 0;
    orc_chain ch = {0, 0, 0};
    int count = 0;
    long long used = 0;
    for (int y = 1; y <= H && count >= 0; y++) {
        int prev = 0;
        for (int x = 1; x <= W; x++) {
            int p = img[(size_t)y * step + x];
            if (p != prev) {
                int is_hole = 0, start = 0;
                if (prev == 0 && p == 1) start = 1;
                else if (p == 0 && prev >= 1) { is_hole = 1; start = 1; }
                if (start) {
                    orc_follow(img, step, x - is_hole, y, is_hole, &ch);
                    if (count >= max_contours || used + (long long)ch.n > max_pts) { count = -1; break; }
                    for (size_t i = 0; i < ch.n; i++) { pts[2 * (used + i)] = ch.p[i].x - 1; pts[2 * (used + i) + 1] = ch.p[i].y - 1; }
                    lens[count++] = (int)ch.n;
                    used += (long long)ch.n;
                    p = img[(size_t)y * step + x];
                }
            }
            prev = p;
        }
    }
    free(ch.p); free(img);
    return count;
}

ORC_API double orc_contour_area(const int* pts, int n)
{
    orc_chain c = {(orc_pt*)pts, (size_t)n, (size_t)n};
    return orc_chain_area(&c);
}

/* cv2.drawContours(img, [contour], 0, value, cv2.FILLED) on a one-channel image. */
ORC_API void orc_draw_contour_filled(const int* pts, int n, int H, int W, uint8_t* img, int value)
{
    orc_chain c = {(orc_pt*)pts, (size_t)n, (size_t)n};
    uint8_t* obj = (uint8_t*)calloc((size_t)H * W, 1);
    orc_fill(&c, H, W, obj);
    for (size_t i = 0; i < (size_t)H * W; i++) if (obj[i]) img[i] = (uint8_t)value;
    free(obj);
}

/* ------------------------------------------------------------------------------------------
 * HumanSegAgent pre / post processing (unscreen/binseg/agent_human.py:262-312)
 * ---------------------------------------------------------------------------------------- */

/* cv::getAffineTransform: the 6x6 system [x y 1 0 0 0; 0 0 0 x y 1] solved by cv::solve(DECOMP_LU) in double
 * (partial pivoting, d = -1/pivot elimination, back substitution).  src / dst: 3 points (x, y). */
static void orc_get_affine_transform(const double* src, const double* dst, double* M)
{
    double a[36], b[6];
    for (int i = 0; i < 3; i++) {
        int j = i * 12, k = i * 12 + 6;
        a[j] = a[k + 3] = src[2 * i];
        a[j + 1] = a[k + 4] = src[2 * i + 1];
        a[j + 2] = a[k + 5] = 1;
        a[j + 3] = a[j + 4] = a[j + 5] = 0;
        a[k] = a[k + 1] = a[k + 2] = 0;
        b[i * 2] = dst[2 * i];
        b[i * 2 + 1] = dst[2 * i + 1];
    }
    const int m = 6;
    for (int i = 0; i < m; i++) {
        int k = i;
        for (int j = i + 1; j < m; j++) if (fabs(a[j * m + i]) > fabs(a[k * m + i])) k = j;
        if (k != i) {
            for (int j = i; j < m; j++) { double t = a[i * m + j]; a[i * m + j] = a[k * m + j]; a[k * m + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / a[i * m + i];
        for (int j = i + 1; j < m; j++) {
            double alpha = a[j * m + i] * d;
            for (k = i + 1; k < m; k++) a[j * m + k] += alpha * a[i * m + k];
            b[j] += alpha * b[i];
        }
    }
    for (int i = m - 1; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < m; k++) s -= a[i * m + k] * b[k];
        b[i] = s / a[i * m + i];
    }
    for (int i = 0; i < 6; i++) M[i] = b[i];
}

/* agent_human.py:247-260 _box2cs / _xywh2cs with box [0, 0, w-1, h-1] and aspect ratio 1, then
 * :91-125 get_affine_transform(center, scale, rot = 0, output_size = (in_h, in_w), inv): the two point triangles
 * (all coordinates are integers or half-integers, exact in float32) and cv2.getAffineTransform.
 * M maps image -> network input when inv = 0, network input -> image when inv = 1. */
ORC_API void orc_humanseg_affine(int w, int h, int in_w, int in_h, int inv, double* M)
{
    float cx = (float)(0 + (w - 1) * 0.5), cy = (float)(0 + (h - 1) * 0.5);
    double bw = w - 1, bh = h - 1;
    if (bw > 1.0 * bh) bh = bw * 1.0 / 1.0; else if (bw < 1.0 * bh) bw = bh * 1.0;
    float src_w = (float)bw;
    (void)bh;
    double src_dir1 = (double)src_w * -0.5;                       /* rot = 0: src_dir = [0, -src_w / 2] */
    float dst_dir1 = (float)((in_w - 1) * -0.5);
    float s0x = cx, s0y = cy;
    float s1x = (float)((double)cx + 0.0), s1y = (float)((double)cy + src_dir1);
    float d0x = (float)((in_w - 1) * 0.5), d0y = (float)((in_h - 1) * 0.5);
    float d1x = d0x + 0.f, d1y = d0y + dst_dir1;
    /* get_3rd_point(a, b) = b + (-(a - b).y, (a - b).x) */
    float s2x = s1x + -(s0y - s1y), s2y = s1y + (s0x - s1x);
    float d2x = d1x + -(d0y - d1y), d2y = d1y + (d0x - d1x);
    double src[6] = {s0x, s0y, s1x, s1y, s2x, s2y}, dst[6] = {d0x, d0y, d1x, d1y, d2x, d2y};
    if (inv) orc_get_affine_transform(dst, src, M); else orc_get_affine_transform(src, dst, M);
}

/* cv::warpAffine without WARP_INVERSE_MAP first inverts the 2x3 matrix (in place, double). */
static void orc_invert_affine(const double* Min, double* M)
{
    for (int i = 0; i < 6; i++) M[i] = Min[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    double b1 = -M[0] * M[2] - M[1] * M[5];
    double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

/* Fixed-point source coordinate of destination pixel (x, y): 10-bit products, 5 fractional bits kept
 * (cv::WarpAffineInvoker, AB_BITS = 10, INTER_BITS = 5, round_delta = 16). */
static void orc_warp_coord(const double* M, int x, int y, int* sx, int* sy, int* fx, int* fy)
{
    int X0 = orc_round_i((M[1] * y + M[2]) * 1024) + 16, Y0 = orc_round_i((M[4] * y + M[5]) * 1024) + 16;
    int X = (X0 + orc_round_i(M[0] * x * 1024)) >> 5, Y = (Y0 + orc_round_i(M[3] * x * 1024)) >> 5;
    int ix = X >> 5, iy = Y >> 5;
    *sx = ix < -32768 ? -32768 : (ix > 32767 ? 32767 : ix);
    *sy = iy < -32768 ? -32768 : (iy > 32767 ? 32767 : iy);
    *fx = X & 31; *fy = Y & 31;
}

/* cv2.warpAffine(img, M, (dw, dh), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT, borderValue=0) on uint8:
 * 15-bit weights (32-fx)(32-fy)*32 ..., (sum + 2^14) >> 15; taps outside the image read 0. */
ORC_API void orc_warp_affine_u8(const uint8_t* src, int sh, int sw, int cn, const double* Mfwd, uint8_t* dst, int dh, int dw)
{
    double M[6];
    orc_invert_affine(Mfwd, M);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sx, sy, fx, fy;
            orc_warp_coord(M, x, y, &sx, &sy, &fx, &fy);
            int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            for (int c = 0; c < cn; c++) {
                int v[4];
                for (int t = 0; t < 4; t++) {
                    int xx = sx + (t & 1), yy = sy + (t >> 1);
                    v[t] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[((size_t)yy * sw + xx) * cn + c] : 0;
                }
                int r = (v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15;
                dst[((size_t)y * dw + x) * cn + c] = orc_sat_u8(r);
            }
        }
}

/* The float32 form (one channel of interleaved data): weights (1 - fy/32)(1 - fx/32) ... as float products,
 * v0*w0 + v1*w1 + v2*w2 + v3*w3 summed left to right. */
ORC_API void orc_warp_affine_f32(const float* src, int sh, int sw, int cn, const double* Mfwd, float* dst, int dh, int dw)
{
    double M[6];
    orc_invert_affine(Mfwd, M);
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            int sx, sy, fx, fy;
            orc_warp_coord(M, x, y, &sx, &sy, &fx, &fy);
            float ax = fx * (1.f / 32), ay = fy * (1.f / 32);
            float wx0 = 1.f - ax, wx1 = ax, wy0 = 1.f - ay, wy1 = ay;
            float w[4] = {wy0 * wx0, wy0 * wx1, wy1 * wx0, wy1 * wx1};
            for (int c = 0; c < cn; c++) {
                float v[4];
                for (int t = 0; t < 4; t++) {
                    int xx = sx + (t & 1), yy = sy + (t >> 1);
                    v[t] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[((size_t)yy * sw + xx) * cn + c] : 0.f;
                }
                float r = v[0] * w[0];
                r = r + v[1] * w[1];
                r = r + v[2] * w[2];
                r = r + v[3] * w[3];
                dst[((size_t)y * dw + x) * cn + c] = r;
            }
        }
}

/* torch.nn.Upsample(size, mode='bilinear', align_corners=True) (agent_human.py:299-300), float32:
 * src = dst * (in-1)/(out-1) in float, i = (int)src, lambda = src - i,
 * val = h0*(w0*v00 + w1*v01) + h1*(w0*v10 + w1*v11).  src: (C, ih, iw) planes with plane pitch `cstride`
 * elements and pixel pitch `pstride` (NCHW: cstride = ih*iw, pstride = 1; NHWC: cstride = 1, pstride = C_buf).
 * dst: (oh, ow, C) interleaved. */
ORC_API void orc_upsample_ac_f32(const float* src, int C, int ih, int iw, size_t cstride, size_t pstride, float* dst, int oh, int ow)
{
    float rh = oh > 1 ? (float)(ih - 1) / (oh - 1) : 0.f, rw = ow > 1 ? (float)(iw - 1) / (ow - 1) : 0.f;
    for (int y = 0; y < oh; y++) {
        float h1r = rh * y;
        int h1 = (int)h1r, h1p = h1 < ih - 1 ? 1 : 0;
        float h1l = h1r - h1, h0l = 1.f - h1l;
        for (int x = 0; x < ow; x++) {
            float w1r = rw * x;
            int w1 = (int)w1r, w1p = w1 < iw - 1 ? 1 : 0;
            float w1l = w1r - w1, w0l = 1.f - w1l;
            for (int c = 0; c < C; c++) {
                const float* p = src + (size_t)c * cstride;
                float v00 = p[((size_t)h1 * iw + w1) * pstride], v01 = p[((size_t)h1 * iw + w1 + w1p) * pstride];
                float v10 = p[((size_t)(h1 + h1p) * iw + w1) * pstride], v11 = p[((size_t)(h1 + h1p) * iw + w1 + w1p) * pstride];
                float top = w0l * v00 + w1l * v01, bot = w0l * v10 + w1l * v11;
                dst[((size_t)y * ow + x) * C + c] = h0l * top + h1l * bot;
            }
        }
    }
}

/* agent_human.py:277-294: warp the frame into the in_h x in_w network input (affine about the image centre,
 * LINEAR, zero border), imnormalize (BGR->RGB, /255, -mean, *1/std), NCHW float32. */
ORC_API void orc_humanseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, float* out)
{
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    double M[6];
    orc_humanseg_affine(W, H, in_w, in_h, 0, M);
    uint8_t* warped = (uint8_t*)malloc((size_t)in_h * in_w * 3);
    orc_warp_affine_u8(frame, H, W, 3, M, warped, in_h, in_w);
    size_t n = (size_t)in_h * in_w;
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) {
            float x = (float)warped[3 * i + (2 - c)] / 255.f;
            x = x - (float)mean[c];
            out[(size_t)c * n + i] = x * (float)(1 / std[c]);
        }
    free(warped);
}

/* agent_human.py:296-311 + transform_logits :58-73: logits (C, lh, lw) -> bilinear (align_corners) to in_h x in_w ->
 * per-channel inverse affine warp to the W x H frame -> argmax over channels (first maximum wins) -> 255 where > 0. */
ORC_API void orc_humanseg_post(const float* logits, int C, int lh, int lw, size_t cstride, size_t pstride,
                               int in_h, int in_w, int H, int W, uint8_t* out)
{
    double M[6];
    orc_humanseg_affine(W, H, in_w, in_h, 1, M);
    float* up = (float*)malloc(sizeof(float) * (size_t)in_h * in_w * C);
    float* full = (float*)malloc(sizeof(float) * (size_t)H * W * C);
    orc_upsample_ac_f32(logits, C, lh, lw, cstride, pstride, up, in_h, in_w);
    orc_warp_affine_f32(up, in_h, in_w, C, M, full, H, W);
    for (size_t i = 0; i < (size_t)H * W; i++) {
        int best = 0;
        for (int c = 1; c < C; c++) if (full[i * C + c] > full[i * C + best]) best = c;
        out[i] = best > 0 ? 255 : 0;
    }
    free(up); free(full);
}

/* ------------------------------------------------------------------------------------------
 * one bg_step stage-3 frame, pixel stages only (tools/unscreen/bg_offline.py:145-176); the
 * matting network runs between pre and post and is supplied by the caller.
 * ---------------------------------------------------------------------------------------- */
ORC_API const char* orc_version(void) { return "usc-oracle 1 (OpenCV-4.x generic-path restatement, parity unpinned)"; }

/* ============================================================================================
 * green mode (tools/unscreen/green.py:70-126): colour filtering, colour correction, harmonization helpers
 * ============================================================================================ */

/* cv2.COLOR_BGR2Lab on uint8 (unscreen/utils/imgprocess.py:281,283 color_correct; unscreen/harmonization/agent.py:63-64).
 * OpenCV's RGB2Lab_b integer path: lab_shift 12, gamma_shift 3; sRGB gamma table (256 x u16, scale 255*8), cube-root table
 * (256*3/2*8 = 3072 x u16, scale 2^15), XYZ coefficients = round(4096 * sRGB2XYZ_D65 / D65 white point),
 * L = descale(296*fY - Lshift), a = descale(500*(fX-fY) + 128<<15), b = descale(200*(fY-fZ) + 128<<15), descale by 15 bits.
 * PARITY UNPINNED: OpenCV 4 builds the two tables with its softfloat pow / cbrt; here libm powf / cbrtf (float, then
 * round-half-even).  A table entry could differ by 1 where softfloat and libm differ in the last ulp next to a .5 tie. */
static uint16_t orc_srgb_gamma_b[256], orc_lab_cbrt_b[3072];
static int orc_lab_coeffs[9], orc_lab_ready = 0;
ORC_API void orc_lab_tables(uint16_t* gamma256, uint16_t* cbrt3072, int* coeffs9)
{
    static const double xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double white[3] = {0.950456, 1., 1.088754};
    for (int i = 0; i < 256; i++) {
        float x = (float)i / 255.f;
        float g = x <= 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f);
        gamma256[i] = (uint16_t)orc_round_i((double)(2040.f * g));
    }
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    for (int i = 0; i < 3072; i++) {
        float x = (float)i / 2040.f;
        float v = x < lthresh ? fmaf(x, lscale, lbias) : cbrtf(x);
        cbrt3072[i] = (uint16_t)orc_round_i((double)(32768.f * v));
    }
    for (int i = 0; i < 3; i++) {                       /* BGR source order: blueIdx = 0 */
        coeffs9[i * 3 + 2] = orc_round_i(4096. * xyz[i * 3 + 0] / white[i]);
        coeffs9[i * 3 + 1] = orc_round_i(4096. * xyz[i * 3 + 1] / white[i]);
        coeffs9[i * 3 + 0] = orc_round_i(4096. * xyz[i * 3 + 2] / white[i]);
    }
}
static inline int orc_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
static inline void orc_bgr2lab_px(const uint8_t* s, uint8_t* d)
{
    if (!orc_lab_ready) { orc_lab_tables(orc_srgb_gamma_b, orc_lab_cbrt_b, orc_lab_coeffs); orc_lab_ready = 1; }
    const int* C = orc_lab_coeffs;
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    int c0 = orc_srgb_gamma_b[s[0]], c1 = orc_srgb_gamma_b[s[1]], c2 = orc_srgb_gamma_b[s[2]];
    int fX = orc_lab_cbrt_b[orc_descale(c0 * C[0] + c1 * C[1] + c2 * C[2], 12)];
    int fY = orc_lab_cbrt_b[orc_descale(c0 * C[3] + c1 * C[4] + c2 * C[5], 12)];
    int fZ = orc_lab_cbrt_b[orc_descale(c0 * C[6] + c1 * C[7] + c2 * C[8], 12)];
    d[0] = orc_sat_u8(orc_descale(Lscale * fY + Lshift, 15));
    d[1] = orc_sat_u8(orc_descale(500 * (fX - fY) + 128 * (1 << 15), 15));
    d[2] = orc_sat_u8(orc_descale(200 * (fY - fZ) + 128 * (1 << 15), 15));
}
ORC_API void orc_bgr2lab_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    for (size_t i = 0; i < n; i++) orc_bgr2lab_px(src + 3 * i, dst + 3 * i);
}

/* cv2.COLOR_LAB2BGR on uint8 (unscreen/harmonization/agent.py:73 foreground_toning).  OpenCV 4's Lab2RGB_b runs its integer
 * path (Lab2RGBinteger, color_lab.cpp; enableBitExactness): BASE = 2^14,
 *   (y, ify) = LabToYF_b[L]: L <= 20: y = round(L*BASE*180 / (17*29^3)), ify = round(BASE*(16/116 + L*5/(3*17*29)));
 *                            else fy = L*100*BASE/(255*116) + 16*BASE/116 (float), ify = round(fy), y = round(fy^3 / BASE^2)
 *   adiv = ((5*a*53687 + 128) >> 13) - 128*BASE/500 ; bdiv = ((b*41943 + 16) >> 9) - 128*BASE/200 + 1
 *   x = abToXZ(ify + adiv), z = abToXZ(ify - bdiv): v <= 3390: v*108/841 - BASE*16/116*108/841 (C integer division),
 *                                                     else v*v/BASE*v/BASE
 *   (r, g, b) = descale(C . (x, y, z), 14) with C = round(4096 * XYZ2sRGB_D65 * white), clamped to [0, 4095], through the
 *   4096-entry inverse sRGB gamma table round(255 * invgamma(i / 4096)).
 * PARITY UNPINNED: restated from the published source; the float table entries use libm (OpenCV: its softfloat). */
static uint16_t orc_lab_yf_b[512], orc_lab_invgamma_b[4096];
static int orc_lab_inv_coeffs[9], orc_lab_inv_ready = 0;
ORC_API void orc_lab_inv_tables(uint16_t* yf512, uint16_t* invgamma4096, int* coeffs9)
{
    static const double m[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double white[3] = {0.950456, 1., 1.088754};
    const int BASE = 1 << 14;
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) {
            y = orc_round_i((double)((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29)));
            ify = orc_round_i((double)((float)BASE * (16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29))));
        } else {
            float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / 116.f;
            ify = orc_round_i((double)fy);
            y = orc_round_i((double)(fy * fy * fy / (float)((double)BASE * BASE)));
        }
        yf512[2 * i] = (uint16_t)y;
        yf512[2 * i + 1] = (uint16_t)ify;
    }
    for (int i = 0; i < 4096; i++) {
        float x = (1.f / 4096.f) * (float)i;
        double xd = (double)x;
        float v = (float)(xd <= 0.0031308 ? xd * 12.92 : pow(xd, 1. / 2.4) * 1.055 - 0.055);
        invgamma4096[i] = (uint16_t)orc_round_i((double)(255.f * v));
    }
    for (int i = 0; i < 3; i++) {            /* rows of coeffs9: B, G, R (blueIdx = 0); column i scaled by the white point */
        coeffs9[0 + i] = orc_round_i(4096. * m[6 + i] * white[i]);
        coeffs9[3 + i] = orc_round_i(4096. * m[3 + i] * white[i]);
        coeffs9[6 + i] = orc_round_i(4096. * m[0 + i] * white[i]);
    }
}
static inline int orc_ab_to_xz(int v)
{
    const int BASE = 1 << 14;
    if (v <= 3390) return v * 108 / 841 - BASE * 16 / 116 * 108 / 841;
    return v * v / BASE * v / BASE;
}
static inline void orc_lab2bgr_px(const uint8_t* s, uint8_t* d)
{
    if (!orc_lab_inv_ready) { orc_lab_inv_tables(orc_lab_yf_b, orc_lab_invgamma_b, orc_lab_inv_coeffs); orc_lab_inv_ready = 1; }
    const int BASE = 1 << 14;
    int y = orc_lab_yf_b[2 * s[0]], ify = orc_lab_yf_b[2 * s[0] + 1];
    int adiv = ((5 * s[1] * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
    int bdiv = ((s[2] * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
    int x = orc_ab_to_xz(ify + adiv), z = orc_ab_to_xz(ify - bdiv);
    const int* C = orc_lab_inv_coeffs;
    for (int c = 0; c < 3; c++) {
        int v = orc_descale(C[3 * c] * x + C[3 * c + 1] * y + C[3 * c + 2] * z, 14);
        v = v < 0 ? 0 : (v > 4095 ? 4095 : v);
        d[c] = (uint8_t)orc_lab_invgamma_b[v];
    }
}
ORC_API void orc_lab2bgr_u8(const uint8_t* src, uint8_t* dst, size_t n)
{
    for (size_t i = 0; i < n; i++) orc_lab2bgr_px(src + 3 * i, dst + 3 * i);
}

/* unscreen/colorfiltering/agent.py:201-230 get_prob_by_gmm for the 256 possible sample values: float32 throughout,
 * x = (s - mean)/std ; y = 1/(std*sqrt(2 pi)) * exp(-0.5 * x*x) ; prob = sum_k w_k * y_k (torch.mm; summed here in
 * component order without fma - the BLAS order is unspecified, agreement is to float rounding). */
ORC_API void orc_gmm_lut(const double* means, const double* stds, const double* weights, int K, float* lut256)
{
    const float s2pi = (float)sqrt(2 * 3.141592653589793);
    for (int v = 0; v < 256; v++) {
        float acc = 0.f;
        for (int k = 0; k < K; k++) {
            float m = (float)means[k], sd = (float)stds[k], w = (float)weights[k];
            float x = ((float)v - m) / sd;
            float y = (1.f / (sd * s2pi)) * expf(-0.5f * (x * x));
            acc = acc + w * y;
        }
        lut256[v] = acc;
    }
}

/* unscreen/colorfiltering/agent.py:232-257 get_alpha_by_gmm on an HSV image (n pixels), per-channel likelihood tables
 * lut_bg / lut_fg [3][256]: prob products in f32, cube root (torch.pow(., 1/3.): restated as the correctly rounded
 * float of the double cube root), fg/(bg+fg+1e-6), clip(prob*255, 0, 255) truncated to uint8. */
ORC_API void orc_gmm_alpha(const uint8_t* hsv, size_t n, const float* lut_bg, const float* lut_fg, uint8_t* alpha, float* prob_out)
{
    for (size_t i = 0; i < n; i++) {
        float bg = 1.f, fg = 1.f;
        for (int c = 0; c < 3; c++) {
            bg = bg * lut_bg[c * 256 + hsv[3 * i + c]];
            fg = fg * lut_fg[c * 256 + hsv[3 * i + c]];
        }
        bg = (float)cbrt((double)bg);
        fg = (float)cbrt((double)fg);
        float p = fg / (bg + fg + 1e-6f);
        if (prob_out) prob_out[i] = p;
        float a = p * 255.f;
        a = a < 0.f ? 0.f : (a > 255.f ? 255.f : a);
        alpha[i] = (uint8_t)a;
    }
}

/* unscreen/colorfiltering/agent.py:259-280 postprocess, first half: score_thr = mean(alpha[(alpha>128)&(mask>0)]) * thr_ratio
 * in float64 (an empty set gives nan: nothing is zeroed); alpha[alpha < score_thr] = 0.  In place. */
ORC_API void orc_cf_threshold(uint8_t* alpha, const uint8_t* mask, size_t n, double thr_ratio)
{
    unsigned long long sum = 0, cnt = 0;
    for (size_t i = 0; i < n; i++) if (alpha[i] > 128 && mask[i] > 0) { sum += alpha[i]; cnt++; }
    double thr = ((double)sum / (double)cnt) * thr_ratio;        /* 0/0 = nan */
    for (size_t i = 0; i < n; i++) if ((double)alpha[i] < thr) alpha[i] = 0;
}

/* torch.nn.functional.interpolate(x, (H, W)) default mode='nearest' source index (ATen UpSample.h nearest_idx). */
static inline int orc_torch_nearest_idx(int o, int in, int out)
{
    if (out == in) return o;
    if (out == 2 * in) return o >> 1;
    float scale = (float)in / (float)out;
    int s = (int)floorf((float)o * scale);
    return s < in - 1 ? s : in - 1;
}

/* unscreen/utils/imgprocess.py:263-300 color_correct.  Working size (th, tw) = get_target_size(H, W, long_side);
 * Lab distance of the (bilinear-resized) frame to the background colour on the a/b channels in float32, min-max
 * normalised, square-rooted until the mean over {alpha_r > 0, dist > 0} reaches mean_exp (the mean is accumulated in
 * float64 here; torch's float32 cascade sum could decide differently only when the mean is within ~1e-6 of mean_exp),
 * zeroed where the resized alpha is 0, nearest-upsampled (torch indices) and multiplied into alpha, truncated. */
ORC_API int orc_color_correct(const uint8_t* img, const uint8_t* alpha, int H, int W, const uint8_t* bg_color, int long_side,
                              double mean_exp, uint8_t* out, int* n_sqrt_out)
{
    int th, tw;
    orc_get_target_size(H, W, long_side, 1, &th, &tw);
    size_t n = (size_t)th * tw;
    uint8_t* small = (uint8_t*)malloc(n * 3);
    uint8_t* ar = (uint8_t*)malloc(n);
    float* dist = (float*)malloc(n * sizeof(float));
    orc_resize_linear_u8(img, H, W, 3, small, th, tw);
    orc_resize_linear_u8(alpha, H, W, 1, ar, th, tw);
    uint8_t bl[3];
    orc_bgr2lab_px(bg_color, bl);
    float ba = (float)bl[1] / 255.f, bb = (float)bl[2] / 255.f;
    float dmin = INFINITY, dmax = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        uint8_t lab[3];
        orc_bgr2lab_px(small + 3 * i, lab);
        float da = (float)lab[1] / 255.f - ba, db = (float)lab[2] / 255.f - bb;
        float d = sqrtf(da * da + db * db);
        dist[i] = d;
        if (d < dmin) dmin = d;
        if (d > dmax) dmax = d;
    }
    float range = dmax - dmin;
    for (size_t i = 0; i < n; i++) dist[i] = (dist[i] - dmin) / range;
    int iters = 0;
    for (;; iters++) {
        double s = 0; size_t c = 0;
        for (size_t i = 0; i < n; i++) if (ar[i] > 0 && dist[i] > 0.f) { s += dist[i]; c++; }
        float mean = (float)(s / (double)c);
        if (!(mean < (float)mean_exp)) break;
        for (size_t i = 0; i < n; i++) dist[i] = sqrtf(dist[i]);
        if (iters > 64) break;
    }
    if (n_sqrt_out) *n_sqrt_out = iters;
    for (size_t i = 0; i < n; i++) if (ar[i] == 0) dist[i] = 0.f;
    for (int y = 0; y < H; y++) {
        int sy = orc_torch_nearest_idx(y, th, H);
        for (int x = 0; x < W; x++) {
            int sx = orc_torch_nearest_idx(x, tw, W);
            float v = (float)alpha[(size_t)y * W + x] * dist[(size_t)sy * tw + sx];
            out[(size_t)y * W + x] = (uint8_t)v;        /* numpy astype(uint8) of a value in [0, 255] (nan -> 0 on x86) */
        }
    }
    free(small); free(ar); free(dist);
    return 0;
}

/* cv2.boxFilter(img, -1, (k, k)) on uint8 (unscreen/harmonization/agent.py:91-92, 110-111): normalised box, anchor centre,
 * BORDER_REFLECT_101; 8-bit sums up to 256 taps go through a 16-bit column sum with the integer reciprocal
 * D = ((s + divDelta) * divScale) >> 16.  PARITY UNPINNED (restated from the OpenCV 4 generic path; an IPP build may differ). */
ORC_API void orc_box_filter_u8(const uint8_t* src, int H, int W, int cn, int k, uint8_t* dst)
{
    int d = k * k, a = k / 2;
    double scalef = 65536.0 / d;
    int divScale = (int)floor(scalef), divDelta = d / 2;
    if (scalef - divScale < 0.5) divDelta++; else divScale++;
    if (k == 1) { divScale = 65536; divDelta = 0; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            for (int c = 0; c < cn; c++) {
                int s = 0;
                for (int j = -a; j < k - a; j++) {
                    int yy = y + j;
                    if (H == 1) yy = 0; else { while (yy < 0 || yy >= H) yy = yy < 0 ? -yy : 2 * (H - 1) - yy; }
                    for (int i = -a; i < k - a; i++) {
                        int xx = x + i;
                        if (W == 1) xx = 0; else { while (xx < 0 || xx >= W) xx = xx < 0 ? -xx : 2 * (W - 1) - xx; }
                        s += src[((size_t)yy * W + xx) * cn + c];
                    }
                }
                dst[((size_t)y * W + x) * cn + c] = (uint8_t)(((unsigned)(s + divDelta) * (unsigned)divScale) >> 16);
            }
}

/* ============================================================================================
 * SegAgent (unscreen/binseg/agent.py): resize/pad, normalise, overlapping crops (+ mirrored copies), network,
 * softmax, float64 merge, inverse pad-resize, argmax
 * ============================================================================================ */

/* unscreen/binseg/agent.py:70-100 SegAgent.get_target_size: long side -> input_long_side, short side scaled (rounded up to
 * `division`), then each side raised to the crop size. */
ORC_API void orc_segagent_target_size(int h, int w, int long_side, int division, int crop_h, int crop_w, int* th, int* tw)
{
    int t_h, t_w;
    if (h > w) {
        t_h = long_side;
        t_w = (int)((double)long_side * w / h);
        if (t_w % division != 0) t_w = (t_w / division + 1) * division;
    } else {
        t_w = long_side;
        t_h = (int)((double)long_side * h / w);
        if (t_h % division != 0) t_h = (t_h / division + 1) * division;
    }
    if (crop_h > t_h) t_h = crop_h;
    if (crop_w > t_w) t_w = crop_w;
    *th = t_h; *tw = t_w;
}

/* unscreen/binseg/agent.py:102-141 grid_crop: locations (s_h, e_h, s_w, e_w) row-major over the grid, stride
 * ceil(crop * stride_ratio), the last crop of a row / column aligned to the far edge; with flip every crop is followed by
 * its mirrored copy whose location is (s_h, e_h, e_w, s_w).  Returns the number of crops (locs may be NULL to count). */
ORC_API int orc_grid_crop_locations(int h, int w, int crop_h, int crop_w, double stride_ratio, int flip, int* locs)
{
    int stride_h = (int)ceil(crop_h * stride_ratio), stride_w = (int)ceil(crop_w * stride_ratio);
    int grid_h = (int)(ceil((double)(h - crop_h) / stride_h) + 1), grid_w = (int)(ceil((double)(w - crop_w) / stride_w) + 1);
    int n = 0;
    for (int ih = 0; ih < grid_h; ih++)
        for (int iw = 0; iw < grid_w; iw++) {
            int e_h = ih * stride_h + crop_h; if (e_h > h) e_h = h;
            int s_h = e_h - crop_h;
            int e_w = iw * stride_w + crop_w; if (e_w > w) e_w = w;
            int s_w = e_w - crop_w;
            if (locs) { locs[4 * n] = s_h; locs[4 * n + 1] = e_h; locs[4 * n + 2] = s_w; locs[4 * n + 3] = e_w; }
            n++;
            if (flip) {
                if (locs) { locs[4 * n] = s_h; locs[4 * n + 1] = e_h; locs[4 * n + 2] = e_w; locs[4 * n + 3] = s_w; }
                n++;
            }
        }
    return n;
}

static inline int orc_reflect(int p, int n)          /* BORDER_REFLECT, any distance: ...cba|abc...xyz|zyx... */
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

/* unscreen/binseg/agent.py:176-198 preprocess for ONE crop: pad_resize(img, (th, tw)) (utils/imgprocess.py:67-97: bilinear
 * to (nh, nw) + BORDER_REFLECT bottom/right), imnormalize (:135-161), crop loc = (s_h, e_h, s_w, e_w), mirrored when
 * s_w > e_w (cv2.flip(crop, 1)).  out: (crop_h, crop_w, 4) float32 NHWC, channels R, G, B, 0. */
ORC_API void orc_segagent_pre(const uint8_t* img, int H, int W, int th, int tw, const int* loc, float* out)
{
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    int nh, nw;
    orc_pad_resize_geometry(H, W, th, tw, &nh, &nw);
    uint8_t* r = (uint8_t*)malloc((size_t)nh * nw * 3);
    orc_resize_linear_u8(img, H, W, 3, r, nh, nw);
    int s_h = loc[0], e_h = loc[1], s_w = loc[2], e_w = loc[3];
    int flip = s_w > e_w;
    if (flip) { int t = s_w; s_w = e_w; e_w = t; }
    int ch = e_h - s_h, cw = e_w - s_w;
    for (int y = 0; y < ch; y++)
        for (int x = 0; x < cw; x++) {
            int px = s_w + (flip ? cw - 1 - x : x), py = s_h + y;
            const uint8_t* p = r + ((size_t)orc_reflect(py, nh) * nw + orc_reflect(px, nw)) * 3;
            float* o = out + ((size_t)y * cw + x) * 4;
            for (int c = 0; c < 3; c++) {
                float v = (float)p[2 - c] / 255.f;
                v = v - (float)mean[c];
                o[c] = v * (float)(1 / std[c]);
            }
            o[3] = 0.f;
        }
    free(r);
}

/* exp on float32 restated with IEEE basic operations only (Cephes expf: Cody-Waite reduction by ln 2 and a degree-5
 * polynomial), so that the HIP kernel and this oracle agree bit for bit.  torch's CPU softmax uses Sleef's expf; both are
 * within ~1 ulp of exp, i.e. the softmax below agrees with the reference's to float rounding, not bit for bit. */
static inline float orc_det_expf(float x)
{
    if (x < -87.f) return 0.f;
    if (x > 88.f) return INFINITY;
    float kf = rintf(x * 1.44269504088896341f);
    float r = x - kf * 0.693359375f;
    r = r - kf * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    p = p * (r * r) + r;
    p = p + 1.f;
    return ldexpf(p, (int)kf);
}

/* F.interpolate(x, size, mode='bilinear', align_corners=False) source coordinate (ATen area_pixel_compute_source_index). */
static inline void orc_bilinear_nc_tap(int d, int in, int out, int* i0, int* i1, float* l0, float* l1)
{
    float scale = (float)in / (float)out;
    float s = scale * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    *i0 = (int)s;
    *i1 = *i0 + (*i0 < in - 1 ? 1 : 0);
    *l1 = s - (float)*i0;
    *l0 = 1.f - *l1;
}

/* F.interpolate(..., mode='bilinear', align_corners=False) (deeplab/_deeplab.py:51-55, deeplab/utils.py:20-21), float32.
 * src (ih, iw) pixels with pixel pitch `pstride` and channel pitch `cstride` elements; dst (oh, ow, C) interleaved. */
ORC_API void orc_upsample_nc_f32(const float* src, int C, int ih, int iw, size_t cstride, size_t pstride, float* dst, int oh, int ow)
{
    for (int y = 0; y < oh; y++) {
        int y0, y1; float hy0, hy1;
        orc_bilinear_nc_tap(y, ih, oh, &y0, &y1, &hy0, &hy1);
        for (int x = 0; x < ow; x++) {
            int x0, x1; float wx0, wx1;
            orc_bilinear_nc_tap(x, iw, ow, &x0, &x1, &wx0, &wx1);
            for (int c = 0; c < C; c++) {
                const float* p = src + (size_t)c * cstride;
                float top = wx0 * p[((size_t)y0 * iw + x0) * pstride] + wx1 * p[((size_t)y0 * iw + x1) * pstride];
                float bot = wx0 * p[((size_t)y1 * iw + x0) * pstride] + wx1 * p[((size_t)y1 * iw + x1) * pstride];
                dst[((size_t)y * ow + x) * C + c] = hy0 * top + hy1 * bot;
            }
        }
    }
}

/* unscreen/binseg/agent.py:200-237 postprocess + :143-174 inv_grid_crop + utils/imgprocess.py:100-132 inv_pad_resize.
 * logits: ncrops x (lh, lw, cbuf) float32 NHWC network outputs (2 classes in channels 0, 1) at the classifier's resolution;
 * the model's own final upsample to the crop size (deeplab/utils.py:20-21, align_corners=False) is applied here.
 * softmax over the 2 classes in float32; crops are summed in order into a float64 (th, tw, 2) map (mirrored crops
 * un-mirrored), divided by the cover count; the pad is cut, the map resized to (H, W) by cv2.resize on float64 (float
 * weights, double products), argmax (first maximum) * 255. */
ORC_API void orc_segagent_post(const float* logits, int ncrops, int lh, int lw, int cbuf, const int* locs, int crop_h, int crop_w,
                               int th, int tw, int H, int W, uint8_t* out, double* score_out)
{
    size_t n = (size_t)th * tw;
    double* pred = (double*)calloc(n * 2, sizeof(double));
    double* cnt = (double*)calloc(n, sizeof(double));
    float* up = (float*)malloc(sizeof(float) * (size_t)crop_h * crop_w * 2);
    for (int i = 0; i < ncrops; i++) {
        const float* lg = logits + (size_t)i * lh * lw * cbuf;
        orc_upsample_nc_f32(lg, 2, lh, lw, 1, cbuf, up, crop_h, crop_w);
        int s_h = locs[4 * i], s_w = locs[4 * i + 2], e_w = locs[4 * i + 3];
        int flip = s_w > e_w;
        int x0 = flip ? e_w : s_w;
        for (int y = 0; y < crop_h; y++)
            for (int x = 0; x < crop_w; x++) {
                int sx = flip ? crop_w - 1 - x : x;
                float a = up[((size_t)y * crop_w + sx) * 2], b = up[((size_t)y * crop_w + sx) * 2 + 1];
                float m = a > b ? a : b;
                float ea = orc_det_expf(a - m), eb = orc_det_expf(b - m);
                float s = ea + eb;
                size_t o = (size_t)(s_h + y) * tw + (x0 + x);
                pred[2 * o] += (double)(ea / s);
                pred[2 * o + 1] += (double)(eb / s);
                cnt[o] += 1.0;
            }
    }
    for (size_t i = 0; i < n; i++) { pred[2 * i] /= cnt[i]; pred[2 * i + 1] /= cnt[i]; }
    int rh, rw;
    orc_inv_pad_geometry(th, tw, H, W, &rh, &rw);
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * W);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * H);
    int same = (rh == H && rw == W);
    orc_linear_taps(W, rw, tx, 1);
    orc_linear_taps(H, rh, ty, 0);
    for (int y = 0; y < H; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, rh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, rh - 1);
        double b0 = ty[y].f0, b1 = ty[y].f1;
        for (int x = 0; x < W; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < rw ? x0 + 1 : x0;
            double a0 = tx[x].f0, a1 = tx[x].f1;
            double v[2];
            for (int c = 0; c < 2; c++) {
                if (same) { v[c] = pred[((size_t)y * tw + x) * 2 + c]; continue; }
                double r0 = pred[((size_t)y0 * tw + x0) * 2 + c] * a0 + pred[((size_t)y0 * tw + x1) * 2 + c] * a1;
                double r1 = pred[((size_t)y1 * tw + x0) * 2 + c] * a0 + pred[((size_t)y1 * tw + x1) * 2 + c] * a1;
                v[c] = r0 * b0 + r1 * b1;
            }
            if (score_out) { score_out[((size_t)y * W + x) * 2] = v[0]; score_out[((size_t)y * W + x) * 2 + 1] = v[1]; }
            out[(size_t)y * W + x] = v[1] > v[0] ? 255 : 0;
        }
    }
    free(tx); free(ty); free(pred); free(cnt); free(up);
}

/* ============================================================================================
 * STMAgent (unscreen/stm/agent.py): pre / post processing around the space-time memory network
 * ============================================================================================ */

/* log on float32 restated with IEEE basic operations only (Cephes logf), the partner of orc_det_expf. */
static inline float orc_det_logf(float x)
{
    if (!(x > 0.f)) return x == 0.f ? -INFINITY : NAN;
    int e;
    float m = frexpf(x, &e);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else m = m - 1.0f;
    float z = m * m;
    float y = 7.0376836292e-2f;
    y = y * m + -1.1514610310e-1f;
    y = y * m + 1.1676998740e-1f;
    y = y * m + -1.2420140846e-1f;
    y = y * m + 1.4249322787e-1f;
    y = y * m + -1.6668057665e-1f;
    y = y * m + 2.0000714765e-1f;
    y = y * m + -2.4999993993e-1f;
    y = y * m + 3.3333331174e-1f;
    y = y * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    z = z + 0.693359375f * fe;
    return z;
}

/* unscreen/stm/agent.py:62-98 preprocess of one frame (+ optionally the first frame's mask): pad_resize to (in_h, in_w) =
 * get_target_size(h, w, long_side, 16) (bilinear + BORDER_REFLECT), imnormalize; mask0 = resized > 127, channel 3 = object
 * (1 / 0), channel 4 = background.  out: (in_h, in_w, 8) float32 NHWC (R, G, B, m, o, 0, 0, 0); mask == NULL: m = o = 0. */
ORC_API void orc_stm_pre(const uint8_t* frame, const uint8_t* mask, int H, int W, int in_h, int in_w, float* out)
{
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    int nh, nw;
    orc_pad_resize_geometry(H, W, in_h, in_w, &nh, &nw);
    uint8_t* fr = (uint8_t*)malloc((size_t)nh * nw * 3);
    uint8_t* mr = mask ? (uint8_t*)malloc((size_t)nh * nw) : NULL;
    orc_resize_linear_u8(frame, H, W, 3, fr, nh, nw);
    if (mask) orc_resize_linear_u8(mask, H, W, 1, mr, nh, nw);
    for (int y = 0; y < in_h; y++)
        for (int x = 0; x < in_w; x++) {
            size_t s = (size_t)orc_reflect(y, nh) * nw + orc_reflect(x, nw);
            float* o = out + ((size_t)y * in_w + x) * 8;
            for (int c = 0; c < 3; c++) {
                float v = (float)fr[3 * s + (2 - c)] / 255.f;
                v = v - (float)mean[c];
                o[c] = v * (float)(1 / std[c]);
            }
            int fg = mask ? mr[s] > 127 : 0;
            o[3] = mask ? (float)fg : 0.f;
            o[4] = mask ? (float)(1 - fg) : 0.f;
            o[5] = o[6] = o[7] = 0.f;
        }
    free(fr); free(mr);
}

/* From the decoder's 1/4-resolution logits p2 (lh, lw, cbuf; classes in channels 0, 1) to the frame mask:
 * x4 bilinear (align_corners=False, stm/model.py:135-136), softmax -> ps (:215), soft aggregation (:195-202: em = (1-ps, ps)
 * clamped to [1e-7, 1-1e-7], logit = log(em / (1 - em))), softmax again (stm/agent.py:164) -> prob (in_h, in_w, 2) float32;
 * inv_pad_resize (utils/imgprocess.py:100-132: cut the pad, cv2.resize float32 bilinear) and argmax * 255 (stm/agent.py:172-201). */
ORC_API void orc_stm_post(const float* p2, int lh, int lw, int cbuf, int in_h, int in_w, int H, int W, uint8_t* out, float* prob_out)
{
    size_t n = (size_t)in_h * in_w;
    float* up = (float*)malloc(sizeof(float) * n * 2);
    float* prob = prob_out ? prob_out : (float*)malloc(sizeof(float) * n * 2);
    orc_upsample_nc_f32(p2, 2, lh, lw, 1, cbuf, up, in_h, in_w);
    for (size_t i = 0; i < n; i++) {
        float a = up[2 * i], b = up[2 * i + 1];
        float m = a > b ? a : b;
        float ea = orc_det_expf(a - m), eb = orc_det_expf(b - m);
        float ps = eb / (ea + eb);
        float e0 = 1.f - ps, e1 = ps;
        const float lo = 1e-7f, hi = 1.f - 1e-7f;
        e0 = e0 < lo ? lo : (e0 > hi ? hi : e0);
        e1 = e1 < lo ? lo : (e1 > hi ? hi : e1);
        float l0 = orc_det_logf(e0 / (1.f - e0)), l1 = orc_det_logf(e1 / (1.f - e1));
        float mm = l0 > l1 ? l0 : l1;
        float x0 = orc_det_expf(l0 - mm), x1 = orc_det_expf(l1 - mm);
        float s = x0 + x1;
        prob[2 * i] = x0 / s;
        prob[2 * i + 1] = x1 / s;
    }
    int rh, rw;
    orc_inv_pad_geometry(in_h, in_w, H, W, &rh, &rw);
    int same = (rh == H && rw == W);
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * W);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * H);
    orc_linear_taps(W, rw, tx, 1);
    orc_linear_taps(H, rh, ty, 0);
    for (int y = 0; y < H; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, rh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, rh - 1);
        float b0 = ty[y].f0, b1 = ty[y].f1;
        for (int x = 0; x < W; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < rw ? x0 + 1 : x0;
            float a0 = tx[x].f0, a1 = tx[x].f1;
            float v[2];
            for (int c = 0; c < 2; c++) {
                if (same) { v[c] = prob[((size_t)y * in_w + x) * 2 + c]; continue; }
                float r0 = prob[((size_t)y0 * in_w + x0) * 2 + c] * a0 + prob[((size_t)y0 * in_w + x1) * 2 + c] * a1;
                float r1 = prob[((size_t)y1 * in_w + x0) * 2 + c] * a0 + prob[((size_t)y1 * in_w + x1) * 2 + c] * a1;
                v[c] = r0 * b0 + r1 * b1;
            }
            out[(size_t)y * W + x] = v[1] > v[0] ? 255 : 0;
        }
    }
    free(tx); free(ty); free(up);
    if (!prob_out) free(prob);
}

/* ============================================================================================
 * NumPy-level helpers of unscreen/utils/imgprocess.py that the reference's tools and agents import
 * (tools/unscreen/bg_offline.py:16, tools/replace/replace.py:14, unscreen/iseg/agent.py:5)
 * ============================================================================================ */

/* unscreen/utils/imgprocess.py:67-97 pad_resize on uint8 (1 or 3 channels): cv2.resize (default INTER_LINEAR) to the
 * aspect-preserving size, then cv2.copyMakeBorder(0, pad_h, 0, pad_w, BORDER_REFLECT). */
ORC_API void orc_pad_resize_u8(const uint8_t* src, int h, int w, int cn, uint8_t* dst, int th, int tw)
{
    int nh, nw;
    orc_pad_resize_geometry(h, w, th, tw, &nh, &nw);
    uint8_t* tmp = (uint8_t*)malloc((size_t)nh * nw * cn);
    orc_resize_linear_u8(src, h, w, cn, tmp, nh, nw);
    orc_pad_reflect_u8(tmp, nh, nw, cn, dst, th, tw);
    free(tmp);
}

/* cv2.resize (INTER_LINEAR) on float32 / float64 images with cn interleaved channels and a source row pitch (in elements):
 * same size -> copy; exact 2x decimation -> the 2x2 INTER_AREA average (generic path: ((S00 + S01) + S10) + S11, times the
 * float 0.25); otherwise float weights in both passes (cv::HResizeLinear / VResizeLinear with AT = float; WT = float for
 * float32 data, double for float64 data), each product and sum rounded once. */
ORC_API void orc_resize_linear_f32c(const float* src, int sh, int sw, int cn, size_t pitch, float* dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {
        for (int y = 0; y < dh; y++) memcpy(dst + (size_t)y * dw * cn, src + (size_t)y * pitch, sizeof(float) * (size_t)dw * cn);
        return;
    }
    if (orc_is_area_fast_2x(sh, sw, dh, dw)) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const float* p = src + (size_t)(2 * y) * pitch + (size_t)(2 * x) * cn + c;
                    float s = p[0] + p[cn];
                    s = s + p[pitch];
                    s = s + p[pitch + cn];
                    dst[((size_t)y * dw + x) * cn + c] = s * 0.25f;
                }
        return;
    }
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * dw);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * dh);
    orc_linear_taps(dw, sw, tx, 1);
    orc_linear_taps(dh, sh, ty, 0);
    for (int y = 0; y < dh; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, sh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, sh - 1);
        float b0 = ty[y].f0, b1 = ty[y].f1;
        for (int x = 0; x < dw; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < sw ? x0 + 1 : x0;
            float a0 = tx[x].f0, a1 = tx[x].f1;
            for (int c = 0; c < cn; c++) {
                float r0 = src[(size_t)y0 * pitch + (size_t)x0 * cn + c] * a0 + src[(size_t)y0 * pitch + (size_t)x1 * cn + c] * a1;
                float r1 = src[(size_t)y1 * pitch + (size_t)x0 * cn + c] * a0 + src[(size_t)y1 * pitch + (size_t)x1 * cn + c] * a1;
                dst[((size_t)y * dw + x) * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
    free(tx); free(ty);
}

ORC_API void orc_resize_linear_f64c(const double* src, int sh, int sw, int cn, size_t pitch, double* dst, int dh, int dw)
{
    if (sh == dh && sw == dw) {
        for (int y = 0; y < dh; y++) memcpy(dst + (size_t)y * dw * cn, src + (size_t)y * pitch, sizeof(double) * (size_t)dw * cn);
        return;
    }
    if (orc_is_area_fast_2x(sh, sw, dh, dw)) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const double* p = src + (size_t)(2 * y) * pitch + (size_t)(2 * x) * cn + c;
                    double s = p[0] + p[cn];
                    s = s + p[pitch];
                    s = s + p[pitch + cn];
                    dst[((size_t)y * dw + x) * cn + c] = s * (double)0.25f;
                }
        return;
    }
    orc_tap* tx = (orc_tap*)malloc(sizeof(orc_tap) * dw);
    orc_tap* ty = (orc_tap*)malloc(sizeof(orc_tap) * dh);
    orc_linear_taps(dw, sw, tx, 1);
    orc_linear_taps(dh, sh, ty, 0);
    for (int y = 0; y < dh; y++) {
        int y0 = orc_clampi(ty[y].ofs, 0, sh - 1), y1 = orc_clampi(ty[y].ofs + 1, 0, sh - 1);
        double b0 = ty[y].f0, b1 = ty[y].f1;
        for (int x = 0; x < dw; x++) {
            int x0 = tx[x].ofs, x1 = x0 + 1 < sw ? x0 + 1 : x0;
            double a0 = tx[x].f0, a1 = tx[x].f1;
            for (int c = 0; c < cn; c++) {
                double r0 = src[(size_t)y0 * pitch + (size_t)x0 * cn + c] * a0 + src[(size_t)y0 * pitch + (size_t)x1 * cn + c] * a1;
                double r1 = src[(size_t)y1 * pitch + (size_t)x0 * cn + c] * a0 + src[(size_t)y1 * pitch + (size_t)x1 * cn + c] * a1;
                dst[((size_t)y * dw + x) * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
    free(tx); free(ty);
}

/* unscreen/utils/imgprocess.py:100-132 inv_pad_resize: cut the bottom / right pad of an (h, w) canvas, cv2.resize the rest to
 * (ori_h, ori_w).  elem: 0 = uint8, 1 = float32, 2 = float64 (the reference passes uint8 masks, the float32 alpha of
 * vmatting/agent.py:127 and iseg/agent.py:117, and the float64 scores of binseg/agent.py:233). */
ORC_API void orc_inv_pad_resize(const void* src, int h, int w, int cn, int elem, void* dst, int oh, int ow)
{
    int rh, rw;
    orc_inv_pad_geometry(h, w, oh, ow, &rh, &rw);
    if (elem == 0) {
        uint8_t* crop = (uint8_t*)malloc((size_t)rh * rw * cn);
        for (int y = 0; y < rh; y++) memcpy(crop + (size_t)y * rw * cn, (const uint8_t*)src + (size_t)y * w * cn, (size_t)rw * cn);
        orc_resize_linear_u8(crop, rh, rw, cn, (uint8_t*)dst, oh, ow);
        free(crop);
    } else if (elem == 1) {
        orc_resize_linear_f32c((const float*)src, rh, rw, cn, (size_t)w * cn, (float*)dst, oh, ow);
    } else {
        orc_resize_linear_f64c((const double*)src, rh, rw, cn, (size_t)w * cn, (double*)dst, oh, ow);
    }
}

/* unscreen/utils/imgprocess.py:135-161 imnormalize: cv2.cvtColor(BGR2RGB) IN PLACE on the argument when to_rgb, then
 * float32(img) / 255., cv2.subtract(img, mean) and cv2.multiply(img, 1 / std) - OpenCV turns the float64 scalars into float32
 * for a float32 array, so all three steps are float32 operations; 1 / std is evaluated in float64 first. */
ORC_API void orc_imnormalize(uint8_t* img, size_t n, const double* mean, const double* std, int to_rgb, float* out)
{
    float meanf[3], stdinvf[3];
    for (int c = 0; c < 3; c++) { meanf[c] = (float)mean[c]; stdinvf[c] = (float)(1 / std[c]); }
    for (size_t i = 0; i < n; i++) {
        if (to_rgb) { uint8_t t = img[3 * i]; img[3 * i] = img[3 * i + 2]; img[3 * i + 2] = t; }
        for (int c = 0; c < 3; c++) {
            float x = (float)img[3 * i + c] / 255.f;
            x = x - meanf[c];
            out[3 * i + c] = x * stdinvf[c];
        }
    }
}

/* cv2.moments of an 8-bit single-channel image, spatial moments up to order 1 (m00, m10, m01): exact integer sums
 * sum(I), sum(x I), sum(y I) returned as doubles (OpenCV accumulates 8-bit rows in integers). */
ORC_API void orc_moments_u8(const uint8_t* img, int h, int w, double* m)
{
    unsigned long long m00 = 0, m10 = 0, m01 = 0;
    for (int y = 0; y < h; y++) {
        unsigned long long r0 = 0, r1 = 0;
        for (int x = 0; x < w; x++) { r0 += img[(size_t)y * w + x]; r1 += (unsigned long long)x * img[(size_t)y * w + x]; }
        m00 += r0; m10 += r1; m01 += r0 * (unsigned long long)y;
    }
    m[0] = (double)m00; m[1] = (double)m10; m[2] = (double)m01;
}

/* unscreen/utils/imgprocess.py:7-14 get_center: cv2.cvtColor(img, COLOR_RGB2GRAY) - the first channel gets the red weight -,
 * cv2.moments, cX = int(m10 / (m00 + 1e-6)), cY = int(m01 / (m00 + 1e-6)). */
ORC_API void orc_get_center(const uint8_t* img3, int h, int w, int* cx, int* cy)
{
    size_t n = (size_t)h * w;
    uint8_t* g = (uint8_t*)malloc(n);
    for (size_t i = 0; i < n; i++) g[i] = orc_gray(img3[3 * i + 2], img3[3 * i + 1], img3[3 * i]);
    double m[3];
    orc_moments_u8(g, h, w, m);
    *cx = (int)(m[1] / (m[0] + 1e-6));
    *cy = (int)(m[2] / (m[0] + 1e-6));
    free(g);
}

/* unscreen/utils/imgprocess.py:17-30 get_mask: BGR2GRAY, cv2.threshold(25, 255, THRESH_BINARY): 255 where gray > 25. */
ORC_API void orc_get_mask(const uint8_t* img3, size_t n, int thr, uint8_t* out)
{
    for (size_t i = 0; i < n; i++) out[i] = orc_gray(img3[3 * i], img3[3 * i + 1], img3[3 * i + 2]) > thr ? 255 : 0;
}

/* ============================================================================================
 * ISegAgent, no-BRS predictor (unscreen/iseg/agent.py, iseg/brs/brs_predictor.py, iseg/brs/models/distmap.py): pixel stages
 * ============================================================================================ */

/* agent.py:92-104 preprocess (pad_resize to (in_h, in_w), imnormalize) + the model's input stage: DistMaps (distmap.py:80-114,
 * spatial_scale 1, norm_radius 260: tanh(2 sqrt(min over the clicks of ((y - py)^2 + (x - px)^2) / 260^2)), absent clicks (-1, -1)
 * count as distance^2 = 1e6) and rgb_conv (distmap.py:16-22: Conv 5 -> 8, LeakyReLU(0.2), BatchNorm, Conv 8 -> 3; the BatchNorm is
 * folded into the second layer by the caller).  points: 2 P (y, x) pairs in network-input coordinates, positive slots first.
 * tanh is restated with orc_det_expf (1 - 2 / (exp(2u) + 1), u clamped at 20) so that the HIP kernel agrees bit for bit; it is
 * within float rounding of torch.tanh.  out: (in_h, in_w, 4) float32 NHWC (R', G', B', 0); with flip a second, mirrored image. */
ORC_API void orc_iseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, const float* points, int P, const float* rgb_params,
                          int flip, float* out)
{
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    size_t n = (size_t)in_h * in_w;
    uint8_t* img = (uint8_t*)malloc(n * 3);
    orc_pad_resize_u8(frame, H, W, 3, img, in_h, in_w);
    float meanf[3], stdinvf[3];
    for (int c = 0; c < 3; c++) { meanf[c] = (float)mean[c]; stdinvf[c] = (float)(1 / std[c]); }
    const float* w1 = rgb_params; const float* b1 = w1 + 40; const float* w2 = b1 + 8; const float* b2 = w2 + 24;
    for (int y = 0; y < in_h; y++)
        for (int x = 0; x < in_w; x++) {
            size_t i = (size_t)y * in_w + x;
            float v[5];
            for (int c = 0; c < 3; c++) {
                float f = (float)img[3 * i + (2 - c)] / 255.f;
                f = f - meanf[c];
                v[c] = f * stdinvf[c];
            }
            for (int g = 0; g < 2; g++) {
                float best = 1e6f;
                for (int k = 0; k < P; k++) {
                    float qy = points[2 * (g * P + k)], qx = points[2 * (g * P + k) + 1];
                    float d = 1e6f;
                    if (!((qy > qx ? qy : qx) < 0.f)) {
                        float dy = ((float)y - qy) / 260.f, dx = ((float)x - qx) / 260.f;
                        d = dy * dy + dx * dx;
                    }
                    best = d < best ? d : best;
                }
                float u = sqrtf(best) * 2.f;
                u = u > 20.f ? 20.f : u;
                v[3 + g] = 1.f - 2.f / (orc_det_expf(2.f * u) + 1.f);
            }
            float h[8], r[3];
            for (int j = 0; j < 8; j++) {
                float s = b1[j];
                for (int c = 0; c < 5; c++) s = s + w1[j * 5 + c] * v[c];
                h[j] = s > 0.f ? s : s * 0.2f;
            }
            for (int k = 0; k < 3; k++) {
                float s = b2[k];
                for (int j = 0; j < 8; j++) s = s + w2[k * 8 + j] * h[j];
                r[k] = s;
            }
            float* o = out + 4 * i;
            o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = 0.f;
            if (flip) {
                float* m = out + 4 * (n + (size_t)y * in_w + (in_w - 1 - x));
                m[0] = r[0]; m[1] = r[1]; m[2] = r[2]; m[3] = 0.f;
            }
        }
    free(img);
}

/* brs_predictor.py:46-51 + transforms.py:28-29, 61-66 + agent.py:116-119: channel 0 of the (nimg, lh, lw, cbuf) logits is upsampled
 * (bilinear, align_corners) to (in_h, in_w); with flip the second image is mirrored back and the pair averaged ON THE LOGITS (the
 * transforms are undone in reverse order), then the sigmoid (restated with orc_det_expf); prob_out (in_h, in_w).  mask (may be
 * NULL): inv_pad_resize of prob to (H, W) (cv2.resize on float32), > thr (float32 compare) -> 255 / 0. */
ORC_API void orc_iseg_post(const float* logits, int lh, int lw, int cbuf, int in_h, int in_w, int flip, int H, int W, double thr,
                           float* prob_out, uint8_t* mask)
{
    size_t n = (size_t)in_h * in_w;
    float* up0 = (float*)malloc(sizeof(float) * n);
    float* up1 = flip ? (float*)malloc(sizeof(float) * n) : NULL;
    orc_upsample_ac_f32(logits, 1, lh, lw, 0, (size_t)cbuf, up0, in_h, in_w);
    if (flip) orc_upsample_ac_f32(logits + (size_t)lh * lw * cbuf, 1, lh, lw, 0, (size_t)cbuf, up1, in_h, in_w);
    for (int y = 0; y < in_h; y++)
        for (int x = 0; x < in_w; x++) {
            float v = up0[(size_t)y * in_w + x];
            if (flip) v = 0.5f * (v + up1[(size_t)y * in_w + (in_w - 1 - x)]);
            prob_out[(size_t)y * in_w + x] = 1.f / (1.f + orc_det_expf(-v));
        }
    free(up0); free(up1);
    if (!mask) return;
    float* full = (float*)malloc(sizeof(float) * (size_t)H * W);
    orc_inv_pad_resize(prob_out, in_h, in_w, 1, 1, full, H, W);
    float t = (float)thr;
    for (size_t i = 0; i < (size_t)H * W; i++) mask[i] = full[i] > t ? 255 : 0;
    free(full);
}
