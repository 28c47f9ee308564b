// usc_pixel.hip - hand-written gfx950 kernels + the C ABI of libunscreen_hip.so (include/unscreen_hip.h).
//
// All of these stages are HBM-bound byte work (1-28 B per pixel, O(10-100) integer ops per byte), so the
// design rules are the memory ones: flat 4-pixel (12/4-byte) coalesced accesses, LDS only where a stencil
// needs neighbours (morphology tiles with halo), LDS-resident lookup tables for the 8-bit HSV conversion,
// grids of >> 256 workgroups, no host synchronisation, everything on the caller's stream.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

// --------------------------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
int usc::fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
int usc::check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(USC_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return USC_OK;
}
extern "C" const char* usc_last_error(void) { return g_err; }
extern "C" int usc_abi_version(void) { return 1; }

// --------------------------------------------------------------------------------------------
// host geometry (pure functions)
// --------------------------------------------------------------------------------------------
extern "C" void usc_get_target_size(int h, int w, int long_side, int division, int* th, int* tw) {
    // int(float(L) * short / long), rounded UP to a multiple of `division`; square goes to the width branch.
    if (h > w) {
        int t = (int)((double)long_side * w / h);
        if (t % division) t = (t / division + 1) * division;
        *th = long_side; *tw = t;
    } else {
        int t = (int)((double)long_side * h / w);
        if (t % division) t = (t / division + 1) * division;
        *tw = long_side; *th = t;
    }
}
extern "C" void usc_pad_resize_geometry(int h, int w, int target_h, int target_w, int* new_h, int* new_w) {
    if ((double)h / w > (double)target_h / target_w) { *new_h = target_h; *new_w = (int)((double)target_h * w / h); }
    else { *new_w = target_w; *new_h = (int)((double)target_w * h / w); }
}
extern "C" void usc_inv_pad_geometry(int h, int w, int ori_h, int ori_w, int* rh, int* rw) {
    if ((double)ori_h / ori_w > (double)h / w) { *rh = h; *rw = (int)((double)h * ori_w / ori_h); }
    else { *rw = w; *rh = (int)((double)w * ori_h / ori_w); }
}

// --------------------------------------------------------------------------------------------
// structuring element (cv::getStructuringElement(MORPH_ELLIPSE, (k,k)), anchor (k/2, k/2))
// --------------------------------------------------------------------------------------------
constexpr int kMaxK = 9;
struct StructElem {
    int n, ksize, anchor;
    int8_t dy[kMaxK * kMaxK];
    int8_t dx[kMaxK * kMaxK];
};
static StructElem make_ellipse(int k) {
    StructElem se;
    memset(&se, 0, sizeof(se));
    se.ksize = k; se.anchor = k / 2;
    int r = k / 2, c = k / 2;
    double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < k; i++) {
        int dy = i - r, j1 = 0, j2 = 0;
        if (abs(dy) <= r) {
            int dx = (int)lrint(c * sqrt((r * r - dy * dy) * inv_r2));
            j1 = c - dx > 0 ? c - dx : 0;
            j2 = c + dx + 1 < k ? c + dx + 1 : k;
        }
        for (int j = j1; j < j2; j++) { se.dy[se.n] = (int8_t)(i - se.anchor); se.dx[se.n] = (int8_t)(j - se.anchor); se.n++; }
    }
    return se;
}

// --------------------------------------------------------------------------------------------
// LDS tile morphology
// --------------------------------------------------------------------------------------------
constexpr int kTileH = 16, kTileW = 64, kMorphThreads = 256;

struct TileGeom {             // extended (core + halo) tile of one workgroup
    int eh, ew, pitch;        // rows, cols, LDS row pitch (bytes)
    int top, left;            // halo before the core
};
static TileGeom tile_geom(const StructElem& se, int iters) {
    TileGeom t;
    int before = iters * se.anchor, after = iters * (se.ksize - 1 - se.anchor);
    t.top = before; t.left = before;
    t.eh = kTileH + before + after;
    t.ew = kTileW + before + after;
    t.pitch = (t.ew + 3) & ~3;
    return t;
}

// `iters` passes of the SE over an LDS tile, ping-ponging a <-> b.  Positions outside the image hold the
// identity (0 for max, 255 for min) in every pass - that is cv2's "border never wins" rule per iteration.
// Taps that fall outside the extended tile are skipped; the error this causes stays inside the halo.
// Returns the buffer holding the result.
template <bool ERODE>
__device__ uint8_t* morph_passes(uint8_t* a, uint8_t* b, const TileGeom& tg, int gy0, int gx0, int H, int W,
                                 const StructElem& se, int iters) {
    const int total = tg.eh * tg.ew;
    for (int it = 0; it < iters; it++) {
        for (int i = threadIdx.x; i < total; i += blockDim.x) {
            int ty = i / tg.ew, tx = i - ty * tg.ew;
            int gy = gy0 + ty, gx = gx0 + tx;
            int best = ERODE ? 255 : 0;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                for (int k = 0; k < se.n; k++) {
                    int yy = ty + se.dy[k], xx = tx + se.dx[k];
                    if (yy < 0 || yy >= tg.eh || xx < 0 || xx >= tg.ew) continue;
                    int v = a[yy * tg.pitch + xx];
                    best = ERODE ? min(best, v) : max(best, v);
                }
            }
            b[ty * tg.pitch + tx] = (uint8_t)best;
        }
        __syncthreads();
        uint8_t* t = a; a = b; b = t;
    }
    return a;
}

// Plain morphology: dst = morph(src).  MODE_GATE additionally applies the bg_offline.py:160 gate
// alpha = gray(seg) * (dilated // 255) in the epilogue.
enum { MODE_PLAIN = 0, MODE_GATE = 1 };
template <int MODE>
__global__ __launch_bounds__(kMorphThreads) void morph_tile_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                   int H, int W, StructElem se, TileGeom tg, int iters, int op,
                                                                   const uint8_t* __restrict__ seg, int seg_cn) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    uint8_t* a = lds;
    uint8_t* b = lds + tg.eh * tg.pitch;
    const int y0 = blockIdx.y * kTileH, x0 = blockIdx.x * kTileW;
    const int gy0 = y0 - tg.top, gx0 = x0 - tg.left;
    const uint8_t ident = op == USC_MORPH_ERODE ? 255 : 0;
    for (int i = threadIdx.x; i < tg.eh * tg.ew; i += blockDim.x) {
        int ty = i / tg.ew, tx = i - ty * tg.ew;
        int gy = gy0 + ty, gx = gx0 + tx;
        uint8_t v = ident;
        if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = src[(size_t)gy * W + gx];
        a[ty * tg.pitch + tx] = v;
    }
    __syncthreads();
    uint8_t* r = op == USC_MORPH_ERODE ? morph_passes<true>(a, b, tg, gy0, gx0, H, W, se, iters)
                                       : morph_passes<false>(a, b, tg, gy0, gx0, H, W, se, iters);
    for (int i = threadIdx.x; i < kTileH * kTileW; i += blockDim.x) {
        int ty = i / kTileW, tx = i - ty * kTileW;
        int gy = y0 + ty, gx = x0 + tx;
        if (gy >= H || gx >= W) continue;
        int v = r[(ty + tg.top) * tg.pitch + tx + tg.left];
        size_t o = (size_t)gy * W + gx;
        if (MODE == MODE_GATE) {
            int s = seg_cn == 3 ? gray_u8(seg[3 * o], seg[3 * o + 1], seg[3 * o + 2]) : seg[o];
            v = (s * (v / 255)) & 0xff;
        }
        dst[o] = (uint8_t)v;
    }
}

static int launch_morph(int mode, const uint8_t* src, uint8_t* dst, int H, int W, int ksize, int iters, int op,
                        const uint8_t* seg, int seg_cn, hipStream_t st) {
    if (ksize < 1 || ksize > kMaxK) return fail(USC_E_UNSUPPORTED, "morph: ksize %d outside [1,%d]", ksize, kMaxK);
    if (iters < 0) return fail(USC_E_ARG, "morph: iters %d < 0", iters);
    StructElem se = make_ellipse(ksize);
    TileGeom tg = tile_geom(se, iters);
    size_t lds = 2 * (size_t)tg.eh * tg.pitch;
    if (lds > 60 * 1024) return fail(USC_E_UNSUPPORTED, "morph: ksize %d x iters %d needs %zu B of LDS", ksize, iters, lds);
    dim3 grid((W + kTileW - 1) / kTileW, (H + kTileH - 1) / kTileH);
    if (mode == MODE_GATE)
        hipLaunchKernelGGL(morph_tile_kernel<MODE_GATE>, grid, dim3(kMorphThreads), lds, st, src, dst, H, W, se, tg, iters, op, seg, seg_cn);
    else
        hipLaunchKernelGGL(morph_tile_kernel<MODE_PLAIN>, grid, dim3(kMorphThreads), lds, st, src, dst, H, W, se, tg, iters, op, seg, seg_cn);
    return check_launch("morph_tile_kernel");
}

extern "C" size_t usc_morph_ws_bytes(int, int, int, int) { return 0; }

extern "C" int usc_morph_u8(const uint8_t* src, uint8_t* dst, int H, int W, int ksize, int iters, int op,
                            void*, size_t, usc_stream_t stream) {
    if (!src || !dst || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_morph_u8: null pointer or empty image");
    if (src == dst) return fail(USC_E_ARG, "usc_morph_u8: src and dst alias");
    if (op != USC_MORPH_DILATE && op != USC_MORPH_ERODE) return fail(USC_E_ARG, "usc_morph_u8: op %d", op);
    return launch_morph(MODE_PLAIN, src, dst, H, W, ksize, iters, op, nullptr, 1, (hipStream_t)stream);
}

// --------------------------------------------------------------------------------------------
// trimap
// --------------------------------------------------------------------------------------------

// ---- bit-packed binary morphology ---------------------------------------------------------------
// Both hot uses of morphology are binary in disguise: the trimap thresholds the grey dilation / erosion at
// 127/128 (min / max filters commute with a threshold), and the difference gate only asks whether the dilated
// value reaches 255.  One uint32 holds 32 horizontal pixels (bit i <-> x0 + i), a workgroup owns a tile of
// kBitH rows x kBitWW words plus halo rows and one halo word (32 px) per side, and one SE pass is a handful of
// shifts / ORs per word instead of k*k byte compares per pixel.
constexpr int kBitH = 16, kBitWW = 4, kBitEWW = kBitWW + 2, kBitThreads = 1024;     // 16 waves: the ~80 64-pixel chunks of a tile's byte loads go out in ONE round (round 5; 4 waves needed four round trips)

struct BitSE {                // per SE row: contiguous run [lo, hi] of dx (lo > hi: empty row)
    int nrows, anchor, top, bottom;
    int8_t lo[kMaxK], hi[kMaxK];
};
static BitSE make_bit_se(int ksize, int iters) {
    StructElem se = make_ellipse(ksize);
    BitSE b;
    memset(&b, 0, sizeof(b));
    b.nrows = ksize; b.anchor = se.anchor;
    for (int i = 0; i < ksize; i++) { b.lo[i] = 1; b.hi[i] = 0; }
    for (int k = 0; k < se.n; k++) {
        int i = se.dy[k] + se.anchor;
        if (b.lo[i] > b.hi[i]) { b.lo[i] = se.dx[k]; b.hi[i] = se.dx[k]; }
        else { if (se.dx[k] < b.lo[i]) b.lo[i] = se.dx[k]; if (se.dx[k] > b.hi[i]) b.hi[i] = se.dx[k]; }
    }
    b.top = iters * se.anchor;
    b.bottom = iters * (ksize - 1 - se.anchor);
    return b;
}
// the horizontal halo is one word: iters * max(|dx|) must fit in 32 bits
static bool bit_se_fits(int ksize, int iters) { return iters * (ksize / 2) <= 32 && iters * (ksize - 1 - ksize / 2) <= 32; }

// `iters` passes over the LDS bit tile.  D accumulates OR (dilation, outside = 0), E accumulates AND (erosion,
// outside = 1); positions outside the image are forced back to the identity after every pass.
template <bool WITH_ERODE, int EWW = kBitEWW>
__device__ void bit_morph_passes(uint32_t*& d, uint32_t*& d2, uint32_t*& e, uint32_t*& e2, const uint32_t* inimg,
                                 int ER, const BitSE& se, int iters) {
    constexpr int kBitEWW = EWW;              // words per extended row of THIS tile shape (shadows the default)
    for (int it = 0; it < iters; it++) {
        for (int idx = threadIdx.x; idx < ER * kBitEWW; idx += blockDim.x) {
            const int r = idx / kBitEWW, w = idx - r * kBitEWW;
            uint32_t accD = 0, accE = 0xffffffffu;
            for (int i = 0; i < se.nrows; i++) {
                const int lo = se.lo[i], hi = se.hi[i];
                const int rr = r + i - se.anchor;
                if (lo > hi || rr < 0 || rr >= ER) continue;
                const int base = rr * kBitEWW + w;
                const uint32_t dc = d[base], dl = w > 0 ? d[base - 1] : 0u, dr = w < kBitEWW - 1 ? d[base + 1] : 0u;
                uint32_t ec = 0, el = 0, er = 0;
                if (WITH_ERODE) {
                    ec = e[base]; el = w > 0 ? e[base - 1] : 0xffffffffu; er = w < kBitEWW - 1 ? e[base + 1] : 0xffffffffu;
                }
                for (int dx = lo; dx <= hi; dx++) {
                    if (dx == 0) { accD |= dc; if (WITH_ERODE) accE &= ec; }
                    else if (dx > 0) {
                        accD |= (dc >> dx) | (dr << (32 - dx));
                        if (WITH_ERODE) accE &= (ec >> dx) | (er << (32 - dx));
                    } else {
                        accD |= (dc << (-dx)) | (dl >> (32 + dx));
                        if (WITH_ERODE) accE &= (ec << (-dx)) | (el >> (32 + dx));
                    }
                }
            }
            const uint32_t in = inimg[idx];
            d2[idx] = accD & in;
            if (WITH_ERODE) e2[idx] = accE | ~in;
        }
        __syncthreads();
        uint32_t* t = d; d = d2; d2 = t;
        if (WITH_ERODE) { t = e; e = e2; e2 = t; }
    }
}

// Stage 1: NEAREST decimation + threshold to bits + dilate + erode + classify, one low-res tile per workgroup.
__global__ __launch_bounds__(kBitThreads) void trimap_small_kernel(const uint8_t* __restrict__ mask, int H, int W,
                                                                   uint8_t* __restrict__ tri, int h, int w,
                                                                   double ify, double ifx, BitSE se, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    const int ER = kBitH + se.top + se.bottom;
    const int plane = ER * kBitEWW;
    uint32_t* lds = reinterpret_cast<uint32_t*>(lds_raw);
    uint32_t *d = lds, *d2 = lds + plane, *e = lds + 2 * plane, *e2 = lds + 3 * plane, *inimg = lds + 4 * plane;
    const int y0 = blockIdx.y * kBitH, x0 = blockIdx.x * (kBitWW * 32);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
    const int chunks_per_row = kBitEWW / 2;                      // 64 pixels = 2 words per wave-instruction
    // kLoadBatch chunks per round: their byte loads are all issued before the first ballot needs one (a wave has ~20 chunks; one
    // memory round trip per chunk was most of this kernel's duration)
    constexpr int kLoadBatch = 5;
    for (int c0 = wave; c0 < ER * chunks_per_row; c0 += nwaves * kLoadBatch) {
        uint8_t val[kLoadBatch];
        bool inside[kLoadBatch];
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
            const int c = c0 + u * nwaves;
            const int r = c / chunks_per_row, cx = c - r * chunks_per_row;
            const int ly = y0 - se.top + r, lx = x0 - 32 + cx * 64 + lane;
            inside[u] = c < ER * chunks_per_row && ly >= 0 && ly < h && lx >= 0 && lx < w;
            val[u] = 0;
            if (inside[u]) {
                const int sy = min((int)floor(ly * ify), H - 1);     // cv::resizeNN
                const int sx = min((int)floor(lx * ifx), W - 1);
                val[u] = mask[(size_t)sy * W + sx];
            }
        }
#pragma unroll
        for (int u = 0; u < kLoadBatch; u++) {
            const int c = c0 + u * nwaves;
            if (c >= ER * chunks_per_row) continue;              // (uniform per wave)
            const int r = c / chunks_per_row, cx = c - r * chunks_per_row;
            const bool in = inside[u], b = in && val[u] > 127;
            const unsigned long long bal = __ballot(b), inb = __ballot(in);
            if (lane == 0) {
                const int o = r * kBitEWW + 2 * cx;
                d[o] = (uint32_t)bal; d[o + 1] = (uint32_t)(bal >> 32);
                e[o] = (uint32_t)(bal | ~inb); e[o + 1] = (uint32_t)((bal | ~inb) >> 32);
                inimg[o] = (uint32_t)inb; inimg[o + 1] = (uint32_t)(inb >> 32);
            }
        }
    }
    __syncthreads();
    bit_morph_passes<true>(d, d2, e, e2, inimg, ER, se, iters);
    // epilogue: 8 pixels per lane.  trimap = 128; [eroded > 127] = 255; [dilated < 128] = 0  (agent.py:54,57-58)
    for (int idx = threadIdx.x; idx < kBitH * kBitWW * 4; idx += blockDim.x) {
        const int row = idx / (kBitWW * 4), seg = idx - row * (kBitWW * 4);
        const int ly = y0 + row, lx = x0 + seg * 8;
        if (ly >= h || lx >= w) continue;
        const int o = (row + se.top) * kBitEWW + 1 + (seg >> 2), sh = (seg & 3) * 8;
        const uint32_t db = (d[o] >> sh) & 0xff, eb = (e[o] >> sh) & 0xff;
        uint32_t out[8];
#pragma unroll
        for (int k = 0; k < 8; k++) out[k] = ((db >> k) & 1) ? (((eb >> k) & 1) ? 255u : 128u) : 0u;
        uint8_t* dst = tri + (size_t)ly * w + lx;
        if (lx + 8 <= w && (((size_t)ly * w + lx) & 7) == 0) {
            uint2 v;
            v.x = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
            v.y = out[4] | (out[5] << 8) | (out[6] << 16) | (out[7] << 24);
            *reinterpret_cast<uint2*>(dst) = v;
        } else for (int k = 0; k < 8 && lx + k < w; k++) dst[k] = (uint8_t)out[k];
    }
}

// 8-bit cv::resize(INTER_LINEAR) of a single-channel image, one value.
__device__ __forceinline__ int resize_u8_c1(const uint8_t* __restrict__ s, int sh, int sw, int y, int x,
                                            int mode, double scale_y, double scale_x) {
    if (mode == RS_COPY) return s[(size_t)y * sw + x];
    if (mode == RS_AREA2) {
        const uint8_t* p = s + (size_t)(2 * y) * sw + 2 * x;
        return (p[0] + p[1] + p[sw] + p[sw + 1] + 2) >> 2;
    }
    Tap tx = linear_tap_x(x, scale_x, sw), ty = linear_tap(y, scale_y);
    int x0 = tx.s, x1 = min(x0 + 1, sw - 1);
    int y0 = clampi(ty.s, 0, sh - 1), y1 = clampi(ty.s + 1, 0, sh - 1);
    int a0 = coef11(1.f - tx.f), a1 = coef11(tx.f), b0 = coef11(1.f - ty.f), b1 = coef11(ty.f);
    return bilinear_u8(s[(size_t)y0 * sw + x0], s[(size_t)y0 * sw + x1], s[(size_t)y1 * sw + x0], s[(size_t)y1 * sw + x1],
                       a0, a1, b0, b1);
}

// Stage 2: bilinear back to H x W (trimap/agent.py:59) and collapse the blur to 128 (:60). 4 pixels per lane.
__global__ __launch_bounds__(256) void trimap_up_kernel(const uint8_t* __restrict__ tri, int h, int w,
                                                        uint8_t* __restrict__ out, int H, int W,
                                                        int mode, double scale_y, double scale_x) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += gridDim.x * blockDim.x) {
        const uint32_t i0 = q * 4;
        uint32_t y = i0 / (uint32_t)W, x = i0 - y * (uint32_t)W;      // one 32-bit division per 4 pixels
        const int cnt = (int)min(4u, n - i0);
        uint32_t packed = 0;
        for (int k = 0; k < cnt; k++) {
            int v = resize_u8_c1(tri, h, w, (int)y, (int)x, mode, scale_y, scale_x);
            if (v > 0 && v < 255) v = 128;
            packed |= (uint32_t)v << (8 * k);
            if (++x == (uint32_t)W) { x = 0; y++; }
        }
        if (cnt == 4) *reinterpret_cast<uint32_t*>(out + i0) = packed;
        else for (int k = 0; k < cnt; k++) out[i0 + k] = (packed >> (8 * k)) & 0xff;
    }
}

extern "C" size_t usc_trimap_ws_bytes(int H, int W, int long_side) {
    int h, w;
    usc_get_target_size(H, W, long_side, 1, &h, &w);
    return ((size_t)h * w + 255) & ~(size_t)255;
}

extern "C" int usc_trimap_u8(const uint8_t* mask, int H, int W, int long_side, int ksize, int iters, uint8_t* out,
                             void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!mask || !out || H <= 0 || W <= 0 || long_side <= 0) return fail(USC_E_ARG, "usc_trimap_u8: bad argument");
    if (ksize < 1 || ksize > kMaxK) return fail(USC_E_UNSUPPORTED, "usc_trimap_u8: ksize %d outside [1,%d]", ksize, kMaxK);
    if (iters < 0) return fail(USC_E_ARG, "usc_trimap_u8: iters < 0");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_trimap_u8: image of %dx%d pixels", H, W);
    int h, w;
    usc_get_target_size(H, W, long_side, 1, &h, &w);
    if (h <= 0 || w <= 0) return fail(USC_E_ARG, "usc_trimap_u8: degenerate low-res size %dx%d", h, w);
    if (!ws || ws_bytes < (size_t)h * w) return fail(USC_E_WORKSPACE, "usc_trimap_u8: workspace %zu < %zu", ws_bytes, (size_t)h * w);
    hipStream_t st = (hipStream_t)stream;
    uint8_t* tri = (uint8_t*)ws;
    if (!bit_se_fits(ksize, iters)) return fail(USC_E_UNSUPPORTED, "usc_trimap_u8: ksize %d x iters %d exceeds the 32-pixel halo", ksize, iters);
    BitSE se = make_bit_se(ksize, iters);
    size_t lds = 5 * (size_t)(kBitH + se.top + se.bottom) * kBitEWW * sizeof(uint32_t);
    if (lds > 60 * 1024) return fail(USC_E_UNSUPPORTED, "usc_trimap_u8: ksize %d x iters %d needs %zu B of LDS", ksize, iters, lds);
    double ifx = 1. / ((double)w / W), ify = 1. / ((double)h / H);
    dim3 grid((w + kBitWW * 32 - 1) / (kBitWW * 32), (h + kBitH - 1) / kBitH);
    hipLaunchKernelGGL(trimap_small_kernel, grid, dim3(kBitThreads), lds, st, mask, H, W, tri, h, w, ify, ifx, se, iters);
    int rc = check_launch("trimap_small_kernel");
    if (rc) return rc;
    ResizeGeom rg = resize_geom(h, w, H, W);
    size_t quads = ((size_t)H * W + 3) / 4;
    hipLaunchKernelGGL(trimap_up_kernel, dim3(grid_for(quads, 256)), dim3(256), 0, st, tri, h, w, out, H, W, rg.mode, rg.scale_y, rg.scale_x);
    return check_launch("trimap_up_kernel");
}

// --------------------------------------------------------------------------------------------
// HSV in-range test and the background-aware trimap (green mode)
// --------------------------------------------------------------------------------------------

// Host copy of the 8-bit BGR -> HSV integer formula (one colour; the kernels use the same formula per pixel).
static void host_bgr2hsv(int b, int g, int r, int* hsv) {
    int v = b > g ? b : g; if (r > v) v = r;
    int vmin = b < g ? b : g; if (r < vmin) vmin = r;
    const int diff = v - vmin;
    const int sdiv = v ? div_round_half_even(255 << 12, v) : 0;
    const int hdiv = diff ? div_round_half_even((180 << 12) / 6, diff) : 0;
    const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
    const int s = (diff * sdiv + (1 << 11)) >> 12;
    int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    h = (h * hdiv + (1 << 11)) >> 12;
    if (h < 0) h += 180;
    hsv[0] = h < 0 ? 0 : (h > 255 ? 255 : h); hsv[1] = s; hsv[2] = v;
}

struct HsvRange { int lo[3], hi[3], half[3]; };

// bgmask (1 byte / pixel: in range or not); optionally the (mask > 0) and fuzzy = (mask > 0) & bgmask counts.
// 4 pixels per lane; per-wave ballots + one atomic pair per workgroup.
__global__ __launch_bounds__(256) void inrange_hsv_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ bg_img,
                                                          HsvRange rg, const uint8_t* __restrict__ mask, size_t n,
                                                          uint8_t* __restrict__ out, unsigned long long* __restrict__ counts) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    __shared__ unsigned int s_cnt[2];
    load_hsv_tables(s_sdiv, s_hdiv);
    if (threadIdx.x < 2) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    unsigned int nm = 0, nf = 0;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t i0 = q * 4;
        const int cnt = (int)min((size_t)4, n - i0);
        uint32_t packed = 0;
        for (int k = 0; k < cnt; k++) {
            const size_t i = i0 + k;
            int h, s, v, lo[3], hi[3];
            bgr2hsv_u8(img[3 * i], img[3 * i + 1], img[3 * i + 2], s_sdiv, s_hdiv, h, s, v);
            if (bg_img) {
                int bh[3];
                bgr2hsv_u8(bg_img[3 * i], bg_img[3 * i + 1], bg_img[3 * i + 2], s_sdiv, s_hdiv, bh[0], bh[1], bh[2]);
#pragma unroll
                for (int c = 0; c < 3; c++) { lo[c] = clampi(bh[c] - rg.half[c], 10, 255); hi[c] = clampi(bh[c] + rg.half[c], 10, 255); }
            } else {
#pragma unroll
                for (int c = 0; c < 3; c++) { lo[c] = rg.lo[c]; hi[c] = rg.hi[c]; }
            }
            const int in = (h >= lo[0] && h <= hi[0] && s >= lo[1] && s <= hi[1] && v >= lo[2] && v <= hi[2]) ? 1 : 0;
            packed |= (uint32_t)in << (8 * k);
            if (mask) { const int m = mask[i] > 0; nm += m; nf += m & in; }
        }
        if (cnt == 4) *reinterpret_cast<uint32_t*>(out + i0) = packed;
        else for (int k = 0; k < cnt; k++) out[i0 + k] = (packed >> (8 * k)) & 0xff;
    }
    if (counts) {
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) { nm += __shfl_down(nm, off, kWave); nf += __shfl_down(nf, off, kWave); }
        if ((threadIdx.x & (kWave - 1)) == 0) { atomicAdd(&s_cnt[0], nm); atomicAdd(&s_cnt[1], nf); }
        __syncthreads();
        if (threadIdx.x == 0) {
            if (s_cnt[0]) atomicAdd(&counts[0], (unsigned long long)s_cnt[0]);
            if (s_cnt[1]) atomicAdd(&counts[1], (unsigned long long)s_cnt[1]);
        }
    }
}

// trimap/agent.py:94-100 on the device: counts decide (uniformly) whether the fuzzy pixels are cut out of the mask
// before the trimap (mode 0) and forced to 128 after it (mode 1).  ratio = double(nf) / double(nm) > 0.1.
__global__ __launch_bounds__(256) void fuzzy_apply_kernel(const uint8_t* __restrict__ mask, const uint8_t* __restrict__ bgmask,
                                                          const unsigned long long* __restrict__ counts, size_t n, int mode,
                                                          uint8_t* __restrict__ io) {
    const unsigned long long nm = counts[0], nf = counts[1];
    const bool trust_mask = nm == 0 || ((double)nf / (double)nm > 0.1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const bool fuzzy = mask[i] > 0 && bgmask[i];
        if (mode == 0) io[i] = (!trust_mask && fuzzy) ? 0 : mask[i];
        else if (!trust_mask && fuzzy) io[i] = 128;
    }
}

static int make_range(const uint8_t* bg_color_host, const int* winsize, HsvRange& rg) {
    for (int c = 0; c < 3; c++) rg.half[c] = winsize[c] / 2;
    if (bg_color_host) {
        int hsv[3];
        host_bgr2hsv(bg_color_host[0], bg_color_host[1], bg_color_host[2], hsv);
        for (int c = 0; c < 3; c++) {
            int lo = hsv[c] - rg.half[c], hi = hsv[c] + rg.half[c];
            rg.lo[c] = lo < 10 ? 10 : (lo > 255 ? 255 : lo);
            rg.hi[c] = hi < 10 ? 10 : (hi > 255 ? 255 : hi);
        }
    }
    return 0;
}

extern "C" int usc_inrange_hsv_u8(const uint8_t* img, const uint8_t* bg_img, const uint8_t* bg_color_host, const int* winsize,
                                  int H, int W, uint8_t* out, usc_stream_t stream) {
    if (!img || !out || !winsize || H <= 0 || W <= 0 || (!bg_img == !bg_color_host))
        return fail(USC_E_ARG, "usc_inrange_hsv_u8: need img, out, winsize and exactly one of bg_img / bg_color_host");
    HsvRange rg;
    make_range(bg_color_host, winsize, rg);
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(inrange_hsv_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream,
                       img, bg_img, rg, (const uint8_t*)nullptr, n, out, (unsigned long long*)nullptr);
    return check_launch("inrange_hsv_kernel");
}

extern "C" size_t usc_trimap_withbg_ws_bytes(int H, int W, int long_side) {
    const size_t n = (((size_t)H * W) + 255) & ~(size_t)255;
    return 256 + 2 * n + usc_trimap_ws_bytes(H, W, long_side);
}

extern "C" int usc_trimap_withbg_u8(const uint8_t* mask, const uint8_t* img, const uint8_t* bg_img, const uint8_t* bg_color_host,
                                    const int* winsize, int H, int W, int long_side, int ksize, int iters, uint8_t* out,
                                    void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!mask || !img || !out || !winsize || H <= 0 || W <= 0 || (!bg_img == !bg_color_host))
        return fail(USC_E_ARG, "usc_trimap_withbg_u8: bad argument");
    if (!ws || ws_bytes < usc_trimap_withbg_ws_bytes(H, W, long_side)) return fail(USC_E_WORKSPACE, "usc_trimap_withbg_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)H * W, npad = (n + 255) & ~(size_t)255;
    unsigned long long* counts = (unsigned long long*)ws;
    uint8_t* bgmask = (uint8_t*)ws + 256;
    uint8_t* ens = bgmask + npad;
    uint8_t* tws = ens + npad;
    HsvRange rg;
    make_range(bg_color_host, winsize, rg);
    if (hipMemsetAsync(counts, 0, 16, st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_trimap_withbg_u8: memset failed");
    hipLaunchKernelGGL(inrange_hsv_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, st, img, bg_img, rg, mask, n, bgmask, counts);
    int rc = check_launch("inrange_hsv_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(fuzzy_apply_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, mask, bgmask, counts, n, 0, ens);
    rc = check_launch("fuzzy_apply_kernel");
    if (rc) return rc;
    rc = usc_trimap_u8(ens, H, W, long_side, ksize, iters, out, tws, ws_bytes - 256 - 2 * npad, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(fuzzy_apply_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, mask, bgmask, counts, n, 1, out);
    return check_launch("fuzzy_apply_kernel");
}

// --------------------------------------------------------------------------------------------
// background fusion + difference mask
// --------------------------------------------------------------------------------------------

// Background fusion, difference threshold, 4x4-ellipse dilation x2 and the gate (bg_offline.py:150-160) in ONE launch (rounds 1-2 ran a
// pointwise kernel that wrote a 1 byte / pixel difference image and a tile kernel that read it back): a workgroup owns 16 rows x 248 pixels.
// It loads 22 rows x 256 pixels (halo: 4 rows / pixels before, 2 after - the reach of the 4x4 ellipse applied twice) of frame, bg_i
// and bg_always as 12-byte groups (lane l of a wave = pixels 4l .. 4l + 3 of the row: 768 contiguous bytes per wave and image),
// computes the fused background and the thresholded grey difference in registers, writes the background of its core pixels,
// packs the 4 difference bits of every lane into 32-pixel words with three lane exchanges, dilates the bit tile in LDS and
// finishes alpha = gray(seg) where the dilation reached.  The 1 byte / pixel difference image never exists.
constexpr int kFgRows = 16, kFgCoreLanes = 62, kFgEWW = 8, kFgThreads = 256;

__device__ __forceinline__ int fuse_bg_px(int bi, int ba, bool has_always, float beta, float beta1) {
    if (!has_always) return bi;
    const float p0 = (float)bi * beta;
    const float p1 = beta1 * (float)ba;
    return (int)(p0 + p1);
}

__global__ __launch_bounds__(kFgThreads) void bgdiff_gate_kernel(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ bg_i,
                                                                 const uint8_t* __restrict__ bg_always, float beta, float beta1, int thr,
                                                                 const uint8_t* __restrict__ seg, int seg_cn, int H, int W,
                                                                 uint8_t* __restrict__ out_bg, uint8_t* __restrict__ alpha, BitSE se, int iters) {
    __shared__ uint32_t lds[3 * (kFgRows + 2 * kMaxK) * kFgEWW];
    const int ER = kFgRows + se.top + se.bottom;
    const int plane = ER * kFgEWW;
    uint32_t *d = lds, *d2 = lds + plane, *inimg = lds + 2 * plane, *unused = nullptr, *unused2 = nullptr;
    const int y0 = blockIdx.y * kFgRows, x0 = blockIdx.x * (kFgCoreLanes * 4);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, nwaves = blockDim.x / kWave;
    const int gx0 = x0 - 4 + 4 * lane;                          // first pixel of this lane's group (may be < 0 in the first tile)
    const bool core_lane = lane >= 1 && lane <= kFgCoreLanes;
    const bool has_always = bg_always != nullptr;
    // Every wave owns the rows r = wave, wave + 4, ...: all of their 12-byte loads are issued before the first is used (one
    // memory round trip per wave instead of one per row: at 2 workgroups per CU nothing else would hide the latency).
    constexpr int kRowsPerWave = (kFgRows + 6 + 3) / 4;        // ER = 16 + 4 + 2 rows (the 4x4 ellipse applied twice): checked by the host
    Px4 F[kRowsPerWave], BI[kRowsPerWave], BA[kRowsPerWave];
    bool fast[kRowsPerWave];
#pragma unroll
    for (int j = 0; j < kRowsPerWave; j++) {
        const int r = wave + nwaves * j, gy = y0 - se.top + r;
        const size_t p = (size_t)gy * W + gx0;
        fast[j] = r < ER && gy >= 0 && gy < H && gx0 >= 0 && gx0 + 4 <= W && (p & 3) == 0;
        if (fast[j]) {
            F[j] = *reinterpret_cast<const Px4*>(frame + 3 * p);
            BI[j] = *reinterpret_cast<const Px4*>(bg_i + 3 * p);
            if (has_always) BA[j] = *reinterpret_cast<const Px4*>(bg_always + 3 * p);
        }
    }
#pragma unroll
    for (int j = 0; j < kRowsPerWave; j++) {
        const int r = wave + nwaves * j;
        if (r >= ER) continue;
        const int gy = y0 - se.top + r;
        uint32_t nib = 0, inb = 0;
        const bool core = core_lane && gy >= y0 && gy < y0 + kFgRows;
        if (fast[j]) {
            const size_t p = (size_t)gy * W + gx0;
            int f[12], bi[12], ba[12], o[12];
            unpack12(F[j], f);
            unpack12(BI[j], bi);
            if (has_always) unpack12(BA[j], ba);
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int dd[3];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const int k = 3 * q + c;
                    const int b = fuse_bg_px(bi[k], has_always ? ba[k] : 0, has_always, beta, beta1);
                    o[k] = b;
                    dd[c] = abs(f[k] - b);
                }
                int g = gray_u8(dd[0], dd[1], dd[2]);
                if (g > thr) g = 255;
                nib |= (uint32_t)(g == 255) << q;
            }
            inb = 0xf;
            if (core) *reinterpret_cast<Px4*>(out_bg + 3 * p) = pack12(o);
        } else if (gy >= 0 && gy < H && gx0 + 3 >= 0 && gx0 < W) {
            for (int q = 0; q < 4; q++) {
                const int gx = gx0 + q;
                if (gx < 0 || gx >= W) continue;
                const size_t i = (size_t)gy * W + gx;
                int dd[3];
                for (int c = 0; c < 3; c++) {
                    const int b = fuse_bg_px(bg_i[3 * i + c], has_always ? bg_always[3 * i + c] : 0, has_always, beta, beta1);
                    if (core) out_bg[3 * i + c] = (uint8_t)b;
                    dd[c] = abs((int)frame[3 * i + c] - b);
                }
                int g = gray_u8(dd[0], dd[1], dd[2]);
                if (g > thr) g = 255;
                nib |= (uint32_t)(g == 255) << q;
                inb |= 1u << q;
            }
        }
        uint32_t vd = nib << (4 * (lane & 7)), vi = inb << (4 * (lane & 7));
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { vd |= __shfl_xor(vd, m, kWave); vi |= __shfl_xor(vi, m, kWave); }
        if ((lane & 7) == 0) { d[r * kFgEWW + (lane >> 3)] = vd; inimg[r * kFgEWW + (lane >> 3)] = vi; }
    }
    __syncthreads();
    bit_morph_passes<false, kFgEWW>(d, d2, unused, unused2, inimg, ER, se, iters);
    if (!core_lane) return;
    for (int r = se.top + wave; r < se.top + kFgRows; r += nwaves) {
        const int gy = y0 - se.top + r;
        if (gy >= H || gx0 >= W) continue;
        const uint32_t db = (d[r * kFgEWW + (lane >> 3)] >> (4 * (lane & 7))) & 0xf;
        const size_t p = (size_t)gy * W + gx0;
        if (gx0 + 4 <= W && (p & 3) == 0) {
            uint32_t out = 0;
            if (seg_cn == 3) {
                int c0[12];
                unpack12(*reinterpret_cast<const Px4*>(seg + 3 * p), c0);
#pragma unroll
                for (int k = 0; k < 4; k++) out |= (((db >> k) & 1) ? (uint32_t)gray_u8(c0[3 * k], c0[3 * k + 1], c0[3 * k + 2]) : 0u) << (8 * k);
            } else {
                const uint32_t s0 = *reinterpret_cast<const uint32_t*>(seg + p);
#pragma unroll
                for (int k = 0; k < 4; k++) out |= (((db >> k) & 1) ? ((s0 >> (8 * k)) & 0xffu) : 0u) << (8 * k);
            }
            *reinterpret_cast<uint32_t*>(alpha + p) = out;
        } else {
            for (int k = 0; k < 4 && gx0 + k < W; k++) {
                const size_t q = p + k;
                const int sv = seg_cn == 3 ? gray_u8(seg[3 * q], seg[3 * q + 1], seg[3 * q + 2]) : seg[q];
                alpha[q] = ((db >> k) & 1) ? (uint8_t)sv : 0;
            }
        }
    }
}

extern "C" size_t usc_bgdiff_ws_bytes(int, int) { return 0; }            // (kept in the ABI: the fused kernel needs no scratch)

extern "C" int usc_bgdiff_mask_u8(const uint8_t* frame, const uint8_t* bg_i, const uint8_t* bg_always,
                                  const uint8_t* seg, int seg_cn, double beta, int thr, int H, int W,
                                  uint8_t* out_bg, uint8_t* out_alpha, void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!frame || !bg_i || !seg || !out_bg || !out_alpha || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_bgdiff_mask_u8: bad argument");
    if (seg_cn != 1 && seg_cn != 3) return fail(USC_E_ARG, "usc_bgdiff_mask_u8: seg_cn %d", seg_cn);
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_bgdiff_mask_u8: image too large");
    hipStream_t st = (hipStream_t)stream;
    (void)ws; (void)ws_bytes;                                              // (the 1 byte / pixel difference image of the two-kernel form is gone)
    // `1 - beta` is formed in double by the reference and only then meets the float32 image.
    BitSE se = make_bit_se(4, 2);
    if (se.top + se.bottom != 6 || se.top > 4) return fail(USC_E_UNSUPPORTED, "usc_bgdiff_mask_u8: structuring element reach");
    dim3 grid((W + kFgCoreLanes * 4 - 1) / (kFgCoreLanes * 4), (H + kFgRows - 1) / kFgRows);
    hipLaunchKernelGGL(bgdiff_gate_kernel, grid, dim3(kFgThreads), 0, st, frame, bg_i, bg_always, (float)beta, (float)(1 - beta), thr, seg, seg_cn,
                       H, W, out_bg, out_alpha, se, 2);
    return check_launch("bgdiff_gate_kernel");
}

// --------------------------------------------------------------------------------------------
// matting pre / post processing
// --------------------------------------------------------------------------------------------

// One work-item per network-input pixel: reflect-pad index -> resized image -> taps in the full-res sources.
__global__ __launch_bounds__(256) void vmat_pre_kernel(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ alpha_pre,
                                                       const uint8_t* __restrict__ trimap, int H, int W, int nh, int nw,
                                                       int inH, int inW, int mode, double scale_y, double scale_x,
                                                       float* __restrict__ out, int layout,
                                                       float m0, float m1, float m2, float s0, float s1, float s2) {
    const uint32_t n = (uint32_t)inH * (uint32_t)inW;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int py = (int)(i / (uint32_t)inW), px = (int)(i - (uint32_t)py * (uint32_t)inW);
        int y = py < nh ? py : 2 * nh - 1 - py;        // BORDER_REFLECT: the edge sample repeats
        int x = px < nw ? px : 2 * nw - 1 - px;
        int bgr[3], tri, ap;
        if (mode == RS_COPY) {
            size_t o = (size_t)y * W + x;
            bgr[0] = frame[3 * o]; bgr[1] = frame[3 * o + 1]; bgr[2] = frame[3 * o + 2];
            tri = trimap[o]; ap = alpha_pre[o];
        } else if (mode == RS_AREA2) {
            size_t o = (size_t)(2 * y) * W + 2 * x;
            const uint8_t* f0 = frame + 3 * o;
            const uint8_t* f1 = f0 + 3 * (size_t)W;
#pragma unroll
            for (int c = 0; c < 3; c++) bgr[c] = (f0[c] + f0[3 + c] + f1[c] + f1[3 + c] + 2) >> 2;
            tri = (trimap[o] + trimap[o + 1] + trimap[o + W] + trimap[o + W + 1] + 2) >> 2;
            ap = (alpha_pre[o] + alpha_pre[o + 1] + alpha_pre[o + W] + alpha_pre[o + W + 1] + 2) >> 2;
        } else {
            Tap tx = linear_tap_x(x, scale_x, W), ty = linear_tap(y, scale_y);
            int x0 = tx.s, x1 = min(x0 + 1, W - 1);
            int y0 = clampi(ty.s, 0, H - 1), y1 = clampi(ty.s + 1, 0, H - 1);
            int a0 = coef11(1.f - tx.f), a1 = coef11(tx.f), b0 = coef11(1.f - ty.f), b1 = coef11(ty.f);
            size_t o00 = (size_t)y0 * W + x0, o01 = (size_t)y0 * W + x1, o10 = (size_t)y1 * W + x0, o11 = (size_t)y1 * W + x1;
#pragma unroll
            for (int c = 0; c < 3; c++)
                bgr[c] = bilinear_u8(frame[3 * o00 + c], frame[3 * o01 + c], frame[3 * o10 + c], frame[3 * o11 + c], a0, a1, b0, b1);
            tri = bilinear_u8(trimap[o00], trimap[o01], trimap[o10], trimap[o11], a0, a1, b0, b1);
            ap = bilinear_u8(alpha_pre[o00], alpha_pre[o01], alpha_pre[o10], alpha_pre[o11], a0, a1, b0, b1);
        }
        float v[7];
        v[0] = ((float)bgr[2] / 255.f - m0) * s0;       // R
        v[1] = ((float)bgr[1] / 255.f - m1) * s1;       // G
        v[2] = ((float)bgr[0] / 255.f - m2) * s2;       // B
        v[3] = (float)((double)ap / 255.);
        v[4] = tri == 0 ? 1.f : 0.f;
        v[5] = (tri > 0 && tri < 255) ? 1.f : 0.f;
        v[6] = tri == 255 ? 1.f : 0.f;
        if (layout == USC_LAYOUT_NCHW) {
#pragma unroll
            for (int c = 0; c < 7; c++) out[(size_t)c * n + i] = v[c];
        } else if (layout == USC_LAYOUT_NHWC8) {
            float4* o = reinterpret_cast<float4*>(out + (size_t)i * 8);
            o[0] = make_float4(v[0], v[1], v[2], v[3]);
            o[1] = make_float4(v[4], v[5], v[6], 0.f);
        } else {
#pragma unroll
            for (int c = 0; c < 7; c++) out[(size_t)i * 7 + c] = v[c];
        }
    }
}

extern "C" int usc_vmat_pre(const uint8_t* frame, const uint8_t* alpha_pre, const uint8_t* trimap, int H, int W,
                            int inH, int inW, float* out7, int layout, usc_stream_t stream) {
    if (!frame || !alpha_pre || !trimap || !out7 || H <= 0 || W <= 0 || inH <= 0 || inW <= 0) return fail(USC_E_ARG, "usc_vmat_pre: bad argument");
    if (layout != USC_LAYOUT_NCHW && layout != USC_LAYOUT_NHWC && layout != USC_LAYOUT_NHWC8) return fail(USC_E_ARG, "usc_vmat_pre: layout %d", layout);
    if (too_big(H, W) || too_big(inH, inW)) return fail(USC_E_UNSUPPORTED, "usc_vmat_pre: image too large");
    int nh, nw;
    usc_pad_resize_geometry(H, W, inH, inW, &nh, &nw);
    if (nh <= 0 || nw <= 0 || nh > inH || nw > inW) return fail(USC_E_ARG, "usc_vmat_pre: resized size %dx%d does not fit %dx%d", nh, nw, inH, inW);
    if (inH - nh > nh || inW - nw > nw) return fail(USC_E_UNSUPPORTED, "usc_vmat_pre: reflect pad larger than the image");
    ResizeGeom rg = resize_geom(H, W, nh, nw);
    static const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
    size_t n = (size_t)inH * inW;
    hipLaunchKernelGGL(vmat_pre_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, frame, alpha_pre, trimap,
                       H, W, nh, nw, inH, inW, rg.mode, rg.scale_y, rg.scale_x, out7, layout,
                       (float)mean[0], (float)mean[1], (float)mean[2],
                       (float)(1 / stdv[0]), (float)(1 / stdv[1]), (float)(1 / stdv[2]));
    return check_launch("vmat_pre_kernel");
}

// Float bilinear back to full resolution + known-region clamp + *255 truncation; 4 pixels per lane.
// (one quad of consecutive pixels i0 .. i0 + cnt - 1 -> the packed alpha bytes; shared by vmat_post_kernel and vmat_post_fg_kernel)
__device__ __forceinline__ uint32_t vmat_post_quad(const float* __restrict__ pred, int rh, int rw, int pitch, const uint8_t* __restrict__ trimap, int W,
                                                   double scale_y, double scale_x, int copy, uint32_t i0, int cnt, float* __restrict__ alpha_f32) {
    uint32_t packed = 0;
    uint32_t yy = i0 / (uint32_t)W, xx = i0 - yy * (uint32_t)W;
    for (int k = 0; k < cnt; k++) {
        const uint32_t i = i0 + k;
        const int y = (int)yy, x = (int)xx;
        if (++xx == (uint32_t)W) { xx = 0; yy++; }
        float a;
        if (copy) a = pred[(size_t)y * pitch + x];
        else {
            Tap tx = linear_tap_x(x, scale_x, rw), ty = linear_tap(y, scale_y);
            int x0 = tx.s, x1 = min(x0 + 1, rw - 1);
            int y0 = clampi(ty.s, 0, rh - 1), y1 = clampi(ty.s + 1, 0, rh - 1);
            float a0 = 1.f - tx.f, a1 = tx.f, b0 = 1.f - ty.f, b1 = ty.f;
            float r0 = pred[(size_t)y0 * pitch + x0] * a0 + pred[(size_t)y0 * pitch + x1] * a1;
            float r1 = pred[(size_t)y1 * pitch + x0] * a0 + pred[(size_t)y1 * pitch + x1] * a1;
            a = r0 * b0 + r1 * b1;
        }
        int t = trimap[i];
        if (t == 0) a = 0.f;
        if (t == 255) a = 1.f;
        if (alpha_f32) alpha_f32[i] = a;
        float s = a * 255.f;
        int u = (int)s;                         // astype(uint8) of a value in [0, 255]: truncation
        packed |= (uint32_t)(u & 0xff) << (8 * k);
    }
    return packed;
}

__global__ __launch_bounds__(256) void vmat_post_kernel(const float* __restrict__ pred, int rh, int rw, int pitch,
                                                        const uint8_t* __restrict__ trimap, int H, int W,
                                                        double scale_y, double scale_x, int copy,
                                                        uint8_t* __restrict__ alpha_u8, float* __restrict__ alpha_f32) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += gridDim.x * blockDim.x) {
        const uint32_t i0 = q * 4;
        const int cnt = (int)min(4u, n - i0);
        const uint32_t packed = vmat_post_quad(pred, rh, rw, pitch, trimap, W, scale_y, scale_x, copy, i0, cnt, alpha_f32);
        if (cnt == 4) *reinterpret_cast<uint32_t*>(alpha_u8 + i0) = packed;
        else for (int k = 0; k < cnt; k++) alpha_u8[i0 + k] = (packed >> (8 * k)) & 0xff;
    }
}

extern "C" int usc_vmat_post(const float* pred, int inH, int inW, const uint8_t* trimap, int H, int W,
                             uint8_t* alpha_u8, float* alpha_f32, usc_stream_t stream) {
    if (!pred || !trimap || !alpha_u8 || H <= 0 || W <= 0 || inH <= 0 || inW <= 0) return fail(USC_E_ARG, "usc_vmat_post: bad argument");
    if (too_big(H, W) || too_big(inH, inW)) return fail(USC_E_UNSUPPORTED, "usc_vmat_post: image too large");
    int rh, rw;
    usc_inv_pad_geometry(inH, inW, H, W, &rh, &rw);
    if (rh <= 0 || rw <= 0 || rh > inH || rw > inW) return fail(USC_E_ARG, "usc_vmat_post: un-padded size %dx%d", rh, rw);
    double scale_x = 1. / ((double)W / rw), scale_y = 1. / ((double)H / rh);
    int copy = (rh == H && rw == W);
    size_t quads = ((size_t)H * W + 3) / 4;
    hipLaunchKernelGGL(vmat_post_kernel, dim3(grid_for(quads, 256)), dim3(256), 0, (hipStream_t)stream, pred, rh, rw, inW, trimap,
                       H, W, scale_y, scale_x, copy, alpha_u8, alpha_f32);
    return check_launch("vmat_post_kernel");
}

// --------------------------------------------------------------------------------------------
// foreground / background extraction in HSV, composite
// --------------------------------------------------------------------------------------------

__device__ __forceinline__ void fg_pixel(int fb, int fg_, int fr, int bb, int bg_, int br, int alpha,
                                         const int* s_sdiv, const int* s_hdiv, int* out /*3*/) {
    int ih[3], bh[3], oh[3];
    bgr2hsv_u8(fb, fg_, fr, s_sdiv, s_hdiv, ih[0], ih[1], ih[2]);
    bgr2hsv_u8(bb, bg_, br, s_sdiv, s_hdiv, bh[0], bh[1], bh[2]);
    float a = (float)alpha / 255.f;
    float w = 1.f - a;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float t = w * (float)bh[c];
        float f = (float)ih[c] - t;
        f = fminf(fmaxf(f, 0.f), 255.f);
        oh[c] = (int)f;
    }
    hsv2bgr_u8(oh[0], oh[1], oh[2], out[0], out[1], out[2]);
}

__global__ __launch_bounds__(256) void get_fg_kernel(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ alpha,
                                                     const uint8_t* __restrict__ bg, int paste_thr, size_t n,
                                                     uint8_t* __restrict__ out_fg, uint8_t* __restrict__ out_bg) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    load_hsv_tables(s_sdiv, s_hdiv);
    __syncthreads();
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += (size_t)gridDim.x * blockDim.x) {
        size_t i0 = q * 4;
        if (i0 + 4 <= n) {
            int f[12], b[12], o[12];
            unpack12(*reinterpret_cast<const Px4*>(frame + 3 * i0), f);
            unpack12(*reinterpret_cast<const Px4*>(bg + 3 * i0), b);
            uint32_t a4 = *reinterpret_cast<const uint32_t*>(alpha + i0);
#pragma unroll
            for (int p = 0; p < 4; p++) {
                int a = (a4 >> (8 * p)) & 0xff;
                if (a < paste_thr) { b[3 * p] = f[3 * p]; b[3 * p + 1] = f[3 * p + 1]; b[3 * p + 2] = f[3 * p + 2]; }
                fg_pixel(f[3 * p], f[3 * p + 1], f[3 * p + 2], b[3 * p], b[3 * p + 1], b[3 * p + 2], a, s_sdiv, s_hdiv, o + 3 * p);
            }
            *reinterpret_cast<Px4*>(out_fg + 3 * i0) = pack12(o);
            if (out_bg) *reinterpret_cast<Px4*>(out_bg + 3 * i0) = pack12(b);
        } else {
            for (size_t i = i0; i < n; i++) {
                int a = alpha[i], o[3], b[3];
                for (int c = 0; c < 3; c++) b[c] = a < paste_thr ? frame[3 * i + c] : bg[3 * i + c];
                fg_pixel(frame[3 * i], frame[3 * i + 1], frame[3 * i + 2], b[0], b[1], b[2], a, s_sdiv, s_hdiv, o);
                for (int c = 0; c < 3; c++) { out_fg[3 * i + c] = (uint8_t)o[c]; if (out_bg) out_bg[3 * i + c] = (uint8_t)b[c]; }
            }
        }
    }
}

extern "C" int usc_get_fg_u8(const uint8_t* frame, const uint8_t* alpha, const uint8_t* bg, int paste_thr, int H, int W,
                             uint8_t* out_fg, uint8_t* out_bg, usc_stream_t stream) {
    if (!frame || !alpha || !bg || !out_fg || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_get_fg_u8: bad argument");
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(get_fg_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, frame, alpha, bg, paste_thr, n, out_fg, out_bg);
    return check_launch("get_fg_kernel");
}

// vmat_post + get_fg of the frame loop's back half as ONE launch (reference vmatting/agent.py:163-173 followed by fgfuncs.py get_fg): the
// alpha quad stays in a register between the two, one launch and one pass over alpha less per frame.  Same device code as the two kernels.
__global__ __launch_bounds__(256) void vmat_post_fg_kernel(const float* __restrict__ pred, int rh, int rw, int pitch,
                                                           const uint8_t* __restrict__ trimap, int H, int W, double scale_y, double scale_x, int copy,
                                                           const uint8_t* __restrict__ frame, const uint8_t* __restrict__ bg, int paste_thr,
                                                           uint8_t* __restrict__ alpha_u8, float* __restrict__ alpha_f32, uint8_t* __restrict__ out_fg) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    load_hsv_tables(s_sdiv, s_hdiv);
    __syncthreads();
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q * 4 < n; q += gridDim.x * blockDim.x) {
        const uint32_t i0 = q * 4;
        const int cnt = (int)min(4u, n - i0);
        const uint32_t a4 = vmat_post_quad(pred, rh, rw, pitch, trimap, W, scale_y, scale_x, copy, i0, cnt, alpha_f32);
        if (cnt == 4) {
            *reinterpret_cast<uint32_t*>(alpha_u8 + i0) = a4;
            int f[12], b[12], o[12];
            unpack12(*reinterpret_cast<const Px4*>(frame + 3 * (size_t)i0), f);
            unpack12(*reinterpret_cast<const Px4*>(bg + 3 * (size_t)i0), b);
#pragma unroll
            for (int p = 0; p < 4; p++) {
                int a = (a4 >> (8 * p)) & 0xff;
                if (a < paste_thr) { b[3 * p] = f[3 * p]; b[3 * p + 1] = f[3 * p + 1]; b[3 * p + 2] = f[3 * p + 2]; }
                fg_pixel(f[3 * p], f[3 * p + 1], f[3 * p + 2], b[3 * p], b[3 * p + 1], b[3 * p + 2], a, s_sdiv, s_hdiv, o + 3 * p);
            }
            *reinterpret_cast<Px4*>(out_fg + 3 * (size_t)i0) = pack12(o);
        } else {
            for (int k = 0; k < cnt; k++) {
                const size_t i = (size_t)i0 + k;
                const int a = (a4 >> (8 * k)) & 0xff;
                alpha_u8[i] = (uint8_t)a;
                int o[3], b[3];
                for (int c = 0; c < 3; c++) b[c] = a < paste_thr ? frame[3 * i + c] : bg[3 * i + c];
                fg_pixel(frame[3 * i], frame[3 * i + 1], frame[3 * i + 2], b[0], b[1], b[2], a, s_sdiv, s_hdiv, o);
                for (int c = 0; c < 3; c++) out_fg[3 * i + c] = (uint8_t)o[c];
            }
        }
    }
}

extern "C" int usc_vmat_post_fg(const float* pred, int inH, int inW, const uint8_t* trimap, const uint8_t* frame, const uint8_t* bg, int paste_thr,
                                int H, int W, uint8_t* alpha_u8, float* alpha_f32, uint8_t* out_fg, usc_stream_t stream) {
    if (!pred || !trimap || !frame || !bg || !alpha_u8 || !out_fg || H <= 0 || W <= 0 || inH <= 0 || inW <= 0) return fail(USC_E_ARG, "usc_vmat_post_fg: bad argument");
    if (too_big(H, W) || too_big(inH, inW)) return fail(USC_E_UNSUPPORTED, "usc_vmat_post_fg: image too large");
    int rh, rw;
    usc_inv_pad_geometry(inH, inW, H, W, &rh, &rw);
    if (rh <= 0 || rw <= 0 || rh > inH || rw > inW) return fail(USC_E_ARG, "usc_vmat_post_fg: un-padded size %dx%d", rh, rw);
    double scale_x = 1. / ((double)W / rw), scale_y = 1. / ((double)H / rh);
    int copy = (rh == H && rw == W);
    size_t quads = ((size_t)H * W + 3) / 4;
    hipLaunchKernelGGL(vmat_post_fg_kernel, dim3(grid_for(quads, 256)), dim3(256), 0, (hipStream_t)stream, pred, rh, rw, inW, trimap,
                       H, W, scale_y, scale_x, copy, frame, bg, paste_thr, alpha_u8, alpha_f32, out_fg);
    return check_launch("vmat_post_fg_kernel");
}

__global__ __launch_bounds__(256) void get_bg_kernel(const uint8_t* __restrict__ alpha, const uint8_t* __restrict__ bg, size_t n,
                                                     uint8_t* __restrict__ out) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    load_hsv_tables(s_sdiv, s_hdiv);
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int bh[3], oh[3], o[3];
        bgr2hsv_u8(bg[3 * i], bg[3 * i + 1], bg[3 * i + 2], s_sdiv, s_hdiv, bh[0], bh[1], bh[2]);
        float a = (float)alpha[i] / 255.f;
        float w = 1.f - a;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float f = w * (float)bh[c];
            f = fminf(fmaxf(f, 0.f), 255.f);
            oh[c] = (int)f;
        }
        hsv2bgr_u8(oh[0], oh[1], oh[2], o[0], o[1], o[2]);
        out[3 * i] = (uint8_t)o[0]; out[3 * i + 1] = (uint8_t)o[1]; out[3 * i + 2] = (uint8_t)o[2];
    }
}

extern "C" int usc_get_bg_u8(const uint8_t* alpha, const uint8_t* bg, int H, int W, uint8_t* out, usc_stream_t stream) {
    if (!alpha || !bg || !out || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_get_bg_u8: bad argument");
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(get_bg_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, alpha, bg, n, out);
    return check_launch("get_bg_kernel");
}

__global__ __launch_bounds__(256) void blend_kernel(const uint8_t* __restrict__ fg, const uint8_t* __restrict__ mask, int mask_cn,
                                                    const uint8_t* __restrict__ bg, size_t n, uint8_t* __restrict__ out) {
    __shared__ double s_m[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_m[i] = (double)i / 255.;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            double m = s_m[mask_cn == 3 ? mask[3 * i + c] : mask[i]];
            double r = (double)fg[3 * i + c] * m + (double)bg[3 * i + c] * (1 - m);
            out[3 * i + c] = (uint8_t)(int)r;
        }
    }
}

extern "C" int usc_blend_u8(const uint8_t* fg, const uint8_t* mask, int mask_cn, const uint8_t* bg, int H, int W,
                            uint8_t* out, usc_stream_t stream) {
    if (!fg || !mask || !bg || !out || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_blend_u8: bad argument");
    if (mask_cn != 1 && mask_cn != 3) return fail(USC_E_ARG, "usc_blend_u8: mask_cn %d", mask_cn);
    size_t n = (size_t)H * W;
    hipLaunchKernelGGL(blend_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, fg, mask, mask_cn, bg, n, out);
    return check_launch("blend_kernel");
}

// --------------------------------------------------------------------------------------------
// exist_foreground reduction
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_ge_kernel(const uint8_t* __restrict__ mask, size_t n, int thr, unsigned long long* out) {
    unsigned int local = 0;
    const size_t nvec = n / 16;
    const uint4* v = reinterpret_cast<const uint4*>(mask);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        uint4 q = v[i];
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int b = 0; b < 4; b++) local += (int)((w[k] >> (8 * b)) & 0xff) >= thr;
    }
    if (blockIdx.x == 0)
        for (size_t i = nvec * 16 + threadIdx.x; i < n; i += blockDim.x) local += mask[i] >= thr;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) local += __shfl_down(local, off, kWave);
    __shared__ unsigned int s_part[256 / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) s_part[threadIdx.x / kWave] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int i = 0; i < 256 / kWave; i++) t += s_part[i];
        if (t) atomicAdd(out, t);
    }
}

extern "C" int usc_count_ge_u8(const uint8_t* mask, size_t n, int thr, unsigned long long* out_count, usc_stream_t stream) {
    if (!mask || !out_count || n == 0) return fail(USC_E_ARG, "usc_count_ge_u8: bad argument");
    if (((uintptr_t)mask & 15) != 0) return fail(USC_E_ARG, "usc_count_ge_u8: mask must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out_count, 0, sizeof(unsigned long long), st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_count_ge_u8: memset failed");
    hipLaunchKernelGGL(count_ge_kernel, dim3(grid_for((n + 15) / 16, 256)), dim3(256), 0, st, mask, n, thr, out_count);
    return check_launch("count_ge_kernel");
}

// --------------------------------------------------------------------------------------------
// generic 8-bit resize / affine warp / composite (replace mode: tools/replace/replace.py:59-82)
// --------------------------------------------------------------------------------------------

// dst (oh x ow) = window [y_off, y_off + oh) x [x_off, x_off + ow) of the virtual resize of src to (dh x dw).
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int cn,
                                                        uint8_t* __restrict__ dst, int oh, int ow, int y_off, int x_off,
                                                        int interp, int mode, double scale_y, double scale_x, double ify, double ifx) {
    const uint32_t n = (uint32_t)oh * (uint32_t)ow;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)ow), x = (int)(i - (uint32_t)y * (uint32_t)ow);
        for (int c = 0; c < cn; c++)
            dst[(size_t)i * cn + c] = (uint8_t)resize_sample_u8(src, sh, sw, cn, c, y + y_off, x + x_off, interp, mode, scale_y, scale_x, ify, ifx);
    }
}

static int launch_resize(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, int oh, int ow, int y_off, int x_off,
                         int interp, hipStream_t st, double inv_scale = 0.) {
    ResizeGeom rg = resize_geom(sh, sw, dh, dw);
    if (inv_scale > 0) rg.scale_x = rg.scale_y = 1. / inv_scale;       // cv2.resize(src, None, fx, fy): the caller's factors map the coordinates
    const double ifx = 1. / ((double)dw / sw), ify = 1. / ((double)dh / sh);
    hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for((size_t)oh * ow, 256)), dim3(256), 0, st, src, sh, sw, cn, dst, oh, ow, y_off, x_off,
                       interp, rg.mode, rg.scale_y, rg.scale_x, ify, ifx);
    return check_launch("resize_u8_kernel");
}

extern "C" int usc_resize_u8(const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw, int interp, usc_stream_t stream) {
    if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || (cn != 1 && cn != 3) || interp < 0 || interp > 2)
        return fail(USC_E_ARG, "usc_resize_u8: bad argument");
    if (too_big(sh, sw) || too_big(dh, dw)) return fail(USC_E_UNSUPPORTED, "usc_resize_u8: image too large");
    return launch_resize(src, sh, sw, cn, dst, dh, dw, dh, dw, 0, 0, interp, (hipStream_t)stream);
}

extern "C" int usc_rescale_fg_u8(const uint8_t* src, int h, int w, int cn, double scale, uint8_t* dst, usc_stream_t stream) {
    if (!src || !dst || h <= 0 || w <= 0 || (cn != 1 && cn != 3) || !(scale >= 1.0)) return fail(USC_E_ARG, "usc_rescale_fg_u8: bad argument (scale >= 1)");
    const int nh = (int)lrint(h * scale), nw = (int)lrint(w * scale);
    const int h_off = (int)((nh - h) / 2.0), w_off = (int)((nw - w) / 2.0);
    return launch_resize(src, h, w, cn, dst, nh, nw, h, w, h_off, w_off, 2, (hipStream_t)stream, scale);
}

// fgfuncs.py:172-214 composite_fgbg (extend = False), the background resize fused in.
__global__ __launch_bounds__(256) void composite_kernel(const uint8_t* __restrict__ fg, const uint8_t* __restrict__ alpha, int fh, int fw,
                                                        const uint8_t* __restrict__ bg, int bh, int bw, int nbh, int nbw, int top, int left,
                                                        int mode, double scale_y, double scale_x, uint8_t* __restrict__ out) {
    const uint32_t n = (uint32_t)fh * (uint32_t)fw;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)fw), x = (int)(i - (uint32_t)y * (uint32_t)fw);
        double a = (double)alpha[i] / 255.;
        if (a > 0.9) a = 1;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int b = resize_sample_u8(bg, bh, bw, 3, c, y + top, x + left, 1, mode, scale_y, scale_x, 0., 0.);
            double v = (double)fg[(size_t)i * 3 + c] + (double)b * (1 - a);
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
            out[(size_t)i * 3 + c] = (uint8_t)(int)v;
        }
    }
}

extern "C" int usc_composite_fgbg_u8(const uint8_t* fg, const uint8_t* alpha, int fh, int fw, const uint8_t* bg, int bh, int bw,
                                     uint8_t* out, usc_stream_t stream) {
    if (!fg || !alpha || !bg || !out || fh <= 0 || fw <= 0 || bh <= 0 || bw <= 0) return fail(USC_E_ARG, "usc_composite_fgbg_u8: bad argument");
    int nbw, nbh;
    if ((double)fh / fw > (double)bh / bw) { nbh = fh; nbw = (int)((double)bw * nbh / bh); }
    else { nbw = fw; nbh = (int)((double)bh * nbw / bw); }
    int left = nbw / 2 - fw / 2, top = nbh / 2 - fh / 2;
    if (left < 0) left = 0;
    if (top < 0) top = 0;
    if (top + fh > nbh || left + fw > nbw) return fail(USC_E_UNSUPPORTED, "usc_composite_fgbg_u8: resized background does not cover the foreground");
    ResizeGeom rg = resize_geom(bh, bw, nbh, nbw);
    hipLaunchKernelGGL(composite_kernel, dim3(grid_for((size_t)fh * fw, 256)), dim3(256), 0, (hipStream_t)stream, fg, alpha, fh, fw, bg, bh, bw,
                       nbh, nbw, top, left, rg.mode, rg.scale_y, rg.scale_x, out);
    return check_launch("composite_kernel");
}
