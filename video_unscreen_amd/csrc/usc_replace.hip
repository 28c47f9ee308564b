// usc_replace.hip - the per-frame body of the replace tool as ONE launch (reference tools/replace/replace.py:59-82):
//
//     dst_fg_shift   = rescale_fg(shift_fg(dst_fg,   dx, dy), 1.2)          unscreen/utils/imgprocess.py:55-64, :40-52
//     dst_mask_shift = rescale_fg(shift_fg(dst_mask, dx, dy), 1.2)
//     res     = u8(dst_fg_shift * m + bg * (1 - m)),  m = dst_mask_shift / 255 in float64        replace.py:74-76
//     compare = concatenate((adaptive_resize(src_image, dst_fg), res), axis=1)                   replace.py:64, :82
//
// The chain of stand-alone kernels (usc_warp_affine_u8 x2, usc_rescale_fg_u8 x2, usc_blend_u8, a concatenation, usc_resize_u8) moves
// ~60 bytes per pixel through HBM; the algorithm needs 21 (fg, mask, bg, source frame in; res and the side-by-side image out).
// Here a workgroup owns 16 x 64 output pixels: it evaluates cv2.warpAffine (AB_BITS 10 / INTER_BITS 5 coordinates, 15-bit
// weights, zero border) for the window of shifted pixels its cubic taps touch - with a zoom > 1 that window is SMALLER than the
// tile - into LDS as 8-byte pixels (fg b g r 0, mask b g r 0), then runs cv2.resize(INTER_CUBIC)'s 11-bit fixed-point 4 x 4 taps from
// LDS (one ds_read_b64 per tap), the float64 blend and the two stores.  Same integer arithmetic as the stand-alone kernels:
// bit-identical results (tests/test_gpu_replace.py).  HBM-bound.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_host.h"

namespace {
using namespace usc;

struct Affine { double m[6]; };

void invert_affine(const double* in, double* M) {       // cv::warpAffine without WARP_INVERSE_MAP
    for (int i = 0; i < 6; i++) M[i] = in[i];
    double D = M[0] * M[4] - M[1] * M[3];
    D = D != 0 ? 1. / D : 0;
    const double A11 = M[4] * D, A22 = M[0] * D;
    M[0] = A11; M[1] *= -D;
    M[3] *= -D; M[4] = A22;
    const double b1 = -M[0] * M[2] - M[1] * M[5];
    const double b2 = -M[3] * M[2] - M[4] * M[5];
    M[2] = b1; M[5] = b2;
}

__device__ __forceinline__ void warp_coord(const Affine& A, int x, int y, int& sx, int& sy, int& fx, int& fy) {
    const int X0 = __double2int_rn((A.m[1] * y + A.m[2]) * 1024) + 16, Y0 = __double2int_rn((A.m[4] * y + A.m[5]) * 1024) + 16;
    const int X = (X0 + __double2int_rn(A.m[0] * x * 1024)) >> 5, Y = (Y0 + __double2int_rn(A.m[3] * x * 1024)) >> 5;
    sx = min(max(X >> 5, -32768), 32767);
    sy = min(max(Y >> 5, -32768), 32767);
    fx = X & 31; fy = Y & 31;
}

constexpr int kTH = 16, kTW = 64;                 // output tile
constexpr int kWH = kTH + 3, kWW = kTW + 3;       // largest window of shifted pixels a tile's cubic taps touch (zoom >= 1)

// 8 / 2 bytes at any byte address (global memory takes unaligned accesses): the 2 x 1 neighbourhood of a 3-channel / 1-channel tap pair
struct __attribute__((packed, aligned(1))) Raw8 { uint32_t lo, hi; };
struct __attribute__((packed, aligned(1))) Raw2 { uint16_t v; };

// One shifted pixel: cv2.warpAffine of fg (3 channels) and mask (MASK_CN channels) at (y, x) -> x = fg b | g << 8 | r << 16, y = mask likewise.
// The two taps of a source row are 6 (2) adjacent bytes: one 8-byte (2-byte) load per row and array when both columns are inside the image.
template <int MASK_CN>
__device__ __forceinline__ uint2 warp_px(const uint8_t* __restrict__ fg, const uint8_t* __restrict__ mask, int H, int W, uint32_t nbytes, const Affine& A, int y, int x) {
    int sx, sy, fx, fy;
    warp_coord(A, x, y, sx, sy, fx, fy);
    const int w[4] = {(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32};
    int acc[6] = {0, 0, 0, 0, 0, 0};
    if (sx >= 0 && sx + 1 < W) {
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int yy = sy + r;
            if (yy < 0 || yy >= H) continue;
            const uint32_t p = (uint32_t)yy * (uint32_t)W + (uint32_t)sx;
            uint32_t lo, hi;
            if (3 * p + 8 <= nbytes) { const Raw8 q = *reinterpret_cast<const Raw8*>(fg + 3 * p); lo = q.lo; hi = q.hi; }
            else { lo = fg[3 * p] | (fg[3 * p + 1] << 8) | (fg[3 * p + 2] << 16) | ((uint32_t)fg[3 * p + 3] << 24); hi = fg[3 * p + 4] | (fg[3 * p + 5] << 8); }
            const int w0 = w[2 * r], w1 = w[2 * r + 1];
            acc[0] += (int)(lo & 0xff) * w0 + (int)(lo >> 24) * w1;
            acc[1] += (int)((lo >> 8) & 0xff) * w0 + (int)(hi & 0xff) * w1;
            acc[2] += (int)((lo >> 16) & 0xff) * w0 + (int)((hi >> 8) & 0xff) * w1;
            if (MASK_CN == 3) {
                if (3 * p + 8 <= nbytes) { const Raw8 q = *reinterpret_cast<const Raw8*>(mask + 3 * p); lo = q.lo; hi = q.hi; }
                else { lo = mask[3 * p] | (mask[3 * p + 1] << 8) | (mask[3 * p + 2] << 16) | ((uint32_t)mask[3 * p + 3] << 24); hi = mask[3 * p + 4] | (mask[3 * p + 5] << 8); }
                acc[3] += (int)(lo & 0xff) * w0 + (int)(lo >> 24) * w1;
                acc[4] += (int)((lo >> 8) & 0xff) * w0 + (int)(hi & 0xff) * w1;
                acc[5] += (int)((lo >> 16) & 0xff) * w0 + (int)((hi >> 8) & 0xff) * w1;
            } else {
                const uint32_t m2 = reinterpret_cast<const Raw2*>(mask + p)->v;
                acc[3] += (int)(m2 & 0xff) * w0 + (int)(m2 >> 8) * w1;
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int xx = sx + (t & 1), yy = sy + (t >> 1);
            if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
                const uint32_t p = (uint32_t)yy * (uint32_t)W + (uint32_t)xx;
                acc[0] += fg[3 * p] * w[t]; acc[1] += fg[3 * p + 1] * w[t]; acc[2] += fg[3 * p + 2] * w[t];
                if (MASK_CN == 3) { acc[3] += mask[3 * p] * w[t]; acc[4] += mask[3 * p + 1] * w[t]; acc[5] += mask[3 * p + 2] * w[t]; }
                else acc[3] += mask[p] * w[t];
            }
        }
    }
    int v[6];
#pragma unroll
    for (int c = 0; c < 6; c++) v[c] = min(max((acc[c] + (1 << 14)) >> 15, 0), 255);
    if (MASK_CN == 1) v[4] = v[5] = v[3];
    return make_uint2((uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16), (uint32_t)v[3] | ((uint32_t)v[4] << 8) | ((uint32_t)v[5] << 16));
}

template <int MASK_CN>
__global__ __launch_bounds__(256) void replace_compose_kernel(const uint8_t* __restrict__ fg, const uint8_t* __restrict__ mask, int H, int W, Affine A,
                                                              double scale_y, double scale_x, int h_off, int w_off,
                                                              const uint8_t* __restrict__ bg, const uint8_t* __restrict__ src, int sh, int sw,
                                                              int src_mode, double src_sy, double src_sx,
                                                              uint8_t* __restrict__ res, uint8_t* __restrict__ cmp, int aligned) {
    __shared__ uint2 s_px[kWH * kWW];
    __shared__ double s_m[256];
    const int tiles_x = (W + kTW - 1) / kTW;
    const int ty0 = (blockIdx.x / tiles_x) * kTH, tx0 = (blockIdx.x % tiles_x) * kTW;
    const int ty1 = min(ty0 + kTH, H) - 1, tx1 = min(tx0 + kTW, W) - 1;         // last output row / column of the tile
    // window of shifted pixels: rows / columns (first tap - 1) .. (last tap + 2), clamped the way the resize clamps its taps
    const int wy0 = clampi(linear_tap(ty0 + h_off, scale_y).s - 1, 0, H - 1), wy1 = clampi(linear_tap(ty1 + h_off, scale_y).s + 2, 0, H - 1);
    const int wx0 = clampi(linear_tap(tx0 + w_off, scale_x).s - 1, 0, W - 1), wx1 = clampi(linear_tap(tx1 + w_off, scale_x).s + 2, 0, W - 1);
    const int wh = wy1 - wy0 + 1, ww = wx1 - wx0 + 1;                           // <= kWH, kWW for any zoom >= 1
    const uint32_t nbytes = 3u * (uint32_t)H * (uint32_t)W;
    s_m[threadIdx.x] = (double)threadIdx.x / 255.;
    for (int i = threadIdx.x; i < wh * ww; i += 256) {
        const int r = i / ww, c = i - r * ww;
        s_px[r * kWW + c] = warp_px<MASK_CN>(fg, mask, H, W, nbytes, A, wy0 + r, wx0 + c);     // nbytes bounds the 8-byte loads of the 3-channel arrays (a 1-channel mask is read two bytes at a time: sx + 1 < W keeps them inside)
    }
    __syncthreads();
    // 4 output pixels per lane along x: 16 lanes cover a tile row
    const int y = ty0 + (int)(threadIdx.x >> 4), xq = tx0 + 4 * (int)(threadIdx.x & 15);
    if (y > ty1 || xq > tx1) return;
    const int npx = min(4, tx1 - xq + 1);
    const bool quad = npx == 4 && (W & 3) == 0 && aligned;       // 12-byte packed accesses (dword-aligned tensors whose rows are dword-aligned: W % 4 == 0)
    const uint32_t p0 = (uint32_t)y * (uint32_t)W + (uint32_t)xq;
    int bgv[12], sv[12];
    if (quad) {                                        // issued before the taps are read: in flight during the arithmetic
        unpack12(*reinterpret_cast<const Px4*>(bg + 3 * p0), bgv);
        if (cmp && src_mode == RS_COPY) unpack12(*reinterpret_cast<const Px4*>(src + 3 * p0), sv);
    } else {
        for (int k = 0; k < 12; k++) bgv[k] = bg[3 * (p0 + min(k / 3, npx - 1)) + k % 3];
    }
    if (cmp && !(quad && src_mode == RS_COPY)) {
#pragma unroll
        for (int k = 0; k < 12; k++) sv[k] = resize_sample_u8(src, sh, sw, 3, k % 3, y, min(xq + k / 3, tx1), 1, src_mode, src_sy, src_sx, 0., 0.);
    }
    const Tap tyy = linear_tap(y + h_off, scale_y);
    int cy[4];
    cubic_coeffs(tyy.f, cy);
    int rows[4];
#pragma unroll
    for (int j = 0; j < 4; j++) rows[j] = (clampi(tyy.s - 1 + j, 0, H - 1) - wy0) * kWW;
    int out[12];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int x = min(xq + k, tx1);
        const Tap txx = linear_tap(x + w_off, scale_x);
        int cx[4];
        cubic_coeffs(txx.f, cx);
        int cols[4];
#pragma unroll
        for (int i = 0; i < 4; i++) cols[i] = clampi(txx.s - 1 + i, 0, W - 1) - wx0;
        int acc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int row[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint2 p = s_px[rows[j] + cols[i]];
                row[0] += (int)(p.x & 0xff) * cx[i]; row[1] += (int)((p.x >> 8) & 0xff) * cx[i]; row[2] += (int)((p.x >> 16) & 0xff) * cx[i];
                row[3] += (int)(p.y & 0xff) * cx[i]; row[4] += (int)((p.y >> 8) & 0xff) * cx[i]; row[5] += (int)((p.y >> 16) & 0xff) * cx[i];
            }
            // |row| <= 4 * 255 * 2331 < 2^23 and |cy| < 2^12: the 24-bit multiplier gives the exact 32-bit product (a full-rate
            // instruction; the 32 x 32 multiply the compiler must otherwise assume runs at a quarter of the rate)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[c] += __mul24(row[c], cy[j]);
        }
        int v[6];
#pragma unroll
        for (int c = 0; c < 6; c++) v[c] = min(max((acc[c] + (1 << 21)) >> 22, 0), 255);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double m = s_m[v[3 + c]];
            const double r = (double)v[c] * m + (double)bgv[3 * k + c] * (1 - m);
            out[3 * k + c] = (int)r;
        }
    }
    if (quad) {
        const Px4 o = pack12(out);
        if (res) *reinterpret_cast<Px4*>(res + 3 * p0) = o;
        if (cmp) {
            uint8_t* row = cmp + 3 * ((size_t)y * 2 * W);
            *reinterpret_cast<Px4*>(row + 3 * (size_t)xq) = pack12(sv);
            *reinterpret_cast<Px4*>(row + 3 * (size_t)(W + xq)) = o;
        }
    } else {
        for (int k = 0; k < 3 * npx; k++) {
            if (res) res[3 * (size_t)p0 + k] = (uint8_t)out[k];
            if (cmp) {
                uint8_t* row = cmp + 3 * ((size_t)y * 2 * W);
                row[3 * (size_t)xq + k] = (uint8_t)sv[k];
                row[3 * (size_t)(W + xq) + k] = (uint8_t)out[k];
            }
        }
    }
}

}  // namespace

extern "C" int usc_replace_compose_u8(const uint8_t* fg, const uint8_t* mask, int mask_cn, int H, int W, const double* M2x3_host, double scale,
                                      const uint8_t* bg, const uint8_t* src, int sh, int sw, uint8_t* res, uint8_t* compare,
                                      usc_stream_t stream) {
    if (!fg || !mask || !bg || !M2x3_host || H <= 0 || W <= 0 || (mask_cn != 1 && mask_cn != 3) || !(scale >= 1.0) || (!res && !compare))
        return fail(USC_E_ARG, "usc_replace_compose_u8: bad argument (scale >= 1, at least one output)");
    if (compare && (!src || sh <= 0 || sw <= 0)) return fail(USC_E_ARG, "usc_replace_compose_u8: the side-by-side output needs the source frame");
    if ((long long)H * W * 6 >= (1ll << 32) || (src && too_big(sh, sw))) return fail(USC_E_UNSUPPORTED, "usc_replace_compose_u8: image too large (32-bit byte offsets)");
    const int aligned = (((uintptr_t)bg | (uintptr_t)res | (uintptr_t)compare | (uintptr_t)src) & 3) == 0;      // else byte accesses (e.g. image k of a packed stack of odd-sized images)
    Affine A;
    invert_affine(M2x3_host, A.m);
    // rescale_fg (imgprocess.py:46-51): cv2.resize(fx = fy = scale, INTER_CUBIC) -> size = round(h * scale), then the centre crop
    const int nh = (int)lrint(H * scale), nw = (int)lrint(W * scale);
    const int h_off = (int)((nh - H) / 2.0), w_off = (int)((nw - W) / 2.0);
    ResizeGeom rz = resize_geom(H, W, nh, nw);
    rz.scale_x = rz.scale_y = 1. / scale;              // cv2.resize(img, None, fx = fy = scale): the factor itself maps the coordinates
    ResizeGeom rs = {RS_COPY, 1., 1.};
    if (compare) rs = resize_geom(sh, sw, H, W);
    const int tiles = ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
    if (mask_cn == 3)
        hipLaunchKernelGGL(replace_compose_kernel<3>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, fg, mask, H, W, A, rz.scale_y, rz.scale_x, h_off, w_off,
                           bg, src, sh, sw, rs.mode, rs.scale_y, rs.scale_x, res, compare, aligned);
    else
        hipLaunchKernelGGL(replace_compose_kernel<1>, dim3(tiles), dim3(256), 0, (hipStream_t)stream, fg, mask, H, W, A, rz.scale_y, rz.scale_x, h_off, w_off,
                           bg, src, sh, sw, rs.mode, rs.scale_y, rs.scale_x, res, compare, aligned);
    return check_launch("replace_compose_kernel");
}
