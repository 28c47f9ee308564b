// usc_device.h - device-side arithmetic shared by the pixel kernels (gfx950 only).
//
// Every routine here is the integer / single-rounded-float arithmetic of one OpenCV 8-bit primitive the
// reference calls (see include/unscreen_hip.h for the reference file:line of each stage).  The file is
// compiled with -ffp-contract=off: a*b+c must round twice, as NumPy / torch-CPU / OpenCV's scalar paths do.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace usc {

constexpr int kWave = 64;

// ---- compile-time tables of cv::RGB2HSV_b (hsv_shift = 12): round-half-even of (255<<12)/i and (180<<12)/(6 i)
constexpr int div_round_half_even(int a, int b) {
    int q = a / b, r = a % b;
    return (2 * r > b) ? q + 1 : ((2 * r == b) ? q + (q & 1) : q);
}
struct HsvTables {
    int sdiv[256];
    int hdiv[256];
    constexpr HsvTables() : sdiv{}, hdiv{} {
        sdiv[0] = 0; hdiv[0] = 0;
        for (int i = 1; i < 256; i++) {
            sdiv[i] = div_round_half_even(255 << 12, i);
            hdiv[i] = div_round_half_even((180 << 12) / 6, i);
        }
    }
};
__device__ const HsvTables g_hsv_tables{};

// Block-cooperative copy of the two 256-entry tables into LDS (2 KiB); callers __syncthreads() after.
__device__ __forceinline__ void load_hsv_tables(int* s_sdiv, int* s_hdiv) {
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        s_sdiv[i] = g_hsv_tables.sdiv[i];
        s_hdiv[i] = g_hsv_tables.hdiv[i];
    }
}

// cv2.COLOR_BGR2GRAY, 8-bit (15-bit fixed point).
__device__ __forceinline__ int gray_u8(int b, int g, int r) {
    return (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15;
}

// cv2.COLOR_BGR2HSV, 8-bit, H in [0,180).
__device__ __forceinline__ void bgr2hsv_u8(int b, int g, int r, const int* s_sdiv, const int* s_hdiv,
                                           int& h, int& s, int& v) {
    v = max(b, max(g, r));
    int vmin = min(b, min(g, r));
    int diff = v - vmin;
    int vr = (v == r) ? -1 : 0, vg = (v == g) ? -1 : 0;
    s = (diff * s_sdiv[v] + (1 << 11)) >> 12;
    int hh = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
    hh = (hh * s_hdiv[diff] + (1 << 11)) >> 12;
    hh += hh < 0 ? 180 : 0;
    h = min(max(hh, 0), 255);
}

__device__ __forceinline__ int sat_round_u8(float x) {      // saturate_cast<uchar>(float): rint, clamp
    int i = (int)rintf(x);
    return min(max(i, 0), 255);
}

// cv2.COLOR_HSV2BGR, 8-bit in / 8-bit out (float six-sector formula, hscale = 6/180).
__device__ __forceinline__ void hsv2bgr_u8(int hi, int si, int vi, int& b, int& g, int& r) {
    float h = (float)hi, s = (float)si * (1.f / 255.f), v = (float)vi * (1.f / 255.f);
    float fb, fg, fr;
    if (s == 0.f) {
        fb = fg = fr = v;
    } else {
        h *= (6.f / 180.f);
        if (h >= 6.f) h -= 6.f;                 // h <= 255*6/180 = 8.5: one step is enough
        int sector = (int)floorf(h);
        h -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
        float t0 = v;
        float t1 = v * (1.f - s);
        float t2 = v * (1.f - s * h);
        float t3 = v * (1.f - s * (1.f - h));
        // sector_data = {{1,3,0},{1,0,2},{3,0,1},{0,2,1},{0,1,3},{2,1,0}} (b, g, r)
        switch (sector) {
            case 0:  fb = t1; fg = t3; fr = t0; break;
            case 1:  fb = t1; fg = t0; fr = t2; break;
            case 2:  fb = t3; fg = t0; fr = t1; break;
            case 3:  fb = t0; fg = t2; fr = t1; break;
            case 4:  fb = t0; fg = t1; fr = t3; break;
            default: fb = t2; fg = t1; fr = t0; break;
        }
    }
    b = sat_round_u8(fb * 255.f);
    g = sat_round_u8(fg * 255.f);
    r = sat_round_u8(fr * 255.f);
}

// One axis of cv::resize INTER_LINEAR: source index and fraction for destination coordinate d.
// scale = 1 / (dn / sn) is computed by the host in double and passed in.
struct Tap { int s; float f; };
__device__ __forceinline__ Tap linear_tap(int d, double scale) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    return Tap{s, f - (float)s};
}
// x-axis variant: weights zeroed at the edges (cv::resize xmin / xmax handling).
__device__ __forceinline__ Tap linear_tap_x(int d, double scale, int sn) {
    Tap t = linear_tap(d, scale);
    if (t.s < 0) { t.s = 0; t.f = 0.f; }
    if (t.s + 1 >= sn) { t.s = sn - 1; t.f = 0.f; }
    return t;
}
__device__ __forceinline__ int coef11(float w) {             // saturate_cast<short>(w * 2048)
    int i = (int)rintf(w * 2048.f);
    return min(max(i, -32768), 32767);
}

// 8-bit bilinear sample, cv::resize fixed-point arithmetic.  p00..p11 are the four taps.
__device__ __forceinline__ int bilinear_u8(int p00, int p01, int p10, int p11, int a0, int a1, int b0, int b1) {
    int r0 = p00 * a0 + p01 * a1;
    int r1 = p10 * a0 + p11 * a1;
    return (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// Unaligned-safe packed 4-pixel access: 12 bytes of BGR at a 4-byte aligned address.
struct __attribute__((packed, aligned(4))) Px4 { uint32_t w[3]; };

__device__ __forceinline__ void unpack12(const Px4& p, int* c /*12*/) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        c[4 * k + 0] = p.w[k] & 0xff;
        c[4 * k + 1] = (p.w[k] >> 8) & 0xff;
        c[4 * k + 2] = (p.w[k] >> 16) & 0xff;
        c[4 * k + 3] = p.w[k] >> 24;
    }
}
__device__ __forceinline__ Px4 pack12(const int* c) {
    Px4 p;
#pragma unroll
    for (int k = 0; k < 3; k++)
        p.w[k] = (uint32_t)c[4 * k] | ((uint32_t)c[4 * k + 1] << 8) | ((uint32_t)c[4 * k + 2] << 16) |
                 ((uint32_t)c[4 * k + 3] << 24);
    return p;
}

// How cv::resize(INTER_LINEAR, 8-bit) treats a (src -> dst) size pair (chosen on the host, usc_host.h resize_geom).
enum ResizeMode { RS_COPY = 0, RS_AREA2 = 1, RS_LINEAR = 2 };

__device__ __forceinline__ void cubic_coeffs(float x, int* c) {      // cv::interpolateCubic, A = -0.75, 11-bit fixed point
    const float A = -0.75f;
    const float f0 = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    const float f1 = ((A + 2) * x - (A + 3)) * x * x + 1;
    const float f2 = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    const float f3 = 1.f - f0 - f1 - f2;
    c[0] = coef11(f0); c[1] = coef11(f1); c[2] = coef11(f2); c[3] = coef11(f3);
}

// One channel value of cv::resize at destination (y, x) of a virtual (dh x dw) image.  interp: 0 nearest, 1 linear
// (mode = copy / 2x2 box / fixed-point bilinear as OpenCV would choose), 2 cubic.
__device__ __forceinline__ int resize_sample_u8(const uint8_t* __restrict__ s, int sh, int sw, int cn, int c, int y, int x,
                                                int interp, int mode, double scale_y, double scale_x, double ify, double ifx) {
    if (interp == 0) {
        const int sy = min((int)floor(y * ify), sh - 1), sx = min((int)floor(x * ifx), sw - 1);
        return s[((size_t)sy * sw + sx) * cn + c];
    }
    if (interp == 1) {
        if (mode == RS_COPY) return s[((size_t)y * sw + x) * cn + c];
        if (mode == RS_AREA2) {
            const uint8_t* p = s + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
            return (p[0] + p[cn] + p[(size_t)sw * cn] + p[(size_t)sw * cn + cn] + 2) >> 2;
        }
        Tap tx = linear_tap_x(x, scale_x, sw), ty = linear_tap(y, scale_y);
        const int x0 = tx.s, x1 = min(x0 + 1, sw - 1);
        const int y0 = clampi(ty.s, 0, sh - 1), y1 = clampi(ty.s + 1, 0, sh - 1);
        return bilinear_u8(s[((size_t)y0 * sw + x0) * cn + c], s[((size_t)y0 * sw + x1) * cn + c],
                           s[((size_t)y1 * sw + x0) * cn + c], s[((size_t)y1 * sw + x1) * cn + c],
                           coef11(1.f - tx.f), coef11(tx.f), coef11(1.f - ty.f), coef11(ty.f));
    }
    const Tap tx = linear_tap(x, scale_x), ty = linear_tap(y, scale_y);
    int cx[4], cy[4];
    cubic_coeffs(tx.f, cx);
    cubic_coeffs(ty.f, cy);
    int acc = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int yy = clampi(ty.s - 1 + j, 0, sh - 1);
        int row = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) row += s[((size_t)yy * sw + clampi(tx.s - 1 + i, 0, sw - 1)) * cn + c] * cx[i];
        acc += row * cy[j];
    }
    return min(max((acc + (1 << 21)) >> 22, 0), 255);
}

}  // namespace usc
