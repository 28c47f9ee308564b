// usc_stm.hip - STMAgent (reference unscreen/stm/agent.py) around the space-time memory network:
//   pre     : pad_resize (fixed-point bilinear + BORDER_REFLECT) of frame and first-frame mask, imnormalize, mask > 127 ->
//             object / background planes; one lane per network pixel, NHWC with 8 channels (R, G, B, m, o, 0, 0, 0) so the
//             7x7 stems read it through usc_im2col_nhwc_f32 (3 channels for the query encoder, 5 for the memory encoder);
//   pack    : a row-major (K, N) matrix living in HBM (the memory keys / values) -> the [K/4][N][4] operand layout of the MFMA
//             convolution kernel, so the two matrix products of the memory read (stm/model.py:137-156) are 1x1 layers of
//             usc_conv_* whose "weights" are refreshed on the device every frame;
//   softmax : row softmax of the (query position) x (memory position) affinity, one wave per row, in place;
//   post    : x4 bilinear + softmax + soft aggregation + softmax (per network pixel), then inv_pad_resize + argmax.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

__device__ __forceinline__ int reflect_any(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

// exp / log restated with IEEE basic operations only (Cephes expf / logf): identical bits on the host oracle and here.
__device__ __forceinline__ float det_expf(float x) {
    if (x < -87.f) return 0.f;
    if (x > 88.f) return INFINITY;
    const float kf = rintf(x * 1.44269504088896341f);
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

__device__ __forceinline__ float det_logf(float x) {
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
    const float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    z = m + y;
    z = z + 0.693359375f * fe;
    return z;
}

struct NcTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ NcTap nc_tap(int d, int in, float scale) {
    float s = scale * ((float)d + 0.5f) - 0.5f;
    if (s < 0.f) s = 0.f;
    NcTap t;
    t.i0 = (int)s;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = s - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

}  // namespace

// --------------------------------------------------------------------------------------------
// pre (stm/agent.py:62-98)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stm_pre_kernel(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ mask, int H, int W,
                                                      int nh, int nw, int mode, double scale_y, double scale_x, int in_h, int in_w,
                                                      float* __restrict__ out) {
    const float mean[3] = {(float)0.485, (float)0.456, (float)0.406};
    const float stdinv[3] = {(float)(1 / 0.229), (float)(1 / 0.224), (float)(1 / 0.225)};
    const uint32_t n = (uint32_t)in_h * (uint32_t)in_w;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)in_w), x = (int)(i - (uint32_t)y * (uint32_t)in_w);
        const int py = reflect_any(y, nh), px = reflect_any(x, nw);
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int u = resize_sample_u8(frame, H, W, 3, 2 - c, py, px, 1, mode, scale_y, scale_x, 0., 0.);
            float t = (float)u / 255.f;
            t = t - mean[c];
            v[c] = t * stdinv[c];
        }
        float m = 0.f, o = 0.f;
        if (mask) {
            const int fg = resize_sample_u8(mask, H, W, 1, 0, py, px, 1, mode, scale_y, scale_x, 0., 0.) > 127;
            m = (float)fg;
            o = (float)(1 - fg);
        }
        float4* dst = reinterpret_cast<float4*>(out) + 2 * (size_t)i;
        dst[0] = make_float4(v[0], v[1], v[2], m);
        dst[1] = make_float4(o, 0.f, 0.f, 0.f);
    }
}

extern "C" int usc_stm_pre(const uint8_t* frame, const uint8_t* mask, int H, int W, int in_h, int in_w, float* out, usc_stream_t stream) {
    if (!frame || !out || H <= 0 || W <= 0 || in_h <= 0 || in_w <= 0) return fail(USC_E_ARG, "usc_stm_pre: bad argument");
    if (too_big(H, W) || too_big(in_h, in_w)) return fail(USC_E_UNSUPPORTED, "usc_stm_pre: image too large");
    int nh, nw;
    usc_pad_resize_geometry(H, W, in_h, in_w, &nh, &nw);
    const ResizeGeom rg = resize_geom(H, W, nh, nw);
    hipLaunchKernelGGL(stm_pre_kernel, dim3(grid_for((size_t)in_h * in_w, 256)), dim3(256), 0, (hipStream_t)stream, frame, mask, H, W, nh, nw,
                       rg.mode, rg.scale_y, rg.scale_x, in_h, in_w, out);
    return check_launch("stm_pre_kernel");
}

// --------------------------------------------------------------------------------------------
// operand packing for device-resident "weights": dst[k/4][n][k%4] = src[k*sk + n*sn] (k < K, n < N), zero elsewhere
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kn_kernel(const float* __restrict__ src, int K, int N, long long sk, long long sn, int Kpad,
                                                      int Npad, float* __restrict__ dst) {
    const uint32_t n = (uint32_t)(Kpad / 4) * (uint32_t)Npad;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int k4 = (int)(i / (uint32_t)Npad), col = (int)(i - (uint32_t)k4 * (uint32_t)Npad);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * k4 + j;
            v[j] = (k < K && col < N) ? src[(long long)k * sk + (long long)col * sn] : 0.f;
        }
        reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int usc_pack_kn_f32(const float* src, int K, int N, long long sk, long long sn, int Kpad, int Npad, float* dst,
                               usc_stream_t stream) {
    if (!src || !dst || K <= 0 || N <= 0 || Kpad < K || Npad < N || Kpad % 32 || Npad % 32)
        return fail(USC_E_ARG, "usc_pack_kn_f32: bad argument (Kpad >= K, Npad >= N, both multiples of 32)");
    if ((long long)(Kpad / 4) * Npad >= (1ll << 32)) return fail(USC_E_UNSUPPORTED, "usc_pack_kn_f32: too large");
    hipLaunchKernelGGL(pack_kn_kernel, dim3(grid_for((size_t)(Kpad / 4) * Npad, 256)), dim3(256), 0, (hipStream_t)stream, src, K, N, sk, sn,
                       Kpad, Npad, dst);
    return check_launch("pack_kn_kernel");
}

// --------------------------------------------------------------------------------------------
// row softmax of x / div over the first `cols` of `ld` columns (columns beyond stay / become 0), one wave per row, in place
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_softmax_kernel(float* __restrict__ x, int rows, int cols, int ld, float div) {
    const int lane = threadIdx.x & (kWave - 1);
    const int row = blockIdx.x * (256 / kWave) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* r = x + (size_t)row * ld;
    float m = -INFINITY;
    for (int c = lane; c < cols; c += kWave) m = fmaxf(m, r[c] / div);
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, kWave));
    float s = 0.f;
    for (int c = lane; c < cols; c += kWave) {
        const float e = expf(r[c] / div - m);
        r[c] = e;
        s += e;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
    for (int c = lane; c < cols; c += kWave) r[c] = r[c] / s;
    for (int c = cols + lane; c < ld; c += kWave) r[c] = 0.f;
}

extern "C" int usc_row_softmax_f32(float* x, int rows, int cols, int ld, float div, usc_stream_t stream) {
    if (!x || rows <= 0 || cols <= 0 || ld < cols || !(div > 0.f)) return fail(USC_E_ARG, "usc_row_softmax_f32: bad argument");
    hipLaunchKernelGGL(row_softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, div);
    return check_launch("row_softmax_kernel");
}

// --------------------------------------------------------------------------------------------
// post (stm/model.py:135-136, 195-219; stm/agent.py:164, 172-201)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stm_prob_kernel(const float* __restrict__ p2, int lh, int lw, int cbuf, int in_h, int in_w,
                                                       float* __restrict__ prob) {
    const uint32_t n = (uint32_t)in_h * (uint32_t)in_w;
    const float rh = (float)lh / (float)in_h, rw = (float)lw / (float)in_w;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)in_w), x = (int)(i - (uint32_t)y * (uint32_t)in_w);
        const NcTap ty = nc_tap(y, lh, rh), tx = nc_tap(x, lw, rw);
        const float2 v00 = *reinterpret_cast<const float2*>(p2 + ((size_t)ty.i0 * lw + tx.i0) * cbuf);
        const float2 v01 = *reinterpret_cast<const float2*>(p2 + ((size_t)ty.i0 * lw + tx.i1) * cbuf);
        const float2 v10 = *reinterpret_cast<const float2*>(p2 + ((size_t)ty.i1 * lw + tx.i0) * cbuf);
        const float2 v11 = *reinterpret_cast<const float2*>(p2 + ((size_t)ty.i1 * lw + tx.i1) * cbuf);
        const float a = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
        const float b = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
        const float m = a > b ? a : b;
        const float ea = det_expf(a - m), eb = det_expf(b - m);
        const float ps = eb / (ea + eb);
        const float lo = 1e-7f, hi = 1.f - 1e-7f;
        float e0 = 1.f - ps, e1 = ps;
        e0 = e0 < lo ? lo : (e0 > hi ? hi : e0);
        e1 = e1 < lo ? lo : (e1 > hi ? hi : e1);
        const float l0 = det_logf(e0 / (1.f - e0)), l1 = det_logf(e1 / (1.f - e1));
        const float mm = l0 > l1 ? l0 : l1;
        const float x0 = det_expf(l0 - mm), x1 = det_expf(l1 - mm);
        const float s = x0 + x1;
        reinterpret_cast<float2*>(prob)[i] = make_float2(x0 / s, x1 / s);
    }
}

__global__ __launch_bounds__(256) void stm_mask_kernel(const float* __restrict__ prob, int in_w, int rh, int rw, int H, int W, int same,
                                                       double scale_y, double scale_x, uint8_t* __restrict__ out) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        float v0, v1;
        if (same) {
            const float2 p = reinterpret_cast<const float2*>(prob)[(size_t)y * in_w + x];
            v0 = p.x; v1 = p.y;
        } else {
            const Tap tx = linear_tap_x(x, scale_x, rw), ty = linear_tap(y, scale_y);
            const int x0 = tx.s, x1 = min(x0 + 1, rw - 1);
            const int y0 = clampi(ty.s, 0, rh - 1), y1 = clampi(ty.s + 1, 0, rh - 1);
            const float a0 = 1.f - tx.f, a1 = tx.f, b0 = 1.f - ty.f, b1 = ty.f;
            const float2 p00 = reinterpret_cast<const float2*>(prob)[(size_t)y0 * in_w + x0];
            const float2 p01 = reinterpret_cast<const float2*>(prob)[(size_t)y0 * in_w + x1];
            const float2 p10 = reinterpret_cast<const float2*>(prob)[(size_t)y1 * in_w + x0];
            const float2 p11 = reinterpret_cast<const float2*>(prob)[(size_t)y1 * in_w + x1];
            const float t0 = p00.x * a0 + p01.x * a1, u0 = p10.x * a0 + p11.x * a1;
            const float t1 = p00.y * a0 + p01.y * a1, u1 = p10.y * a0 + p11.y * a1;
            v0 = t0 * b0 + u0 * b1;
            v1 = t1 * b0 + u1 * b1;
        }
        out[i] = v1 > v0 ? 255 : 0;
    }
}

static int launch_stm_mask(const float* prob, int in_h, int in_w, int H, int W, uint8_t* out, hipStream_t st) {
    int rh, rw;
    usc_inv_pad_geometry(in_h, in_w, H, W, &rh, &rw);
    const int same = (rh == H && rw == W);
    const double scale_x = 1. / ((double)W / rw), scale_y = 1. / ((double)H / rh);
    hipLaunchKernelGGL(stm_mask_kernel, dim3(grid_for((size_t)H * W, 256)), dim3(256), 0, st, prob, in_w, rh, rw, H, W, same, scale_y, scale_x, out);
    return check_launch("stm_mask_kernel");
}

extern "C" int usc_stm_mask_from_prob(const float* prob, int in_h, int in_w, int H, int W, uint8_t* out, usc_stream_t stream) {
    if (!prob || !out || in_h <= 0 || in_w <= 0 || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_stm_mask_from_prob: bad argument");
    if (too_big(H, W) || too_big(in_h, in_w)) return fail(USC_E_UNSUPPORTED, "usc_stm_mask_from_prob: image too large");
    return launch_stm_mask(prob, in_h, in_w, H, W, out, (hipStream_t)stream);
}

extern "C" int usc_stm_post(const float* p2, int lh, int lw, int cbuf, int in_h, int in_w, int H, int W, uint8_t* out, float* prob,
                            usc_stream_t stream) {
    if (!p2 || !out || !prob || lh <= 0 || lw <= 0 || cbuf < 2 || cbuf % 2 || in_h <= 0 || in_w <= 0 || H <= 0 || W <= 0)
        return fail(USC_E_ARG, "usc_stm_post: bad argument (prob: in_h*in_w*2 floats of device scratch / output)");
    if (too_big(H, W) || too_big(in_h, in_w)) return fail(USC_E_UNSUPPORTED, "usc_stm_post: image too large");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stm_prob_kernel, dim3(grid_for((size_t)in_h * in_w, 256)), dim3(256), 0, st, p2, lh, lw, cbuf, in_h, in_w, prob);
    return launch_stm_mask(prob, in_h, in_w, H, W, out, st);
}
