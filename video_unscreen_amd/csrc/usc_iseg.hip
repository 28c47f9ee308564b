// usc_iseg.hip - pixel stages and the depthwise convolution of ISegAgent's no-BRS forward (reference unscreen/iseg/agent.py:92-119,
// iseg/brs/brs_predictor.py:35-106, iseg/brs/models/distmap.py:16-22, 70-124, iseg/brs/models/basic_blocks.py:89-121).
//   pre   : pad_resize(frame) -> imnormalize -> click distance maps (DistMaps) -> rgb_conv (1x1 5->8, LeakyReLU 0.2, BN, 1x1 8->3),
//           one lane per network pixel, written as the image AND its mirror (the predictor's flip test-time augmentation mirrors
//           image and clicks, so the second batch entry is the mirror of the first: one evaluation feeds both).
//   dw    : depthwise 3x3 (groups = channels, padding 1) on NHWC float32 - 9 multiply-adds per output, not a dense contraction: a
//           float4-per-lane direct kernel, HBM-bound, kept off the matrix cores.
//   prob  : bilinear (align_corners) up of the (flip-paired) logits to the network input size, mean of the pair, sigmoid.
//   mask  : inv_pad_resize(prob) (cv2.resize on float32) > prob_thresh -> 0 / 255.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

__device__ __forceinline__ int reflect_any(int p, int n) {      // cv::borderInterpolate(BORDER_REFLECT)
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

// exp restated with IEEE basic operations only (Cephes expf): identical bits on the host oracle and here.
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

struct IsegPreParams {
    float py[2 * USC_ISEG_MAX_CLICKS], px[2 * USC_ISEG_MAX_CLICKS];    // first P positive, then P negative; (-1, -1) = absent
    int P;
    float w1[8][5], b1[8], w2[3][8], b2[3];                            // rgb_conv, BN folded into the second layer
    float mean[3], stdinv[3];
};

__global__ __launch_bounds__(256) void iseg_pre_kernel(const uint8_t* __restrict__ frame, int H, int W, int in_h, int in_w, int nh, int nw,
                                                       int mode, double scale_y, double scale_x, IsegPreParams prm, int flip,
                                                       float* __restrict__ out) {
    const uint32_t n = (uint32_t)in_h * (uint32_t)in_w;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)in_w), x = (int)(i - (uint32_t)y * (uint32_t)in_w);
        const int ry = reflect_any(y, nh), rx = reflect_any(x, nw);
        float v[5];
#pragma unroll
        for (int c = 0; c < 3; c++) {                                   // channel c of the RGB image = BGR channel 2 - c
            const int p = resize_sample_u8(frame, H, W, 3, 2 - c, ry, rx, 1, mode, scale_y, scale_x, 0., 0.);
            float f = (float)p / 255.f;
            f = f - prm.mean[c];
            v[c] = f * prm.stdinv[c];
        }
        const float fy = (float)y, fx = (float)x;
#pragma unroll
        for (int g = 0; g < 2; g++) {                                   // distmap.py:80-114, spatial_scale = 1, norm_radius = 260
            float best = 1e6f;
            for (int k = 0; k < prm.P; k++) {
                const float qy = prm.py[g * prm.P + k], qx = prm.px[g * prm.P + k];
                float d = 1e6f;
                if (!(fmaxf(qy, qx) < 0.f)) {
                    const float dy = (fy - qy) / 260.f, dx = (fx - qx) / 260.f;
                    d = dy * dy + dx * dx;
                }
                best = fminf(best, d);
            }
            float u = sqrtf(best) * 2.f;                                // tanh(u), u >= 0, as 1 - 2 / (exp(2u) + 1)
            u = u > 20.f ? 20.f : u;
            v[3 + g] = 1.f - 2.f / (det_expf(2.f * u) + 1.f);
        }
        float h[8];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            float s = prm.b1[j];
#pragma unroll
            for (int c = 0; c < 5; c++) s = s + prm.w1[j][c] * v[c];
            h[j] = s > 0.f ? s : s * 0.2f;
        }
        float4 o;
        float r[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float s = prm.b2[k];
#pragma unroll
            for (int j = 0; j < 8; j++) s = s + prm.w2[k][j] * h[j];
            r[k] = s;
        }
        o.x = r[0]; o.y = r[1]; o.z = r[2]; o.w = 0.f;
        reinterpret_cast<float4*>(out)[i] = o;
        if (flip) reinterpret_cast<float4*>(out)[(size_t)n + (size_t)y * in_w + (in_w - 1 - x)] = o;
    }
}

// depthwise 3x3, padding 1, NHWC, `batch` images stacked along y; weights [9][C] (tap-major: ky * 3 + kx), taps summed in that order
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float* __restrict__ src, int h, int w, int C, int batch, const float* __restrict__ wt,
                                                        float* __restrict__ dst) {
    const int c4n = C / 4;
    const uint32_t n = (uint32_t)batch * h * w * c4n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c4 = (int)(i % (uint32_t)c4n);
        const uint32_t p = i / (uint32_t)c4n;
        const int x = (int)(p % (uint32_t)w);
        const uint32_t row = p / (uint32_t)w;
        const int y = (int)(row % (uint32_t)h);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            const int yy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const int xx = x + kx - 1;
                if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                const float4 a = reinterpret_cast<const float4*>(src + ((size_t)(row - y + yy) * w + xx) * C)[c4];
                const float4 k = reinterpret_cast<const float4*>(wt + (size_t)(ky * 3 + kx) * C)[c4];
                acc.x = acc.x + a.x * k.x; acc.y = acc.y + a.y * k.y; acc.z = acc.z + a.z * k.z; acc.w = acc.w + a.w * k.w;
            }
        }
        reinterpret_cast<float4*>(dst + (size_t)p * C)[c4] = acc;
    }
}

// torch upsample_bilinear2d(align_corners=True) of channel 0 of image `img` at (y, x)
__device__ __forceinline__ float up_ac(const float* __restrict__ lg, int lh, int lw, int cbuf, int img, int y, int x, float rh, float rw) {
    const float* __restrict__ p = lg + (size_t)img * lh * lw * cbuf;
    const float h1r = rh * y, w1r = rw * x;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < lh - 1 ? 1 : 0, w1p = w1 < lw - 1 ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    const float v00 = p[((size_t)h1 * lw + w1) * cbuf], v01 = p[((size_t)h1 * lw + w1 + w1p) * cbuf];
    const float v10 = p[((size_t)(h1 + h1p) * lw + w1) * cbuf], v11 = p[((size_t)(h1 + h1p) * lw + w1 + w1p) * cbuf];
    const float top = w0l * v00 + w1l * v01, bot = w0l * v10 + w1l * v11;
    return h0l * top + h1l * bot;
}

// brs_predictor.py:35-51 + transforms.py:28-29, 61-66: prediction = sigmoid(0.5 * (up(l0) + mirror(up(l1)))) (the transforms are undone
// in reverse order: the flip pair is averaged on the LOGITS, then the sigmoid); without flip sigmoid(up(l0)).
__global__ __launch_bounds__(256) void iseg_prob_kernel(const float* __restrict__ lg, int lh, int lw, int cbuf, int in_h, int in_w, int flip,
                                                        float rh, float rw, float* __restrict__ prob) {
    const uint32_t n = (uint32_t)in_h * (uint32_t)in_w;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)in_w), x = (int)(i - (uint32_t)y * (uint32_t)in_w);
        float v = up_ac(lg, lh, lw, cbuf, 0, y, x, rh, rw);
        if (flip) v = 0.5f * (v + up_ac(lg, lh, lw, cbuf, 1, y, in_w - 1 - x, rh, rw));
        prob[i] = 1.f / (1.f + det_expf(-v));
    }
}

// agent.py:116-119: cv2.resize of the un-padded float32 probability to the frame size, > prob_thresh (float32 compare) -> 255 / 0
__global__ __launch_bounds__(256) void iseg_mask_kernel(const float* __restrict__ prob, int rh, int rw, int pitch, int H, int W, int mode,
                                                        double scale_y, double scale_x, float thr, uint8_t* __restrict__ mask) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        float v;
        if (mode == RS_COPY) v = prob[(size_t)y * pitch + x];
        else if (mode == RS_AREA2) {
            const float* p = prob + (size_t)(2 * y) * pitch + 2 * x;
            float s = p[0] + p[1];
            s = s + p[pitch];
            s = s + p[pitch + 1];
            v = s * 0.25f;
        } else {
            const Tap tx = linear_tap_x(x, scale_x, rw), ty = linear_tap(y, scale_y);
            const int x0 = tx.s, x1 = min(x0 + 1, rw - 1);
            const int y0 = clampi(ty.s, 0, rh - 1), y1 = clampi(ty.s + 1, 0, rh - 1);
            const float a0 = 1.f - tx.f, a1 = tx.f, b0 = 1.f - ty.f, b1 = ty.f;
            const float r0 = prob[(size_t)y0 * pitch + x0] * a0 + prob[(size_t)y0 * pitch + x1] * a1;
            const float r1 = prob[(size_t)y1 * pitch + x0] * a0 + prob[(size_t)y1 * pitch + x1] * a1;
            v = r0 * b0 + r1 * b1;
        }
        mask[i] = v > thr ? 255 : 0;
    }
}

}  // namespace

extern "C" int usc_iseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, const float* points_yx_host, int P,
                            const float* rgb_params_host, int flip, float* out, usc_stream_t stream) {
    if (!frame || !out || !points_yx_host || !rgb_params_host || H <= 0 || W <= 0 || in_h <= 0 || in_w <= 0 || P < 1 || P > USC_ISEG_MAX_CLICKS)
        return fail(USC_E_ARG, "usc_iseg_pre: bad argument (1 <= P <= %d)", USC_ISEG_MAX_CLICKS);
    if (too_big(H, W) || too_big(2 * in_h, in_w)) return fail(USC_E_UNSUPPORTED, "usc_iseg_pre: image too large");
    static const double mean[3] = {0.485, 0.456, 0.406}, std[3] = {0.229, 0.224, 0.225};
    IsegPreParams prm;
    memset(&prm, 0, sizeof(prm));
    prm.P = P;
    for (int k = 0; k < 2 * P; k++) { prm.py[k] = points_yx_host[2 * k]; prm.px[k] = points_yx_host[2 * k + 1]; }
    const float* q = rgb_params_host;                  // w1 (8 x 5), b1 (8), w2 (3 x 8), b2 (3)
    memcpy(prm.w1, q, sizeof(prm.w1)); q += 40;
    memcpy(prm.b1, q, sizeof(prm.b1)); q += 8;
    memcpy(prm.w2, q, sizeof(prm.w2)); q += 24;
    memcpy(prm.b2, q, sizeof(prm.b2));
    for (int c = 0; c < 3; c++) { prm.mean[c] = (float)mean[c]; prm.stdinv[c] = (float)(1 / std[c]); }
    int nh, nw;
    usc_pad_resize_geometry(H, W, in_h, in_w, &nh, &nw);
    if (nh <= 0 || nw <= 0) return fail(USC_E_ARG, "usc_iseg_pre: empty resized image");
    const ResizeGeom rg = resize_geom(H, W, nh, nw);
    hipLaunchKernelGGL(iseg_pre_kernel, dim3(grid_for((size_t)in_h * in_w, 256)), dim3(256), 0, (hipStream_t)stream, frame, H, W, in_h, in_w, nh, nw,
                       rg.mode, rg.scale_y, rg.scale_x, prm, flip ? 1 : 0, out);
    return check_launch("iseg_pre_kernel");
}

extern "C" int usc_dwconv3x3_nhwc_f32(const float* src, int h, int w, int C, int batch, const float* w9c, float* dst, usc_stream_t stream) {
    if (!src || !dst || !w9c || h <= 0 || w <= 0 || C <= 0 || C % 4 || batch < 1 || src == dst) return fail(USC_E_ARG, "usc_dwconv3x3_nhwc_f32: bad argument (C % 4 == 0)");
    if ((long long)batch * h * w * (C / 4) >= (1ll << 31)) return fail(USC_E_UNSUPPORTED, "usc_dwconv3x3_nhwc_f32: tensor too large");
    hipLaunchKernelGGL(dwconv3x3_kernel, dim3(grid_for((size_t)batch * h * w * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, h, w, C, batch, w9c, dst);
    return check_launch("dwconv3x3_kernel");
}

extern "C" int usc_iseg_post(const float* logits, int lh, int lw, int cbuf, int in_h, int in_w, int flip, int H, int W, double prob_thresh,
                             float* prob, uint8_t* mask, usc_stream_t stream) {
    if (!logits || !prob || lh <= 0 || lw <= 0 || cbuf <= 0 || in_h <= 0 || in_w <= 0) return fail(USC_E_ARG, "usc_iseg_post: bad argument");
    if (too_big(in_h, in_w) || (mask && too_big(H, W))) return fail(USC_E_UNSUPPORTED, "usc_iseg_post: image too large");
    hipStream_t st = (hipStream_t)stream;
    const float rh = in_h > 1 ? (float)(lh - 1) / (in_h - 1) : 0.f, rw = in_w > 1 ? (float)(lw - 1) / (in_w - 1) : 0.f;
    hipLaunchKernelGGL(iseg_prob_kernel, dim3(grid_for((size_t)in_h * in_w, 256)), dim3(256), 0, st, logits, lh, lw, cbuf, in_h, in_w, flip ? 1 : 0, rh, rw, prob);
    int rc = check_launch("iseg_prob_kernel");
    if (rc || !mask) return rc;
    if (H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_iseg_post: bad frame size");
    int ch, cw;
    usc_inv_pad_geometry(in_h, in_w, H, W, &ch, &cw);
    if (ch <= 0 || cw <= 0 || ch > in_h || cw > in_w) return fail(USC_E_ARG, "usc_iseg_post: empty un-padded region");
    const ResizeGeom rg = resize_geom(ch, cw, H, W);
    hipLaunchKernelGGL(iseg_mask_kernel, dim3(grid_for((size_t)H * W, 256)), dim3(256), 0, st, prob, ch, cw, in_w, H, W, rg.mode, rg.scale_y, rg.scale_x,
                       (float)prob_thresh, mask);
    return check_launch("iseg_mask_kernel");
}
