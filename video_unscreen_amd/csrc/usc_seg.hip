// usc_seg.hip - pre / post processing of the human-parsing segmentation stage
// (reference unscreen/binseg/agent_human.py:262-312, transform_logits :58-73, get_affine_transform :91-125).
//
//   pre : affine warp of the BGR frame into the 473 x 473 network input (cv2.warpAffine fixed-point coordinates and
//         15-bit bilinear weights, zero border) fused with imnormalize; output NHWC with 4 channels (R, G, B, 0) so the
//         first convolution reads 16-byte pixels.
//   post: (1) torch Upsample(bilinear, align_corners=True) of the 20 class logits to 473 x 473, channel-interleaved
//         so the 20 values of one tap are 80 contiguous bytes; (2) one lane per frame pixel: inverse affine warp of
//         all 20 channels (float bilinear, zero border) + argmax + (> 0 -> 255).  The reference materialises a
//         H x W x 20 float tensor on the CPU for this (166 MB at 1080p, 663 MB at 4K); here it never exists.
// All coordinate math follows cv::WarpAffineInvoker (AB_BITS = 10, INTER_BITS = 5) in double / int on the device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/unscreen_hip.h"

namespace {

thread_local char g_seg_err[256] = "";
int seg_fail(int code, const char* msg) { snprintf(g_seg_err, sizeof(g_seg_err), "%s", msg); return code; }

struct Affine { double m[6]; };

// cv::getAffineTransform = cv::solve(DECOMP_LU) of the 6x6 system, in double.
void get_affine_transform(const double* src, const double* dst, double* M) {
    double a[36], b[6];
    for (int i = 0; i < 3; i++) {
        const int j = i * 12, k = i * 12 + 6;
        a[j] = a[k + 3] = src[2 * i];
        a[j + 1] = a[k + 4] = src[2 * i + 1];
        a[j + 2] = a[k + 5] = 1;
        a[j + 3] = a[j + 4] = a[j + 5] = 0;
        a[k] = a[k + 1] = a[k + 2] = 0;
        b[2 * i] = dst[2 * i];
        b[2 * i + 1] = dst[2 * i + 1];
    }
    for (int i = 0; i < 6; i++) {
        int piv = i;
        for (int j = i + 1; j < 6; j++) if (fabs(a[j * 6 + i]) > fabs(a[piv * 6 + i])) piv = j;
        if (piv != i) {
            for (int j = i; j < 6; j++) { const double t = a[i * 6 + j]; a[i * 6 + j] = a[piv * 6 + j]; a[piv * 6 + j] = t; }
            const double t = b[i]; b[i] = b[piv]; b[piv] = t;
        }
        const double d = -1 / a[i * 6 + i];
        for (int j = i + 1; j < 6; j++) {
            const double alpha = a[j * 6 + i] * d;
            for (int k = i + 1; k < 6; k++) a[j * 6 + k] += alpha * a[i * 6 + k];
            b[j] += alpha * b[i];
        }
    }
    for (int i = 5; i >= 0; i--) {
        double s = b[i];
        for (int k = i + 1; k < 6; k++) s -= a[i * 6 + k] * b[k];
        b[i] = s / a[i * 6 + i];
    }
    for (int i = 0; i < 6; i++) M[i] = b[i];
}

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

__global__ __launch_bounds__(256) void seg_pre_kernel(const uint8_t* __restrict__ frame, int H, int W, Affine A, int in_h, int in_w,
                                                      float* __restrict__ out, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int n = in_h * in_w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / in_w, x = i - y * in_w;
        int sx, sy, fx, fy;
        warp_coord(A, x, y, sx, sy, fx, fy);
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        int bgr[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            int v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int xx = sx + (t & 1), yy = sy + (t >> 1);
                v[t] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? frame[((size_t)yy * W + xx) * 3 + c] : 0;
            }
            const int r = (v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15;
            bgr[c] = min(max(r, 0), 255);
        }
        float4 o;
        o.x = ((float)bgr[2] / 255.f - m0) * s0;
        o.y = ((float)bgr[1] / 255.f - m1) * s1;
        o.z = ((float)bgr[0] / 255.f - m2) * s2;
        o.w = 0.f;
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

// torch upsample_bilinear2d (align_corners = True) of class c at (y, x) of the (in_h, in_w) network-input grid: src = dst * (in - 1) /
// (out - 1) in float, weights and sums exactly as ATen's CPU kernel forms them.
__device__ __forceinline__ float up_ac_value(const float* __restrict__ logits, int Cbuf, int lh, int lw, int c, int y, int x, float rh, float rw) {
    const float h1r = rh * y, w1r = rw * x;
    const int h1 = (int)h1r, w1 = (int)w1r;
    const int h1p = h1 < lh - 1 ? 1 : 0, w1p = w1 < lw - 1 ? 1 : 0;
    const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
    const float v00 = logits[((size_t)h1 * lw + w1) * Cbuf + c], v01 = logits[((size_t)h1 * lw + w1 + w1p) * Cbuf + c];
    const float v10 = logits[((size_t)(h1 + h1p) * lw + w1) * Cbuf + c], v11 = logits[((size_t)(h1 + h1p) * lw + w1 + w1p) * Cbuf + c];
    const float top = w0l * v00 + w1l * v01, bot = w0l * v10 + w1l * v11;
    return h0l * top + h1l * bot;
}

// The whole post stage in one launch (agent_human.py:58-73, 296-310): a workgroup owns 16 x 64 frame pixels.  It finds the box of
// network-input pixels its inverse-warp taps touch, evaluates the align_corners upsample of the C class logits for that box into
// LDS (at 1080p: ~6 x 18 pixels x 20 classes = 8.6 KB, pixel stride 80 B: the 16 lanes of a ds_read_b128 phase hit 64 distinct
// banks), then every lane warps its 4 pixels from LDS (float bilinear, zero border) and takes the argmax.  Rounds 1-2 wrote the
// upsampled 473 x 473 x 20 tensor (17.9 MB) to HBM in a first kernel and gathered from it in a second (62 MB fetched per frame).
// When the box does not fit the LDS budget (frames smaller than the network input are magnified) the taps are evaluated on the fly.
constexpr int kSpTileH = 16, kSpGroups = 16, kSpCap = 320;

// R: row groups per workgroup (tile = 16 R rows x 64 pixels).  The staging pass (the box's cells x C bilinear evaluations from the
// logits, a chain of dependent global loads) is what a workgroup waits for; on a frame much larger than the network input the box
// of a 16 x 64 tile is a handful of cells, so taller tiles share it between R times the pixels (round 5: 3840 x 2160, R = 4).
template <int C, int R>
__global__ __launch_bounds__(256) void seg_post_fused_kernel(const float* __restrict__ logits, int Cbuf, int lh, int lw, int in_h, int in_w,
                                                             float rh, float rw, Affine A, uint8_t* __restrict__ out, int H, int W) {
    __shared__ __attribute__((aligned(16))) float s_up[kSpCap * C];
    __shared__ int s_box[4];
    const int t = threadIdx.x;
    const int yb = blockIdx.y * (kSpTileH * R) + (t >> 4), x0 = blockIdx.x * (kSpGroups * 4) + (t & 15) * 4;
    if (t < 4) s_box[t] = (t & 1) ? -0x7fffffff : 0x7fffffff;
    __syncthreads();
    int bx0 = 0x7fffffff, bx1 = -0x7fffffff, by0 = 0x7fffffff, by1 = -0x7fffffff;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int y = yb + r * kSpTileH;
        const int cnt = (y < H && x0 < W) ? min(4, W - x0) : 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k < cnt) {
                int sx, sy, fx, fy;
                warp_coord(A, x0 + k, y, sx, sy, fx, fy);
                bx0 = min(bx0, min(max(sx, 0), in_w - 1)); bx1 = max(bx1, min(max(sx + 1, 0), in_w - 1));
                by0 = min(by0, min(max(sy, 0), in_h - 1)); by1 = max(by1, min(max(sy + 1, 0), in_h - 1));
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        bx0 = min(bx0, __shfl_xor(bx0, off, 64)); bx1 = max(bx1, __shfl_xor(bx1, off, 64));
        by0 = min(by0, __shfl_xor(by0, off, 64)); by1 = max(by1, __shfl_xor(by1, off, 64));
    }
    if ((t & 63) == 0) { atomicMin(&s_box[0], bx0); atomicMax(&s_box[1], bx1); atomicMin(&s_box[2], by0); atomicMax(&s_box[3], by1); }
    __syncthreads();
    bx0 = s_box[0]; bx1 = s_box[1]; by0 = s_box[2]; by1 = s_box[3];
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1;
    const bool staged = bx1 >= bx0 && by1 >= by0 && (long long)bw * bh <= kSpCap;
    if (staged) {
        const int total = bw * bh * C;
        for (int idx = t; idx < total; idx += 256) {
            const int c = idx % C, p = idx / C;
            const int py = p / bw, px = p - py * bw;
            s_up[idx] = up_ac_value(logits, Cbuf, lh, lw, c, by0 + py, bx0 + px, rh, rw);
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int r = 0; r < R; r++) {
        const int y = yb + r * kSpTileH;
        const int cnt = (y < H && x0 < W) ? min(4, W - x0) : 0;
        if (cnt == 0) continue;
        uint32_t packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k >= cnt) continue;
            int sxk, syk, fxk, fyk;
            warp_coord(A, x0 + k, y, sxk, syk, fxk, fyk);
            const float ax = fxk * (1.f / 32), ay = fyk * (1.f / 32);
            const float wx0 = 1.f - ax, wx1 = ax, wy0 = 1.f - ay, wy1 = ay;
            const float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
            const bool in0 = sxk >= 0 && sxk < in_w, in1 = sxk + 1 >= 0 && sxk + 1 < in_w;
            const bool jn0 = syk >= 0 && syk < in_h, jn1 = syk + 1 >= 0 && syk + 1 < in_h;
            const int cx0 = min(max(sxk, 0), in_w - 1), cx1 = min(max(sxk + 1, 0), in_w - 1);
            const int cy0 = min(max(syk, 0), in_h - 1), cy1 = min(max(syk + 1, 0), in_h - 1);
            const bool m00 = in0 && jn0, m01 = in1 && jn0, m10 = in0 && jn1, m11 = in1 && jn1;
            float best = 0.f;
            int besti = 0;
            if (staged) {
                const float4* p00 = reinterpret_cast<const float4*>(s_up + ((cy0 - by0) * bw + (cx0 - bx0)) * C);
                const float4* p01 = reinterpret_cast<const float4*>(s_up + ((cy0 - by0) * bw + (cx1 - bx0)) * C);
                const float4* p10 = reinterpret_cast<const float4*>(s_up + ((cy1 - by0) * bw + (cx0 - bx0)) * C);
                const float4* p11 = reinterpret_cast<const float4*>(s_up + ((cy1 - by0) * bw + (cx1 - bx0)) * C);
#pragma unroll
                for (int c4 = 0; c4 < C / 4; c4++) {
                    const float4 a = p00[c4], b = p01[c4], c = p10[c4], d = p11[c4];
                    const float va[4] = {a.x, a.y, a.z, a.w}, vb[4] = {b.x, b.y, b.z, b.w}, vc[4] = {c.x, c.y, c.z, c.w}, vd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        // out-of-image taps read the border value 0 (select, not multiply: -0 / NaN safe)
                        const float t0 = m00 ? va[e] : 0.f, t1 = m01 ? vb[e] : 0.f, t2 = m10 ? vc[e] : 0.f, t3 = m11 ? vd[e] : 0.f;
                        float rr = t0 * w0;
                        rr = rr + t1 * w1;
                        rr = rr + t2 * w2;
                        rr = rr + t3 * w3;
                        const int ci = c4 * 4 + e;
                        if (ci == 0 || rr > best) { best = rr; besti = ci; }       // np.argmax: the first maximum wins
                    }
                }
            } else {
                for (int ci = 0; ci < C; ci++) {
                    const float t0 = m00 ? up_ac_value(logits, Cbuf, lh, lw, ci, cy0, cx0, rh, rw) : 0.f;
                    const float t1 = m01 ? up_ac_value(logits, Cbuf, lh, lw, ci, cy0, cx1, rh, rw) : 0.f;
                    const float t2 = m10 ? up_ac_value(logits, Cbuf, lh, lw, ci, cy1, cx0, rh, rw) : 0.f;
                    const float t3 = m11 ? up_ac_value(logits, Cbuf, lh, lw, ci, cy1, cx1, rh, rw) : 0.f;
                    float rr = t0 * w0;
                    rr = rr + t1 * w1;
                    rr = rr + t2 * w2;
                    rr = rr + t3 * w3;
                    if (ci == 0 || rr > best) { best = rr; besti = ci; }
                }
            }
            packed |= (uint32_t)(besti > 0 ? 255 : 0) << (8 * k);
        }
        const size_t i0 = (size_t)y * W + x0;
        if (cnt == 4 && (i0 & 3) == 0) *reinterpret_cast<uint32_t*>(out + i0) = packed;
        else for (int k = 0; k < cnt; k++) out[i0 + k] = (packed >> (8 * k)) & 0xff;
    }
}

// F.interpolate(mode='bilinear', align_corners=True) on NHWC float32, 4 channels per lane, written into channels
// [c0, c0 + C) of a destination with ld channels per pixel (human_parse.py:108, 147-150, 186: PSP priors, edge and
// decoder features are upsampled straight into their concat buffers).
__global__ __launch_bounds__(256) void upsample_ac_nhwc_kernel(const float* __restrict__ src0, int sh, int sw, int C,
                                                               float* __restrict__ dst0, int dh, int dw, int ld, int c0,
                                                               float rh, float rw, int batch) {
    const int c4n = C / 4, per = dh * dw * c4n, n = per * batch;
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const int img = i0 / per, i = i0 - img * per;            // images are stacked along y in src and dst
        const float* __restrict__ src = src0 + (size_t)img * sh * sw * C;
        float* __restrict__ dst = dst0 + (size_t)img * dh * dw * ld;
        const int c4 = i % c4n, p = i / c4n;
        const int y = p / dw, x = p - y * dw;
        const float h1r = rh * y, w1r = rw * x;
        const int h1 = (int)h1r, w1 = (int)w1r;
        const int h1p = h1 < sh - 1 ? 1 : 0, w1p = w1 < sw - 1 ? 1 : 0;
        const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
        const float4 v00 = reinterpret_cast<const float4*>(src + ((size_t)h1 * sw + w1) * C)[c4];
        const float4 v01 = reinterpret_cast<const float4*>(src + ((size_t)h1 * sw + w1 + w1p) * C)[c4];
        const float4 v10 = reinterpret_cast<const float4*>(src + ((size_t)(h1 + h1p) * sw + w1) * C)[c4];
        const float4 v11 = reinterpret_cast<const float4*>(src + ((size_t)(h1 + h1p) * sw + w1 + w1p) * C)[c4];
        float4 o;
        o.x = h0l * (w0l * v00.x + w1l * v01.x) + h1l * (w0l * v10.x + w1l * v11.x);
        o.y = h0l * (w0l * v00.y + w1l * v01.y) + h1l * (w0l * v10.y + w1l * v11.y);
        o.z = h0l * (w0l * v00.z + w1l * v01.z) + h1l * (w0l * v10.z + w1l * v11.z);
        o.w = h0l * (w0l * v00.w + w1l * v01.w) + h1l * (w0l * v10.w + w1l * v11.w);
        *reinterpret_cast<float4*>(dst + (size_t)p * ld + c0 + 4 * c4) = o;
    }
}

// conv3x3(upsample(p)) without the convolution.  A 3x3 conv (padding 1) over a bilinearly upsampled low-resolution map p
// (s x s x Cin, s <= 6: the PSP priors of human_parse.py:83-111) is linear in both steps and the 1x1 channel mixing of a tap
// commutes with the spatial upsampling:  conv(up(p)) = sum_taps shift_tap(up(p W_tap)).  The caller computes Q = p [W_0 .. W_8]
// (one tiny 1x1 layer with 9 C output channels) and this kernel adds, for every output pixel, the nine upsampled Q_tap sampled
// at the tap's neighbour (zero outside the image = the conv's zero padding): 36 fused multiply-adds per element instead of
// 9 Cin.  dst (+)= the sum; the interpolation arithmetic is that of upsample_ac_nhwc_kernel (align_corners=True) or of
// upsample_nc_nhwc_kernel (align_corners=False: the DeepLabv3+ decoder, deeplab/_deeplab.py:51-55).
template <bool ALIGN>
__global__ __launch_bounds__(256) void upsample_taps_kernel(const float* __restrict__ q0, int sh, int sw, int C,
                                                            float* __restrict__ dst0, int dh, int dw, float rh, float rw, int batch,
                                                            int accumulate) {
    const int c4n = C / 4, per = dh * dw * c4n, n = per * batch;
    for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += gridDim.x * blockDim.x) {
        const int img = i0 / per, i = i0 - img * per;
        const float* __restrict__ q = q0 + (size_t)img * sh * sw * 9 * C;
        float* __restrict__ dst = dst0 + (size_t)img * dh * dw * C;
        const int c4 = i % c4n, p = i / c4n;
        const int y = p / dw, x = p - y * dw;
        float4 acc = accumulate ? reinterpret_cast<const float4*>(dst + (size_t)p * C)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; t++) {
            const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
            if (yy < 0 || yy >= dh || xx < 0 || xx >= dw) continue;
            float h1r, w1r;
            if (ALIGN) { h1r = rh * yy; w1r = rw * xx; }
            else {                                          // ATen area_pixel_compute_source_index, align_corners=False
                h1r = rh * ((float)yy + 0.5f) - 0.5f; w1r = rw * ((float)xx + 0.5f) - 0.5f;
                if (h1r < 0.f) h1r = 0.f;
                if (w1r < 0.f) w1r = 0.f;
            }
            const int h1 = (int)h1r, w1 = (int)w1r;
            const int h1p = h1 < sh - 1 ? 1 : 0, w1p = w1 < sw - 1 ? 1 : 0;
            const float h1l = h1r - h1, h0l = 1.f - h1l, w1l = w1r - w1, w0l = 1.f - w1l;
            const float* base = q + (size_t)t * C;
            const float4 v00 = reinterpret_cast<const float4*>(base + ((size_t)h1 * sw + w1) * 9 * C)[c4];
            const float4 v01 = reinterpret_cast<const float4*>(base + ((size_t)h1 * sw + w1 + w1p) * 9 * C)[c4];
            const float4 v10 = reinterpret_cast<const float4*>(base + ((size_t)(h1 + h1p) * sw + w1) * 9 * C)[c4];
            const float4 v11 = reinterpret_cast<const float4*>(base + ((size_t)(h1 + h1p) * sw + w1 + w1p) * 9 * C)[c4];
            acc.x += h0l * (w0l * v00.x + w1l * v01.x) + h1l * (w0l * v10.x + w1l * v11.x);
            acc.y += h0l * (w0l * v00.y + w1l * v01.y) + h1l * (w0l * v10.y + w1l * v11.y);
            acc.z += h0l * (w0l * v00.z + w1l * v01.z) + h1l * (w0l * v10.z + w1l * v11.z);
            acc.w += h0l * (w0l * v00.w + w1l * v01.w) + h1l * (w0l * v10.w + w1l * v11.w);
        }
        reinterpret_cast<float4*>(dst + (size_t)p * C)[c4] = acc;
    }
}

// cv2.warpAffine(img, M, (dw, dh), INTER_LINEAR, BORDER_CONSTANT 0) on uint8, 1 or 3 channels (imgprocess.py:55-64 shift_fg).
__global__ __launch_bounds__(256) void warp_affine_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int cn, Affine A,
                                                             uint8_t* __restrict__ dst, int dh, int dw) {
    const uint32_t n = (uint32_t)dh * (uint32_t)dw;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)dw), x = (int)(i - (uint32_t)y * (uint32_t)dw);
        int sx, sy, fx, fy;
        warp_coord(A, x, y, sx, sy, fx, fy);
        const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
        for (int c = 0; c < cn; c++) {
            int v[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int xx = sx + (t & 1), yy = sy + (t >> 1);
                v[t] = (xx >= 0 && xx < sw && yy >= 0 && yy < sh) ? src[((size_t)yy * sw + xx) * cn + c] : 0;
            }
            const int r = (v[0] * w00 + v[1] * w01 + v[2] * w10 + v[3] * w11 + (1 << 14)) >> 15;
            dst[(size_t)i * cn + c] = (uint8_t)min(max(r, 0), 255);
        }
    }
}

int grid_for(long long items, int threads) {
    long long b = (items + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > 256 * 16) b = 256 * 16;
    return (int)b;
}

}  // namespace

extern "C" const char* usc_seg_last_error(void) { return g_seg_err; }

extern "C" void usc_humanseg_affine(int w, int h, int in_w, int in_h, int inv, double* M) {
    // agent_human.py:247-260 (box [0, 0, w-1, h-1], aspect ratio 1) and :91-125 (rot = 0, shift = 0): every triangle
    // coordinate is an integer or half-integer, so the float32 arrays of the reference hold them exactly.
    const float cx = (float)((w - 1) * 0.5), cy = (float)((h - 1) * 0.5);
    const double side = (w - 1) > (h - 1) ? (w - 1) : (h - 1);
    const float src_w = (float)side;
    const float s0x = cx, s0y = cy, s1x = cx, s1y = (float)((double)cy + (double)src_w * -0.5);
    const float d0x = (float)((in_w - 1) * 0.5), d0y = (float)((in_h - 1) * 0.5);
    const float d1x = d0x, d1y = d0y + (float)((in_w - 1) * -0.5);
    const float s2x = s1x + -(s0y - s1y), s2y = s1y + (s0x - s1x);
    const float d2x = d1x + -(d0y - d1y), d2y = d1y + (d0x - d1x);
    const double src[6] = {s0x, s0y, s1x, s1y, s2x, s2y}, dst[6] = {d0x, d0y, d1x, d1y, d2x, d2y};
    if (inv) get_affine_transform(dst, src, M); else get_affine_transform(src, dst, M);
}

extern "C" int usc_humanseg_pre(const uint8_t* frame, int H, int W, int in_h, int in_w, float* out_nhwc4, usc_stream_t stream) {
    if (!frame || !out_nhwc4 || H <= 0 || W <= 0 || in_h <= 0 || in_w <= 0) return seg_fail(USC_E_ARG, "usc_humanseg_pre: bad argument");
    if ((long long)H * W >= (1ll << 29)) return seg_fail(USC_E_UNSUPPORTED, "usc_humanseg_pre: image too large");
    double M[6];
    Affine A;
    usc_humanseg_affine(W, H, in_w, in_h, 0, M);
    invert_affine(M, A.m);
    static const double mean[3] = {0.485, 0.456, 0.406}, stdv[3] = {0.229, 0.224, 0.225};
    hipLaunchKernelGGL(seg_pre_kernel, dim3(grid_for((long long)in_h * in_w, 256)), dim3(256), 0, (hipStream_t)stream, frame, H, W, A,
                       in_h, in_w, out_nhwc4, (float)mean[0], (float)mean[1], (float)mean[2],
                       (float)(1 / stdv[0]), (float)(1 / stdv[1]), (float)(1 / stdv[2]));
    return hipGetLastError() == hipSuccess ? USC_OK : seg_fail(USC_E_LAUNCH, "seg_pre_kernel launch failed");
}

extern "C" size_t usc_humanseg_post_ws_bytes(int, int, int) { return 0; }      // (kept in the ABI: the fused kernel stages in LDS)

extern "C" int usc_humanseg_post(const float* logits, int C, int Cbuf, int lh, int lw, int in_h, int in_w, int H, int W,
                                 uint8_t* out, void* ws, size_t ws_bytes, usc_stream_t stream) {
    (void)ws; (void)ws_bytes;
    if (!logits || !out || lh <= 0 || lw <= 0 || H <= 0 || W <= 0 || in_h <= 1 || in_w <= 1 || Cbuf < C)
        return seg_fail(USC_E_ARG, "usc_humanseg_post: bad argument");
    if (C != 20) return seg_fail(USC_E_UNSUPPORTED, "usc_humanseg_post: built for the 20 LIP classes");
    if ((long long)H * W >= (1ll << 31)) return seg_fail(USC_E_UNSUPPORTED, "usc_humanseg_post: image too large");
    const float rh = (float)(lh - 1) / (in_h - 1), rw = (float)(lw - 1) / (in_w - 1);
    double M[6];
    Affine A;
    usc_humanseg_affine(W, H, in_w, in_h, 1, M);
    invert_affine(M, A.m);
    // rows per workgroup: as tall as the box of network-input cells under a 16 R x 64 tile still fits the LDS budget with a margin
    // (A.m[0], A.m[4]: input cells per frame pixel; a tile that overflows anyway evaluates its taps on the fly, so this is a speed choice only)
    int R = 1;
    for (int cand = 4; cand > 1; cand >>= 1) {
        const double tw = kSpGroups * 4, th = kSpTileH * cand;          // box of the inverse-warped tile: |A| applied to its extent
        if ((tw * fabs(A.m[0]) + th * fabs(A.m[1]) + 3) * (tw * fabs(A.m[3]) + th * fabs(A.m[4]) + 3) <= 0.8 * kSpCap) { R = cand; break; }
    }
    const dim3 grid((W + kSpGroups * 4 - 1) / (kSpGroups * 4), (H + kSpTileH * R - 1) / (kSpTileH * R));
    if (R == 4) hipLaunchKernelGGL((seg_post_fused_kernel<20, 4>), grid, dim3(256), 0, (hipStream_t)stream, logits, Cbuf, lh, lw, in_h, in_w, rh, rw, A, out, H, W);
    else if (R == 2) hipLaunchKernelGGL((seg_post_fused_kernel<20, 2>), grid, dim3(256), 0, (hipStream_t)stream, logits, Cbuf, lh, lw, in_h, in_w, rh, rw, A, out, H, W);
    else hipLaunchKernelGGL((seg_post_fused_kernel<20, 1>), grid, dim3(256), 0, (hipStream_t)stream, logits, Cbuf, lh, lw, in_h, in_w, rh, rw, A, out, H, W);
    return hipGetLastError() == hipSuccess ? USC_OK : seg_fail(USC_E_LAUNCH, "seg_post_fused_kernel launch failed");
}

extern "C" int usc_upsample_ac_nhwc_batch_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0,
                                              int batch, usc_stream_t stream) {
    if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || C <= 0 || C % 4 || ld % 4 || c0 % 4 || c0 + C > ld || batch < 1)
        return seg_fail(USC_E_ARG, "usc_upsample_ac_nhwc_f32: bad argument (C, ld, c0 must be multiples of 4, c0 + C <= ld)");
    if ((long long)batch * dh * dw * (C / 4) >= (1ll << 31)) return seg_fail(USC_E_UNSUPPORTED, "usc_upsample_ac_nhwc_f32: too large");
    const float rh = dh > 1 ? (float)(sh - 1) / (dh - 1) : 0.f, rw = dw > 1 ? (float)(sw - 1) / (dw - 1) : 0.f;
    hipLaunchKernelGGL(upsample_ac_nhwc_kernel, dim3(grid_for((long long)batch * dh * dw * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                       src, sh, sw, C, dst, dh, dw, ld, c0, rh, rw, batch);
    return hipGetLastError() == hipSuccess ? USC_OK : seg_fail(USC_E_LAUNCH, "upsample_ac_nhwc_kernel launch failed");
}

extern "C" int usc_upsample_taps_accum_f32(const float* q, int sh, int sw, int C, float* dst, int dh, int dw, int batch, int accumulate,
                                           int align_corners, usc_stream_t stream) {
    if (!q || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || C <= 0 || C % 4 || batch < 1)
        return seg_fail(USC_E_ARG, "usc_upsample_taps_accum_f32: bad argument (C must be a multiple of 4)");
    if ((long long)batch * dh * dw * (C / 4) >= (1ll << 31)) return seg_fail(USC_E_UNSUPPORTED, "usc_upsample_taps_accum_f32: too large");
    const dim3 grid(grid_for((long long)batch * dh * dw * (C / 4), 256));
    if (align_corners) {
        const float rh = dh > 1 ? (float)(sh - 1) / (dh - 1) : 0.f, rw = dw > 1 ? (float)(sw - 1) / (dw - 1) : 0.f;
        hipLaunchKernelGGL(upsample_taps_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, sh, sw, C, dst, dh, dw, rh, rw, batch, accumulate);
    } else {
        hipLaunchKernelGGL(upsample_taps_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q, sh, sw, C, dst, dh, dw,
                           (float)sh / (float)dh, (float)sw / (float)dw, batch, accumulate);
    }
    return hipGetLastError() == hipSuccess ? USC_OK : seg_fail(USC_E_LAUNCH, "upsample_taps_kernel launch failed");
}

extern "C" int usc_upsample_ac_nhwc_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0,
                                        usc_stream_t stream) {
    return usc_upsample_ac_nhwc_batch_f32(src, sh, sw, C, dst, dh, dw, ld, c0, 1, stream);
}

extern "C" int usc_warp_affine_u8(const uint8_t* src, int sh, int sw, int cn, const double* M2x3, uint8_t* dst, int dh, int dw,
                                  usc_stream_t stream) {
    if (!src || !dst || !M2x3 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || (cn != 1 && cn != 3)) return seg_fail(USC_E_ARG, "usc_warp_affine_u8: bad argument");
    if ((long long)dh * dw >= (1ll << 31)) return seg_fail(USC_E_UNSUPPORTED, "usc_warp_affine_u8: image too large");
    Affine A;
    invert_affine(M2x3, A.m);
    hipLaunchKernelGGL(warp_affine_u8_kernel, dim3(grid_for((long long)dh * dw, 256)), dim3(256), 0, (hipStream_t)stream, src, sh, sw, cn, A, dst, dh, dw);
    return hipGetLastError() == hipSuccess ? USC_OK : seg_fail(USC_E_LAUNCH, "warp_affine_u8_kernel launch failed");
}
