// usc_dropin.hip - the NumPy-level helpers of unscreen/utils/imgprocess.py that the reference's tools and agents import by
// name (tools/unscreen/bg_offline.py:16, tools/replace/replace.py:14, unscreen/iseg/agent.py:5): pad_resize, inv_pad_resize,
// imnormalize, get_center, get_mask.  The frame loops use the fused forms (usc_vmat_pre / post, usc_segagent_pre / post ...);
// these are the stand-alone ones behind the drop-in names.  HBM-bound byte / float work: one lane per output pixel, flat
// grid-stride, wave reductions + one atomic per workgroup for the moments.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

__device__ __forceinline__ int reflect_idx(int p, int n) {      // cv::borderInterpolate(BORDER_REFLECT): bounces until inside
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p - 1 : 2 * n - 1 - p;
    return p;
}

// imgprocess.py:67-97: dst (th x tw) = copyMakeBorder(resize(src, (nw, nh)), 0, th - nh, 0, tw - nw, BORDER_REFLECT)
__global__ __launch_bounds__(256) void pad_resize_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int cn, uint8_t* __restrict__ dst,
                                                            int th, int tw, int nh, int nw, int mode, double scale_y, double scale_x) {
    const uint32_t n = (uint32_t)th * (uint32_t)tw;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)tw), x = (int)(i - (uint32_t)y * (uint32_t)tw);
        const int ry = reflect_idx(y, nh), rx = reflect_idx(x, nw);
        for (int c = 0; c < cn; c++)
            dst[(size_t)i * cn + c] = (uint8_t)resize_sample_u8(src, sh, sw, cn, c, ry, rx, 1, mode, scale_y, scale_x, 0., 0.);
    }
}

// cv2.resize(INTER_LINEAR) of the top-left (sh x sw) window of a pitched image.  T = uint8_t: 11-bit fixed point; float /
// double: float weights, products and sums in T (cv::HResizeLinear / VResizeLinear, AT = float).
template <typename T>
__global__ __launch_bounds__(256) void resize_window_kernel(const T* __restrict__ src, int sh, int sw, int cn, size_t pitch, T* __restrict__ dst,
                                                            int dh, int dw, int mode, double scale_y, double scale_x) {
    const uint32_t n = (uint32_t)dh * (uint32_t)dw;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)dw), x = (int)(i - (uint32_t)y * (uint32_t)dw);
        if (mode == RS_COPY) {
            for (int c = 0; c < cn; c++) dst[(size_t)i * cn + c] = src[(size_t)y * pitch + (size_t)x * cn + c];
            continue;
        }
        if (mode == RS_AREA2) {
            for (int c = 0; c < cn; c++) {
                const T* p = src + (size_t)(2 * y) * pitch + (size_t)(2 * x) * cn + c;
                if constexpr (sizeof(T) == 1) {
                    dst[(size_t)i * cn + c] = (T)(((int)p[0] + (int)p[cn] + (int)p[pitch] + (int)p[pitch + cn] + 2) >> 2);
                } else {
                    T s = p[0] + p[cn];
                    s = s + p[pitch];
                    s = s + p[pitch + cn];
                    dst[(size_t)i * cn + c] = s * (T)0.25f;
                }
            }
            continue;
        }
        const Tap tx = linear_tap_x(x, scale_x, sw), ty = linear_tap(y, scale_y);
        const int x0 = tx.s, x1 = min(x0 + 1, sw - 1);
        const int y0 = clampi(ty.s, 0, sh - 1), y1 = clampi(ty.s + 1, 0, sh - 1);
        for (int c = 0; c < cn; c++) {
            const T p00 = src[(size_t)y0 * pitch + (size_t)x0 * cn + c], p01 = src[(size_t)y0 * pitch + (size_t)x1 * cn + c];
            const T p10 = src[(size_t)y1 * pitch + (size_t)x0 * cn + c], p11 = src[(size_t)y1 * pitch + (size_t)x1 * cn + c];
            if constexpr (sizeof(T) == 1) {
                dst[(size_t)i * cn + c] = (T)bilinear_u8(p00, p01, p10, p11, coef11(1.f - tx.f), coef11(tx.f), coef11(1.f - ty.f), coef11(ty.f));
            } else {
                const T a0 = (T)(1.f - tx.f), a1 = (T)tx.f, b0 = (T)(1.f - ty.f), b1 = (T)ty.f;
                const T r0 = p00 * a0 + p01 * a1;
                const T r1 = p10 * a0 + p11 * a1;
                dst[(size_t)i * cn + c] = r0 * b0 + r1 * b1;
            }
        }
    }
}

// imgprocess.py:135-161: BGR -> RGB in place on the argument (to_rgb), float32(px) / 255, - mean, * (1 / std): three float32
// operations (OpenCV turns the double scalars into float32 for a float32 array).
__global__ __launch_bounds__(256) void imnormalize_kernel(uint8_t* __restrict__ img, size_t n, float m0, float m1, float m2, float s0, float s1,
                                                          float s2, int to_rgb, float* __restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        int c0 = img[3 * i], c1 = img[3 * i + 1], c2 = img[3 * i + 2];
        if (to_rgb) {
            const int t = c0; c0 = c2; c2 = t;
            img[3 * i] = (uint8_t)c0;
            img[3 * i + 2] = (uint8_t)c2;
        }
        float x = (float)c0 / 255.f; x = x - m0; out[3 * i] = x * s0;
        x = (float)c1 / 255.f; x = x - m1; out[3 * i + 1] = x * s1;
        x = (float)c2 / 255.f; x = x - m2; out[3 * i + 2] = x * s2;
    }
}

// imgprocess.py:7-14 / :17-30: gray of a 3-channel image (rgb_order: the FIRST channel carries the red weight, cv2.COLOR_RGB2GRAY),
// optional threshold (gray > thr -> 255 else 0) written to `mask`, optional spatial moments m00, m10, m01 (exact integer sums).
// One image per blockIdx.y (img0 / img1, moments at mom + 3 * blockIdx.y): the centroid pass of the replace tool asks for the moments of two
// masks per frame (tools/replace/replace.py:28-41).  4 pixels = 12 bytes per lane and step; per-thread sums in 64 bits, one wave
// reduction, ONE atomic per workgroup and moment - with kMomBlocks workgroups per image (the L2 serialises same-address atomics: the
// first version's 4096 workgroups spent 25 us on them at 1080p).
constexpr int kMomBlocks = 256;
__global__ __launch_bounds__(256) void gray_moments_kernel(const uint8_t* __restrict__ img0, const uint8_t* __restrict__ img1, int H, int W, int rgb_order, int thr,
                                                           uint8_t* __restrict__ mask, unsigned long long* __restrict__ mom, int aligned) {
    const uint8_t* __restrict__ img = blockIdx.y ? img1 : img0;
    if (mom) mom += 3 * blockIdx.y;
    const uint32_t n = (uint32_t)H * (uint32_t)W, nq = n / 4;
    unsigned long long m00 = 0, m10 = 0, m01 = 0;
    for (uint32_t q = blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += gridDim.x * blockDim.x) {
        int c[12];
        if (aligned) unpack12(*reinterpret_cast<const Px4*>(img + 12 * (size_t)q), c);
        else {                                                      // (image k of a packed stack of odd-sized images: any byte address)
#pragma unroll
            for (int k = 0; k < 12; k++) c[k] = img[12 * (size_t)q + k];
        }
        uint32_t y = (4 * q) / (uint32_t)W, x = 4 * q - y * (uint32_t)W;
        uint32_t s0 = 0, s1 = 0, s2 = 0, mk = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t g = (uint32_t)(rgb_order ? gray_u8(c[3 * k + 2], c[3 * k + 1], c[3 * k]) : gray_u8(c[3 * k], c[3 * k + 1], c[3 * k + 2]));
            mk |= ((int)g > thr ? 255u : 0u) << (8 * k);
            s0 += g; s1 += x * g; s2 += y * g;                     // < 4 * 255 * 2^15: 32 bits hold a quad's sums (too_big() bounds H W)
            if (++x >= (uint32_t)W) { x = 0; y++; }
        }
        if (mask) {
            if (aligned) *reinterpret_cast<uint32_t*>(mask + 4 * (size_t)q) = mk;
            else { mask[4 * (size_t)q] = (uint8_t)mk; mask[4 * (size_t)q + 1] = (uint8_t)(mk >> 8); mask[4 * (size_t)q + 2] = (uint8_t)(mk >> 16); mask[4 * (size_t)q + 3] = (uint8_t)(mk >> 24); }
        }
        m00 += s0; m10 += s1; m01 += s2;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {                // the last n % 4 pixels
        const uint32_t i = 4 * nq + threadIdx.x, y = i / (uint32_t)W, x = i - y * (uint32_t)W;
        const int c0 = img[3 * (size_t)i], c1 = img[3 * (size_t)i + 1], c2 = img[3 * (size_t)i + 2];
        const uint32_t g = (uint32_t)(rgb_order ? gray_u8(c2, c1, c0) : gray_u8(c0, c1, c2));
        if (mask) mask[i] = (int)g > thr ? 255 : 0;
        m00 += g; m10 += (unsigned long long)x * g; m01 += (unsigned long long)y * g;
    }
    if (!mom) return;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        m00 += __shfl_down(m00, off, kWave);
        m10 += __shfl_down(m10, off, kWave);
        m01 += __shfl_down(m01, off, kWave);
    }
    __shared__ unsigned long long s_part[3][256 / kWave];
    const int wave = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { s_part[0][wave] = m00; s_part[1][wave] = m10; s_part[2][wave] = m01; }
    __syncthreads();
    if (threadIdx.x < 3) {
        unsigned long long t = 0;
        for (int i = 0; i < 256 / kWave; i++) t += s_part[threadIdx.x][i];
        if (t) atomicAdd(&mom[threadIdx.x], t);          // integer sums: the order of the atomics does not matter
    }
}

template <typename T>
int launch_window(const T* src, int sh, int sw, int cn, size_t pitch, T* dst, int dh, int dw, hipStream_t st, const char* what) {
    const ResizeGeom rg = resize_geom(sh, sw, dh, dw);
    hipLaunchKernelGGL(resize_window_kernel<T>, dim3(grid_for((size_t)dh * dw, 256)), dim3(256), 0, st, src, sh, sw, cn, pitch, dst, dh, dw,
                       rg.mode, rg.scale_y, rg.scale_x);
    return check_launch(what);
}

}  // namespace

extern "C" int usc_pad_resize_u8(const uint8_t* src, int h, int w, int cn, uint8_t* dst, int target_h, int target_w, usc_stream_t stream) {
    if (!src || !dst || h <= 0 || w <= 0 || target_h <= 0 || target_w <= 0 || (cn != 1 && cn != 3)) return fail(USC_E_ARG, "usc_pad_resize_u8: bad argument");
    if (too_big(h, w) || too_big(target_h, target_w)) return fail(USC_E_UNSUPPORTED, "usc_pad_resize_u8: image too large");
    int nh, nw;
    usc_pad_resize_geometry(h, w, target_h, target_w, &nh, &nw);
    if (nh <= 0 || nw <= 0) return fail(USC_E_ARG, "usc_pad_resize_u8: empty resized image");
    const ResizeGeom rg = resize_geom(h, w, nh, nw);
    hipLaunchKernelGGL(pad_resize_u8_kernel, dim3(grid_for((size_t)target_h * target_w, 256)), dim3(256), 0, (hipStream_t)stream, src, h, w, cn, dst,
                       target_h, target_w, nh, nw, rg.mode, rg.scale_y, rg.scale_x);
    return check_launch("pad_resize_u8_kernel");
}

extern "C" int usc_inv_pad_resize(const void* src, int h, int w, int cn, int elem, void* dst, int ori_h, int ori_w, usc_stream_t stream) {
    if (!src || !dst || h <= 0 || w <= 0 || ori_h <= 0 || ori_w <= 0 || cn < 1 || cn > 4 || elem < 0 || elem > 2)
        return fail(USC_E_ARG, "usc_inv_pad_resize: bad argument");
    if (too_big(h, w) || too_big(ori_h, ori_w)) return fail(USC_E_UNSUPPORTED, "usc_inv_pad_resize: image too large");
    int rh, rw;
    usc_inv_pad_geometry(h, w, ori_h, ori_w, &rh, &rw);
    if (rh <= 0 || rw <= 0 || rh > h || rw > w) return fail(USC_E_ARG, "usc_inv_pad_resize: empty un-padded region");
    hipStream_t st = (hipStream_t)stream;
    const size_t pitch = (size_t)w * cn;
    if (elem == USC_ELEM_U8) return launch_window((const uint8_t*)src, rh, rw, cn, pitch, (uint8_t*)dst, ori_h, ori_w, st, "resize_window_kernel<u8>");
    if (elem == USC_ELEM_F32) return launch_window((const float*)src, rh, rw, cn, pitch, (float*)dst, ori_h, ori_w, st, "resize_window_kernel<f32>");
    return launch_window((const double*)src, rh, rw, cn, pitch, (double*)dst, ori_h, ori_w, st, "resize_window_kernel<f64>");
}

extern "C" int usc_imnormalize_u8(uint8_t* img, size_t n, const double* mean3_host, const double* std3_host, int to_rgb, float* out, usc_stream_t stream) {
    if (!img || !out || !mean3_host || !std3_host || n == 0) return fail(USC_E_ARG, "usc_imnormalize_u8: bad argument");
    float m[3], s[3];
    for (int c = 0; c < 3; c++) { m[c] = (float)mean3_host[c]; s[c] = (float)(1 / std3_host[c]); }
    hipLaunchKernelGGL(imnormalize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, img, n, m[0], m[1], m[2], s[0], s[1], s[2], to_rgb, out);
    return check_launch("imnormalize_kernel");
}

static int launch_moments(const uint8_t* a, const uint8_t* b, int H, int W, int rgb_order, int thr, uint8_t* mask, unsigned long long* mom, hipStream_t st) {
    const int images = b ? 2 : 1;
    if (mom && hipMemsetAsync(mom, 0, 3 * images * sizeof(unsigned long long), st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_gray_moments_u8: memset failed");
    const size_t nq = ((size_t)H * W) / 4;
    int blocks = (int)((nq + 255) / 256);
    blocks = blocks < 1 ? 1 : (blocks > kMomBlocks ? kMomBlocks : blocks);
    const int aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)mask) & 3) == 0;
    hipLaunchKernelGGL(gray_moments_kernel, dim3(blocks, images), dim3(256), 0, st, a, b, H, W, rgb_order, thr, mask, mom, aligned);
    return check_launch("gray_moments_kernel");
}

extern "C" int usc_gray_moments_u8(const uint8_t* img, int H, int W, int rgb_order, int thr, uint8_t* mask, unsigned long long* moments3,
                                   usc_stream_t stream) {
    if (!img || H <= 0 || W <= 0 || (!mask && !moments3)) return fail(USC_E_ARG, "usc_gray_moments_u8: bad argument");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_gray_moments_u8: image too large");
    return launch_moments(img, nullptr, H, W, rgb_order, thr, mask, moments3, (hipStream_t)stream);
}

extern "C" int usc_gray_moments_pair_u8(const uint8_t* img_a, const uint8_t* img_b, int H, int W, int rgb_order, unsigned long long* moments6,
                                        usc_stream_t stream) {
    if (!img_a || !img_b || !moments6 || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_gray_moments_pair_u8: bad argument");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_gray_moments_pair_u8: image too large");
    return launch_moments(img_a, img_b, H, W, rgb_order, 0, nullptr, moments6, (hipStream_t)stream);
}
