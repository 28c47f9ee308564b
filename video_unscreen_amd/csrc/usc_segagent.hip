// usc_segagent.hip - everything of SegAgent (reference unscreen/binseg/agent.py:176-260) that is not a convolution:
//   pre   : pad_resize (fixed-point bilinear + BORDER_REFLECT) + imnormalize + one grid crop (optionally mirrored), fused:
//           one lane per crop pixel gathers straight from the BGR frame, the resized / padded image never exists;
//   im2col: the 7x7 stride-2 stem of the ResNet-50 trunk (Cin = 3, K = 147) as a [pixels][160] matrix, so the stem runs on
//           the same MFMA kernel as every other layer (a 1x1 convolution over 160 "channels");
//   up    : F.interpolate(bilinear, align_corners=False) on NHWC float32 into a channel slice of a concat buffer;
//   post  : the model's final x4 upsample + 2-class softmax + float64 merge of the overlapping crops (mirrored ones
//           un-mirrored) in crop order + cover-count division (one lane per merged pixel: no atomics, fixed order), then
//           cv2.resize(float64) back to the frame + argmax.  The reference holds ncrops x 2 x 513 x 513 floats and a
//           (th, tw, 2) float64 map on the host for this; here only the latter exists (8.3 MB at 1080p).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

constexpr int kMaxCrops = 64;
struct CropTable { int n; int s_h[kMaxCrops], s_w[kMaxCrops], flip[kMaxCrops]; };

__device__ __forceinline__ int reflect_any(int p, int n) {      // BORDER_REFLECT at any distance
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

struct NcTap { int i0, i1; float l0, l1; };
__device__ __forceinline__ NcTap nc_tap(int d, int in, float scale) {   // ATen area_pixel_compute_source_index, align_corners=False
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
// host geometry (agent.py:70-141)
// --------------------------------------------------------------------------------------------
extern "C" void usc_segagent_target_size(int h, int w, int long_side, int division, int crop_h, int crop_w, int* th, int* tw) {
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
    *th = crop_h > t_h ? crop_h : t_h;
    *tw = crop_w > t_w ? crop_w : t_w;
}

extern "C" int usc_grid_crop_locations(int h, int w, int crop_h, int crop_w, double stride_ratio, int flip, int* locs, int max_crops) {
    if (h < crop_h || w < crop_w || crop_h <= 0 || crop_w <= 0 || !(stride_ratio > 0)) return fail(USC_E_ARG, "usc_grid_crop_locations: bad argument");
    const int stride_h = (int)ceil(crop_h * stride_ratio), stride_w = (int)ceil(crop_w * stride_ratio);
    const int grid_h = (int)(ceil((double)(h - crop_h) / stride_h) + 1), grid_w = (int)(ceil((double)(w - crop_w) / stride_w) + 1);
    int n = 0;
    for (int ih = 0; ih < grid_h; ih++)
        for (int iw = 0; iw < grid_w; iw++) {
            int e_h = ih * stride_h + crop_h; if (e_h > h) e_h = h;
            const int s_h = e_h - crop_h;
            int e_w = iw * stride_w + crop_w; if (e_w > w) e_w = w;
            const int s_w = e_w - crop_w;
            for (int f = 0; f < (flip ? 2 : 1); f++) {
                if (locs) {
                    if (n >= max_crops) return fail(USC_E_ARG, "usc_grid_crop_locations: more than max_crops crops");
                    locs[4 * n] = s_h; locs[4 * n + 1] = e_h; locs[4 * n + 2] = f ? e_w : s_w; locs[4 * n + 3] = f ? s_w : e_w;
                }
                n++;
            }
        }
    return n;
}

// --------------------------------------------------------------------------------------------
// pre: one crop (agent.py:176-198)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segagent_pre_kernel(const uint8_t* __restrict__ img, int H, int W, int nh, int nw, int mode,
                                                           double scale_y, double scale_x, int s_h, int s_w, int flip, int ch, int cw,
                                                           float* __restrict__ out) {
    const float mean[3] = {(float)0.485, (float)0.456, (float)0.406};
    const float stdinv[3] = {(float)(1 / 0.229), (float)(1 / 0.224), (float)(1 / 0.225)};
    const uint32_t n = (uint32_t)ch * (uint32_t)cw;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)cw), x = (int)(i - (uint32_t)y * (uint32_t)cw);
        const int py = reflect_any(s_h + y, nh), px = reflect_any(s_w + (flip ? cw - 1 - x : x), nw);
        float4 o;
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int u = resize_sample_u8(img, H, W, 3, 2 - c, py, px, 1, mode, scale_y, scale_x, 0., 0.);
            float t = (float)u / 255.f;
            t = t - mean[c];
            v[c] = t * stdinv[c];
        }
        o.x = v[0]; o.y = v[1]; o.z = v[2]; o.w = 0.f;
        reinterpret_cast<float4*>(out)[i] = o;
    }
}

extern "C" int usc_segagent_pre(const uint8_t* img, int H, int W, int th, int tw, const int* loc_host, float* out, usc_stream_t stream) {
    if (!img || !loc_host || !out || H <= 0 || W <= 0 || th <= 0 || tw <= 0) return fail(USC_E_ARG, "usc_segagent_pre: bad argument");
    if (too_big(H, W) || too_big(th, tw)) return fail(USC_E_UNSUPPORTED, "usc_segagent_pre: image too large");
    int nh, nw;
    usc_pad_resize_geometry(H, W, th, tw, &nh, &nw);
    int s_h = loc_host[0], e_h = loc_host[1], s_w = loc_host[2], e_w = loc_host[3];
    const int flip = s_w > e_w;
    if (flip) { const int t = s_w; s_w = e_w; e_w = t; }
    const int ch = e_h - s_h, cw = e_w - s_w;
    if (ch <= 0 || cw <= 0 || s_h < 0 || s_w < 0 || e_h > th || e_w > tw) return fail(USC_E_ARG, "usc_segagent_pre: crop outside the target");
    const ResizeGeom rg = resize_geom(H, W, nh, nw);
    hipLaunchKernelGGL(segagent_pre_kernel, dim3(grid_for((size_t)ch * cw, 256)), dim3(256), 0, (hipStream_t)stream, img, H, W, nh, nw,
                       rg.mode, rg.scale_y, rg.scale_x, s_h, s_w, flip, ch, cw, out);
    return check_launch("segagent_pre_kernel");
}

// --------------------------------------------------------------------------------------------
// im2col for a k x k stride-s stem on a (IH, IW, cpitch) NHWC input of which the first cuse channels are used
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col_kernel(const float* __restrict__ in, int IH, int IW, int cpitch, int cuse, int kh, int kw,
                                                     int stride, int pad, float* __restrict__ out, int OH, int OW, int Kpad) {
    const int k4n = Kpad / 4;
    const uint32_t n = (uint32_t)OH * (uint32_t)OW * (uint32_t)k4n;
    const int K = kh * kw * cuse;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = i / (uint32_t)k4n;
        const int k4 = (int)(i - p * (uint32_t)k4n);
        const int oy = (int)(p / (uint32_t)OW), ox = (int)(p - (uint32_t)oy * (uint32_t)OW);
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * k4 + j;
            float val = 0.f;
            if (k < K) {
                const int tap = k / cuse, c = k - tap * cuse;
                const int ky = tap / kw, kx = tap - ky * kw;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < IH && ix >= 0 && ix < IW) val = in[((size_t)iy * IW + ix) * cpitch + c];
            }
            v[j] = val;
        }
        reinterpret_cast<float4*>(out)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

extern "C" int usc_im2col_nhwc_f32(const float* in, int IH, int IW, int cpitch, int cuse, int kh, int kw, int stride, int pad, float* out,
                                   int OH, int OW, int Kpad, usc_stream_t stream) {
    if (!in || !out || IH <= 0 || IW <= 0 || cuse <= 0 || cuse > cpitch || kh <= 0 || kw <= 0 || stride <= 0 || pad < 0 || OH <= 0 ||
        OW <= 0 || Kpad % 4 || Kpad < kh * kw * cuse)
        return fail(USC_E_ARG, "usc_im2col_nhwc_f32: bad argument (Kpad >= kh*kw*cuse, multiple of 4)");
    if ((OH - 1) * stride - pad + kh - 1 >= IH + pad || (OW - 1) * stride - pad + kw - 1 >= IW + pad)
        return fail(USC_E_ARG, "usc_im2col_nhwc_f32: output size does not match the input / stride / pad");
    if ((long long)OH * OW * (Kpad / 4) >= (1ll << 32)) return fail(USC_E_UNSUPPORTED, "usc_im2col_nhwc_f32: too large");
    hipLaunchKernelGGL(im2col_kernel, dim3(grid_for((size_t)OH * OW * (Kpad / 4), 256)), dim3(256), 0, (hipStream_t)stream, in, IH, IW,
                       cpitch, cuse, kh, kw, stride, pad, out, OH, OW, Kpad);
    return check_launch("im2col_kernel");
}

// --------------------------------------------------------------------------------------------
// bilinear, align_corners=False, NHWC f32 (deeplab/_deeplab.py:51-55, 148-151)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void upsample_nc_nhwc_kernel(const float* __restrict__ src, int sh, int sw, int C, float* __restrict__ dst,
                                                               int dh, int dw, int ld, int c0, float rh, float rw) {
    const int c4n = C / 4, n = dh * dw * c4n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int c4 = i % c4n, p = i / c4n;
        const int y = p / dw, x = p - y * dw;
        const NcTap ty = nc_tap(y, sh, rh), tx = nc_tap(x, sw, rw);
        const float4 v00 = reinterpret_cast<const float4*>(src + ((size_t)ty.i0 * sw + tx.i0) * C)[c4];
        const float4 v01 = reinterpret_cast<const float4*>(src + ((size_t)ty.i0 * sw + tx.i1) * C)[c4];
        const float4 v10 = reinterpret_cast<const float4*>(src + ((size_t)ty.i1 * sw + tx.i0) * C)[c4];
        const float4 v11 = reinterpret_cast<const float4*>(src + ((size_t)ty.i1 * sw + tx.i1) * C)[c4];
        float4 o;
        o.x = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
        o.y = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
        o.z = ty.l0 * (tx.l0 * v00.z + tx.l1 * v01.z) + ty.l1 * (tx.l0 * v10.z + tx.l1 * v11.z);
        o.w = ty.l0 * (tx.l0 * v00.w + tx.l1 * v01.w) + ty.l1 * (tx.l0 * v10.w + tx.l1 * v11.w);
        *reinterpret_cast<float4*>(dst + (size_t)p * ld + c0 + 4 * c4) = o;
    }
}

extern "C" int usc_upsample_nc_nhwc_f32(const float* src, int sh, int sw, int C, float* dst, int dh, int dw, int ld, int c0,
                                        usc_stream_t stream) {
    if (!src || !dst || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || C <= 0 || C % 4 || ld % 4 || c0 % 4 || c0 + C > ld)
        return fail(USC_E_ARG, "usc_upsample_nc_nhwc_f32: bad argument (C, ld, c0 must be multiples of 4, c0 + C <= ld)");
    if ((long long)dh * dw * (C / 4) >= (1ll << 31)) return fail(USC_E_UNSUPPORTED, "usc_upsample_nc_nhwc_f32: too large");
    hipLaunchKernelGGL(upsample_nc_nhwc_kernel, dim3(grid_for((size_t)dh * dw * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream, src, sh,
                       sw, C, dst, dh, dw, ld, c0, (float)sh / (float)dh, (float)sw / (float)dw);
    return check_launch("upsample_nc_nhwc_kernel");
}

// --------------------------------------------------------------------------------------------
// nn.AdaptiveAvgPool2d(s) on NHWC f32 (human_parse.py:92-96 PSP stages, deeplab/_deeplab.py:166 image pooling): window
// [floor(o*H/s), ceil((o+1)*H/s)) in float32, no atomics: the result does not depend on launch shape or batching.
// --------------------------------------------------------------------------------------------
// A workgroup = one output cell x 32 channels x 8 row partitions: partition p sums rows y0 + p, y0 + p + 8, ... (each row left
// to right), the 8 partial sums are combined in order 0..7 - fixed for any launch shape or batch.
__global__ __launch_bounds__(256) void adaptive_avgpool_kernel(const float* __restrict__ src0, int H, int W, int ld, int c0, int C,
                                                               float* __restrict__ dst0, int s) {
    __shared__ float part[8][32];
    const int cl = threadIdx.x & 31, p = threadIdx.x >> 5;
    const int cgroups = (C + 31) / 32;
    const int cell = blockIdx.x / cgroups, c = (blockIdx.x - cell * cgroups) * 32 + cl;
    const int img = blockIdx.y;                                  // images stacked along y in src, cells stacked in dst
    const float* __restrict__ src = src0 + (size_t)img * H * W * ld;
    const int oy = cell / s, ox = cell - oy * s;
    const int y0 = (oy * H) / s, y1 = ((oy + 1) * H + s - 1) / s, x0 = (ox * W) / s, x1 = ((ox + 1) * W + s - 1) / s;
    float acc = 0.f;
    if (c < C)
        for (int y = y0 + p; y < y1; y += 8)
            for (int x = x0; x < x1; x++) acc += src[((size_t)y * W + x) * ld + c0 + c];
    part[p][cl] = acc;
    __syncthreads();
    if (p == 0 && c < C) {
        float t = part[0][cl];
#pragma unroll
        for (int k = 1; k < 8; k++) t += part[k][cl];
        dst0[((size_t)img * s * s + cell) * C + c] = t / (float)((y1 - y0) * (x1 - x0));
    }
}

extern "C" int usc_adaptive_avgpool_nhwc_batch_f32(const float* src, int H, int W, int ld, int c0, int C, float* dst, int s, int batch,
                                                   usc_stream_t stream) {
    if (!src || !dst || H <= 0 || W <= 0 || C <= 0 || c0 < 0 || c0 + C > ld || s <= 0 || s > H || s > W || batch < 1 || batch > 65535)
        return fail(USC_E_ARG, "usc_adaptive_avgpool_nhwc_f32: bad argument");
    hipLaunchKernelGGL(adaptive_avgpool_kernel, dim3(s * s * ((C + 31) / 32), batch), dim3(256), 0, (hipStream_t)stream, src, H, W, ld, c0, C,
                       dst, s);
    return check_launch("adaptive_avgpool_kernel");
}

extern "C" int usc_adaptive_avgpool_nhwc_f32(const float* src, int H, int W, int ld, int c0, int C, float* dst, int s, usc_stream_t stream) {
    return usc_adaptive_avgpool_nhwc_batch_f32(src, H, W, ld, c0, C, dst, s, 1, stream);
}

// --------------------------------------------------------------------------------------------
// nn.MaxPool2d(kernel_size=3, stride=2, padding=1) on NHWC f32 - the ResNet stems of the four segmentation / tracking networks
// (binseg/human_parse.py:210, binseg/deeplab/backbone/resnet.py:185, stm/model.py:46,58,73,83 (torchvision's resnet.maxpool), iseg/brs/models/resnetv1b.py:191).  One
// lane = one output pixel x 4 channels: nine 16-byte loads (taps outside the image are skipped: the padding is -inf), one 16-byte
// store; consecutive lanes = consecutive channel groups of a pixel, so every load is a contiguous run.  HBM-bound: the input is read
// once (2.25 taps per input element come from L2), e.g. 12 x 237 x 237 x 128 floats of the human-parsing net: 345 MB in, 87 MB out.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float4* __restrict__ src, int H, int W, int C4, float4* __restrict__ dst,
                                                           int OH, int OW, int batch) {
    const size_t total = (size_t)batch * OH * OW * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % OW);
        p /= OW;
        const int oy = (int)(p % OH), img = (int)(p / OH);
        const float4* __restrict__ base = src + (size_t)img * H * W * C4 + c;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            const int y = 2 * oy - 1 + dy;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) {
                const int x = 2 * ox - 1 + dx;
                if (x < 0 || x >= W) continue;
                const float4 v = base[((size_t)y * W + x) * C4];
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        dst[i] = m;
    }
}

extern "C" int usc_maxpool3x3s2_nhwc_f32(const float* src, int H, int W, int C, float* dst, int batch, usc_stream_t stream) {
    if (!src || !dst || H <= 0 || W <= 0 || C <= 0 || (C & 3) || batch < 1 || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return fail(USC_E_ARG, "usc_maxpool3x3s2_nhwc_f32: bad argument (C % 4, 16-byte aligned tensors)");
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;                 // floor((H + 2 - 3) / 2) + 1
    const size_t total = (size_t)batch * OH * OW * (C / 4);
    const size_t blocks = (total + 255) / 256;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const float4*>(src), H, W, C / 4, reinterpret_cast<float4*>(dst), OH, OW, batch);
    return check_launch("maxpool3x3s2_kernel");
}

// --------------------------------------------------------------------------------------------
// post (agent.py:200-237, 143-174; utils/imgprocess.py:100-132)
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void segagent_merge_kernel(const float* __restrict__ logits, CropTable ct, int lh, int lw, int cbuf,
                                                             int crop_h, int crop_w, int th, int tw, double* __restrict__ pred) {
    const uint32_t n = (uint32_t)th * (uint32_t)tw;
    const float rh = (float)lh / (float)crop_h, rw = (float)lw / (float)crop_w;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int Y = (int)(i / (uint32_t)tw), X = (int)(i - (uint32_t)Y * (uint32_t)tw);
        double p0 = 0., p1 = 0., cnt = 0.;
        for (int k = 0; k < ct.n; k++) {
            const int y = Y - ct.s_h[k], x = X - ct.s_w[k];
            if (y < 0 || y >= crop_h || x < 0 || x >= crop_w) continue;
            const int sx = ct.flip[k] ? crop_w - 1 - x : x;
            const NcTap ty = nc_tap(y, lh, rh), tx = nc_tap(sx, lw, rw);
            const float* lg = logits + (size_t)k * lh * lw * cbuf;
            const float2 v00 = *reinterpret_cast<const float2*>(lg + ((size_t)ty.i0 * lw + tx.i0) * cbuf);
            const float2 v01 = *reinterpret_cast<const float2*>(lg + ((size_t)ty.i0 * lw + tx.i1) * cbuf);
            const float2 v10 = *reinterpret_cast<const float2*>(lg + ((size_t)ty.i1 * lw + tx.i0) * cbuf);
            const float2 v11 = *reinterpret_cast<const float2*>(lg + ((size_t)ty.i1 * lw + tx.i1) * cbuf);
            const float a = ty.l0 * (tx.l0 * v00.x + tx.l1 * v01.x) + ty.l1 * (tx.l0 * v10.x + tx.l1 * v11.x);
            const float b = ty.l0 * (tx.l0 * v00.y + tx.l1 * v01.y) + ty.l1 * (tx.l0 * v10.y + tx.l1 * v11.y);
            const float m = a > b ? a : b;
            const float ea = det_expf(a - m), eb = det_expf(b - m);
            const float s = ea + eb;
            p0 += (double)(ea / s);
            p1 += (double)(eb / s);
            cnt += 1.0;
        }
        pred[2 * (size_t)i] = p0 / cnt;
        pred[2 * (size_t)i + 1] = p1 / cnt;
    }
}

__global__ __launch_bounds__(256) void segagent_finish_kernel(const double* __restrict__ pred, int tw, int rh, int rw, int H, int W,
                                                              int same, double scale_y, double scale_x, uint8_t* __restrict__ out,
                                                              double* __restrict__ score_out) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        double v0, v1;
        if (same) {
            v0 = pred[((size_t)y * tw + x) * 2];
            v1 = pred[((size_t)y * tw + x) * 2 + 1];
        } else {
            const Tap tx = linear_tap_x(x, scale_x, rw), ty = linear_tap(y, scale_y);
            const int x0 = tx.s, x1 = min(x0 + 1, rw - 1);
            const int y0 = clampi(ty.s, 0, rh - 1), y1 = clampi(ty.s + 1, 0, rh - 1);
            const double a0 = (double)(1.f - tx.f), a1 = (double)tx.f, b0 = (double)(1.f - ty.f), b1 = (double)ty.f;
            const double* r0 = pred + (size_t)y0 * tw * 2;
            const double* r1 = pred + (size_t)y1 * tw * 2;
            const double t0 = r0[2 * x0] * a0 + r0[2 * x1] * a1, u0 = r1[2 * x0] * a0 + r1[2 * x1] * a1;
            const double t1 = r0[2 * x0 + 1] * a0 + r0[2 * x1 + 1] * a1, u1 = r1[2 * x0 + 1] * a0 + r1[2 * x1 + 1] * a1;
            v0 = t0 * b0 + u0 * b1;
            v1 = t1 * b0 + u1 * b1;
        }
        if (score_out) { score_out[2 * (size_t)i] = v0; score_out[2 * (size_t)i + 1] = v1; }
        out[i] = v1 > v0 ? 255 : 0;
    }
}

extern "C" size_t usc_segagent_post_ws_bytes(int th, int tw) { return (size_t)th * tw * 2 * sizeof(double); }

extern "C" int usc_segagent_post(const float* logits, int ncrops, int lh, int lw, int cbuf, const int* locs_host, int crop_h, int crop_w,
                                 int th, int tw, int H, int W, uint8_t* out, double* score_out, void* ws, size_t ws_bytes,
                                 usc_stream_t stream) {
    if (!logits || !locs_host || !out || !ws || ncrops <= 0 || lh <= 0 || lw <= 0 || cbuf < 2 || cbuf % 2 || crop_h <= 0 || crop_w <= 0 ||
        th < crop_h || tw < crop_w || H <= 0 || W <= 0)
        return fail(USC_E_ARG, "usc_segagent_post: bad argument");
    if (ncrops > kMaxCrops) return fail(USC_E_UNSUPPORTED, "usc_segagent_post: more than %d crops", kMaxCrops);
    if (too_big(H, W) || too_big(th, tw)) return fail(USC_E_UNSUPPORTED, "usc_segagent_post: image too large");
    if (ws_bytes < usc_segagent_post_ws_bytes(th, tw)) return fail(USC_E_WORKSPACE, "usc_segagent_post: workspace too small");
    CropTable ct;
    memset(&ct, 0, sizeof(ct));
    ct.n = ncrops;
    for (int i = 0; i < ncrops; i++) {
        const int s_h = locs_host[4 * i], e_h = locs_host[4 * i + 1], s_w = locs_host[4 * i + 2], e_w = locs_host[4 * i + 3];
        const int flip = s_w > e_w, left = flip ? e_w : s_w, right = flip ? s_w : e_w;
        if (e_h - s_h != crop_h || right - left != crop_w || s_h < 0 || left < 0 || e_h > th || right > tw)
            return fail(USC_E_ARG, "usc_segagent_post: crop %d is not a crop_h x crop_w window of the target", i);
        ct.s_h[i] = s_h; ct.s_w[i] = left; ct.flip[i] = flip;
    }
    hipStream_t st = (hipStream_t)stream;
    double* pred = reinterpret_cast<double*>(ws);
    hipLaunchKernelGGL(segagent_merge_kernel, dim3(grid_for((size_t)th * tw, 256)), dim3(256), 0, st, logits, ct, lh, lw, cbuf, crop_h, crop_w,
                       th, tw, pred);
    int rh, rw;
    usc_inv_pad_geometry(th, tw, H, W, &rh, &rw);
    const int same = (rh == H && rw == W);
    const double scale_x = 1. / ((double)W / rw), scale_y = 1. / ((double)H / rh);
    hipLaunchKernelGGL(segagent_finish_kernel, dim3(grid_for((size_t)H * W, 256)), dim3(256), 0, st, pred, tw, rh, rw, H, W, same, scale_y,
                       scale_x, out, score_out);
    return check_launch("segagent_post kernels");
}
