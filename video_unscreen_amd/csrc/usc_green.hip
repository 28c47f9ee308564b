// usc_green.hip - green-screen mode pixel kernels (tools/unscreen/green.py:70-126 of the reference): colour conversion,
// per-pixel GMM likelihood ratio (unscreen/colorfiltering/agent.py), Lab colour correction (unscreen/utils/imgprocess.py:263-300)
// and the box filter of the harmonization helpers.  All HBM-bound byte work at the 540x960 working size: one lane per pixel,
// lookup tables staged in LDS, wave-level reductions + one atomic per workgroup for the global statistics.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

// --------------------------------------------------------------------------------------------
// 8-bit BGR -> Lab tables (cv::RGB2Lab_b: lab_shift 12, gamma_shift 3)
// --------------------------------------------------------------------------------------------
namespace {

struct LabTables {
    uint16_t gamma[256];      // sRGB gamma, scale 255 * 8
    uint16_t cbrt[3072];      // f(t) of CIE Lab, scale 2^15, argument scale 255 * 8
    int coeffs[9];            // round(4096 * sRGB2XYZ_D65 / D65 white), columns in B, G, R order
    int pad[3];
};
static_assert(sizeof(LabTables) == 512 + 6144 + 48, "LabTables layout");

inline int round_half_even(double v) { return (int)lrint(v); }

void fill_lab_tables(LabTables* t) {
    static const double xyz[9] = {0.412453, 0.357580, 0.180423, 0.212671, 0.715160, 0.072169, 0.019334, 0.119193, 0.950227};
    static const double white[3] = {0.950456, 1., 1.088754};
    for (int i = 0; i < 256; i++) {
        const float x = (float)i / 255.f;
        const float g = x <= 0.04045f ? x / 12.92f : powf((x + 0.055f) / 1.055f, 2.4f);
        t->gamma[i] = (uint16_t)round_half_even((double)(2040.f * g));
    }
    const float lthresh = 216.f / 24389.f, lscale = 841.f / 108.f, lbias = 16.f / 116.f;
    for (int i = 0; i < 3072; i++) {
        const float x = (float)i / 2040.f;
        const float v = x < lthresh ? fmaf(x, lscale, lbias) : cbrtf(x);
        t->cbrt[i] = (uint16_t)round_half_even((double)(32768.f * v));
    }
    for (int i = 0; i < 3; i++) {
        t->coeffs[i * 3 + 2] = round_half_even(4096. * xyz[i * 3 + 0] / white[i]);
        t->coeffs[i * 3 + 1] = round_half_even(4096. * xyz[i * 3 + 1] / white[i]);
        t->coeffs[i * 3 + 0] = round_half_even(4096. * xyz[i * 3 + 2] / white[i]);
    }
    t->pad[0] = t->pad[1] = t->pad[2] = 0;
}

template <typename T>
__host__ __device__ __forceinline__ void bgr2lab_u8(const T* t, int c0, int c1, int c2, int& L, int& a, int& b) {
    const int Lscale = (116 * 255 + 50) / 100, Lshift = -((16 * 255 * (1 << 15) + 50) / 100);
    const int g0 = t->gamma[c0], g1 = t->gamma[c1], g2 = t->gamma[c2];
    const int* C = t->coeffs;
    const int fX = t->cbrt[(g0 * C[0] + g1 * C[1] + g2 * C[2] + (1 << 11)) >> 12];
    const int fY = t->cbrt[(g0 * C[3] + g1 * C[4] + g2 * C[5] + (1 << 11)) >> 12];
    const int fZ = t->cbrt[(g0 * C[6] + g1 * C[7] + g2 * C[8] + (1 << 11)) >> 12];
    const int l = (Lscale * fY + Lshift + (1 << 14)) >> 15;
    const int aa = (500 * (fX - fY) + 128 * (1 << 15) + (1 << 14)) >> 15;
    const int bb = (200 * (fY - fZ) + 128 * (1 << 15) + (1 << 14)) >> 15;
    L = l < 0 ? 0 : (l > 255 ? 255 : l);
    a = aa < 0 ? 0 : (aa > 255 ? 255 : aa);
    b = bb < 0 ? 0 : (bb > 255 ? 255 : bb);
}

// 8-bit Lab -> BGR (cv::Lab2RGB_b, integer path Lab2RGBinteger: BASE 2^14, inverse gamma table of 4096 entries); the
// a / b -> x / z table of OpenCV is a closed form in integers and is evaluated on the fly.
struct LabInvTables {
    uint16_t yf[512];         // (y, ify) per L
    uint16_t invgamma[4096];  // round(255 * sRGB inverse gamma(i / 4096))
    int coeffs[9];            // round(4096 * XYZ2sRGB_D65 * white), rows B, G, R
    int pad[3];
};
static_assert(sizeof(LabInvTables) == 1024 + 8192 + 48, "LabInvTables layout");

void fill_lab_inv_tables(LabInvTables* t) {
    static const double m[9] = {3.240479, -1.53715, -0.498535, -0.969256, 1.875991, 0.041556, 0.055648, -0.204043, 1.057311};
    static const double white[3] = {0.950456, 1., 1.088754};
    const int BASE = 1 << 14;
    for (int i = 0; i < 256; i++) {
        int y, ify;
        if (i <= 20) {
            y = round_half_even((double)((float)(i * BASE * 20 * 9) / (float)(17 * 29 * 29 * 29)));
            ify = round_half_even((double)((float)BASE * (16.f / 116.f + (float)(i * 5) / (float)(3 * 17 * 29))));
        } else {
            const float fy = (float)(i * 100 * BASE) / (float)(255 * 116) + (float)(16 * BASE) / 116.f;
            ify = round_half_even((double)fy);
            y = round_half_even((double)(fy * fy * fy / (float)((double)BASE * BASE)));
        }
        t->yf[2 * i] = (uint16_t)y;
        t->yf[2 * i + 1] = (uint16_t)ify;
    }
    for (int i = 0; i < 4096; i++) {
        const float x = (1.f / 4096.f) * (float)i;
        const double xd = (double)x;
        const float v = (float)(xd <= 0.0031308 ? xd * 12.92 : pow(xd, 1. / 2.4) * 1.055 - 0.055);
        t->invgamma[i] = (uint16_t)round_half_even((double)(255.f * v));
    }
    for (int i = 0; i < 3; i++) {
        t->coeffs[0 + i] = round_half_even(4096. * m[6 + i] * white[i]);
        t->coeffs[3 + i] = round_half_even(4096. * m[3 + i] * white[i]);
        t->coeffs[6 + i] = round_half_even(4096. * m[0 + i] * white[i]);
    }
    t->pad[0] = t->pad[1] = t->pad[2] = 0;
}

__host__ __device__ __forceinline__ int lab_ab_to_xz(int v) {
    constexpr int BASE = 1 << 14;
    if (v <= 3390) return v * 108 / 841 - BASE * 16 / 116 * 108 / 841;
    return v * v / BASE * v / BASE;
}

template <typename T>
__host__ __device__ __forceinline__ void lab2bgr_u8(const T* t, int L, int a, int b, int& o0, int& o1, int& o2) {
    constexpr int BASE = 1 << 14;
    const int y = t->yf[2 * L], ify = t->yf[2 * L + 1];
    const int adiv = ((5 * a * 53687 + (1 << 7)) >> 13) - 128 * BASE / 500;
    const int bdiv = ((b * 41943 + (1 << 4)) >> 9) - 128 * BASE / 200 + 1;
    const int x = lab_ab_to_xz(ify + adiv), z = lab_ab_to_xz(ify - bdiv);
    const int* C = t->coeffs;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int w = (C[3 * c] * x + C[3 * c + 1] * y + C[3 * c + 2] * z + (1 << 13)) >> 14;
        w = w < 0 ? 0 : (w > 4095 ? 4095 : w);
        v[c] = t->invgamma[w];
    }
    o0 = v[0]; o1 = v[1]; o2 = v[2];
}

__device__ __forceinline__ void load_lab_inv_tables(LabInvTables* s, const LabInvTables* __restrict__ g) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s);
    for (int i = threadIdx.x; i < (int)(sizeof(LabInvTables) / 4); i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ void load_lab_tables(LabTables* s, const LabTables* __restrict__ g) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(g);
    uint32_t* dst = reinterpret_cast<uint32_t*>(s);
    for (int i = threadIdx.x; i < (int)(sizeof(LabTables) / 4); i += blockDim.x) dst[i] = src[i];
}

}  // namespace

extern "C" size_t usc_lab_tables_bytes(void) { return sizeof(LabTables); }
extern "C" void usc_lab_tables_host(void* blob) { fill_lab_tables(reinterpret_cast<LabTables*>(blob)); }
extern "C" size_t usc_lab_inv_tables_bytes(void) { return sizeof(LabInvTables); }
extern "C" void usc_lab_inv_tables_host(void* blob) { fill_lab_inv_tables(reinterpret_cast<LabInvTables*>(blob)); }

// --------------------------------------------------------------------------------------------
// cvtColor
// --------------------------------------------------------------------------------------------
template <int CODE>
__global__ __launch_bounds__(256) void cvt_color_kernel(const uint8_t* __restrict__ src, size_t n, uint8_t* __restrict__ dst,
                                                        const LabTables* __restrict__ lab) {
    __shared__ int s_sdiv[256], s_hdiv[256];
    __shared__ LabTables s_lab;
    __shared__ LabInvTables s_inv;
    if (CODE == USC_CVT_BGR2HSV) load_hsv_tables(s_sdiv, s_hdiv);
    if (CODE == USC_CVT_BGR2LAB) load_lab_tables(&s_lab, lab);
    if (CODE == USC_CVT_LAB2BGR) load_lab_inv_tables(&s_inv, reinterpret_cast<const LabInvTables*>(lab));
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int c0 = src[3 * i], c1 = src[3 * i + 1], c2 = src[3 * i + 2];
        int o0, o1, o2;
        if (CODE == USC_CVT_BGR2HSV) bgr2hsv_u8(c0, c1, c2, s_sdiv, s_hdiv, o0, o1, o2);
        else if (CODE == USC_CVT_HSV2BGR) hsv2bgr_u8(c0, c1, c2, o0, o1, o2);
        else if (CODE == USC_CVT_BGR2LAB) bgr2lab_u8(&s_lab, c0, c1, c2, o0, o1, o2);
        else if (CODE == USC_CVT_LAB2BGR) lab2bgr_u8(&s_inv, c0, c1, c2, o0, o1, o2);
        else { o0 = gray_u8(c0, c1, c2); o1 = o2 = 0; }
        if (CODE == USC_CVT_BGR2GRAY) dst[i] = (uint8_t)o0;
        else { dst[3 * i] = (uint8_t)o0; dst[3 * i + 1] = (uint8_t)o1; dst[3 * i + 2] = (uint8_t)o2; }
    }
}

extern "C" int usc_cvt_color_u8(const uint8_t* src, size_t n, int code, uint8_t* dst, const void* lab_tables, usc_stream_t stream) {
    if (!src || !dst || n == 0) return fail(USC_E_ARG, "usc_cvt_color_u8: bad argument");
    if (n >= ((size_t)1 << 31)) return fail(USC_E_UNSUPPORTED, "usc_cvt_color_u8: image too large");
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid_for(n, 256)), b(256);
    const LabTables* lt = reinterpret_cast<const LabTables*>(lab_tables);
    switch (code) {
        case USC_CVT_BGR2HSV: hipLaunchKernelGGL(cvt_color_kernel<USC_CVT_BGR2HSV>, g, b, 0, st, src, n, dst, lt); break;
        case USC_CVT_HSV2BGR: hipLaunchKernelGGL(cvt_color_kernel<USC_CVT_HSV2BGR>, g, b, 0, st, src, n, dst, lt); break;
        case USC_CVT_BGR2GRAY: hipLaunchKernelGGL(cvt_color_kernel<USC_CVT_BGR2GRAY>, g, b, 0, st, src, n, dst, lt); break;
        case USC_CVT_BGR2LAB:
            if (!lt) return fail(USC_E_ARG, "usc_cvt_color_u8: BGR2LAB needs the device copy of usc_lab_tables_host()");
            hipLaunchKernelGGL(cvt_color_kernel<USC_CVT_BGR2LAB>, g, b, 0, st, src, n, dst, lt);
            break;
        case USC_CVT_LAB2BGR:
            if (!lt) return fail(USC_E_ARG, "usc_cvt_color_u8: LAB2BGR needs the device copy of usc_lab_inv_tables_host()");
            hipLaunchKernelGGL(cvt_color_kernel<USC_CVT_LAB2BGR>, g, b, 0, st, src, n, dst, lt);
            break;
        default: return fail(USC_E_ARG, "usc_cvt_color_u8: unknown conversion code %d", code);
    }
    return check_launch("cvt_color_kernel");
}

// --------------------------------------------------------------------------------------------
// colour filtering: GMM likelihood ratio (colorfiltering/agent.py:201-257) and the adaptive threshold (:259-280)
// --------------------------------------------------------------------------------------------
extern "C" int usc_gmm_lut_host(const double* means, const double* stds, const double* weights, int K, float* lut256) {
    if (!means || !stds || !weights || !lut256 || K <= 0) return fail(USC_E_ARG, "usc_gmm_lut_host: bad argument");
    const float s2pi = (float)sqrt(2 * 3.141592653589793);
    for (int v = 0; v < 256; v++) {
        float acc = 0.f;
        for (int k = 0; k < K; k++) {
            const float m = (float)means[k], sd = (float)stds[k], w = (float)weights[k];
            const float x = ((float)v - m) / sd;
            const float y = (1.f / (sd * s2pi)) * expf(-0.5f * (x * x));
            acc = acc + w * y;
        }
        lut256[v] = acc;
    }
    return USC_OK;
}

__global__ __launch_bounds__(256) void gmm_alpha_kernel(const uint8_t* __restrict__ hsv, size_t n, const float* __restrict__ luts,
                                                        uint8_t* __restrict__ alpha, float* __restrict__ prob_out) {
    __shared__ float s_lut[6 * 256];       // bg h, s, v ; fg h, s, v
    for (int i = threadIdx.x; i < 6 * 256; i += blockDim.x) s_lut[i] = luts[i];
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int h = hsv[3 * i], s = hsv[3 * i + 1], v = hsv[3 * i + 2];
        float bg = 1.f * s_lut[h];
        bg = bg * s_lut[256 + s];
        bg = bg * s_lut[512 + v];
        float fg = 1.f * s_lut[768 + h];
        fg = fg * s_lut[1024 + s];
        fg = fg * s_lut[1280 + v];
        bg = (float)cbrt((double)bg);
        fg = (float)cbrt((double)fg);
        const float p = fg / (bg + fg + 1e-6f);
        if (prob_out) prob_out[i] = p;
        float a = p * 255.f;
        a = a < 0.f ? 0.f : (a > 255.f ? 255.f : a);
        alpha[i] = (uint8_t)a;
    }
}

extern "C" int usc_gmm_alpha_u8(const uint8_t* hsv, size_t n, const float* luts, uint8_t* alpha, float* prob_out, usc_stream_t stream) {
    if (!hsv || !luts || !alpha || n == 0) return fail(USC_E_ARG, "usc_gmm_alpha_u8: bad argument");
    hipLaunchKernelGGL(gmm_alpha_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, hsv, n, luts, alpha, prob_out);
    return check_launch("gmm_alpha_kernel");
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
    return v;
}

__global__ __launch_bounds__(256) void cf_stats_kernel(const uint8_t* __restrict__ alpha, const uint8_t* __restrict__ mask, size_t n,
                                                       unsigned long long* __restrict__ stats) {
    unsigned long long sum = 0, cnt = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int a = alpha[i];
        if (a > 128 && mask[i] > 0) { sum += a; cnt++; }
    }
    sum = wave_sum_u64(sum);
    cnt = wave_sum_u64(cnt);
    __shared__ unsigned long long s_sum[4], s_cnt[4];
    if ((threadIdx.x & (kWave - 1)) == 0) { s_sum[threadIdx.x >> 6] = sum; s_cnt[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {                       // one pair of (integer, order-independent) atomics per block
        for (int w = 1; w < 4; w++) { sum += s_sum[w]; cnt += s_cnt[w]; }
        if (cnt) { atomicAdd(&stats[0], sum); atomicAdd(&stats[1], cnt); }
    }
}

__global__ __launch_bounds__(256) void cf_apply_kernel(uint8_t* __restrict__ alpha, size_t n, const unsigned long long* __restrict__ stats,
                                                       double thr_ratio) {
    const double thr = ((double)stats[0] / (double)stats[1]) * thr_ratio;       // 0/0 = nan: nothing is below it
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if ((double)alpha[i] < thr) alpha[i] = 0;
}

extern "C" size_t usc_cf_threshold_ws_bytes(void) { return 256; }

extern "C" int usc_cf_threshold_u8(uint8_t* alpha, const uint8_t* mask, size_t n, double thr_ratio, void* ws, size_t ws_bytes,
                                   usc_stream_t stream) {
    if (!alpha || !mask || !ws || n == 0) return fail(USC_E_ARG, "usc_cf_threshold_u8: bad argument");
    if (ws_bytes < usc_cf_threshold_ws_bytes()) return fail(USC_E_WORKSPACE, "usc_cf_threshold_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned long long* stats = reinterpret_cast<unsigned long long*>(ws);
    if (hipMemsetAsync(stats, 0, 16, st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_cf_threshold_u8: memset failed");
    hipLaunchKernelGGL(cf_stats_kernel, dim3(grid_for(n, 256 * 8)), dim3(256), 0, st, alpha, mask, n, stats);      // 8 px per lane: ~1000 blocks
    hipLaunchKernelGGL(cf_apply_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, alpha, n, stats, thr_ratio);
    return check_launch("cf_threshold kernels");
}

// --------------------------------------------------------------------------------------------
// color_correct (utils/imgprocess.py:263-300)
// --------------------------------------------------------------------------------------------
namespace {

constexpr int kSqrtLevels = 16;           // x^(1/2^15) >= 0.95 for every positive float: the loop always ends within 16 levels
struct CcStats {
    unsigned int dmin_bits, dmax_bits;    // dist >= 0: float order == unsigned order of the bit patterns
    unsigned long long count;
    double sums[kSqrtLevels];
    int levels;                           // number of sqrt passes the reference's while loop would run
};
constexpr int kCcBlocks = 256;            // blocks of the two reduction kernels = rows of the partial-sum table
struct CcPartial { double sums[kSqrtLevels]; unsigned long long count; };

__device__ __forceinline__ int torch_nearest_idx(int o, int in, int out, float scale) {
    if (out == in) return o;
    if (out == 2 * in) return o >> 1;
    const int s = (int)floorf((float)o * scale);
    return min(s, in - 1);
}

}  // namespace

__global__ __launch_bounds__(256) void cc_dist_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ alpha, int H, int W,
                                                      int th, int tw, int mode, double scale_y, double scale_x,
                                                      const LabTables* __restrict__ lab, float ba, float bb,
                                                      float* __restrict__ dist, uint8_t* __restrict__ alpha_r, CcStats* __restrict__ st) {
    __shared__ LabTables s_lab;
    load_lab_tables(&s_lab, lab);
    __syncthreads();
    const uint32_t n = (uint32_t)th * (uint32_t)tw;
    float dmin = INFINITY, dmax = 0.f;
    bool any = false;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)tw), x = (int)(i - (uint32_t)y * (uint32_t)tw);
        const int c0 = resize_sample_u8(img, H, W, 3, 0, y, x, 1, mode, scale_y, scale_x, 0., 0.);
        const int c1 = resize_sample_u8(img, H, W, 3, 1, y, x, 1, mode, scale_y, scale_x, 0., 0.);
        const int c2 = resize_sample_u8(img, H, W, 3, 2, y, x, 1, mode, scale_y, scale_x, 0., 0.);
        alpha_r[i] = (uint8_t)resize_sample_u8(alpha, H, W, 1, 0, y, x, 1, mode, scale_y, scale_x, 0., 0.);
        int L, a, b;
        bgr2lab_u8(&s_lab, c0, c1, c2, L, a, b);
        const float da = (float)a / 255.f - ba, db = (float)b / 255.f - bb;
        const float d = sqrtf(da * da + db * db);
        dist[i] = d;
        dmin = fminf(dmin, d);
        dmax = fmaxf(dmax, d);
        any = true;
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        dmin = fminf(dmin, __shfl_down(dmin, off, kWave));
        dmax = fmaxf(dmax, __shfl_down(dmax, off, kWave));
    }
    (void)any;
    __shared__ float s_min[4], s_max[4];
    if ((threadIdx.x & (kWave - 1)) == 0) { s_min[threadIdx.x >> 6] = dmin; s_max[threadIdx.x >> 6] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {                       // one pair of atomics per block (min / max: order independent)
        for (int w = 1; w < 4; w++) { dmin = fminf(dmin, s_min[w]); dmax = fmaxf(dmax, s_max[w]); }
        if (dmin != INFINITY) {
            atomicMin(&st->dmin_bits, __float_as_uint(dmin));
            atomicMax(&st->dmax_bits, __float_as_uint(dmax));
        }
    }
}

__global__ __launch_bounds__(256) void cc_sums_kernel(float* __restrict__ dist, const uint8_t* __restrict__ alpha_r, uint32_t n,
                                                      const CcStats* __restrict__ st, CcPartial* __restrict__ part) {
    const float dmin = __uint_as_float(st->dmin_bits), range = __uint_as_float(st->dmax_bits) - dmin;
    double sums[kSqrtLevels];
#pragma unroll
    for (int k = 0; k < kSqrtLevels; k++) sums[k] = 0.;
    unsigned long long cnt = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float d = (dist[i] - dmin) / range;
        dist[i] = d;
        if (alpha_r[i] > 0 && d > 0.f) {
            cnt++;
#pragma unroll
            for (int k = 0; k < kSqrtLevels; k++) { sums[k] += (double)d; d = sqrtf(d); }
        }
    }
    // fixed-order reduction (lanes, then the block's 4 waves): the table row of this block, no atomics - the level count
    // below is then the same from run to run
    __shared__ double s_sum[4][kSqrtLevels];
    __shared__ unsigned long long s_cnt[4];
    cnt = wave_sum_u64(cnt);
#pragma unroll
    for (int k = 0; k < kSqrtLevels; k++) {
        double v = sums[k];
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) s_sum[threadIdx.x >> 6][k] = v;
    }
    if ((threadIdx.x & (kWave - 1)) == 0) s_cnt[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x < kSqrtLevels) part[blockIdx.x].sums[threadIdx.x] = ((s_sum[0][threadIdx.x] + s_sum[1][threadIdx.x]) + s_sum[2][threadIdx.x]) + s_sum[3][threadIdx.x];
    if (threadIdx.x == 0) part[blockIdx.x].count = ((s_cnt[0] + s_cnt[1]) + s_cnt[2]) + s_cnt[3];
}

// One block: sum the table rows in row order and run the reference's `while mean(...) < mean_exp: dist = sqrt(dist)` on the sums.
__global__ __launch_bounds__(64) void cc_levels_kernel(const CcPartial* __restrict__ part, int rows, float mean_exp, CcStats* __restrict__ st,
                                                       int* __restrict__ n_sqrt_out) {
    __shared__ double s_tot[kSqrtLevels];
    __shared__ unsigned long long s_count;
    if (threadIdx.x < kSqrtLevels) {
        double v = 0.;
        for (int r = 0; r < rows; r++) v += part[r].sums[threadIdx.x];
        s_tot[threadIdx.x] = v;
        st->sums[threadIdx.x] = v;
    }
    if (threadIdx.x == kSqrtLevels) {
        unsigned long long c = 0;
        for (int r = 0; r < rows; r++) c += part[r].count;
        s_count = c;
        st->count = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int levels = 0;
        const double c = (double)s_count;
        for (; levels < kSqrtLevels; levels++) {
            const float mean = (float)(s_tot[levels] / c);         // count 0 -> nan -> the loop condition is false
            if (!(mean < mean_exp)) break;
        }
        st->levels = levels;
        if (n_sqrt_out) *n_sqrt_out = levels;
    }
}

__global__ __launch_bounds__(256) void cc_apply_kernel(const uint8_t* __restrict__ alpha, int H, int W, const float* __restrict__ dist,
                                                       const uint8_t* __restrict__ alpha_r, int th, int tw, float sy_scale, float sx_scale,
                                                       const CcStats* __restrict__ st, uint8_t* __restrict__ out) {
    const int levels = st->levels;
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        const int sy = torch_nearest_idx(y, th, H, sy_scale), sx = torch_nearest_idx(x, tw, W, sx_scale);
        const uint32_t j = (uint32_t)sy * (uint32_t)tw + (uint32_t)sx;
        float d = dist[j];
        for (int k = 0; k < levels; k++) d = sqrtf(d);
        if (alpha_r[j] == 0) d = 0.f;
        const float v = (float)alpha[i] * d;
        out[i] = (uint8_t)(int)v;
    }
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t usc_color_correct_ws_bytes(int H, int W, int long_side) {
    int th, tw;
    usc_get_target_size(H, W, long_side, 1, &th, &tw);
    const size_t n = (size_t)th * tw;
    return 256 + align256(n * sizeof(float)) + align256(n) + align256(kCcBlocks * sizeof(CcPartial));
}

extern "C" int usc_color_correct_u8(const uint8_t* img, const uint8_t* alpha, int H, int W, const uint8_t* bg_color_host, int long_side,
                                    double mean_exp, const void* lab_tables, uint8_t* out, int* n_sqrt_out, void* ws, size_t ws_bytes,
                                    usc_stream_t stream) {
    if (!img || !alpha || !bg_color_host || !lab_tables || !out || !ws || H <= 0 || W <= 0 || long_side <= 0)
        return fail(USC_E_ARG, "usc_color_correct_u8: bad argument");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_color_correct_u8: image too large");
    if (ws_bytes < usc_color_correct_ws_bytes(H, W, long_side)) return fail(USC_E_WORKSPACE, "usc_color_correct_u8: workspace too small");
    int th, tw;
    usc_get_target_size(H, W, long_side, 1, &th, &tw);
    const size_t n = (size_t)th * tw;
    if (too_big(th, tw)) return fail(USC_E_UNSUPPORTED, "usc_color_correct_u8: working size too large");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = reinterpret_cast<uint8_t*>(ws);
    CcStats* stats = reinterpret_cast<CcStats*>(base);
    float* dist = reinterpret_cast<float*>(base + 256);
    uint8_t* alpha_r = base + 256 + align256(n * sizeof(float));
    CcPartial* part = reinterpret_cast<CcPartial*>(alpha_r + align256(n));
    static_assert(sizeof(CcStats) <= 256, "CcStats fits the header");

    static const LabTables* const host_tab = [] { LabTables* t = new LabTables; fill_lab_tables(t); return t; }();   // built once
    int bL, bA, bB;                         // the background colour is one host pixel: convert it here
    bgr2lab_u8(host_tab, bg_color_host[0], bg_color_host[1], bg_color_host[2], bL, bA, bB);
    const float ba = (float)bA / 255.f, bb = (float)bB / 255.f;

    CcStats init;
    memset(&init, 0, sizeof(init));
    init.dmin_bits = 0x7f800000u;           // +inf
    init.dmax_bits = 0u;
    // a pageable-host async copy is staged by the runtime before the call returns: `init` may live on this stack
    if (hipMemcpyAsync(stats, &init, sizeof(init), hipMemcpyHostToDevice, st) != hipSuccess)
        return fail(USC_E_LAUNCH, "usc_color_correct_u8: stats init failed");
    const ResizeGeom rg = resize_geom(H, W, th, tw);
    hipLaunchKernelGGL(cc_dist_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, img, alpha, H, W, th, tw, rg.mode, rg.scale_y, rg.scale_x,
                       reinterpret_cast<const LabTables*>(lab_tables), ba, bb, dist, alpha_r, stats);
    hipLaunchKernelGGL(cc_sums_kernel, dim3(kCcBlocks), dim3(256), 0, st, dist, alpha_r, (uint32_t)n, stats, part);
    hipLaunchKernelGGL(cc_levels_kernel, dim3(1), dim3(64), 0, st, part, kCcBlocks, (float)mean_exp, stats, n_sqrt_out);
    hipLaunchKernelGGL(cc_apply_kernel, dim3(grid_for((size_t)H * W, 256)), dim3(256), 0, st, alpha, H, W, dist, alpha_r, th, tw,
                       (float)th / (float)H, (float)tw / (float)W, stats, out);
    return check_launch("color_correct kernels");
}

// --------------------------------------------------------------------------------------------
// cv2.boxFilter(img, -1, (k, k)), uint8, BORDER_REFLECT_101 (harmonization/agent.py:91-92, 110-111)
// --------------------------------------------------------------------------------------------
__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

__global__ __launch_bounds__(256) void box_filter_kernel(const uint8_t* __restrict__ src, int H, int W, int cn, int k, int div_scale,
                                                         int div_delta, uint8_t* __restrict__ dst) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    const int a = k / 2;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        for (int c = 0; c < cn; c++) {
            unsigned s = 0;
            for (int j = -a; j < k - a; j++) {
                const int yy = reflect101(y + j, H);
                for (int ii = -a; ii < k - a; ii++) s += src[((size_t)yy * W + reflect101(x + ii, W)) * cn + c];
            }
            dst[(size_t)i * cn + c] = (uint8_t)(((s + (unsigned)div_delta) * (unsigned)div_scale) >> 16);
        }
    }
}

extern "C" int usc_box_filter_u8(const uint8_t* src, int H, int W, int cn, int ksize, uint8_t* dst, usc_stream_t stream) {
    if (!src || !dst || H <= 0 || W <= 0 || (cn != 1 && cn != 3) || ksize < 1 || ksize > 16)
        return fail(USC_E_ARG, "usc_box_filter_u8: bad argument (ksize 1..16: 16-bit column sums)");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_box_filter_u8: image too large");
    const int d = ksize * ksize;
    const double scalef = 65536.0 / d;
    int div_scale = (int)floor(scalef), div_delta = d / 2;
    if (scalef - div_scale < 0.5) div_delta++; else div_scale++;
    if (ksize == 1) { div_scale = 65536; div_delta = 0; }      // scale == 1: the sum is stored as is
    hipLaunchKernelGGL(box_filter_kernel, dim3(grid_for((size_t)H * W, 256)), dim3(256), 0, (hipStream_t)stream, src, H, W, cn, ksize,
                       div_scale, div_delta, dst);
    return check_launch("box_filter_kernel");
}

// --------------------------------------------------------------------------------------------
// overall background of a clip (tools/unscreen/bg_offline.py:99-130, stage 2 - disabled with `if 0:` in the reference)
// --------------------------------------------------------------------------------------------
// raw += float32(frame * float32(1 - mask // 255)) ; cont += (mask < 250)        (bg_offline.py:111-117; mask already dilated)
__global__ __launch_bounds__(256) void bg_accumulate_kernel(const uint8_t* __restrict__ frame, const uint8_t* __restrict__ mask, size_t n,
                                                            float* __restrict__ raw, float* __restrict__ cont) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int m = mask[i];
        const float keep = (float)(1 - m / 255);
#pragma unroll
        for (int c = 0; c < 3; c++) raw[3 * i + c] += (float)frame[3 * i + c] * keep;
        cont[i] += m < 250 ? 1.f : 0.f;
    }
}

extern "C" int usc_bg_accumulate_f32(const uint8_t* frame, const uint8_t* mask_dilated, int H, int W, float* raw, float* cont,
                                     usc_stream_t stream) {
    if (!frame || !mask_dilated || !raw || !cont || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_bg_accumulate_f32: bad argument");
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(bg_accumulate_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, frame, mask_dilated, n, raw, cont);
    return check_launch("bg_accumulate_kernel");
}

// mask_always = cont <= 10 ; bg_always = uint8(clip(raw / max(cont, 1), 0, 255)), zero where always masked   (bg_offline.py:118-123)
__global__ __launch_bounds__(256) void bg_finalize_kernel(const float* __restrict__ raw, const float* __restrict__ cont, size_t n,
                                                          uint8_t* __restrict__ bg_always, uint8_t* __restrict__ mask_always) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double c = (double)cont[i];
        const bool always = c <= 10.;
        const double d = c == 0. ? 1. : c;
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            double v = (double)raw[3 * i + ch] / d;
            v = v < 0. ? 0. : (v > 255. ? 255. : v);
            bg_always[3 * i + ch] = always ? 0 : (uint8_t)(int)v;
        }
        mask_always[i] = always ? 255 : 0;
    }
}

extern "C" int usc_bg_finalize_u8(const float* raw, const float* cont, int H, int W, uint8_t* bg_always, uint8_t* mask_always,
                                  usc_stream_t stream) {
    if (!raw || !cont || !bg_always || !mask_always || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_bg_finalize_u8: bad argument");
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(bg_finalize_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, raw, cont, n, bg_always, mask_always);
    return check_launch("bg_finalize_kernel");
}
