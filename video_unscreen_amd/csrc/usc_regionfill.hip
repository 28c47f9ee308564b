// usc_regionfill.hip - background inpainting of the bg online mode (reference unscreen/utils/region_fill.py:7-63 regionfill,
// called per colour channel from tools/unscreen/bg.py:79): inside the mask solve the discrete Laplace equation
//     n(p) x(p) - sum_{q in N4(p), q in mask} x(q) = sum_{q in N4(p), q not in mask} I(q),     n(p) = number of in-image neighbours,
// i.e. membrane interpolation of the surrounding pixels.  The reference assembles the sparse matrix and calls SuperLU
// (scipy spsolve), seconds per 1080p channel; here the three channels are solved together by conjugate gradients in float64 on
// the pixel grid (the matrix is the 5-point stencil, never stored): three kernels per iteration, dot products reduced in a fixed
// order (per-block partial sums, re-summed by every consumer block) so a run is bit-reproducible.  The direct and the
// iterative solution agree to the tolerance (default: residual reduced by 1e-7, ~1e-3 grey levels), not bit for bit; after
// the reference's astype(uint8) truncation that is +-1 level on the few pixels whose value sits next to an integer.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

constexpr int kBlocks = 1024, kThreads = 256;
struct D3 { double v[3]; };

__device__ __forceinline__ D3 block_sum(D3 a, double* s_part) {         // fixed-order reduction of 3 doubles over the block
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int off = kWave / 2; off > 0; off >>= 1) a.v[c] += __shfl_down(a.v[c], off, kWave);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) { s_part[wave * 3] = a.v[0]; s_part[wave * 3 + 1] = a.v[1]; s_part[wave * 3 + 2] = a.v[2]; }
    __syncthreads();
    D3 t;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        t.v[c] = 0.;
        for (int w = 0; w < kThreads / kWave; w++) t.v[c] += s_part[w * 3 + c];
    }
    return t;
}

__device__ __forceinline__ D3 sum_partials(const double* __restrict__ part, double* s_part) {   // every block re-sums all partials
    D3 a = {{0., 0., 0.}};
    for (int b = threadIdx.x; b < kBlocks; b += kThreads) { a.v[0] += part[3 * b]; a.v[1] += part[3 * b + 1]; a.v[2] += part[3 * b + 2]; }
    return block_sum(a, s_part);
}

struct Nbr { int n; bool in[4]; uint32_t idx[4]; };
__device__ __forceinline__ Nbr neighbours(int y, int x, int H, int W) {
    Nbr nb;
    const int dy[4] = {-1, 1, 0, 0}, dx[4] = {0, 0, -1, 1};
    nb.n = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int yy = y + dy[k], xx = x + dx[k];
        nb.in[k] = yy >= 0 && yy < H && xx >= 0 && xx < W;
        nb.idx[k] = nb.in[k] ? (uint32_t)yy * (uint32_t)W + (uint32_t)xx : 0u;
        nb.n += nb.in[k];
    }
    return nb;
}

}  // namespace

// r = b - A x0 with x0 = I ; p = r ; partial r.r
__global__ __launch_bounds__(kThreads) void rf_init_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, int H, int W,
                                                           double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                           double* __restrict__ part_rr) {
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), xq = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, xq, H, W);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double xc = (double)img[3 * (size_t)i + c];
            double res = -(double)nb.n * xc;                 // b - A x0: every neighbour contributes + its value (known or current guess)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (nb.in[k]) res += (double)img[3 * (size_t)nb.idx[k] + c];
            x[3 * (size_t)i + c] = xc;
            r[3 * (size_t)i + c] = res;
            p[3 * (size_t)i + c] = res;
            acc.v[c] += res * res;
        }
    }
    const D3 t = block_sum(acc, s_part);
    if (threadIdx.x == 0) { part_rr[3 * blockIdx.x] = t.v[0]; part_rr[3 * blockIdx.x + 1] = t.v[1]; part_rr[3 * blockIdx.x + 2] = t.v[2]; }
}

// Ap = A p ; partial p.Ap
__global__ __launch_bounds__(kThreads) void rf_ap_kernel(const uint8_t* __restrict__ mask, int H, int W, const double* __restrict__ p,
                                                         double* __restrict__ Ap, double* __restrict__ part_pap) {
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), xq = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, xq, H, W);
        bool inm[4];
#pragma unroll
        for (int k = 0; k < 4; k++) inm[k] = nb.in[k] && mask[nb.idx[k]];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double pc = p[3 * (size_t)i + c];
            double v = (double)nb.n * pc;
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (inm[k]) v -= p[3 * (size_t)nb.idx[k] + c];
            Ap[3 * (size_t)i + c] = v;
            acc.v[c] += pc * v;
        }
    }
    const D3 t = block_sum(acc, s_part);
    if (threadIdx.x == 0) { part_pap[3 * blockIdx.x] = t.v[0]; part_pap[3 * blockIdx.x + 1] = t.v[1]; part_pap[3 * blockIdx.x + 2] = t.v[2]; }
}

// alpha = rr / pAp ; x += alpha p ; r -= alpha Ap ; partial r.r (new)
__global__ __launch_bounds__(kThreads) void rf_xr_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ p,
                                                         const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                                                         const double* __restrict__ part_rr_old, const double* __restrict__ part_pap,
                                                         double* __restrict__ part_rr_new) {
    __shared__ double s_part[12];
    const D3 rr = sum_partials(part_rr_old, s_part), pap = sum_partials(part_pap, s_part);
    double alpha[3];
#pragma unroll
    for (int c = 0; c < 3; c++) alpha[c] = pap.v[c] > 0. ? rr.v[c] / pap.v[c] : 0.;
    D3 acc = {{0., 0., 0.}};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t j = 3 * (size_t)i + c;
            x[j] += alpha[c] * p[j];
            const double rn = r[j] - alpha[c] * Ap[j];
            r[j] = rn;
            acc.v[c] += rn * rn;
        }
    }
    const D3 t = block_sum(acc, s_part);
    if (threadIdx.x == 0) { part_rr_new[3 * blockIdx.x] = t.v[0]; part_rr_new[3 * blockIdx.x + 1] = t.v[1]; part_rr_new[3 * blockIdx.x + 2] = t.v[2]; }
}

// beta = rr_new / rr_old ; p = r + beta p ; block 0 publishes rr_new (and rr0 on the first call) for the host's convergence test
__global__ __launch_bounds__(kThreads) void rf_p_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ r,
                                                        double* __restrict__ p, const double* __restrict__ part_rr_old,
                                                        const double* __restrict__ part_rr_new, double* __restrict__ scal, int first) {
    __shared__ double s_part[12];
    const D3 ro = sum_partials(part_rr_old, s_part), rn = sum_partials(part_rr_new, s_part);
    double beta[3];
#pragma unroll
    for (int c = 0; c < 3; c++) beta[c] = ro.v[c] > 0. ? rn.v[c] / ro.v[c] : 0.;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
#pragma unroll
        for (int c = 0; c < 3; c++) { scal[3 + c] = rn.v[c]; if (first) scal[c] = ro.v[c]; }
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t j = 3 * (size_t)i + c;
            p[j] = r[j] + beta[c] * p[j];
        }
    }
}

__global__ __launch_bounds__(kThreads) void rf_out_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, uint32_t n,
                                                          const double* __restrict__ x, uint8_t* __restrict__ out) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t j = 3 * (size_t)i + c;
            if (mask[i]) {
                const double v = x[j];
                out[j] = (uint8_t)(int)(v < 0. ? 0. : (v > 255. ? 255. : v));      // astype(np.uint8) truncation (bg.py:81)
            } else {
                out[j] = img[j];
            }
        }
    }
}

static size_t rf_align(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t usc_regionfill_ws_bytes(int H, int W) {
    return 256 + 3 * rf_align((size_t)kBlocks * 3 * sizeof(double)) + 4 * rf_align((size_t)H * W * 3 * sizeof(double));
}

extern "C" int usc_regionfill_u8(const uint8_t* img, const uint8_t* mask, int H, int W, uint8_t* out, double rtol, int max_iters,
                                 int* iters_out, void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!img || !mask || !out || !ws || H <= 0 || W <= 0 || !(rtol > 0.) || max_iters < 1) return fail(USC_E_ARG, "usc_regionfill_u8: bad argument");
    if (too_big(H, W) || (long long)H * W * 3 >= (1ll << 31)) return fail(USC_E_UNSUPPORTED, "usc_regionfill_u8: image too large");
    if (ws_bytes < usc_regionfill_ws_bytes(H, W)) return fail(USC_E_WORKSPACE, "usc_regionfill_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    uint8_t* base = reinterpret_cast<uint8_t*>(ws);
    double* scal = reinterpret_cast<double*>(base);                       // [0..2] = r0.r0, [3..5] = latest r.r
    const size_t psz = rf_align((size_t)kBlocks * 3 * sizeof(double)), vsz = rf_align((size_t)H * W * 3 * sizeof(double));
    double* part_a = reinterpret_cast<double*>(base + 256);
    double* part_b = reinterpret_cast<double*>(base + 256 + psz);
    double* part_pap = reinterpret_cast<double*>(base + 256 + 2 * psz);
    double* x = reinterpret_cast<double*>(base + 256 + 3 * psz);
    double* r = reinterpret_cast<double*>(base + 256 + 3 * psz + vsz);
    double* p = reinterpret_cast<double*>(base + 256 + 3 * psz + 2 * vsz);
    double* Ap = reinterpret_cast<double*>(base + 256 + 3 * psz + 3 * vsz);
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    const dim3 g(kBlocks), b(kThreads);
    hipLaunchKernelGGL(rf_init_kernel, g, b, 0, st, img, mask, H, W, x, r, p, part_a);
    double* rr_old = part_a;
    double* rr_new = part_b;
    int it = 0;
    const int check_every = 32;
    bool done = false;
    while (!done && it < max_iters) {
        const int stop = it + check_every < max_iters ? it + check_every : max_iters;
        for (; it < stop; it++) {
            hipLaunchKernelGGL(rf_ap_kernel, g, b, 0, st, mask, H, W, p, Ap, part_pap);
            hipLaunchKernelGGL(rf_xr_kernel, g, b, 0, st, mask, n, p, Ap, x, r, rr_old, part_pap, rr_new);
            hipLaunchKernelGGL(rf_p_kernel, g, b, 0, st, mask, n, r, p, rr_old, rr_new, scal, it == 0 ? 1 : 0);
            double* t = rr_old; rr_old = rr_new; rr_new = t;
        }
        double h[6];                                                       // the one host round trip per 32 iterations
        if (hipMemcpyAsync(h, scal, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
            return fail(USC_E_LAUNCH, "usc_regionfill_u8: reading the residual failed");
        done = true;
        for (int c = 0; c < 3; c++)
            if (h[c] > 0. && !(h[3 + c] <= rtol * rtol * h[c])) done = false;
    }
    if (iters_out) *iters_out = it;
    hipLaunchKernelGGL(rf_out_kernel, g, b, 0, st, img, mask, n, x, out);
    return check_launch("regionfill kernels");
}
