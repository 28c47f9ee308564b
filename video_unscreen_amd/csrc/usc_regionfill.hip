// usc_regionfill.hip - background inpainting of the bg online mode (reference unscreen/utils/region_fill.py:7-63 regionfill,
// called per colour channel from tools/unscreen/bg.py:79): inside the mask solve the discrete Laplace equation
//     n(p) x(p) - sum_{q in N4(p), q in mask} x(q) = sum_{q in N4(p), q not in mask} I(q),     n(p) = number of in-image neighbours,
// i.e. membrane interpolation of the surrounding pixels.  The reference assembles the sparse matrix and calls SuperLU
// (scipy spsolve), seconds per 1080p channel.  Here the three channels are solved together in float64 by conjugate
// gradients preconditioned with one geometric multigrid V-cycle per iteration - the matrix is the masked 5-point stencil and is
// never stored:
//   * levels: 2x2 cell aggregation (coarse cell in the mask iff any child is), piecewise-constant prolongation P, restriction
//     P^T, coarse operator = the same masked stencil re-discretised on the coarse grid (the Galerkin operator P^T A P of an
//     interior cell is twice that; with piecewise-constant P the unscaled stencil - an over-correction by 2 - is the better
//     preconditioner: measured 24 iterations against 52 at 1080p), weighted-Jacobi smoothing (2 pre, 2 post, omega 0.8) -
//     every piece symmetric, so the V-cycle is a valid CG preconditioner;
//   * ~25 iterations instead of the ~3000 plain CG needs on a 1080p silhouette (condition number ~ width^2);
//   * dot products are reduced in a fixed order (per-block partial sums, re-summed by every consumer block): bit-reproducible.
// The direct and the iterative solution agree to the tolerance (default: residual reduced by 1e-7, ~1e-3 grey levels), not bit
// for bit; after the reference's astype(uint8) truncation that is +-1 level on the few pixels whose value sits next to an integer.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>


#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

constexpr int kBlocks = 256, kThreads = 256;
constexpr int kMaxLevels = 12;
constexpr double kOmega = 0.8;
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

__device__ __forceinline__ void store_partial(double* part, const D3& t) {
    if (threadIdx.x == 0) { part[3 * blockIdx.x] = t.v[0]; part[3 * blockIdx.x + 1] = t.v[1]; part[3 * blockIdx.x + 2] = t.v[2]; }
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

// (A v)(i) / s for one masked pixel and channel block: n v_i - sum of in-mask neighbours.  All loads are issued
// unconditionally (out-of-image neighbours alias cell 0) and de-selected afterwards, so they do not wait for the mask bytes.
__device__ __forceinline__ void stencil3(const uint8_t* __restrict__ mask, const double* __restrict__ v, uint32_t i, const Nbr& nb, double* out3) {
    uint8_t mk[4];
    double nv[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        mk[k] = mask[nb.idx[k]];
#pragma unroll
        for (int c = 0; c < 3; c++) nv[k][c] = v[3 * (size_t)nb.idx[k] + c];
    }
#pragma unroll
    for (int c = 0; c < 3; c++) {
        double a = (double)nb.n * v[3 * (size_t)i + c];
#pragma unroll
        for (int k = 0; k < 4; k++) a -= (nb.in[k] && mk[k]) ? nv[k][c] : 0.;
        out3[c] = a;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------- CG (level 0)
// r = b - A x0 with x0 = I ; partial r.r
__global__ __launch_bounds__(kThreads) void rf_init_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, int H, int W,
                                                           double* __restrict__ x, double* __restrict__ r, double* __restrict__ part_rr) {
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
            acc.v[c] += res * res;
        }
    }
    store_partial(part_rr, block_sum(acc, s_part));
}

// partial a.b over the mask
__global__ __launch_bounds__(kThreads) void rf_dot_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ a,
                                                          const double* __restrict__ b, double* __restrict__ part, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) acc.v[c] += a[3 * (size_t)i + c] * b[3 * (size_t)i + c];
    }
    store_partial(part, block_sum(acc, s_part));
}

// p = z + beta p, beta = rz_new / rz_old (first: p = z)
__global__ __launch_bounds__(kThreads) void rf_p_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ z,
                                                        double* __restrict__ p, const double* __restrict__ part_rz_old,
                                                        const double* __restrict__ part_rz_new, int first, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    double beta[3] = {0., 0., 0.};
    if (!first) {
        const D3 ro = sum_partials(part_rz_old, s_part), rn = sum_partials(part_rz_new, s_part);
#pragma unroll
        for (int c = 0; c < 3; c++) beta[c] = ro.v[c] != 0. ? rn.v[c] / ro.v[c] : 0.;
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t j = 3 * (size_t)i + c;
            p[j] = first ? z[j] : z[j] + beta[c] * p[j];
        }
    }
}

// Ap = A p ; partial p.Ap
__global__ __launch_bounds__(kThreads) void rf_ap_kernel(const uint8_t* __restrict__ mask, int H, int W, const double* __restrict__ p,
                                                         double* __restrict__ Ap, double* __restrict__ part_pap, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), xq = (int)(i - (uint32_t)y * (uint32_t)W);
        double a[3];
        stencil3(mask, p, i, neighbours(y, xq, H, W), a);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            Ap[3 * (size_t)i + c] = a[c];
            acc.v[c] += p[3 * (size_t)i + c] * a[c];
        }
    }
    store_partial(part_pap, block_sum(acc, s_part));
}

// alpha = rz / pAp ; x += alpha p ; r -= alpha Ap ; partial r.r ; block 0 publishes r.r (and r0.r0) for the host
__global__ __launch_bounds__(kThreads) void rf_xr_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ p,
                                                         const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                                                         const double* __restrict__ part_rz, const double* __restrict__ part_pap,
                                                         double* __restrict__ part_rr, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    const D3 rz = sum_partials(part_rz, s_part), pap = sum_partials(part_pap, s_part);
    double alpha[3];
#pragma unroll
    for (int c = 0; c < 3; c++) alpha[c] = pap.v[c] > 0. ? rz.v[c] / pap.v[c] : 0.;
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
    store_partial(part_rr, block_sum(acc, s_part));
}

// Sums r.r and decides on the device.  slot 0 (after rf_init): records r0.r0, resets the control block (done only if there is
// nothing to solve).  slot 1 (after a block of `iters_add` iterations): records r.r, counts the iterations and sets done + converged
// once every channel's residual has dropped by rtol (rtol2 = rtol^2) - from then on every iteration kernel returns at its first
// instruction.  ctl = {done, iterations run, converged}.
__global__ __launch_bounds__(kThreads) void rf_publish_kernel(const double* __restrict__ part_rr, double* __restrict__ scal, int slot,
                                                             int* __restrict__ ctl, double rtol2, int iters_add) {
    __shared__ double s_part[12];
    if (slot == 1 && ctl[0]) return;
    const D3 t = sum_partials(part_rr, s_part);
    if (threadIdx.x == 0) {
        scal[3 * slot] = t.v[0]; scal[3 * slot + 1] = t.v[1]; scal[3 * slot + 2] = t.v[2];
        bool done = true;
        for (int c = 0; c < 3; c++) {
            const double r0 = slot == 0 ? t.v[c] : scal[c];
            if (r0 > 0. && !(t.v[c] <= rtol2 * r0)) done = false;
        }
        if (slot == 0) { ctl[0] = done ? 1 : 0; ctl[1] = 0; ctl[2] = done ? 1 : 0; }
        else { ctl[1] += iters_add; if (done) { ctl[0] = 1; ctl[2] = 1; } }
    }
}

__global__ __launch_bounds__(kThreads) void rf_out_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ mask, uint32_t n,
                                                          const double* __restrict__ x, uint8_t* __restrict__ out, const int* __restrict__ ctl,
                                                          int* __restrict__ status) {
    if (status && blockIdx.x == 0 && threadIdx.x == 0) { status[0] = ctl[1]; status[1] = ctl[2]; }
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

// ---------------------------------------------------------------------------------------------------------------- multigrid
namespace {
struct Level { int H, W; double s; uint8_t* mask; double *z, *zt, *r; };
struct LevelSet { int L, first_tail; Level lv[kMaxLevels]; };
constexpr int kTailThreads = 1024;
constexpr int kTailCells = 2500;           // levels with at most this many cells run inside the single-workgroup tail kernel (one CU: ~12 us per sweep of 8000 cells, measured)
}  // namespace

__global__ __launch_bounds__(kThreads) void mg_mask_down_kernel(const uint8_t* __restrict__ mf, int Hf, int Wf, uint8_t* __restrict__ mc,
                                                                int Hc, int Wc) {
    const uint32_t n = (uint32_t)Hc * (uint32_t)Wc;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int Y = (int)(i / (uint32_t)Wc), X = (int)(i - (uint32_t)Y * (uint32_t)Wc);
        int any = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * Y + (k >> 1), x = 2 * X + (k & 1);
            if (y < Hf && x < Wf) any |= mf[(size_t)y * Wf + x] != 0;
        }
        mc[i] = (uint8_t)any;
    }
}

// z_out = z_in + omega / (s n) (r - s A' z_in)   (z_in == nullptr: z_in = 0).  A' is the unscaled masked stencil.
__device__ __forceinline__ void mg_jacobi_cells(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ zin,
                                                const double* __restrict__ r, double* __restrict__ zout, uint32_t start, uint32_t stride) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = start; i < n; i += stride) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, x, H, W);
        const double w = kOmega / (s * (double)nb.n);
        if (zin) {
            double a[3];
            stencil3(mask, zin, i, nb, a);
#pragma unroll
            for (int c = 0; c < 3; c++) zout[3 * (size_t)i + c] = zin[3 * (size_t)i + c] + w * (r[3 * (size_t)i + c] - s * a[c]);
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) zout[3 * (size_t)i + c] = w * r[3 * (size_t)i + c];
        }
    }
}

__global__ __launch_bounds__(kThreads) void mg_jacobi_kernel(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ zin,
                                                             const double* __restrict__ r, double* __restrict__ zout, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    mg_jacobi_cells(mask, H, W, s, zin, r, zout, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The last sweep of a V-cycle on level 0 and the dot product r . z that follows it in the CG iteration, in one pass: launched with
// kBlocks workgroups, every workgroup sums the cells rf_dot_kernel would have given it, in the same order (same partial sums,
// bit for bit), one launch and one pass over r and z less per iteration.
__global__ __launch_bounds__(kThreads) void mg_jacobi_dot_kernel(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ zin,
                                                                 const double* __restrict__ r, double* __restrict__ zout, double* __restrict__ part,
                                                                 const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, x, H, W);
        const double w = kOmega / (s * (double)nb.n);
        double a[3];
        stencil3(mask, zin, i, nb, a);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double rc = r[3 * (size_t)i + c];
            const double z = zin[3 * (size_t)i + c] + w * (rc - s * a[c]);
            zout[3 * (size_t)i + c] = z;
            acc.v[c] += rc * z;
        }
    }
    store_partial(part, block_sum(acc, s_part));
}

// Two sweeps from zero in one pass: z1 = omega / (s n) r is pointwise, so the second sweep evaluates it on the fly for the
// cell and its neighbours - the same operations in the same order as two mg_jacobi_cells calls, one launch less per level.
__device__ __forceinline__ void mg_jacobi2_cells(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ r,
                                                 double* __restrict__ zout, uint32_t start, uint32_t stride) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = start; i < n; i += stride) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, x, H, W);
        const double w = kOmega / (s * (double)nb.n);
        uint8_t mk[4];
        double z1n[4][3];
        const int dy[4] = {-1, 1, 0, 0}, dx[4] = {0, 0, -1, 1};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mk[k] = mask[nb.idx[k]];
            // the neighbour's own in-image neighbour count
            const int yy = y + dy[k], xx = x + dx[k];
            const int nn = (yy > 0) + (yy < H - 1) + (xx > 0) + (xx < W - 1);
            const double wn = kOmega / (s * (double)nn);
#pragma unroll
            for (int c = 0; c < 3; c++) z1n[k][c] = wn * r[3 * (size_t)nb.idx[k] + c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double z1 = w * r[3 * (size_t)i + c];
            double a = (double)nb.n * z1;
#pragma unroll
            for (int k = 0; k < 4; k++) a -= (nb.in[k] && mk[k]) ? z1n[k][c] : 0.;
            zout[3 * (size_t)i + c] = z1 + w * (r[3 * (size_t)i + c] - s * a);
        }
    }
}

__global__ __launch_bounds__(kThreads) void mg_jacobi2_kernel(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ r,
                                                              double* __restrict__ zout, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    mg_jacobi2_cells(mask, H, W, s, r, zout, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// Coarse correction + one sweep in one pass: z' = z + z_c(parent) is pointwise, evaluated on the fly for the cell and its
// neighbours (same operations, same order as adding the correction in place and then sweeping).  Reads z and z_c, writes zout.
__device__ __forceinline__ void mg_prolong_jacobi_cells(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ z,
                                                        const double* __restrict__ zc, int Wc, const double* __restrict__ r,
                                                        double* __restrict__ zout, uint32_t start, uint32_t stride) {
    const uint32_t n = (uint32_t)H * (uint32_t)W;
    for (uint32_t i = start; i < n; i += stride) {
        if (!mask[i]) continue;
        const int y = (int)(i / (uint32_t)W), x = (int)(i - (uint32_t)y * (uint32_t)W);
        const Nbr nb = neighbours(y, x, H, W);
        const double w = kOmega / (s * (double)nb.n);
        const int dy[4] = {-1, 1, 0, 0}, dx[4] = {0, 0, -1, 1};
        uint8_t mk[4];
        double zn[4][3];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            mk[k] = mask[nb.idx[k]];
            const int yy = nb.in[k] ? y + dy[k] : 0, xx = nb.in[k] ? x + dx[k] : 0;
            const size_t I = (size_t)(yy >> 1) * Wc + (xx >> 1);
#pragma unroll
            for (int c = 0; c < 3; c++) zn[k][c] = z[3 * (size_t)nb.idx[k] + c] + zc[3 * I + c];
        }
        const size_t I0 = (size_t)(y >> 1) * Wc + (x >> 1);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const double zi = z[3 * (size_t)i + c] + zc[3 * I0 + c];
            double a = (double)nb.n * zi;
#pragma unroll
            for (int k = 0; k < 4; k++) a -= (nb.in[k] && mk[k]) ? zn[k][c] : 0.;
            zout[3 * (size_t)i + c] = zi + w * (r[3 * (size_t)i + c] - s * a);
        }
    }
}

__global__ __launch_bounds__(kThreads) void mg_prolong_jacobi_kernel(const uint8_t* __restrict__ mask, int H, int W, double s, const double* __restrict__ z,
                                                                     const double* __restrict__ zc, int Wc, const double* __restrict__ r,
                                                                     double* __restrict__ zout, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    mg_prolong_jacobi_cells(mask, H, W, s, z, zc, Wc, r, zout, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// r_c(I) = sum over the children i of I inside the fine mask of (r_f - s_f A' z_f)(i)
__device__ __forceinline__ void mg_restrict_cells(const uint8_t* __restrict__ mf, int Hf, int Wf, double sf, const double* __restrict__ zf,
                                                  const double* __restrict__ rf, const uint8_t* __restrict__ mc, int Hc, int Wc,
                                                  double* __restrict__ rc, uint32_t start, uint32_t stride) {
    const uint32_t n = (uint32_t)Hc * (uint32_t)Wc;
    for (uint32_t I = start; I < n; I += stride) {
        if (!mc[I]) continue;
        const int Y = (int)(I / (uint32_t)Wc), X = (int)(I - (uint32_t)Y * (uint32_t)Wc);
        double acc[3] = {0., 0., 0.};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * Y + (k >> 1), x = 2 * X + (k & 1);
            if (y >= Hf || x >= Wf) continue;
            const uint32_t i = (uint32_t)y * (uint32_t)Wf + (uint32_t)x;
            if (!mf[i]) continue;
            double a[3];
            stencil3(mf, zf, i, neighbours(y, x, Hf, Wf), a);
#pragma unroll
            for (int c = 0; c < 3; c++) acc[c] += rf[3 * (size_t)i + c] - sf * a[c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) rc[3 * (size_t)I + c] = acc[c];
    }
}

__global__ __launch_bounds__(kThreads) void mg_restrict_kernel(const uint8_t* __restrict__ mf, int Hf, int Wf, double sf, const double* __restrict__ zf,
                                                               const double* __restrict__ rf, const uint8_t* __restrict__ mc, int Hc, int Wc,
                                                               double* __restrict__ rc, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    mg_restrict_cells(mf, Hf, Wf, sf, zf, rf, mc, Hc, Wc, rc, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// The coarse end of the V-cycle (levels first_tail .. L-1, a few thousand cells at most) inside ONE workgroup: the same
// sweeps in the same order as the per-level kernels, separated by workgroup barriers instead of kernel boundaries
// (every cell's arithmetic is identical, so the result does not depend on which path ran).
// The vectors (z, zt, r) of the coarsest levels live in LDS for the duration of the kernel: the tail is a chain of ~33 tiny
// phases separated by barriers, and a phase on global memory costs a round trip to the L2 (~2.5 us) however few cells it has.
constexpr int kTailLdsDoubles = 6144;      // 48 KB: at 1080p the levels 17 x 30, 9 x 15 and 5 x 8 (25 of the 33 phases)
__global__ __launch_bounds__(kTailThreads) void mg_tail_kernel(LevelSet ls, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ double s_buf[kTailLdsDoubles];
    __shared__ int s_off[kMaxLevels];                      // first double of a level's three vectors in s_buf, -1: the level stays in global memory
    const uint32_t t = threadIdx.x, nt = blockDim.x;
    if (t == 0) {
        int used = 0;
        bool open = true;
        for (int l = kMaxLevels - 1; l >= 0; l--) {
            s_off[l] = -1;
            if (l >= ls.L || l <= ls.first_tail || !open) continue;      // (the first tail level receives its r from a kernel before this one)
            const int need = 3 * 3 * ls.lv[l].H * ls.lv[l].W;
            if (used + need > kTailLdsDoubles) { open = false; continue; }
            s_off[l] = used;
            used += need;
        }
    }
    __syncthreads();
    auto level = [&](int l) {
        Level v = ls.lv[l];
        const int o = s_off[l];
        if (o >= 0) {
            const int nv = 3 * v.H * v.W;
            v.z = s_buf + o; v.zt = s_buf + o + nv; v.r = s_buf + o + 2 * nv;
        }
        return v;
    };
    for (int l = ls.first_tail; l < ls.L; l++) {
        const Level v = level(l);
        if (l == ls.L - 1) {
            mg_jacobi_cells(v.mask, v.H, v.W, v.s, nullptr, v.r, v.z, t, nt);
            __syncthreads();
            for (int k = 0; k < 8; k++) {
                mg_jacobi_cells(v.mask, v.H, v.W, v.s, v.z, v.r, v.zt, t, nt);
                __syncthreads();
                mg_jacobi_cells(v.mask, v.H, v.W, v.s, v.zt, v.r, v.z, t, nt);
                __syncthreads();
            }
        } else {
            mg_jacobi2_cells(v.mask, v.H, v.W, v.s, v.r, v.z, t, nt);
            __syncthreads();
            const Level c = level(l + 1);
            mg_restrict_cells(v.mask, v.H, v.W, v.s, v.z, v.r, c.mask, c.H, c.W, c.r, t, nt);
            __syncthreads();
        }
    }
    for (int l = ls.L - 2; l >= ls.first_tail; l--) {
        const Level v = level(l), c = level(l + 1);
        mg_prolong_jacobi_cells(v.mask, v.H, v.W, v.s, v.z, c.z, c.W, v.r, v.zt, t, nt);
        __syncthreads();
        mg_jacobi_cells(v.mask, v.H, v.W, v.s, v.zt, v.r, v.z, t, nt);
        __syncthreads();
    }
}

namespace {

size_t rf_align(size_t v) { return (v + 255) & ~(size_t)255; }

int level_dims(int H, int W, int* hs, int* ws) {
    int L = 0;
    hs[0] = H; ws[0] = W;
    while (L + 1 < kMaxLevels && (hs[L] > 8 || ws[L] > 8)) {
        hs[L + 1] = (hs[L] + 1) / 2;
        ws[L + 1] = (ws[L] + 1) / 2;
        L++;
    }
    return L + 1;
}

int grid_of(size_t n) {                 // element-wise multigrid kernels: one cell per thread up to 8192 workgroups
    size_t b = (n + kThreads - 1) / kThreads;
    return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

// Plain stream launches (no graph, no cache: the entry point keeps no state between calls).
struct Launcher {
    hipStream_t st = nullptr;
    template <typename... P, typename... A>
    void launch(void (*k)(P...), dim3 grid, dim3 block, A... a) {
        static_assert(sizeof...(P) == sizeof...(A), "argument count");
        hipLaunchKernelGGL(k, grid, block, 0, st, static_cast<P>(a)...);
    }
};

struct Solver {
    int H, W, L, first_tail;
    uint32_t n;
    Level lv[kMaxLevels];
    double *scal, *part_rr, *part_pap, *part_rz[2], *x, *p, *Ap;
    int* ctl;                                       // {done, iterations run, converged}: the device-side convergence state
    uint8_t* mask0;

    void carve(void* ws, int H_, int W_) {
        H = H_; W = W_; n = (uint32_t)H * (uint32_t)W;
        uint8_t* cur = reinterpret_cast<uint8_t*>(ws);
        auto take = [&](size_t bytes) { uint8_t* q = cur; cur += rf_align(bytes); return q; };
        scal = reinterpret_cast<double*>(take(256));                   // [0..2] = r0.r0, [3..5] = latest r.r, bytes 128.. = ctl
        ctl = reinterpret_cast<int*>(reinterpret_cast<uint8_t*>(scal) + 128);
        const size_t psz = (size_t)kBlocks * 3 * sizeof(double);
        part_rr = reinterpret_cast<double*>(take(psz));
        part_pap = reinterpret_cast<double*>(take(psz));
        part_rz[0] = reinterpret_cast<double*>(take(psz));
        part_rz[1] = reinterpret_cast<double*>(take(psz));
        const size_t v0 = (size_t)H * W * 3 * sizeof(double);
        x = reinterpret_cast<double*>(take(v0));
        p = reinterpret_cast<double*>(take(v0));
        Ap = reinterpret_cast<double*>(take(v0));
        int hs[kMaxLevels], wsz[kMaxLevels];
        L = level_dims(H, W, hs, wsz);
        const double lvl_scale = 1.;                                   // coarse operator = the unscaled re-discretised stencil (see the header)
        for (int l = 0; l < L; l++) {
            const size_t vb = (size_t)hs[l] * wsz[l] * 3 * sizeof(double);
            lv[l].H = hs[l]; lv[l].W = wsz[l];
            lv[l].s = l == 0 ? 1. : lvl_scale * lv[l - 1].s;
            lv[l].z = reinterpret_cast<double*>(take(vb));
            lv[l].zt = reinterpret_cast<double*>(take(vb));
            lv[l].r = reinterpret_cast<double*>(take(vb));
            lv[l].mask = take((size_t)hs[l] * wsz[l]);                 // level 0: a private copy of the caller's mask
        }
        mask0 = lv[0].mask;
        first_tail = L - 1;
        while (first_tail > 1 && (size_t)lv[first_tail - 1].H * lv[first_tail - 1].W <= (size_t)kTailCells) first_tail--;
    }

    // z_l = V-cycle(r_l): 2 pre-smoothing sweeps from zero, coarse correction, 2 post-smoothing sweeps; 17 sweeps on the last level
    void vcycle(Launcher& q, double* part_rz_new) const {
        const dim3 b(kThreads);
        const int last = first_tail < L ? first_tail : L;          // levels [0, last) run as per-level kernels
        for (int l = 0; l < last; l++) {
            const Level& v = lv[l];
            const dim3 g(grid_of((size_t)v.H * v.W));
            if (l == L - 1) {
                q.launch(mg_jacobi_kernel, g, b, v.mask, v.H, v.W, v.s, (const double*)nullptr, v.r, v.z, ctl);
                for (int k = 0; k < 8; k++) {
                    q.launch(mg_jacobi_kernel, g, b, v.mask, v.H, v.W, v.s, v.z, v.r, v.zt, ctl);
                    q.launch(mg_jacobi_kernel, g, b, v.mask, v.H, v.W, v.s, v.zt, v.r, v.z, ctl);
                }
            } else {
                q.launch(mg_jacobi2_kernel, g, b, v.mask, v.H, v.W, v.s, v.r, v.z, ctl);             // two sweeps from zero
                const Level& c = lv[l + 1];
                q.launch(mg_restrict_kernel, dim3(grid_of((size_t)c.H * c.W)), b, v.mask, v.H, v.W, v.s, v.z, v.r, c.mask, c.H, c.W, c.r, ctl);
            }
        }
        if (first_tail < L) {
            LevelSet ls;
            ls.L = L; ls.first_tail = first_tail;
            for (int l = 0; l < kMaxLevels; l++) ls.lv[l] = lv[l < L ? l : L - 1];
            q.launch(mg_tail_kernel, dim3(1), dim3(kTailThreads), ls, ctl);
        }
        for (int l = last - 1; l >= 0; l--) {
            if (l == L - 1) continue;
            const Level& v = lv[l];
            const dim3 g(grid_of((size_t)v.H * v.W));
            q.launch(mg_prolong_jacobi_kernel, g, b, v.mask, v.H, v.W, v.s, v.z, lv[l + 1].z, lv[l + 1].W, v.r, v.zt, ctl);   // correction + sweep
            if (l == 0 && part_rz_new) q.launch(mg_jacobi_dot_kernel, dim3(kBlocks), b, v.mask, v.H, v.W, v.s, v.zt, v.r, v.z, part_rz_new, ctl);   // + r.z
            else q.launch(mg_jacobi_kernel, g, b, v.mask, v.H, v.W, v.s, v.zt, v.r, v.z, ctl);
        }
    }

    // one preconditioned CG iteration; cur selects which of the two r.z partial arrays holds the previous value
    void iteration(Launcher& q, int first, int& cur) const {
        const dim3 b(kThreads), g(kBlocks);           // reductions always use kBlocks blocks: the partial arrays have that many rows
        double* r = lv[0].r;
        const bool fused = first_tail >= 1 && L >= 2;                                                // level 0 ends with a per-level sweep
        vcycle(q, fused ? part_rz[cur ^ 1] : nullptr);                                               // z = M^-1 r (+ the partial sums of r.z)
        if (!fused) q.launch(rf_dot_kernel, g, b, mask0, n, r, lv[0].z, part_rz[cur ^ 1], ctl);           // r.z (new)
        q.launch(rf_p_kernel, g, b, mask0, n, lv[0].z, p, part_rz[cur], part_rz[cur ^ 1], first, ctl);
        cur ^= 1;
        q.launch(rf_ap_kernel, g, b, mask0, H, W, p, Ap, part_pap, ctl);
        q.launch(rf_xr_kernel, g, b, mask0, n, p, Ap, x, r, part_rz[cur], part_pap, part_rr, ctl);
    }
};

constexpr int kItersPerCheck = 2;         // iterations between two device-side convergence tests

}  // namespace

extern "C" size_t usc_regionfill_ws_bytes(int H, int W) {
    int hs[kMaxLevels], ws[kMaxLevels];
    const int L = level_dims(H, W, hs, ws);
    size_t total = 256 + 4 * rf_align((size_t)kBlocks * 3 * sizeof(double));
    total += 3 * rf_align((size_t)H * W * 3 * sizeof(double));                  // x, p, Ap
    for (int l = 0; l < L; l++)
        total += 3 * rf_align((size_t)hs[l] * ws[l] * 3 * sizeof(double)) + rf_align((size_t)hs[l] * ws[l]);   // z, zt, r, mask per level
    return total;
}

extern "C" int usc_regionfill_u8(const uint8_t* img, const uint8_t* mask, int H, int W, uint8_t* out, double rtol, int first_iter,
                                 int n_iters, int* status, void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!img || !mask || !out || !ws || H <= 0 || W <= 0 || !(rtol > 0.) || first_iter < 0 || n_iters < 0 || (first_iter == 0 && n_iters < 1))
        return fail(USC_E_ARG, "usc_regionfill_u8: bad argument");
    if (too_big(H, W) || (long long)H * W * 3 >= (1ll << 31)) return fail(USC_E_UNSUPPORTED, "usc_regionfill_u8: image too large");
    if (ws_bytes < usc_regionfill_ws_bytes(H, W)) return fail(USC_E_WORKSPACE, "usc_regionfill_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    Solver s;
    s.carve(ws, H, W);
    Launcher q;
    q.st = st;
    const dim3 b(kThreads);
    const double rtol2 = rtol * rtol;
    if (first_iter == 0) {                                      // a new solve: masks of every level, x0, r0, control block
        if (hipMemcpyAsync(s.mask0, mask, (size_t)H * W, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(USC_E_LAUNCH, "usc_regionfill_u8: mask copy failed");
        for (int l = 1; l < s.L; l++)
            q.launch(mg_mask_down_kernel, dim3(grid_of((size_t)s.lv[l].H * s.lv[l].W)), b, s.lv[l - 1].mask, s.lv[l - 1].H, s.lv[l - 1].W,
                     s.lv[l].mask, s.lv[l].H, s.lv[l].W);
        q.launch(rf_init_kernel, dim3(kBlocks), b, img, s.mask0, H, W, s.x, s.lv[0].r, s.part_rr);
        q.launch(rf_publish_kernel, dim3(1), b, s.part_rr, s.scal, 0, s.ctl, rtol2, 0);
    }
    // Iterations [first_iter, first_iter + n_iters), a convergence test on the device after every kItersPerCheck of them; once it has
    // passed, the kernels still in the queue return at their first instruction.  No host synchronisation.
    int cur = first_iter & 1;                                   // which r.z partial array holds the previous value
    int since = 0;
    for (int it = first_iter; it < first_iter + n_iters; it++) {
        s.iteration(q, it == 0 ? 1 : 0, cur);
        if (++since == kItersPerCheck || it + 1 == first_iter + n_iters) {
            q.launch(rf_publish_kernel, dim3(1), b, s.part_rr, s.scal, 1, s.ctl, rtol2, since);
            since = 0;
        }
    }
    q.launch(rf_out_kernel, dim3(kBlocks), b, img, s.mask0, s.n, s.x, out, (const int*)s.ctl, status);
    return check_launch("regionfill kernels");
}
