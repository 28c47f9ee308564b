// usc_regionfill.hip - background inpainting of the bg online mode (reference unscreen/utils/region_fill.py:7-63 regionfill,
// called per colour channel from tools/unscreen/bg.py:79): inside the mask solve the discrete Laplace equation
//     n(p) x(p) - sum_{q in N4(p), q in mask} x(q) = sum_{q in N4(p), q not in mask} I(q),     n(p) = number of in-image neighbours,
// i.e. membrane interpolation of the surrounding pixels.  The reference assembles the sparse matrix and calls SuperLU
// (scipy spsolve), seconds per 1080p channel.  Here the three channels are solved together by conjugate gradients in float64,
// preconditioned with one geometric multigrid V-cycle per iteration - the matrix is the masked 5-point stencil and is never stored:
//   * levels: 2x2 cell aggregation (coarse cell in the mask iff any child is), piecewise-constant prolongation P, restriction
//     P^T, coarse operator = HALF the Galerkin operator P^T A P of the level above (mg_galerkin_kernel: per cell a diagonal and four
//     edge weights summed from the children's; an interior cell gets the 5-point stencil 4 / 1 1 1 1 again - the over-correction by 2
//     that piecewise-constant P needs), weighted-Jacobi smoothing (2 pre, 2 post; 5 sweeps on the coarsest level) - every piece
//     symmetric, so the V-cycle is a valid CG preconditioner.  Rounds 1 - 4 re-discretised the masked stencil on the coarse mask
//     instead, which treats a coarse cell that is partly outside the mask as if it were inside: on a real silhouette's ragged
//     Dirichlet boundary that coarse correction is wrong where it matters, and the iteration count doubled (52 against 24 on the
//     bench's hole; tools/dev/rf_model.py, the NumPy model of this solver that priced the change: 44 -> 24.  The same model had said
//     before that cell-centred bilinear P, red-black Gauss-Seidel or Chebyshev weights do NOT pay: 28 -> 24 / 18 / 25; and a second
//     pair of sweeps each way from level 2 on - built as tiled kernels with a 4-cell ring, 24 -> 20 iterations - costs per cycle
//     what it saves: 5.30 against 5.43 ms per solve, not kept).
//   * 18 - 24 iterations instead of the ~3000 plain CG needs on a 1080p silhouette (condition number ~ width^2);
//   * dot products are reduced in a fixed order (per-block partial sums, re-summed by every consumer block): bit-reproducible.
// Round 4: the V-cycle runs in FLOAT32 (a preconditioner only has to be a fixed symmetric linear operator; the CG vectors x, r, p, A p
// and every dot product stay float64, so the iteration still converges to the same tolerance), and the fine levels run as TILED
// kernels: a workgroup owns 32 x 32 cells, stages the 36 x 36 window it needs in LDS and does a whole half of the V-cycle for its
// tile in one pass - "down": two sweeps from zero + residual + restriction, "up": coarse correction + two sweeps (+ the r.z partial
// sums on level 0) - and the CG's direction update p = z + beta p and A p are one tiled pass too.  Four passes over level 0 per
// iteration instead of seven, two launches per level instead of four, tiles without a masked cell are skipped by a flag.
// The direct and the iterative solution agree to the tolerance (default: residual reduced by 1e-7, ~1e-3 grey levels), not bit
// for bit; after the reference's astype(uint8) truncation that is +-1 level on the few pixels whose value sits next to an integer.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <math.h>
#include <stdint.h>
#include <string.h>


#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

#ifndef USC_RF_VT
#define USC_RF_VT float
#endif
typedef USC_RF_VT VT;          // the V-cycle's scalar (float32; -DUSC_RF_VT=double for A/B runs of the iteration count)

namespace {

#ifndef USC_RF_BLOCKS
#define USC_RF_BLOCKS 1024
#endif
constexpr int kBlocks = USC_RF_BLOCKS, kThreads = 256;    // (kBlocks: rows of every partial-sum array = workgroups of every reducing kernel: four per CU)
constexpr int kMaxLevels = 12;
#ifndef USC_RF_COARSE_PAIRS
#define USC_RF_COARSE_PAIRS 2          // sweeps on the coarsest level (<= 8 x 8 cells): 1 + 2 x this many.  Round 4: 17 -> 5 sweeps, the same
#endif                                 // 52 - 54 iterations on the bench's hole, twelve barrier phases fewer in the tail kernel per iteration
constexpr VT kOmega = (VT)0.8;             // the coarsest level's 17 sweeps
// The two smoothing sweeps of a level use the weights of the degree-2 Chebyshev polynomial on [1/2, 2] (the high-frequency range of
// D^-1 A for the 5-point stencil under 2 x 2 coarsening): damping 0.22 per pair instead of 0.36 for two sweeps with 0.8.  Pre-smoothing
// applies (A, B), post-smoothing (B, A): the V-cycle stays symmetric.  NumPy model of this solver on the bench's hole: 28 -> 25 iterations.
constexpr VT kOmegaA = (VT)1.3895, kOmegaB = (VT)0.5617;
constexpr int TS = 32, WN = TS + 4, W1 = TS + 2;       // tile side; window with a 2-cell / 1-cell ring
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

__device__ __forceinline__ int ncount(int y, int x, int H, int W) { return (y > 0) + (y < H - 1) + (x > 0) + (x < W - 1); }

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

// One tiled pass: p_new = z + beta p_old (beta = rz_new / rz_old; first: p_new = z) on the tile's 34 x 34 window in LDS, then
// A p_new for the tile's own cells and the partial sums of p . A p.  p is double-buffered (a tile reads the ring of its neighbours'
// old values).  kBlocks workgroups walk the tiles in a fixed order: the partial sums are the same on every run.
__global__ __launch_bounds__(kThreads) void rf_pap_kernel(const uint8_t* __restrict__ mask, int H, int W, const VT* __restrict__ z,
                                                          const double* __restrict__ p_old, double* __restrict__ p_new, double* __restrict__ Ap,
                                                          const double* __restrict__ part_rz_old, const double* __restrict__ part_rz_new,
                                                          const double* __restrict__ part_cross, int first, double* __restrict__ part_pap, const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                                          const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    __shared__ double P[W1 * W1 * 3];
    double beta[3] = {0., 0., 0.};
    if (!first) {
        // beta = (r_k . z_k - r_k . z_{k-1}) / (r_{k-1} . z_{k-1}): the Polak-Ribiere form.  With an exactly symmetric linear preconditioner
        // the cross term is zero; the float32 V-cycle is symmetric and linear only up to its rounding, and without the term the bench's
        // ragged hole needed 56 iterations instead of the 46 of a float64 V-cycle.
        const D3 ro = sum_partials(part_rz_old, s_part), rn = sum_partials(part_rz_new, s_part), cr = sum_partials(part_cross, s_part);
#pragma unroll
        for (int c = 0; c < 3; c++) beta[c] = ro.v[c] != 0. ? (rn.v[c] - cr.v[c]) / ro.v[c] : 0.;
    }
    D3 acc = {{0., 0., 0.}};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        if (!active[tile]) continue;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * TS - 1, x0 = tx * TS - 1;
        __syncthreads();                                     // (the previous tile's reads of P are done)
        {   // every load of the window is issued before the first is used (cell 0 stands in for cells outside the image)
            constexpr int NW = (W1 * W1 + kThreads - 1) / kThreads;
            uint32_t g[NW];
            uint8_t mk[NW];
            VT zv[NW][3];
            double pv[NW][3];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads, wy = i / W1, wx = i - wy * W1, y = y0 + wy, x = x0 + wx;
                const bool in = i < W1 * W1 && y >= 0 && y < H && x >= 0 && x < W;
                g[k] = in ? (uint32_t)y * (uint32_t)W + (uint32_t)x : 0u;
                mk[k] = in ? mask[g[k]] : (uint8_t)0;
            }
#pragma unroll
            for (int k = 0; k < NW; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) { zv[k][c] = z[3 * (size_t)g[k] + c]; pv[k][c] = first ? 0. : p_old[3 * (size_t)g[k] + c]; }
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads;
                if (i < W1 * W1) {
#pragma unroll
                    for (int c = 0; c < 3; c++) P[3 * i + c] = mk[k] ? (first ? (double)zv[k][c] : (double)zv[k][c] + beta[c] * pv[k][c]) : 0.;
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS * TS; i += kThreads) {
            const int cy = i / TS, cx = i - cy * TS, y = ty * TS + cy, x = tx * TS + cx;
            if (y >= H || x >= W) continue;
            const uint32_t g = (uint32_t)y * (uint32_t)W + (uint32_t)x;
            if (!mask[g]) continue;
            const int j = (cy + 1) * W1 + cx + 1;
            const double nn = (double)ncount(y, x, H, W);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const double pc = P[3 * j + c];
                const double a = nn * pc - (P[3 * (j - W1) + c] + P[3 * (j + W1) + c] + P[3 * (j - 1) + c] + P[3 * (j + 1) + c]);
                p_new[3 * (size_t)g + c] = pc;
                Ap[3 * (size_t)g + c] = a;
                acc.v[c] += pc * a;
            }
        }
    }
    store_partial(part_pap, block_sum(acc, s_part));
}

// alpha = rz / pAp ; x += alpha p ; r -= alpha Ap ; partial r.r and r_new . z (z = the preconditioned residual of this iteration)
__global__ __launch_bounds__(kThreads) void rf_xr_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ p,
                                                         const double* __restrict__ Ap, double* __restrict__ x, double* __restrict__ r,
                                                         const VT* __restrict__ z, const double* __restrict__ part_rz, const double* __restrict__ part_pap,
                                                         double* __restrict__ part_rr, double* __restrict__ part_cross, const int* __restrict__ ctl) {
    if (ctl[0]) return;                                      // converged: the rest of the enqueued iterations are no-ops
    __shared__ double s_part[12];
    const D3 rz = sum_partials(part_rz, s_part), pap = sum_partials(part_pap, s_part);
    double alpha[3];
#pragma unroll
    for (int c = 0; c < 3; c++) alpha[c] = pap.v[c] > 0. ? rz.v[c] / pap.v[c] : 0.;
    D3 acc = {{0., 0., 0.}}, acz = {{0., 0., 0.}};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const size_t j = 3 * (size_t)i + c;
            x[j] += alpha[c] * p[j];
            const double rn = r[j] - alpha[c] * Ap[j];
            r[j] = rn;
            acc.v[c] += rn * rn;
            acz.v[c] += rn * (double)z[j];
        }
    }
    store_partial(part_rr, block_sum(acc, s_part));
    store_partial(part_cross, block_sum(acz, s_part));
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
struct Level { int H, W, tiles_x, ntiles; VT s; uint8_t *mask, *active; VT *z, *zt, *r, *d; float4* w; int off, moff, cp; };   // d / w: the level's operator (levels >= 1, mg_galerkin_kernel); off / moff / cp: the tail's LDS layout (mg_tail_levels)
struct LevelSet { int L, first_tail, tail_floats, tail_cells; Level lv[kMaxLevels]; };
constexpr int kTailThreads = 1024;
constexpr int kTailLdsBytes = 156 * 1024;   // the tail levels in LDS: 48 bytes per cell of the ringed layout (1080p: 34 x 60, 17 x 30, 9 x 15, 5 x 8 = 3097 ringed cells, 146 KB)
constexpr int kTailDwBytes = 4 * 1024 * 16;   // upper bound of the packed d / w image (12 bytes per ringed tail cell) the tail's start-up copy is written for
constexpr int kTailCells = 2500;           // levels with at most this many cells run inside the single-workgroup tail kernel (one CU: ~12 us per sweep of 8000 cells, measured)
}  // namespace

// active[tile] = 1 iff the tile's own 32 x 32 cells hold a masked cell (one workgroup per tile)
__global__ __launch_bounds__(kThreads) void mg_active_kernel(const uint8_t* __restrict__ mask, int H, int W, uint8_t* __restrict__ active, int tiles_x) {
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    int any = 0;
    for (int i = threadIdx.x; i < TS * TS; i += kThreads) {
        const int y = ty * TS + i / TS, x = tx * TS + (i & (TS - 1));
        if (y < H && x < W) any |= mask[(size_t)y * W + x] != 0;
    }
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) active[blockIdx.x] = (uint8_t)(any != 0);
}

// "Down" half of the V-cycle for one level, tiled: z = two weighted-Jacobi sweeps from zero on r, then the restricted residual
// r_c(I) = sum over the masked children i of I of (r - s A z)(i).  The window of r (36 x 36, zero outside the mask) is staged once;
// z1 = w r is pointwise, z2 needs z1 on a 1-cell ring, the residual z2 on a 1-cell ring: all from LDS.  RT = double on level 0 (the
// CG residual), VT below.  GAL = false (level 0): A is the masked 5-point stencil, n z - the in-mask neighbours.  GAL = true (the coarse
// levels): A is the level's Galerkin operator - per cell a diagonal dg and four edge weights wg (up, down, left, right; zero towards
// cells outside the mask, dg = 0 outside the mask), staged beside the window.
// NT threads per workgroup; workgroup `bid` of `nbid`.
struct TileLds { VT a[WN * WN * 3], b[WN * WN * 3], c[W1 * W1 * 3]; uint8_t m[WN * WN]; };
struct GalLds { VT d[WN * WN]; float4 w[W1 * W1]; };
template <int NT, typename RT, bool GAL>
__device__ __forceinline__ void mg_down_tiles(const uint8_t* __restrict__ mask, int H, int W, VT s, const RT* __restrict__ r,
                                              VT* __restrict__ z, const uint8_t* __restrict__ mc, int Hc, int Wc,
                                              VT* __restrict__ rc, const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                              int bid, int nbid, TileLds& lds, const VT* __restrict__ dg, const float4* __restrict__ wg, GalLds* gal) {
    VT* const R = lds.a; VT* const Z1 = lds.b; VT* const Z2 = lds.c;
    uint8_t* const M = lds.m;
    constexpr int kThreads = NT;
    for (int tile = bid; tile < ntiles; tile += nbid) {
        if (!active[tile]) continue;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * TS - 2, x0 = tx * TS - 2;
        __syncthreads();
        {   // every load of the window is issued before the first is used (cell 0 stands in for cells outside the image)
            constexpr int NW = (WN * WN + kThreads - 1) / kThreads;
            uint32_t g[NW];
            uint8_t mk[NW];
            int nc[NW];
            RT rv[NW][3];
            VT dv[NW];
            float4 wv[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads, wy = i / WN, wx = i - wy * WN, y = y0 + wy, x = x0 + wx;
                const bool in = i < WN * WN && y >= 0 && y < H && x >= 0 && x < W;
                g[k] = in ? (uint32_t)y * (uint32_t)W + (uint32_t)x : 0u;
                if (GAL) {
                    dv[k] = dg[g[k]];
                    wv[k] = wg[g[k]];
                    if (!in) dv[k] = (VT)0;
                } else {
                    mk[k] = in ? mask[g[k]] : (uint8_t)0;
                    nc[k] = in ? ncount(y, x, H, W) : 4;
                }
            }
#pragma unroll
            for (int k = 0; k < NW; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) rv[k][c] = r[3 * (size_t)g[k] + c];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads;
                if (i < WN * WN) {
                    const bool in_mask = GAL ? dv[k] > (VT)0 : mk[k] != 0;
                    M[i] = (uint8_t)in_mask;
                    const VT w = kOmegaA / (s * (GAL ? (in_mask ? dv[k] : (VT)1) : (VT)nc[k]));
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const VT v = in_mask ? (VT)rv[k][c] : (VT)0;
                        R[3 * i + c] = v;
                        Z1[3 * i + c] = w * v;
                    }
                    if (GAL) {
                        gal->d[i] = dv[k];
                        const int wy = i / WN, wx = i - wy * WN;
                        if (wy >= 1 && wy <= W1 && wx >= 1 && wx <= W1) gal->w[(wy - 1) * W1 + wx - 1] = wv[k];
                    }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < W1 * W1; i += kThreads) {
            const int wy = i / W1 + 1, wx = i % W1 + 1, j = wy * WN + wx, y = y0 + wy, x = x0 + wx;
            VT o[3] = {(VT)0, (VT)0, (VT)0};
            if (M[j]) {
                const VT nn = GAL ? gal->d[j] : (VT)ncount(y, x, H, W), w = kOmegaB / (s * nn);
                float4 e = make_float4(1.f, 1.f, 1.f, 1.f);
                if (GAL) e = gal->w[i];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const VT z1 = Z1[3 * j + c];
                    VT a;
                    if (GAL) {
                        a = nn * z1;
                        a -= e.x * Z1[3 * (j - WN) + c]; a -= e.y * Z1[3 * (j + WN) + c]; a -= e.z * Z1[3 * (j - 1) + c]; a -= e.w * Z1[3 * (j + 1) + c];
                    } else {
                        a = nn * z1 - (Z1[3 * (j - WN) + c] + Z1[3 * (j + WN) + c] + Z1[3 * (j - 1) + c] + Z1[3 * (j + 1) + c]);
                    }
                    o[c] = z1 + w * (R[3 * j + c] - s * a);
                }
            }
            Z2[3 * i] = o[0]; Z2[3 * i + 1] = o[1]; Z2[3 * i + 2] = o[2];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS * TS; i += kThreads) {
            const int cy = i / TS, cx = i & (TS - 1), y = ty * TS + cy, x = tx * TS + cx;
            if (y >= H || x >= W || !M[(cy + 2) * WN + cx + 2]) continue;
            const size_t g = (size_t)y * W + x;
            const int k = (cy + 1) * W1 + cx + 1;
            z[3 * g] = Z2[3 * k]; z[3 * g + 1] = Z2[3 * k + 1]; z[3 * g + 2] = Z2[3 * k + 2];
        }
        if (NT == 256 || threadIdx.x < 256) {   // one coarse cell per thread: the tile's 16 x 16 parents
            const int cy = threadIdx.x >> 4, cx = threadIdx.x & 15, Y = ty * (TS / 2) + cy, X = tx * (TS / 2) + cx;
            if (Y < Hc && X < Wc && mc[(size_t)Y * Wc + X]) {
                VT acc[3] = {(VT)0, (VT)0, (VT)0};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int fy = 2 * cy + (q >> 1), fx = 2 * cx + (q & 1);          // child, tile coordinates
                    const int j = (fy + 2) * WN + fx + 2, k = (fy + 1) * W1 + fx + 1;
                    if (!M[j]) continue;                                             // (outside the image or the mask)
                    const VT nn = GAL ? gal->d[j] : (VT)ncount(ty * TS + fy, tx * TS + fx, H, W);
                    float4 e = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (GAL) e = gal->w[k];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        VT a;
                        if (GAL) {
                            a = nn * Z2[3 * k + c];
                            a -= e.x * Z2[3 * (k - W1) + c]; a -= e.y * Z2[3 * (k + W1) + c]; a -= e.z * Z2[3 * (k - 1) + c]; a -= e.w * Z2[3 * (k + 1) + c];
                        } else {
                            a = nn * Z2[3 * k + c] - (Z2[3 * (k - W1) + c] + Z2[3 * (k + W1) + c] + Z2[3 * (k - 1) + c] + Z2[3 * (k + 1) + c]);
                        }
                        acc[c] += R[3 * j + c] - s * a;
                    }
                }
                const size_t G = (size_t)Y * Wc + X;
                rc[3 * G] = acc[0]; rc[3 * G + 1] = acc[1]; rc[3 * G + 2] = acc[2];
            }
        }
    }
}

// level 0
__global__ __launch_bounds__(kThreads) void mg_down0_kernel(const uint8_t* __restrict__ mask, int H, int W, VT s, const double* __restrict__ r,
                                                            VT* __restrict__ z, const uint8_t* __restrict__ mc, int Hc, int Wc,
                                                            VT* __restrict__ rc, const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                                            const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ TileLds lds;
    mg_down_tiles<kThreads, double, false>(mask, H, W, s, r, z, mc, Hc, Wc, rc, active, tiles_x, ntiles, (int)blockIdx.x, (int)gridDim.x, lds, nullptr, nullptr, nullptr);
}
// levels >= 1
__global__ __launch_bounds__(kThreads) void mg_down_kernel(const uint8_t* __restrict__ mask, int H, int W, VT s, const VT* __restrict__ r,
                                                           VT* __restrict__ z, const uint8_t* __restrict__ mc, int Hc, int Wc,
                                                           VT* __restrict__ rc, const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                                           const VT* __restrict__ dg, const float4* __restrict__ wg, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ TileLds lds;
    __shared__ GalLds gal;
    mg_down_tiles<kThreads, VT, true>(mask, H, W, s, r, z, mc, Hc, Wc, rc, active, tiles_x, ntiles, (int)blockIdx.x, (int)gridDim.x, lds, dg, wg, &gal);
}

// "Up" half for one level, tiled: z' = z + z_c(parent), then two weighted-Jacobi sweeps; writes zout (another array: a tile reads
// the ring of its neighbours' z).  DOT (level 0): also the partial sums of r . zout in float64 - kBlocks workgroups walk the tiles in
// a fixed order.  GAL as in mg_down_tiles.
template <int NT, typename RT, bool DOT, bool GAL>
__device__ __forceinline__ D3 mg_up_tiles(const uint8_t* __restrict__ mask, int H, int W, VT s, const VT* __restrict__ z,
                                          const VT* __restrict__ zc, int Wc, const RT* __restrict__ r, VT* __restrict__ zout,
                                          const uint8_t* __restrict__ active, int tiles_x, int ntiles, int bid, int nbid, TileLds& lds,
                                          const VT* __restrict__ dg, const float4* __restrict__ wg, GalLds* gal) {
    VT* const Zp = lds.a; VT* const R = lds.b; VT* const Zt = lds.c;
    uint8_t* const M = lds.m;
    constexpr int kThreads = NT;
    D3 acc = {{0., 0., 0.}};
    for (int tile = bid; tile < ntiles; tile += nbid) {
        if (!active[tile]) continue;
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * TS - 2, x0 = tx * TS - 2;
        __syncthreads();
        {   // every load of the window is issued before the first is used (cell 0 stands in for cells outside the image)
            constexpr int NW = (WN * WN + kThreads - 1) / kThreads;
            uint32_t g[NW], G[NW];
            uint8_t mk[NW];
            VT zv[NW][3], cv[NW][3];
            RT rv[NW][3];
            VT dv[NW];
            float4 wv[NW];
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads, wy = i / WN, wx = i - wy * WN, y = y0 + wy, x = x0 + wx;
                const bool in = i < WN * WN && y >= 0 && y < H && x >= 0 && x < W;
                g[k] = in ? (uint32_t)y * (uint32_t)W + (uint32_t)x : 0u;
                G[k] = in ? (uint32_t)(y >> 1) * (uint32_t)Wc + (uint32_t)(x >> 1) : 0u;
                if (GAL) {
                    dv[k] = dg[g[k]];
                    wv[k] = wg[g[k]];
                    if (!in) dv[k] = (VT)0;
                } else {
                    mk[k] = in ? mask[g[k]] : (uint8_t)0;
                }
            }
#pragma unroll
            for (int k = 0; k < NW; k++)
#pragma unroll
                for (int c = 0; c < 3; c++) { zv[k][c] = z[3 * (size_t)g[k] + c]; cv[k][c] = zc[3 * (size_t)G[k] + c]; rv[k][c] = r[3 * (size_t)g[k] + c]; }
#pragma unroll
            for (int k = 0; k < NW; k++) {
                const int i = threadIdx.x + k * kThreads;
                if (i < WN * WN) {
                    const bool in_mask = GAL ? dv[k] > (VT)0 : mk[k] != 0;
                    M[i] = (uint8_t)in_mask;
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        Zp[3 * i + c] = in_mask ? zv[k][c] + cv[k][c] : (VT)0;
                        R[3 * i + c] = in_mask ? (VT)rv[k][c] : (VT)0;
                    }
                    if (GAL) {
                        gal->d[i] = dv[k];
                        const int wy = i / WN, wx = i - wy * WN;
                        if (wy >= 1 && wy <= W1 && wx >= 1 && wx <= W1) gal->w[(wy - 1) * W1 + wx - 1] = wv[k];
                    }
                }
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < W1 * W1; i += kThreads) {
            const int wy = i / W1 + 1, wx = i % W1 + 1, j = wy * WN + wx, y = y0 + wy, x = x0 + wx;
            VT o[3] = {(VT)0, (VT)0, (VT)0};
            if (M[j]) {
                const VT nn = GAL ? gal->d[j] : (VT)ncount(y, x, H, W), w = kOmegaB / (s * nn);
                float4 e = make_float4(1.f, 1.f, 1.f, 1.f);
                if (GAL) e = gal->w[i];
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const VT zi = Zp[3 * j + c];
                    VT a;
                    if (GAL) {
                        a = nn * zi;
                        a -= e.x * Zp[3 * (j - WN) + c]; a -= e.y * Zp[3 * (j + WN) + c]; a -= e.z * Zp[3 * (j - 1) + c]; a -= e.w * Zp[3 * (j + 1) + c];
                    } else {
                        a = nn * zi - (Zp[3 * (j - WN) + c] + Zp[3 * (j + WN) + c] + Zp[3 * (j - 1) + c] + Zp[3 * (j + 1) + c]);
                    }
                    o[c] = zi + w * (R[3 * j + c] - s * a);
                }
            }
            Zt[3 * i] = o[0]; Zt[3 * i + 1] = o[1]; Zt[3 * i + 2] = o[2];
        }
        __syncthreads();
        for (int i = threadIdx.x; i < TS * TS; i += kThreads) {
            const int cy = i / TS, cx = i & (TS - 1), y = ty * TS + cy, x = tx * TS + cx;
            const int j = (cy + 2) * WN + cx + 2, k = (cy + 1) * W1 + cx + 1;
            if (y >= H || x >= W || !M[j]) continue;
            const VT nn = GAL ? gal->d[j] : (VT)ncount(y, x, H, W), w = kOmegaA / (s * nn);
            float4 e = make_float4(1.f, 1.f, 1.f, 1.f);
            if (GAL) e = gal->w[k];
            const size_t g = (size_t)y * W + x;
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const VT zi = Zt[3 * k + c];
                VT a;
                if (GAL) {
                    a = nn * zi;
                    a -= e.x * Zt[3 * (k - W1) + c]; a -= e.y * Zt[3 * (k + W1) + c]; a -= e.z * Zt[3 * (k - 1) + c]; a -= e.w * Zt[3 * (k + 1) + c];
                } else {
                    a = nn * zi - (Zt[3 * (k - W1) + c] + Zt[3 * (k + W1) + c] + Zt[3 * (k - 1) + c] + Zt[3 * (k + 1) + c]);
                }
                const VT zn = zi + w * (R[3 * j + c] - s * a);
                zout[3 * g + c] = zn;
                if (DOT) acc.v[c] += (double)r[3 * g + c] * (double)zn;
            }
        }
    }
    return acc;
}

// level 0 (with the partial sums of r . z)
__global__ __launch_bounds__(kThreads) void mg_up0_kernel(const uint8_t* __restrict__ mask, int H, int W, VT s, const VT* __restrict__ z,
                                                          const VT* __restrict__ zc, int Wc, const double* __restrict__ r, VT* __restrict__ zout,
                                                          double* __restrict__ part, const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                                          const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ double s_part[12];
    __shared__ TileLds lds;
    const D3 acc = mg_up_tiles<kThreads, double, true, false>(mask, H, W, s, z, zc, Wc, r, zout, active, tiles_x, ntiles, (int)blockIdx.x, (int)gridDim.x, lds,
                                                              nullptr, nullptr, nullptr);
    store_partial(part, block_sum(acc, s_part));
}
// levels >= 1
__global__ __launch_bounds__(kThreads) void mg_up_kernel(const uint8_t* __restrict__ mask, int H, int W, VT s, const VT* __restrict__ z,
                                                         const VT* __restrict__ zc, int Wc, const VT* __restrict__ r, VT* __restrict__ zout,
                                                         const uint8_t* __restrict__ active, int tiles_x, int ntiles,
                                                         const VT* __restrict__ dg, const float4* __restrict__ wg, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ TileLds lds;
    __shared__ GalLds gal;
    mg_up_tiles<kThreads, VT, false, true>(mask, H, W, s, z, zc, Wc, r, zout, active, tiles_x, ntiles, (int)blockIdx.x, (int)gridDim.x, lds, dg, wg, &gal);
}

// level 0 of an image so small that it is the tail's first level: the tail works on float32, the CG residual is float64
__global__ __launch_bounds__(kThreads) void rf_cast_kernel(const double* __restrict__ r, VT* __restrict__ out, size_t n, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (VT)r[i];
}
// ... and its r . z afterwards (the tiled "up" pass of level 0 computes it on larger images)
__global__ __launch_bounds__(kThreads) void rf_dot_kernel(const uint8_t* __restrict__ mask, uint32_t n, const double* __restrict__ a,
                                                          const VT* __restrict__ b, double* __restrict__ part, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ double s_part[12];
    D3 acc = {{0., 0., 0.}};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) acc.v[c] += a[3 * (size_t)i + c] * (double)b[3 * (size_t)i + c];
    }
    store_partial(part, block_sum(acc, s_part));
}

// The coarse end of the V-cycle (levels first_tail .. L-1, a few thousand cells at most) inside ONE workgroup: sweeps separated by
// workgroup barriers instead of kernel boundaries - a chain of ~25 tiny phases.  Every tail level lives in LDS for the duration of
// the cycle, in a layout made for the phases: each level with a one-cell ring around it (pitch W + 2), per cell the three channels
// of z, zt, r, the operator's diagonal d (float; 0 outside the mask and on the ring) and its four edge weights (half precision:
// multiples of 2^-level, exact).  z and zt are zero wherever d is: the stencil is d z - sum w_k z_k with no test, a phase is one LDS
// round trip.  (Round 4's first form addressed LDS or global memory through one generic pointer and tested the mask and the image
// border per neighbour: 1.4 us per phase, 36 us per iteration.)  d and w of all tail levels come as one packed image (mg_tail_pack_kernel,
// once per solve), copied in 16-byte pieces in the same round trip as the first level's r, which the "down" pass of the level above
// wrote; the first level's z goes back to that level's "up" pass (both in the compact global layout).
struct H4 { __half2 a, b; };                 // up, down | left, right
template <int NT>
__device__ __forceinline__ void mg_tail_levels(const LevelSet& ls, const uint8_t* __restrict__ g_dw, uint8_t* __restrict__ smem) {
    VT* const fl = reinterpret_cast<VT*>(smem);
    float* const dd = reinterpret_cast<float*>(smem + (size_t)ls.tail_floats * sizeof(VT));
    H4* const ww = reinterpret_cast<H4*>(dd + ls.tail_cells);
    const int t = (int)threadIdx.x, row0 = t >> 6, col0 = t & 63;
    constexpr int ROWS = NT / 64;
    {   // one round trip to global memory: the packed d / w image and the first level's r are requested, the vectors are zeroed (rings,
        // cells outside the mask) while they travel, then everything is put in place
        const Level& v = ls.lv[ls.first_tail];
        const int Wp = v.W + 2, cells = v.H * v.W;
        constexpr int NR = (kTailCells + NT - 1) / NT;
        VT rv[NR][3];
        int pp[NR];
        float dv[NR];
#pragma unroll
        for (int k = 0; k < NR; k++) {
            const int g = t + k * NT, gg = g < cells ? g : 0, y = gg / v.W, x = gg - y * v.W;
            pp[k] = g < cells ? (y + 1) * Wp + x + 1 : -1;
            dv[k] = reinterpret_cast<const float*>(g_dw)[v.moff + (y + 1) * Wp + x + 1];
#pragma unroll
            for (int c = 0; c < 3; c++) rv[k][c] = v.r[3 * gg + c];
        }
        const int nb16 = ls.tail_cells * 12 / 16;               // (<= 4 x 1024 sixteen-byte pieces: see kTailDwBytes)
        uint4 mv[4];
#pragma unroll
        for (int k = 0; k < 4; k++) mv[k] = t + k * NT < nb16 ? reinterpret_cast<const uint4*>(g_dw)[t + k * NT] : make_uint4(0u, 0u, 0u, 0u);
        const int n4 = ls.tail_floats * (int)sizeof(VT) / 16;
        float4* const q = reinterpret_cast<float4*>(smem);
        for (int i = t; i < n4; i += NT) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (t + k * NT < nb16) reinterpret_cast<uint4*>(dd)[t + k * NT] = mv[k];
        VT* const r = fl + v.off + 6 * v.cp;
#pragma unroll
        for (int k = 0; k < NR; k++)
            if (pp[k] >= 0 && dv[k] > 0.f) { r[3 * pp[k]] = rv[k][0]; r[3 * pp[k] + 1] = rv[k][1]; r[3 * pp[k] + 2] = rv[k][2]; }
    }
    __syncthreads();
    // (A v)(p) for channel c
    auto stencil = [&](const VT* __restrict__ v, int p, int Wp, VT d, const H4& e, int c) {
        VT a = d * v[3 * p + c];
        a -= __low2float(e.a) * v[3 * (p - Wp) + c]; a -= __high2float(e.a) * v[3 * (p + Wp) + c];
        a -= __low2float(e.b) * v[3 * (p - 1) + c]; a -= __high2float(e.b) * v[3 * (p + 1) + c];
        return a;
    };
    // zout = zin + omega / (s d) (r - s A zin)   (zin == nullptr: from zero)
    auto jacobi = [&](const Level& v, const VT* __restrict__ zin, VT* __restrict__ zout, VT omega) {
        const int Wp = v.W + 2;
        const VT* const r = fl + v.off + 6 * v.cp;
        for (int y = row0; y < v.H; y += ROWS)
            for (int x = col0; x < v.W; x += 64) {
                const int p = (y + 1) * Wp + x + 1;
                const VT d = dd[v.moff + p];
                if (!(d > (VT)0)) continue;
                const VT w = omega / (v.s * d);
                if (zin) {
                    const H4 e = ww[v.moff + p];
#pragma unroll
                    for (int c = 0; c < 3; c++) zout[3 * p + c] = zin[3 * p + c] + w * (r[3 * p + c] - v.s * stencil(zin, p, Wp, d, e, c));
                } else {
#pragma unroll
                    for (int c = 0; c < 3; c++) zout[3 * p + c] = w * r[3 * p + c];
                }
            }
        __syncthreads();
    };
    for (int l = ls.first_tail; l < ls.L; l++) {
        const Level& v = ls.lv[l];
        VT* const z = fl + v.off; VT* const zt = z + 3 * v.cp;
        if (l == ls.L - 1) {
            jacobi(v, nullptr, z, kOmega);
            for (int k = 0; k < USC_RF_COARSE_PAIRS; k++) { jacobi(v, z, zt, kOmega); jacobi(v, zt, z, kOmega); }
        } else {
            jacobi(v, nullptr, zt, kOmegaA);
            jacobi(v, zt, z, kOmegaB);
            // r_c(I) = sum over the children i of I inside the fine mask of (r_f - s_f A z_f)(i)
            const Level& cl = ls.lv[l + 1];
            const int Wp = v.W + 2, Wpc = cl.W + 2;
            const VT* const r = z + 6 * v.cp;
            VT* const rc = fl + cl.off + 6 * cl.cp;
            for (int Y = row0; Y < cl.H; Y += ROWS)
                for (int X = col0; X < cl.W; X += 64) {
                    const int P = (Y + 1) * Wpc + X + 1;
                    if (!(dd[cl.moff + P] > (VT)0)) continue;
                    VT acc[3] = {(VT)0, (VT)0, (VT)0};
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const int p = (2 * Y + (k >> 1) + 1) * Wp + 2 * X + (k & 1) + 1;      // (a child outside the image sits on the ring: d = 0)
                        const VT d = dd[v.moff + p];
                        if (!(d > (VT)0)) continue;
                        const H4 e = ww[v.moff + p];
#pragma unroll
                        for (int c = 0; c < 3; c++) acc[c] += r[3 * p + c] - v.s * stencil(z, p, Wp, d, e, c);
                    }
                    rc[3 * P] = acc[0]; rc[3 * P + 1] = acc[1]; rc[3 * P + 2] = acc[2];
                }
            __syncthreads();
        }
    }
    for (int l = ls.L - 2; l >= ls.first_tail; l--) {
        const Level& v = ls.lv[l];
        const Level& cl = ls.lv[l + 1];
        VT* const z = fl + v.off; VT* const zt = z + 3 * v.cp;
        const VT* const zc = fl + cl.off;
        const int Wp = v.W + 2, Wpc = cl.W + 2;
        for (int y = row0; y < v.H; y += ROWS)                  // z += z_c(parent) on the masked cells
            for (int x = col0; x < v.W; x += 64) {
                const int p = (y + 1) * Wp + x + 1;
                if (!(dd[v.moff + p] > (VT)0)) continue;
                const int P = ((y >> 1) + 1) * Wpc + (x >> 1) + 1;
                z[3 * p] += zc[3 * P]; z[3 * p + 1] += zc[3 * P + 1]; z[3 * p + 2] += zc[3 * P + 2];
            }
        __syncthreads();
        jacobi(v, z, zt, kOmegaB);
        jacobi(v, zt, z, kOmegaA);
    }
    {
        const Level& v = ls.lv[ls.first_tail];
        const VT* const z = fl + v.off;
        const int Wp = v.W + 2;
        for (int y = row0; y < v.H; y += ROWS)
            for (int x = col0; x < v.W; x += 64) {
                const int g = y * v.W + x, p = (y + 1) * Wp + x + 1;
                if (!(dd[v.moff + p] > (VT)0)) continue;
                v.z[3 * g] = z[3 * p]; v.z[3 * g + 1] = z[3 * p + 1]; v.z[3 * g + 2] = z[3 * p + 2];
            }
    }
}

__global__ __launch_bounds__(kTailThreads) void mg_tail_kernel(LevelSet ls, const uint8_t* __restrict__ g_dw, const int* __restrict__ ctl) {
    if (ctl[0]) return;
    __shared__ __align__(16) uint8_t smem[kTailLdsBytes];
    mg_tail_levels<kTailThreads>(ls, g_dw, smem);
}

// Once per solve: the Galerkin operator of a coarse level from the level above it.  With piecewise-constant prolongation over 2 x 2
// aggregates, P^T A P of aggregate I has the diagonal sum_{i in I} d_i - (the edge weights between children of I) and, towards the
// neighbouring aggregate J, the sum of the children's edge weights that cross from I to J.  The coarse operator is HALF of that: the
// over-correction by two that piecewise-constant P needs (see the header; an interior cell gets 4 / 1 1 1 1 on every level).  A coarse
// cell that is only partly inside the mask - the ragged Dirichlet boundary of a real silhouette - gets exactly the diagonal and the
// couplings its children have, where re-discretising the stencil on the coarse mask pretends the whole cell is inside: NumPy model
// (tools/dev/rf_model.py) on the bench's hole 44 -> 24 iterations, the kernels 52 -> see DESIGN.md section 3b.  All values are sums of
// multiples of 2^-level: exact in float32, the same on every run.  FINE0: the fine level is level 0 (d = in-image neighbour count,
// w = 1 towards neighbours in the mask).
template <bool FINE0>
__global__ __launch_bounds__(kThreads) void mg_galerkin_kernel(const uint8_t* __restrict__ mf, const VT* __restrict__ df, const float4* __restrict__ wf,
                                                               int Hf, int Wf, uint8_t* __restrict__ mc, VT* __restrict__ dc, float4* __restrict__ wc,
                                                               int Hc, int Wc) {
    const uint32_t n = (uint32_t)Hc * (uint32_t)Wc;
    for (uint32_t I = blockIdx.x * blockDim.x + threadIdx.x; I < n; I += gridDim.x * blockDim.x) {
        const int Y = (int)(I / (uint32_t)Wc), X = (int)(I - (uint32_t)Y * (uint32_t)Wc);
        int any = 0;
        VT dsum = 0, internal = 0, up = 0, dn = 0, lf = 0, rt = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int y = 2 * Y + (k >> 1), x = 2 * X + (k & 1);
            if (y >= Hf || x >= Wf) continue;
            const size_t i = (size_t)y * Wf + x;
            if (!mf[i]) continue;
            any = 1;
            VT d;
            float4 e;
            if (FINE0) {
                d = (VT)ncount(y, x, Hf, Wf);
                e.x = (y > 0 && mf[i - Wf]) ? 1.f : 0.f;
                e.y = (y < Hf - 1 && mf[i + Wf]) ? 1.f : 0.f;
                e.z = (x > 0 && mf[i - 1]) ? 1.f : 0.f;
                e.w = (x < Wf - 1 && mf[i + 1]) ? 1.f : 0.f;
            } else {
                d = df[i];
                e = wf[i];
            }
            dsum += d;
            if ((k >> 1) == 0) { up += e.x; internal += e.y; } else { dn += e.y; internal += e.x; }
            if ((k & 1) == 0) { lf += e.z; internal += e.w; } else { rt += e.w; internal += e.z; }
        }
        mc[I] = (uint8_t)any;
        dc[I] = any ? (dsum - internal) * (VT)0.5 : (VT)0;
        wc[I] = make_float4(up * 0.5f, dn * 0.5f, lf * 0.5f, rt * 0.5f);
    }
}

// Once per solve: d (float) and w (four halves) of the tail levels in the ringed layout mg_tail_levels keeps them in - [d of all tail
// cells][w of all tail cells], a level's cells from its moff.
__global__ __launch_bounds__(kTailThreads) void mg_tail_pack_kernel(LevelSet ls, uint8_t* __restrict__ g_dw) {
    const int t = (int)threadIdx.x;
    float* const dd = reinterpret_cast<float*>(g_dw);
    H4* const ww = reinterpret_cast<H4*>(dd + ls.tail_cells);
    for (int i = t; i < ls.tail_cells * 3; i += kTailThreads) reinterpret_cast<uint32_t*>(g_dw)[i] = 0u;
    __syncthreads();
    for (int l = ls.first_tail; l < ls.L; l++) {
        const Level& v = ls.lv[l];
        for (int g = t; g < v.H * v.W; g += kTailThreads) {
            if (!v.mask[g]) continue;
            const int y = g / v.W, x = g - y * v.W, p = v.moff + (y + 1) * (v.W + 2) + x + 1;
            float d;
            float4 e;
            if (l == 0) {                                      // (an image of at most 8 x 8 cells: level 0 is the tail)
                d = (float)ncount(y, x, v.H, v.W);
                e.x = (y > 0 && v.mask[g - v.W]) ? 1.f : 0.f;
                e.y = (y < v.H - 1 && v.mask[g + v.W]) ? 1.f : 0.f;
                e.z = (x > 0 && v.mask[g - 1]) ? 1.f : 0.f;
                e.w = (x < v.W - 1 && v.mask[g + 1]) ? 1.f : 0.f;
            } else {
                d = v.d[g];
                e = v.w[g];
            }
            dd[p] = d;
            H4 h;
            h.a = __floats2half2_rn(e.x, e.y);
            h.b = __floats2half2_rn(e.z, e.w);
            ww[p] = h;
        }
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

int grid_of(size_t n) {                 // element-wise kernels: one cell per thread up to 8192 workgroups
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
    int H, W, L, first_tail, tail_floats, tail_cells;
    uint32_t n;
    Level lv[kMaxLevels];
    double *scal, *part_rr, *part_pap, *part_rz[2], *part_cross, *x, *p[2], *Ap, *r;
    int* ctl;                                       // {done, iterations run, converged}: the device-side convergence state
    uint8_t *mask0, *tail_dw;                       // tail_dw: the tail levels' operators in the tail's LDS layout (mg_tail_pack_kernel)

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
        part_cross = reinterpret_cast<double*>(take(psz));
        const size_t v0 = (size_t)H * W * 3 * sizeof(double);
        x = reinterpret_cast<double*>(take(v0));
        p[0] = reinterpret_cast<double*>(take(v0));
        p[1] = reinterpret_cast<double*>(take(v0));
        Ap = reinterpret_cast<double*>(take(v0));
        r = reinterpret_cast<double*>(take(v0));
        int hs[kMaxLevels], wsz[kMaxLevels];
        L = level_dims(H, W, hs, wsz);
        for (int l = 0; l < L; l++) {
            const size_t vb = (size_t)hs[l] * wsz[l] * 3 * sizeof(VT);
            lv[l].H = hs[l]; lv[l].W = wsz[l];
            lv[l].tiles_x = (wsz[l] + TS - 1) / TS;
            lv[l].ntiles = lv[l].tiles_x * ((hs[l] + TS - 1) / TS);
            lv[l].s = (VT)1;                                             // coarse operator = the unscaled re-discretised stencil (see the header)
            lv[l].z = reinterpret_cast<VT*>(take(vb));
            lv[l].zt = reinterpret_cast<VT*>(take(vb));
            lv[l].r = reinterpret_cast<VT*>(take(vb));              // (level 0: only used when level 0 is the tail's first level)
            lv[l].mask = take((size_t)hs[l] * wsz[l]);                 // level 0: a private copy of the caller's mask
            lv[l].d = l > 0 ? reinterpret_cast<VT*>(take((size_t)hs[l] * wsz[l] * sizeof(VT))) : nullptr;
            lv[l].w = l > 0 ? reinterpret_cast<float4*>(take((size_t)hs[l] * wsz[l] * sizeof(float4))) : nullptr;
            lv[l].active = take((size_t)lv[l].ntiles);
        }
        mask0 = lv[0].mask;
        tail_dw = take(kTailDwBytes);
        // the tail: the coarsest levels as far as their ringed LDS layout (9 values of VT + d + four half-precision weights per cell) fits
        auto ringed = [&](int l) { return ((size_t)(lv[l].H + 2) * (lv[l].W + 2) + 15) & ~(size_t)15; };
        auto lds_need = [&](int l) { return ringed(l) * (9 * sizeof(VT) + 12); };
        first_tail = L - 1;
        size_t used = lds_need(first_tail);
        while (first_tail > 1 && (size_t)lv[first_tail - 1].H * lv[first_tail - 1].W <= (size_t)kTailCells && used + lds_need(first_tail - 1) <= (size_t)kTailLdsBytes) {
            first_tail--;
            used += lds_need(first_tail);
        }
        int fo = 0, mo = 0;
        for (int l = 0; l < L; l++) {
            lv[l].off = lv[l].moff = lv[l].cp = 0;
            if (l < first_tail) continue;
            lv[l].cp = (int)ringed(l);
            lv[l].off = fo; fo += 9 * lv[l].cp;
            lv[l].moff = mo; mo += lv[l].cp;
        }
        tail_floats = fo;
        tail_cells = mo;
    }

    LevelSet level_set() const {
        LevelSet ls;
        ls.L = L; ls.first_tail = first_tail; ls.tail_floats = tail_floats; ls.tail_cells = tail_cells;
        for (int l = 0; l < kMaxLevels; l++) ls.lv[l] = lv[l < L ? l : L - 1];
        return ls;
    }

    static dim3 tile_grid(const Level& v) { return dim3((unsigned)(v.ntiles < 2048 ? v.ntiles : 2048)); }

    // z = V-cycle(r): the result is lv[0].zt (lv[0].z when level 0 is the tail's first level), the partial sums of r . z in part_rz_new
    const VT* vcycle(Launcher& q, double* part_rz_new) const {
        const dim3 b(kThreads);
        if (first_tail == 0) {                                         // an image of at most 8 x 8 cells: everything inside the tail kernel
            q.launch(rf_cast_kernel, dim3(grid_of((size_t)n * 3)), b, r, lv[0].r, (size_t)n * 3, ctl);
        }
        const LevelSet ls = level_set();
        if (first_tail > 0) {
            const Level &v = lv[0], &c = lv[1];
            q.launch(mg_down0_kernel, tile_grid(v), b, v.mask, v.H, v.W, v.s, (const double*)r, v.z, c.mask, c.H, c.W, c.r, v.active, v.tiles_x, v.ntiles, ctl);
        }
        for (int l = 1; l < first_tail; l++) {
            const Level &v = lv[l], &c = lv[l + 1];
            q.launch(mg_down_kernel, tile_grid(v), b, v.mask, v.H, v.W, v.s, (const VT*)v.r, v.z, c.mask, c.H, c.W, c.r, v.active, v.tiles_x, v.ntiles,
                     (const VT*)v.d, (const float4*)v.w, ctl);
        }
        q.launch(mg_tail_kernel, dim3(1), dim3(kTailThreads), ls, (const uint8_t*)tail_dw, ctl);
        for (int l = first_tail - 1; l >= 1; l--) {
            const Level &v = lv[l], &c = lv[l + 1];
            const VT* zc = l + 1 == first_tail ? c.z : c.zt;
            q.launch(mg_up_kernel, tile_grid(v), b, v.mask, v.H, v.W, v.s, (const VT*)v.z, zc, c.W, (const VT*)v.r, v.zt, v.active, v.tiles_x, v.ntiles,
                     (const VT*)v.d, (const float4*)v.w, ctl);
        }
        if (first_tail > 0) {
            const Level &v = lv[0], &c = lv[1];
            const VT* zc = 1 == first_tail ? c.z : c.zt;
            q.launch(mg_up0_kernel, dim3(kBlocks), b, v.mask, v.H, v.W, v.s, (const VT*)v.z, zc, c.W, (const double*)r, v.zt, part_rz_new, v.active, v.tiles_x, v.ntiles, ctl);
        }
        if (first_tail == 0) {
            q.launch(rf_dot_kernel, dim3(kBlocks), b, mask0, n, (const double*)r, (const VT*)lv[0].z, part_rz_new, ctl);
            return lv[0].z;
        }
        return lv[0].zt;
    }

    // one preconditioned CG iteration; cur selects which of the two r.z partial arrays holds the previous value, pc which p is current
    void iteration(Launcher& q, int first, int& cur, int& pc) const {
        const dim3 b(kThreads), g(kBlocks);           // reductions always use kBlocks blocks: the partial arrays have that many rows
        const VT* z = vcycle(q, part_rz[cur ^ 1]);                                                 // z = M^-1 r and the partial sums of r.z
        const Level& v = lv[0];
        q.launch(rf_pap_kernel, g, b, mask0, H, W, z, (const double*)p[pc], p[pc ^ 1], Ap, (const double*)part_rz[cur], (const double*)part_rz[cur ^ 1],
                 (const double*)part_cross, first, part_pap, v.active, v.tiles_x, v.ntiles, ctl);
        cur ^= 1;
        pc ^= 1;
        q.launch(rf_xr_kernel, g, b, mask0, n, (const double*)p[pc], (const double*)Ap, x, r, z, (const double*)part_rz[cur], (const double*)part_pap, part_rr,
                 part_cross, ctl);
    }
};

constexpr int kItersPerCheck = 2;         // iterations between two device-side convergence tests

}  // namespace

extern "C" size_t usc_regionfill_ws_bytes(int H, int W) {
    int hs[kMaxLevels], ws[kMaxLevels];
    const int L = level_dims(H, W, hs, ws);
    size_t total = 256 + 5 * rf_align((size_t)kBlocks * 3 * sizeof(double));
    total += rf_align(kTailDwBytes);
    total += 5 * rf_align((size_t)H * W * 3 * sizeof(double));                  // x, p (two), Ap, r
    for (int l = 0; l < L; l++) {
        const size_t tiles = (size_t)((ws[l] + TS - 1) / TS) * ((hs[l] + TS - 1) / TS);
        total += 3 * rf_align((size_t)hs[l] * ws[l] * 3 * sizeof(VT)) + rf_align((size_t)hs[l] * ws[l]) + rf_align(tiles);   // z, zt, r, mask, tile flags per level
        if (l > 0) total += rf_align((size_t)hs[l] * ws[l] * sizeof(VT)) + rf_align((size_t)hs[l] * ws[l] * sizeof(float4));       // the level's operator: d, w
    }
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
    if (first_iter == 0) {                                      // a new solve: masks and tile flags of every level, x0, r0, control block
        if (hipMemcpyAsync(s.mask0, mask, (size_t)H * W, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return fail(USC_E_LAUNCH, "usc_regionfill_u8: mask copy failed");
        for (int l = 1; l < s.L; l++) {                          // masks and operators of the coarse levels
            const Level &f = s.lv[l - 1], &c = s.lv[l];
            const dim3 g(grid_of((size_t)c.H * c.W));
            if (l == 1) q.launch(mg_galerkin_kernel<true>, g, b, (const uint8_t*)f.mask, (const VT*)nullptr, (const float4*)nullptr, f.H, f.W, c.mask, c.d, c.w, c.H, c.W);
            else q.launch(mg_galerkin_kernel<false>, g, b, (const uint8_t*)f.mask, (const VT*)f.d, (const float4*)f.w, f.H, f.W, c.mask, c.d, c.w, c.H, c.W);
        }
        for (int l = 0; l < s.L && l < (s.first_tail > 0 ? s.first_tail : 1); l++)
            q.launch(mg_active_kernel, dim3((unsigned)s.lv[l].ntiles), b, (const uint8_t*)s.lv[l].mask, s.lv[l].H, s.lv[l].W, s.lv[l].active, s.lv[l].tiles_x);
        q.launch(mg_tail_pack_kernel, dim3(1), dim3(kTailThreads), s.level_set(), s.tail_dw);
        q.launch(rf_init_kernel, dim3(kBlocks), b, img, s.mask0, H, W, s.x, s.r, s.part_rr);
        q.launch(rf_publish_kernel, dim3(1), b, s.part_rr, s.scal, 0, s.ctl, rtol2, 0);
    }
    // Iterations [first_iter, first_iter + n_iters), a convergence test on the device after every kItersPerCheck of them; once it has
    // passed, the kernels still in the queue return at their first instruction.  No host synchronisation.
    int cur = first_iter & 1;                                   // which r.z partial array holds the previous value
    int pc = first_iter & 1;                                    // which p buffer holds the current direction
    int since = 0;
    for (int it = first_iter; it < first_iter + n_iters; it++) {
        s.iteration(q, it == 0 ? 1 : 0, cur, pc);
        if (++since == kItersPerCheck || it + 1 == first_iter + n_iters) {
            q.launch(rf_publish_kernel, dim3(1), b, s.part_rr, s.scal, 1, s.ctl, rtol2, since);
            since = 0;
        }
    }
    q.launch(rf_out_kernel, dim3(kBlocks), b, img, s.mask0, s.n, s.x, out, (const int*)s.ctl, status);
    return check_launch("regionfill kernels");
}
