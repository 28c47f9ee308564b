// usc_objects.hip - remove_invalid_objects on the device (reference unscreen/utils/maskprocess.py:77-152).
//
// The reference enumerates cv2.findContours(RETR_LIST) borders - sequential border following - and, per border, tests the
// polygon area (contourArea < 100: skip), fills the border (drawContours FILLED) and scores the filled region.  Here the same
// quantities come from connected-component labels, with no tracing and no host round trip:
//   * every border separates one 8-connected foreground component F from one 4-connected background component B (the
//     frame is padded by a background ring, so "outside" is an ordinary component B0).  B = parent(F) gives F's outer
//     border, F = parent(B) gives the border of hole B; parents are read off the pixel left of a component's first pixel in
//     raster order - and union-find with "larger root links to smaller" makes the root of a component exactly that first pixel;
//   * the border's pixel chain is the sequence of foreground pixels met when walking the pixel cracks between F and B
//     (turn left at a concave corner = diagonal step, straight, or turn right around the pixel's own corner = same pixel), so
//     the shoelace sum of the traced polygon is a sum of one integer cross product per crack: int64 atomics, exact;
//   * drawContours(FILLED) of an outer border paints F and everything nested inside it; of a hole border, the hole, what is
//     nested inside it, and the chain itself = the pixels of F that touch B across a crack.  Sums over those regions are
//     per-component sums pushed up the containment tree;
//   * alpha is kept where the pixel's component is covered by a kept border (its own or an ancestor's) or the pixel lies on
//     the chain of a kept hole border.
// Validated against the oracle's border follower on random and structured masks (nested rings, checkerboards, diagonal
// slits, shapes on the frame border, blobs with area around 100): tests/test_gpu_objects.py.  The float64 score sums are
// accumulated with atomics, so they can differ from the host's in the last bit; they only enter threshold comparisons.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

struct Comp {                      // one connected component (foreground or background), 96 bytes
    int first;                     // raster-first pixel (padded index) = union-find root
    int parent;                    // component id of the enclosing component of the other class (-1: outside ring)
    int is_fg;
    int keep;                      // its border passed the area and score tests
    int covered;                   // lies inside a kept border
    unsigned int chain_cnt;        // hole borders: pixels of the parent that touch this hole
    unsigned long long cnt, sg;    // own pixels, sum of seg
    unsigned long long t_cnt, t_sg;          // subtree totals
    unsigned long long chain_sg;
    double sc, t_sc, chain_sc;     // sums of the score map
    long long area2;               // twice the signed polygon area of its border
};

__device__ __forceinline__ int ld_label(const int* lab, int i) { return __hip_atomic_load(&lab[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int find_root(const int* lab, int i) {
    int r = i;
    for (;;) {
        const int p = ld_label(lab, r);
        if (p == r) return r;
        r = p;
    }
}

__device__ __forceinline__ void unite(int* lab, int a, int b) {
    for (;;) {
        a = find_root(lab, a);
        b = find_root(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }        // the larger root hangs under the smaller one
        const int old = atomicMin(&lab[a], b);
        if (old == a) return;
        a = old;
    }
}

__device__ __forceinline__ bool is_fg(const uint8_t* __restrict__ alpha, int H, int W, int py, int px) {     // padded coordinates
    return py >= 1 && py <= H && px >= 1 && px <= W && alpha[(size_t)(py - 1) * W + (px - 1)] != 0;
}

}  // namespace

// Labelling, step 1: rows are cut into 64-pixel segments (one wavefront each); every pixel starts with the index of the
// first pixel of its run inside the segment (same class, contiguous) - one ballot and a few bit operations, no atomics.
__global__ __launch_bounds__(256) void obj_runs_kernel(const uint8_t* __restrict__ alpha, int H, int W, int* __restrict__ lab,
                                                       int* __restrict__ ncomp) {
    const int Wp = W + 2;
    const int py = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & (kWave - 1);
    const bool live = px < Wp;
    const bool f = live && is_fg(alpha, H, W, py, px);
    const unsigned long long C = __ballot(f);
    const unsigned long long S = (C ^ (C << 1)) | 1ull;                      // bit l: lane l starts a run
    const unsigned long long upto = S & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
    const int start = 63 - __clzll((long long)upto);
    if (live) lab[py * Wp + px] = py * Wp + px - lane + start;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *ncomp = 0;
}

// Labelling, step 2: unions only where two runs START to touch - a segment boundary inside a run, the first vertically adjacent
// pair of two runs, and, for the 8-connected foreground, the diagonal contacts at run ends - a few per run instead of a few per
// pixel.  Roots end up at the smallest index of the component = its first pixel in raster order.
__global__ __launch_bounds__(256) void obj_merge_kernel(const uint8_t* __restrict__ alpha, int H, int W, int* __restrict__ lab) {
    const int Wp = W + 2, np = (H + 2) * Wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const int py = i / Wp, px = i - py * Wp;
        const bool c = is_fg(alpha, H, W, py, px);
        const bool seg_start = (px & 63) == 0;
        const bool w_same = px > 0 && is_fg(alpha, H, W, py, px - 1) == c;
        const bool start_i = seg_start || !w_same;
        if (seg_start && w_same) unite(lab, i, i - 1);
        if (py == 0) continue;
        const bool cn = is_fg(alpha, H, W, py - 1, px);
        if (cn == c) {
            const bool start_n = seg_start || px == 0 || is_fg(alpha, H, W, py - 1, px - 1) != cn;
            if (start_i || start_n) unite(lab, i, i - Wp);
        } else if (c) {                                      // foreground below background: diagonal contacts
            if (px > 0 && !w_same && is_fg(alpha, H, W, py - 1, px - 1)) unite(lab, i, i - Wp - 1);
            if (px + 1 < Wp && !is_fg(alpha, H, W, py, px + 1) && is_fg(alpha, H, W, py - 1, px + 1)) unite(lab, i, i - Wp + 1);
        }
    }
}

__global__ __launch_bounds__(256) void obj_flatten_kernel(const uint8_t* __restrict__ alpha, int H, int W, int* __restrict__ lab,
                                                          int* __restrict__ cid, Comp* __restrict__ comps, int* __restrict__ ncomp, int kcap) {
    const int Wp = W + 2, np = (H + 2) * Wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const int r = find_root(lab, i);
        lab[i] = r;
        if (r == i) {                                        // a root: give the component a dense id
            const int id = atomicAdd(ncomp, 1);
            cid[i] = id;
            if (id < kcap) {
                Comp c = {};
                const int py = i / Wp, px = i - py * Wp;
                c.first = i; c.parent = -1; c.is_fg = is_fg(alpha, H, W, py, px) ? 1 : 0;
                comps[id] = c;
            }
        }
    }
}

// own sums of every component + parent links.  Atomics on one address serialise in the L2, and a silhouette is hundreds of
// thousands of pixels of ONE component: every lane therefore walks 16 consecutive pixels and only flushes when the component
// changes, and a wavefront whose lanes all end in the same component combines them first - one atomic triple per 1024 pixels.
__global__ __launch_bounds__(256) void obj_stats_kernel(const uint8_t* __restrict__ seg, const double* __restrict__ score, int H, int W,
                                                        const int* __restrict__ lab, const int* __restrict__ cid, Comp* __restrict__ comps,
                                                        int kcap) {
    constexpr int PPT = 16;
    const int Wp = W + 2, np = (H + 2) * Wp;
    const int c_out = cid[0];
    const int nthreads = (np + PPT - 1) / PPT;
    const int total = (nthreads + 63) & ~63;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        int cur = -1;
        double s = 0.;
        unsigned long long g = 0, n = 0;
        auto flush = [&]() {
            if (cur >= 0 && n) { atomicAdd(&comps[cur].cnt, n); atomicAdd(&comps[cur].sg, g); atomicAdd(&comps[cur].sc, s); }
            s = 0.; g = 0; n = 0;
        };
        const int i0 = t * PPT;
        for (int k = 0; k < PPT; k++) {
            const int i = i0 + k;
            if (i >= np) break;
            const int root = lab[i];
            int c = cid[root];
            if (root == i && i > 0 && c < kcap) comps[c].parent = cid[lab[i - 1]];     // the pixel left of the first pixel
            if (c >= kcap || c == c_out) c = -1;             // the outside ring has no border: its sums are never used
            if (c != cur) { flush(); cur = c; }
            if (c >= 0) {
                const int py = i / Wp, px = i - py * Wp;
                n++;
                if (py >= 1 && py <= H && px >= 1 && px <= W) {
                    const size_t o = (size_t)(py - 1) * W + (px - 1);
                    s += score[o];
                    g += seg[o];
                }
            }
        }
        const int c0 = __shfl(cur, 0, kWave);
        if (__all(cur == c0)) {                              // the wave's open segments belong to one component: combine them
#pragma unroll
            for (int off = kWave / 2; off > 0; off >>= 1) {
                s += __shfl_down(s, off, kWave);
                g += __shfl_down(g, off, kWave);
                n += __shfl_down(n, off, kWave);
            }
            if ((threadIdx.x & (kWave - 1)) == 0) flush();
        } else {
            flush();
        }
    }
}

// one term per crack between a foreground pixel and a background pixel
__global__ __launch_bounds__(256) void obj_cracks_kernel(const uint8_t* __restrict__ alpha, const uint8_t* __restrict__ seg,
                                                         const double* __restrict__ score, int H, int W, const int* __restrict__ lab,
                                                         const int* __restrict__ cid, Comp* __restrict__ comps, int kcap) {
    const int Wp = W + 2;
    const int n = H * W;
    const int dy[4] = {-1, 0, 1, 0}, dx[4] = {0, 1, 0, -1};           // N, E, S, W; the walk along side d goes in direction d + 1
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!alpha[i]) continue;
        const int y = i / W, x = i - y * W, py = y + 1, px = x + 1;
        const int f = cid[lab[py * Wp + px]];
        if (f >= kcap) continue;
        int seen[4];
        int nseen = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const int by = py + dy[d], bx = px + dx[d];
            if (is_fg(alpha, H, W, by, bx)) continue;
            const int b = cid[lab[by * Wp + bx]];
            if (b >= kcap) continue;
            const int wy = dy[(d + 1) & 3], wx = dx[(d + 1) & 3];
            int ny, nx;
            if (is_fg(alpha, H, W, by + wy, bx + wx)) { ny = by + wy; nx = bx + wx; }          // concave corner: diagonal step
            else if (is_fg(alpha, H, W, py + wy, px + wx)) { ny = py + wy; nx = px + wx; }     // straight
            else { ny = py; nx = px; }                                                         // around the pixel's own corner
            const long long term = (long long)x * (ny - 1) - (long long)(nx - 1) * y;
            if (comps[b].parent == f) {                      // B is a hole of F: the hole's border
                atomicAdd((unsigned long long*)&comps[b].area2, (unsigned long long)term);
                bool dup = false;
                for (int k = 0; k < nseen; k++) dup |= seen[k] == b;
                if (!dup) {
                    seen[nseen++] = b;
                    atomicAdd(&comps[b].chain_cnt, 1u);
                    atomicAdd(&comps[b].chain_sg, (unsigned long long)seg[i]);
                    atomicAdd(&comps[b].chain_sc, score[i]);
                }
            } else {                                         // B encloses F: F's outer border
                atomicAdd((unsigned long long*)&comps[f].area2, (unsigned long long)term);
            }
        }
    }
}

// subtree totals: every component adds its own sums to all its ancestors
__global__ __launch_bounds__(256) void obj_tree_sum_kernel(Comp* __restrict__ comps, const int* __restrict__ ncomp, int kcap) {
    const int K = min(*ncomp, kcap);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        const unsigned long long n = comps[c].cnt, g = comps[c].sg;
        const double s = comps[c].sc;
        atomicAdd(&comps[c].t_cnt, n); atomicAdd(&comps[c].t_sg, g); atomicAdd(&comps[c].t_sc, s);
        int a = comps[c].parent;
        for (int depth = 0; a >= 0 && comps[a].parent >= 0 && depth < (1 << 20); depth++) {      // stop below the outside ring
            atomicAdd(&comps[a].t_cnt, n); atomicAdd(&comps[a].t_sg, g); atomicAdd(&comps[a].t_sc, s);
            a = comps[a].parent;
        }
    }
}

// maskprocess.py:121-147: area test, saliency / consensus test
__global__ __launch_bounds__(256) void obj_decide_kernel(Comp* __restrict__ comps, const int* __restrict__ ncomp, int kcap, double denom,
                                                         double saliency_thr, double consensus_thr) {
    const int K = min(*ncomp, kcap);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        Comp& q = comps[c];
        int keep = 0;
        if (q.parent >= 0) {                                  // the outside ring has no border
            const long long a2 = q.area2 < 0 ? -q.area2 : q.area2;
            if ((double)a2 * 0.5 >= 100.) {
                double ssum = q.t_sc, gsum = (double)q.t_sg, cnt = (double)q.t_cnt;
                if (!q.is_fg) { ssum += q.chain_sc; gsum += (double)q.chain_sg; cnt += (double)q.chain_cnt; }
                const double sal = ssum / denom, cons = (gsum / cnt) / 255.;
                keep = ((sal > saliency_thr && cons > consensus_thr) || sal > saliency_thr * 10) ? 1 : 0;
            }
        }
        q.keep = keep;
    }
}

__global__ __launch_bounds__(256) void obj_cover_kernel(Comp* __restrict__ comps, const int* __restrict__ ncomp, int kcap) {
    const int K = min(*ncomp, kcap);
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < K; c += gridDim.x * blockDim.x) {
        int cov = 0, a = c;
        for (int depth = 0; a >= 0 && depth < (1 << 20) && !cov; depth++) { cov = comps[a].keep; a = comps[a].parent; }
        comps[c].covered = cov;
    }
}

__global__ __launch_bounds__(256) void obj_apply_kernel(uint8_t* __restrict__ alpha, int H, int W, const int* __restrict__ lab,
                                                        const int* __restrict__ cid, const Comp* __restrict__ comps, int kcap) {
    const int Wp = W + 2;
    const int n = H * W;
    const int dy[4] = {-1, 0, 1, 0}, dx[4] = {0, 1, 0, -1};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!alpha[i]) continue;
        const int y = i / W, x = i - y * W, py = y + 1, px = x + 1;
        const int f = cid[lab[py * Wp + px]];
        bool valid = f < kcap && comps[f].covered;
#pragma unroll
        for (int d = 0; d < 4 && !valid; d++) {
            const int by = py + dy[d], bx = px + dx[d];
            if (is_fg(alpha, H, W, by, bx)) continue;         // (neighbours are only ever cleared to 0 if they were foreground: see below)
            const int b = cid[lab[by * Wp + bx]];
            valid = b < kcap && comps[b].keep && comps[b].parent == f;
        }
        if (!valid) alpha[i] = 0;
    }
}

// The apply kernel reads neighbours' alpha while others may already have been cleared: is_fg() of a neighbour could flip from
// foreground to background mid-kernel and its label (a foreground label) would then be read as if it were a hole.  The test
// `comps[b].parent == f` with b a foreground component's id can never hold (a foreground component's parent is a background
// component), so a cleared neighbour is never mistaken for a kept hole.

// maskprocess.py:155-189 get_score_map / build_score_map on the device (float64, same operations as the host builder)
__device__ __forceinline__ double lin(double a, double b, int n, int i) {
    if (n == 1) return a;
    if (i == n - 1) return b;
    return (double)i * ((b - a) / (double)(n - 1)) + a;
}

__global__ __launch_bounds__(256) void score_raw_kernel(int H, int W, int y0, int x0, double* __restrict__ out, unsigned long long* __restrict__ mx) {
    const int n = H * W;
    double m = 0.;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const double cv = x >= x0 ? lin(0, 1, W - x0, x - x0) : lin(1, 0, x0, x);
        const double rv = y >= y0 ? lin(0, 1, H - y0, y - y0) : lin(1, 0, y0, y);
        const double s = sqrt(cv * cv + rv * rv);
        out[i] = s;
        m = fmax(m, s);
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) m = fmax(m, __shfl_down(m, off, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0) atomicMax(mx, (unsigned long long)__double_as_longlong(m));      // non-negative doubles order like integers
}

__global__ __launch_bounds__(256) void score_norm_kernel(int n, double* __restrict__ out, const unsigned long long* __restrict__ mx) {
    const double m = __longlong_as_double((long long)*mx);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (m - out[i]) / m;
}

extern "C" int usc_score_map_f64(int H, int W, double center_y, double center_x, double* out, void* ws8, usc_stream_t stream) {
    if (!out || !ws8 || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_score_map_f64: bad argument (ws8: 8 bytes of device scratch)");
    if (too_big(H, W)) return fail(USC_E_UNSUPPORTED, "usc_score_map_f64: image too large");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws8, 0, 8, st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_score_map_f64: memset failed");
    const int n = H * W;
    hipLaunchKernelGGL(score_raw_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, st, H, W, (int)(H * center_y), (int)(W * center_x), out,
                       reinterpret_cast<unsigned long long*>(ws8));
    hipLaunchKernelGGL(score_norm_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, st, n, out, reinterpret_cast<unsigned long long*>(ws8));
    return check_launch("score_map kernels");
}

static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static int comp_cap(int H, int W) { return (int)(((size_t)(H + 2) * (W + 2)) / 2 + 8); }     // isolated pixels on every other row / column: np / 4 + their holes

extern "C" size_t usc_remove_invalid_objects_ws_bytes(int H, int W) {
    const size_t np = (size_t)(H + 2) * (W + 2);
    return 256 + 2 * al256(np * sizeof(int)) + al256((size_t)comp_cap(H, W) * sizeof(Comp));
}

extern "C" int usc_remove_invalid_objects_u8(uint8_t* alpha, const uint8_t* seg, int H, int W, const double* score, double saliency_thr,
                                             double consensus_thr, void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (!alpha || !score || !ws || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_remove_invalid_objects_u8: bad argument");
    if ((long long)(H + 2) * (W + 2) >= (1ll << 30)) return fail(USC_E_UNSUPPORTED, "usc_remove_invalid_objects_u8: image too large");
    if (ws_bytes < usc_remove_invalid_objects_ws_bytes(H, W)) return fail(USC_E_WORKSPACE, "usc_remove_invalid_objects_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int np = (H + 2) * (W + 2), kcap = comp_cap(H, W);
    uint8_t* base = reinterpret_cast<uint8_t*>(ws);
    int* ncomp = reinterpret_cast<int*>(base);
    int* lab = reinterpret_cast<int*>(base + 256);
    int* cid = reinterpret_cast<int*>(base + 256 + al256((size_t)np * sizeof(int)));
    Comp* comps = reinterpret_cast<Comp*>(base + 256 + 2 * al256((size_t)np * sizeof(int)));
    if (!seg) seg = alpha;                                  // consensus against the input itself: alpha is only written by the last kernel
    const dim3 b(256), gp(grid_for((size_t)np, 256)), gn(grid_for((size_t)H * W, 256)), gk(grid_for((size_t)kcap, 256));
    hipLaunchKernelGGL(obj_runs_kernel, dim3((W + 2 + 255) / 256, H + 2), b, 0, st, alpha, H, W, lab, ncomp);
    hipLaunchKernelGGL(obj_merge_kernel, gp, b, 0, st, alpha, H, W, lab);
    hipLaunchKernelGGL(obj_flatten_kernel, gp, b, 0, st, alpha, H, W, lab, cid, comps, ncomp, kcap);
    hipLaunchKernelGGL(obj_stats_kernel, dim3(grid_for((size_t)(np + 15) / 16, 256)), b, 0, st, seg, score, H, W, lab, cid, comps, kcap);
    hipLaunchKernelGGL(obj_cracks_kernel, gn, b, 0, st, alpha, seg, score, H, W, lab, cid, comps, kcap);
    hipLaunchKernelGGL(obj_tree_sum_kernel, gk, b, 0, st, comps, ncomp, kcap);
    hipLaunchKernelGGL(obj_decide_kernel, gk, b, 0, st, comps, ncomp, kcap, (double)H * (double)W, saliency_thr, consensus_thr);
    hipLaunchKernelGGL(obj_cover_kernel, gk, b, 0, st, comps, ncomp, kcap);
    hipLaunchKernelGGL(obj_apply_kernel, gn, b, 0, st, alpha, H, W, lab, cid, comps, kcap);
    return check_launch("remove_invalid_objects kernels");
}
