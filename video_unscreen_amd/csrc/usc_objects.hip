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
// slits, shapes on the frame border, blobs with area around 100): tests/test_gpu_objects.py.  Everything is confined to the
// bounding box of the foreground; the score sums are kept in 90-bit fixed point (two integer limbs, exact whatever the order of
// the atomics), so a run is bit-reproducible and agrees with the host's float64 sums far below any threshold margin.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <math.h>
#include <stdint.h>

#include "../../include/unscreen_hip.h"
#include "usc_device.h"
#include "usc_host.h"

using namespace usc;

namespace {

struct Comp {                      // one connected component (foreground or background), 112 bytes
    int first;                     // raster-first pixel (domain index) = union-find root
    int parent;                    // component id of the enclosing component of the other class (-1: outside ring)
    int is_fg;
    int keep;                      // its border passed the area and score tests
    int covered;                   // lies inside a kept border
    unsigned int chain_cnt;        // hole borders: pixels of the parent that touch this hole
    unsigned long long cnt, sg;    // own pixels, sum of seg
    unsigned long long t_cnt, t_sg;          // subtree totals
    unsigned long long chain_sg;
    // sums of the score map in fixed point, two limbs: score * 2^26 = hi + lo * 2^-32 (hi, lo integers; the lo limb truncated at
    // 2^-58): integer sums are exact whatever the order, so the sums - and the threshold decisions - are the same on every run
    unsigned long long sc_hi, sc_lo, t_sc_hi, t_sc_lo, chain_sc_hi, chain_sc_lo;
    long long area2;               // twice the signed polygon area of its border
};

// The work is restricted to the bounding box of alpha != 0, padded by a one-pixel background ring ("the domain"): outside it
// there is only background that belongs to the outside component.  hdr (ints): [0] number of components, [1..4] the box as
// maxima (H - ymin, W - xmin, ymax + 1, xmax + 1), so that a memset to zero initialises it; [3] == 0: no foreground at all.
struct Dom { int y0, x0, Hb, Wb; };
__device__ __forceinline__ bool load_dom(const int* __restrict__ hdr, int H, int W, Dom& d) {
    const int m2 = hdr[3];
    if (m2 == 0) return false;
    d.y0 = H - hdr[1]; d.x0 = W - hdr[2]; d.Hb = m2 - d.y0; d.Wb = hdr[4] - d.x0;
    return true;
}

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

// domain coordinates: (py, px) in [0, Hb + 2) x [0, Wb + 2), the image pixel is (y0 + py - 1, x0 + px - 1)
__device__ __forceinline__ bool is_fg(const uint8_t* __restrict__ alpha, int W, const Dom& d, int py, int px) {
    return py >= 1 && py <= d.Hb && px >= 1 && px <= d.Wb && alpha[(size_t)(d.y0 + py - 1) * W + (d.x0 + px - 1)] != 0;
}

// maskprocess.py:155-189 get_score_map / build_score_map for one pixel (float64, the operations of the table builder below)
__host__ __device__ __forceinline__ double lin(double a, double b, int n, int i) {
    if (n == 1) return a;
    if (i == n - 1) return b;
    return (double)i * ((b - a) / (double)(n - 1)) + a;
}
__host__ __device__ __forceinline__ double score_raw(int y, int x, int H, int W, int cy, int cx) {
    const double cv = x >= cx ? lin(0, 1, W - cx, x - cx) : lin(1, 0, cx, x);
    const double rv = y >= cy ? lin(0, 1, H - cy, y - cy) : lin(1, 0, cy, y);
    return sqrt(cv * cv + rv * rv);
}
__device__ __forceinline__ void score_fixed(int y, int x, int H, int W, int cy, int cx, double m, unsigned long long& hi, unsigned long long& lo) {
    const double s = (m - score_raw(y, x, H, W, cy, cx)) / m;             // the table's value, in [0, 1]
    const double t = s * 67108864.;                                        // 2^26: exact scaling
    const double f = floor(t);
    hi = (unsigned long long)f;
    lo = (unsigned long long)((t - f) * 4294967296.);                      // exact difference, exact scaling, truncation
}

// A small per-workgroup table in LDS that combines contributions to the same component before they go to global memory:
// atomics on one address serialise in the L2, and a silhouette is hundreds of thousands of pixels of ONE component.
constexpr int kSlots = 32;
struct BlockSums {
    int key[kSlots];
    unsigned long long v[4][kSlots];
};
__device__ __forceinline__ void sums_init(BlockSums& t) {
    for (int i = threadIdx.x; i < kSlots; i += blockDim.x) { t.key[i] = -1; t.v[0][i] = t.v[1][i] = t.v[2][i] = t.v[3][i] = 0ull; }
}
__device__ __forceinline__ int sums_slot(BlockSums& t, int c) {          // -1: table full (the caller goes to global memory)
    int h = c & (kSlots - 1);
    for (int probe = 0; probe < kSlots; probe++) {
        const int old = atomicCAS(&t.key[h], -1, c);
        if (old == -1 || old == c) return h;
        h = (h + 1) & (kSlots - 1);
    }
    return -1;
}

}  // namespace

// Bounding box of alpha != 0 (maxima of H - y, W - x, y + 1, x + 1; hdr zeroed by the caller)
// (`copy_to`: the scan also deposits the mask there - the caller's "filter a copy of src" costs no pass of its own)
__global__ __launch_bounds__(256) void obj_bbox_kernel(const uint8_t* __restrict__ alpha, uint8_t* __restrict__ copy_to, int H, int W,
                                                       int* __restrict__ hdr) {
    const int n = H * W;
    int m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    for (int i = (blockIdx.x * blockDim.x + threadIdx.x) * 16; i < n; i += gridDim.x * blockDim.x * 16) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i + 16 <= n && ((reinterpret_cast<uintptr_t>(alpha) + i) & 15) == 0) v = *reinterpret_cast<const uint4*>(alpha + i);
        else { uint8_t* pv = reinterpret_cast<uint8_t*>(&v); for (int k = 0; k < 16 && i + k < n; k++) pv[k] = alpha[i + k]; }
        if (copy_to) {
            if (i + 16 <= n && ((reinterpret_cast<uintptr_t>(copy_to) + i) & 15) == 0) *reinterpret_cast<uint4*>(copy_to + i) = v;
            else { const uint8_t* pv = reinterpret_cast<const uint8_t*>(&v); for (int k = 0; k < 16 && i + k < n; k++) copy_to[i + k] = pv[k]; }
        }
        if ((v.x | v.y | v.z | v.w) == 0) continue;
        // first and last non-zero byte of the group (one division for the group; a group that wraps to the next row is split)
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
        int kf = 0, kl = 15;
        for (int q = 0; q < 4; q++) if (w4[q]) { kf = 4 * q + (__ffs((int)w4[q]) - 1) / 8; break; }
        for (int q = 3; q >= 0; q--) if (w4[q]) { kl = 4 * q + (31 - __clz((int)w4[q])) / 8; break; }
        const int y = i / W, x = i - y * W;
        if (x + kl < W) {                                    // the common case: the group's non-zero bytes lie in one row
            m0 = max(m0, H - y); m2 = max(m2, y + 1); m1 = max(m1, W - (x + kf)); m3 = max(m3, x + kl + 1);
        } else {
            const uint8_t* pv = reinterpret_cast<const uint8_t*>(&v);
            for (int k = kf; k <= kl; k++) {
                if (!pv[k]) continue;
                const int yy = (i + k) / W, xx = (i + k) - yy * W;
                m0 = max(m0, H - yy); m1 = max(m1, W - xx); m2 = max(m2, yy + 1); m3 = max(m3, xx + 1);
            }
        }
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        m0 = max(m0, __shfl_down(m0, off, kWave)); m1 = max(m1, __shfl_down(m1, off, kWave));
        m2 = max(m2, __shfl_down(m2, off, kWave)); m3 = max(m3, __shfl_down(m3, off, kWave));
    }
    // one set of atomics per workgroup (they all hit the same four words: a 1080p silhouette used to issue ~2000 of them,
    // which serialise in the L2 - 50 us for a 2 MB scan)
    __shared__ int s_m[4][4];
    const int wv = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) { s_m[wv][0] = m0; s_m[wv][1] = m1; s_m[wv][2] = m2; s_m[wv][3] = m3; }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int m = max(max(s_m[0][threadIdx.x], s_m[1][threadIdx.x]), max(s_m[2][threadIdx.x], s_m[3][threadIdx.x]));
        if (m > 0) atomicMax(&hdr[1 + threadIdx.x], m);
    }
}

// Labelling, step 1: one wavefront per domain row walks it in 64-pixel segments; every pixel starts with the index of the first
// pixel of its run (same class, contiguous) - one ballot and a few bit operations per segment, the run that crosses a segment
// boundary carries its start along.  No atomics, and no unions inside a row are left for step 2.
__global__ __launch_bounds__(256) void obj_runs_kernel(const uint8_t* __restrict__ alpha, int H, int W, const int* __restrict__ hdr,
                                                       int* __restrict__ lab) {
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2;
    const int lane = threadIdx.x & (kWave - 1);
    const int py = blockIdx.x * 4 + threadIdx.x / kWave;
    if (py >= d.Hb + 2) return;
    int carry_start = 0;                                     // start of the run the previous segment ended in
    bool carry_c = false;
    for (int x0 = 0; x0 < Wp; x0 += kWave) {
        const int px = x0 + lane;
        const bool live = px < Wp;
        const bool f = live && is_fg(alpha, W, d, py, px);
        const unsigned long long C = __ballot(f);
        unsigned long long S = C ^ (C << 1);                                  // bit l (l > 0): lane l starts a run
        const bool first_starts = x0 == 0 || ((C & 1ull) != 0) != carry_c;
        S = (S & ~1ull) | (first_starts ? 1ull : 0ull);
        const unsigned long long upto = S & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
        const int start = upto ? py * Wp + x0 + (63 - __clzll((long long)upto)) : carry_start;
        if (live) lab[py * Wp + px] = start;
        const int last = min(kWave, Wp - x0) - 1;                             // the segment's last live lane
        carry_start = __shfl(start, last, kWave);
        carry_c = ((C >> last) & 1ull) != 0;
    }
}

// Labelling, step 2: unions only where two runs START to touch - the first vertically adjacent pair of two runs and, for the
// 8-connected foreground, the diagonal contacts at run ends - a few per run instead of a few per pixel.  Roots end up at the
// smallest index of the component = its first pixel in raster order.
// One lane per domain pixel, one wavefront per 64-pixel segment of a row (the geometry of obj_runs_kernel): the classes of the
// segment and of the segment above come from two ballots, every neighbour test is a bit of them; only the lanes at the two
// ends of a segment read a neighbouring segment's pixel.
__global__ __launch_bounds__(256) void obj_merge_kernel(const uint8_t* __restrict__ alpha, int H, int W, const int* __restrict__ hdr,
                                                        int* __restrict__ lab) {
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2;
    const int py = blockIdx.y, px = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & (kWave - 1);
    if (py >= d.Hb + 2 || (int)blockIdx.x * 256 >= Wp) return;
    const bool live = px < Wp;
    const bool c = live && is_fg(alpha, W, d, py, px);
    const bool cn = live && py > 0 && is_fg(alpha, W, d, py - 1, px);
    const unsigned long long C = __ballot(c), CN = __ballot(cn);
    if (!live) return;
    const int i = py * Wp + px;
    // the pixel to the left: the previous lane, or (lane 0) the last pixel of the previous segment
    const bool wc = lane > 0 ? ((C >> (lane - 1)) & 1ull) != 0 : (px > 0 && is_fg(alpha, W, d, py, px - 1));
    const bool w_same = px > 0 && wc == c;
    const bool start_i = !w_same;
    if (py == 0) return;
    if (cn == c) {
        bool start_n = px == 0;
        if (!start_i && !start_n) {
            const bool nwc = lane > 0 ? ((CN >> (lane - 1)) & 1ull) != 0 : is_fg(alpha, W, d, py - 1, px - 1);
            start_n = nwc != cn;
        }
        if (start_i || start_n) unite(lab, i, i - Wp);
    } else if (c) {                                          // foreground below background: diagonal contacts
        if (px > 0 && !w_same) {
            const bool nw = lane > 0 ? ((CN >> (lane - 1)) & 1ull) != 0 : is_fg(alpha, W, d, py - 1, px - 1);
            if (nw) unite(lab, i, i - Wp - 1);
        }
        if (px + 1 < Wp) {
            const bool e = lane < kWave - 1 ? ((C >> (lane + 1)) & 1ull) != 0 : is_fg(alpha, W, d, py, px + 1);
            if (!e) {
                const bool ne = lane < kWave - 1 ? ((CN >> (lane + 1)) & 1ull) != 0 : is_fg(alpha, W, d, py - 1, px + 1);
                if (ne) unite(lab, i, i - Wp + 1);
            }
        }
    }
}

__global__ __launch_bounds__(256) void obj_flatten_kernel(const uint8_t* __restrict__ alpha, int H, int W, int* __restrict__ hdr,
                                                          int* __restrict__ lab, int* __restrict__ cid, Comp* __restrict__ comps, int kcap) {
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2, np = (d.Hb + 2) * Wp;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < np; i += gridDim.x * blockDim.x) {
        const int r = find_root(lab, i);
        lab[i] = r;
        if (r == i) {                                        // a root: give the component a dense id
            const int id = atomicAdd(&hdr[0], 1);
            cid[i] = id;
            if (id < kcap) {
                Comp c = {};
                const int py = i / Wp, px = i - py * Wp;
                c.first = i; c.parent = -1; c.is_fg = is_fg(alpha, W, d, py, px) ? 1 : 0;
                comps[id] = c;
            }
        }
    }
}

// Own sums of every component + parent links.  A workgroup takes 4096 consecutive domain pixels: the component ids are fetched
// with coalesced loads into LDS, every lane then walks 16 consecutive pixels and flushes a run of equal ids into the workgroup's
// LDS table, and the table goes to global memory with one atomic per field and component - ~150 same-address atomics for a
// 1080p silhouette instead of one per 64 pixels.
constexpr int kStatsSpan = 4096;
__global__ __launch_bounds__(256) void obj_stats_kernel(const uint8_t* __restrict__ seg, int H, int W, const int* __restrict__ hdr,
                                                        const int* __restrict__ lab, const int* __restrict__ cid, Comp* __restrict__ comps,
                                                        int kcap, int cy, int cx, double smax) {
    __shared__ int s_c[kStatsSpan];
    __shared__ BlockSums tab;
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2, np = (d.Hb + 2) * Wp;
    const int base = blockIdx.x * kStatsSpan;
    if (base >= np) return;
    const int c_out = cid[0];
    sums_init(tab);
    for (int k = threadIdx.x; k < kStatsSpan; k += 256) {
        const int i = base + k;
        int c = -1;
        if (i < np) {
            const int root = lab[i];
            c = cid[root];
            if (root == i && i > 0 && c < kcap) comps[c].parent = cid[lab[i - 1]];     // the pixel left of the first pixel
            if (c >= kcap || c == c_out) c = -1;             // the outside ring has no border: its sums are never used
        }
        s_c[k] = c;
    }
    __syncthreads();
    constexpr int PPT = kStatsSpan / 256;
    int cur = -1;
    unsigned long long n = 0, g = 0, hi = 0, lo = 0;
    auto flush = [&]() {
        if (cur >= 0 && n) {
            const int sl = sums_slot(tab, cur);
            if (sl >= 0) { atomicAdd(&tab.v[0][sl], n); atomicAdd(&tab.v[1][sl], g); atomicAdd(&tab.v[2][sl], hi); atomicAdd(&tab.v[3][sl], lo); }
            else { atomicAdd(&comps[cur].cnt, n); atomicAdd(&comps[cur].sg, g); atomicAdd(&comps[cur].sc_hi, hi); atomicAdd(&comps[cur].sc_lo, lo); }
        }
        n = g = hi = lo = 0;
    };
    const int i0 = base + threadIdx.x * PPT;
    int py = i0 / Wp, px = i0 - py * Wp;
    for (int k = 0; k < PPT; k++) {
        const int c = s_c[threadIdx.x * PPT + k];
        if (c != cur) { flush(); cur = c; }
        if (c >= 0) {
            n++;
            if (py >= 1 && py <= d.Hb && px >= 1 && px <= d.Wb) {
                const int y = d.y0 + py - 1, x = d.x0 + px - 1;
                unsigned long long h2, l2;
                score_fixed(y, x, H, W, cy, cx, smax, h2, l2);
                hi += h2; lo += l2;
                g += seg[(size_t)y * W + x];
            }
        }
        if (++px == Wp) { px = 0; py++; }
    }
    flush();
    __syncthreads();
    if (threadIdx.x < kSlots && tab.key[threadIdx.x] >= 0) {
        Comp& q = comps[tab.key[threadIdx.x]];
        atomicAdd(&q.cnt, tab.v[0][threadIdx.x]); atomicAdd(&q.sg, tab.v[1][threadIdx.x]);
        atomicAdd(&q.sc_hi, tab.v[2][threadIdx.x]); atomicAdd(&q.sc_lo, tab.v[3][threadIdx.x]);
    }
}

// one term per crack between a foreground pixel and a background pixel; the shoelace terms of one border are combined in the
// workgroup's LDS table first (a silhouette's outer border is thousands of cracks of ONE component)
__global__ __launch_bounds__(256) void obj_cracks_kernel(const uint8_t* __restrict__ alpha, const uint8_t* __restrict__ seg, int H, int W,
                                                         const int* __restrict__ hdr, const int* __restrict__ lab, const int* __restrict__ cid,
                                                         Comp* __restrict__ comps, int kcap, int cy, int cx, double smax) {
    __shared__ BlockSums tab;
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2;
    const int n = d.Hb * d.Wb;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    sums_init(tab);
    __syncthreads();
    const int dy[4] = {-1, 0, 1, 0}, dx[4] = {0, 1, 0, -1};           // N, E, S, W; the walk along side d goes in direction d + 1
    if (i < n) {
        const int yy = i / d.Wb, xx = i - yy * d.Wb, py = yy + 1, px = xx + 1;           // domain-interior coordinates
        const int y = d.y0 + yy, x = d.x0 + xx;                                           // image coordinates
        if (alpha[(size_t)y * W + x]) {
            const int f = cid[lab[py * Wp + px]];
            if (f < kcap) {
                int seen[4];
                int nseen = 0;
#pragma unroll
                for (int dd = 0; dd < 4; dd++) {
                    const int by = py + dy[dd], bx = px + dx[dd];
                    if (is_fg(alpha, W, d, by, bx)) continue;
                    const int b = cid[lab[by * Wp + bx]];
                    if (b >= kcap) continue;
                    const int wy = dy[(dd + 1) & 3], wx = dx[(dd + 1) & 3];
                    int ny, nx;
                    if (is_fg(alpha, W, d, by + wy, bx + wx)) { ny = by + wy; nx = bx + wx; }          // concave corner: diagonal step
                    else if (is_fg(alpha, W, d, py + wy, px + wx)) { ny = py + wy; nx = px + wx; }     // straight
                    else { ny = py; nx = px; }                                                         // around the pixel's own corner
                    // the shoelace sum of a closed chain does not depend on the origin: domain coordinates will do
                    const long long term = (long long)xx * (ny - 1) - (long long)(nx - 1) * yy;
                    const bool hole = comps[b].parent == f;   // B is a hole of F: the hole's border; else B encloses F: F's outer border
                    const int target = hole ? b : f;
                    const int sl = sums_slot(tab, target);
                    if (sl >= 0) atomicAdd(&tab.v[0][sl], (unsigned long long)term);
                    else atomicAdd((unsigned long long*)&comps[target].area2, (unsigned long long)term);
                    if (hole) {
                        bool dup = false;
                        for (int k = 0; k < nseen; k++) dup |= seen[k] == b;
                        if (!dup) {
                            seen[nseen++] = b;
                            unsigned long long h2, l2;
                            score_fixed(y, x, H, W, cy, cx, smax, h2, l2);
                            atomicAdd(&comps[b].chain_cnt, 1u);
                            atomicAdd(&comps[b].chain_sg, (unsigned long long)seg[(size_t)y * W + x]);
                            atomicAdd(&comps[b].chain_sc_hi, h2);
                            atomicAdd(&comps[b].chain_sc_lo, l2);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kSlots && tab.key[threadIdx.x] >= 0 && tab.v[0][threadIdx.x] != 0ull)
        atomicAdd((unsigned long long*)&comps[tab.key[threadIdx.x]].area2, tab.v[0][threadIdx.x]);
}

// Subtree totals (every component adds its own sums to all its ancestors), then maskprocess.py:121-147 (area test, saliency /
// consensus test), then "is a kept component among my ancestors": three passes over the component table with a dependency
// between them - ONE workgroup walks them with barriers in between (a clean mask has a handful of components; one launch
// instead of three, and a single CU busy for a few microseconds, matters more to the frame loop than the width of the pass).
__global__ __launch_bounds__(1024) void obj_tree_decide_cover_kernel(Comp* __restrict__ comps, const int* __restrict__ hdr, int kcap, double denom,
                                                                     double saliency_thr, double consensus_thr) {
    const int K = min(hdr[0], kcap);
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        const unsigned long long n = comps[c].cnt, g = comps[c].sg, sh = comps[c].sc_hi, sl = comps[c].sc_lo;
        atomicAdd(&comps[c].t_cnt, n); atomicAdd(&comps[c].t_sg, g); atomicAdd(&comps[c].t_sc_hi, sh); atomicAdd(&comps[c].t_sc_lo, sl);
        int a = comps[c].parent;
        for (int depth = 0; a >= 0 && comps[a].parent >= 0 && depth < (1 << 20); depth++) {      // stop below the outside ring
            atomicAdd(&comps[a].t_cnt, n); atomicAdd(&comps[a].t_sg, g); atomicAdd(&comps[a].t_sc_hi, sh); atomicAdd(&comps[a].t_sc_lo, sl);
            a = comps[a].parent;
        }
    }
    __threadfence();
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        Comp& q = comps[c];
        int keep = 0;
        if (q.parent >= 0) {                                  // the outside ring has no border
            const long long a2 = q.area2 < 0 ? -q.area2 : q.area2;
            if ((double)a2 * 0.5 >= 100.) {
                unsigned long long sh = __hip_atomic_load(&q.t_sc_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), sl = __hip_atomic_load(&q.t_sc_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                double gsum = (double)__hip_atomic_load(&q.t_sg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), cnt = (double)__hip_atomic_load(&q.t_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (!q.is_fg) { sh += q.chain_sc_hi; sl += q.chain_sc_lo; gsum += (double)q.chain_sg; cnt += (double)q.chain_cnt; }
                const double ssum = ((double)sh + (double)sl * (1. / 4294967296.)) * (1. / 67108864.);
                const double sal = ssum / denom, cons = (gsum / cnt) / 255.;
                keep = ((sal > saliency_thr && cons > consensus_thr) || sal > saliency_thr * 10) ? 1 : 0;
            }
        }
        __hip_atomic_store(&q.keep, keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __threadfence();
    __syncthreads();
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        int cov = 0, a = c;
        for (int depth = 0; a >= 0 && depth < (1 << 20) && !cov; depth++) { cov = __hip_atomic_load(&comps[a].keep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); a = comps[a].parent; }
        comps[c].covered = cov;
    }
}

__global__ __launch_bounds__(256) void obj_apply_kernel(uint8_t* __restrict__ alpha, int H, int W, const int* __restrict__ hdr,
                                                        const int* __restrict__ lab, const int* __restrict__ cid, const Comp* __restrict__ comps,
                                                        int kcap) {
    Dom d;
    if (!load_dom(hdr, H, W, d)) return;
    const int Wp = d.Wb + 2;
    const int n = d.Hb * d.Wb;
    const int dy[4] = {-1, 0, 1, 0}, dx[4] = {0, 1, 0, -1};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int yy = i / d.Wb, xx = i - yy * d.Wb, py = yy + 1, px = xx + 1;
        const size_t o = (size_t)(d.y0 + yy) * W + (d.x0 + xx);
        if (!alpha[o]) continue;
        const int f = cid[lab[py * Wp + px]];
        bool valid = f < kcap && comps[f].covered;
#pragma unroll
        for (int dd = 0; dd < 4 && !valid; dd++) {
            const int by = py + dy[dd], bx = px + dx[dd];
            if (is_fg(alpha, W, d, by, bx)) continue;         // (neighbours are only ever cleared to 0 if they were foreground: see below)
            const int b = cid[lab[by * Wp + bx]];
            valid = b < kcap && comps[b].keep && comps[b].parent == f;
        }
        if (!valid) alpha[o] = 0;
    }
}

// The apply kernel reads neighbours' alpha while others may already have been cleared: is_fg() of a neighbour could flip from
// foreground to background mid-kernel and its label (a foreground label) would then be read as if it were a hole.  The test
// `comps[b].parent == f` with b a foreground component's id can never hold (a foreground component's parent is a background
// component), so a cleared neighbour is never mistaken for a kept hole.

// maskprocess.py:155-189 get_score_map / build_score_map as a table on the device (float64, same operations as the host builder)
__global__ __launch_bounds__(256) void score_raw_kernel(int H, int W, int y0, int x0, double* __restrict__ out, unsigned long long* __restrict__ mx) {
    const int n = H * W;
    double m = 0.;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / W, x = i - y * W;
        const double s = score_raw(y, x, H, W, y0, x0);
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

extern "C" int usc_remove_invalid_objects_u8(const uint8_t* src, uint8_t* alpha, const uint8_t* seg, int H, int W, double center_y, double center_x,
                                             double saliency_thr, double consensus_thr, void* ws, size_t ws_bytes, usc_stream_t stream) {
    if (src == alpha) src = nullptr;
    if (!alpha || !ws || H <= 0 || W <= 0) return fail(USC_E_ARG, "usc_remove_invalid_objects_u8: bad argument");
    if ((long long)(H + 2) * (W + 2) >= (1ll << 30)) return fail(USC_E_UNSUPPORTED, "usc_remove_invalid_objects_u8: image too large");
    if (ws_bytes < usc_remove_invalid_objects_ws_bytes(H, W)) return fail(USC_E_WORKSPACE, "usc_remove_invalid_objects_u8: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int np = (H + 2) * (W + 2), kcap = comp_cap(H, W);
    uint8_t* base = reinterpret_cast<uint8_t*>(ws);
    int* hdr = reinterpret_cast<int*>(base);
    int* lab = reinterpret_cast<int*>(base + 256);
    int* cid = reinterpret_cast<int*>(base + 256 + al256((size_t)np * sizeof(int)));
    Comp* comps = reinterpret_cast<Comp*>(base + 256 + 2 * al256((size_t)np * sizeof(int)));
    if (!seg) seg = src ? src : alpha;                      // consensus against the input itself: alpha is only written by the last kernel
    // the score map is evaluated per pixel where it is needed (closed form, maskprocess.py:155-178); its normalising maximum sits
    // at a corner of the image and is computed here with the same float64 operations
    const int cy = (int)(H * center_y), cx = (int)(W * center_x);
    double smax = 0.;
    for (int k = 0; k < 4; k++) smax = fmax(smax, score_raw((k & 1) ? H - 1 : 0, (k & 2) ? W - 1 : 0, H, W, cy, cx));
    if (hipMemsetAsync(hdr, 0, 256, st) != hipSuccess) return fail(USC_E_LAUNCH, "usc_remove_invalid_objects_u8: memset failed");
    const dim3 b(256), gp(grid_for((size_t)np, 256)), gn(grid_for((size_t)H * W, 256));
    hipLaunchKernelGGL(obj_bbox_kernel, dim3((unsigned)std::min<size_t>(128, ((size_t)(H * W + 15) / 16 + 255) / 256)), b, 0, st, src ? src : alpha, src ? alpha : (uint8_t*)nullptr, H, W, hdr);
    hipLaunchKernelGGL(obj_runs_kernel, dim3((H + 2 + 3) / 4), b, 0, st, alpha, H, W, hdr, lab);
    hipLaunchKernelGGL(obj_merge_kernel, dim3((W + 2 + 255) / 256, H + 2), b, 0, st, alpha, H, W, hdr, lab);
    hipLaunchKernelGGL(obj_flatten_kernel, gp, b, 0, st, alpha, H, W, hdr, lab, cid, comps, kcap);
    hipLaunchKernelGGL(obj_stats_kernel, dim3((np + kStatsSpan - 1) / kStatsSpan), b, 0, st, seg, H, W, hdr, lab, cid, comps, kcap, cy, cx, smax);
    hipLaunchKernelGGL(obj_cracks_kernel, dim3((H * W + 255) / 256), b, 0, st, alpha, seg, H, W, hdr, lab, cid, comps, kcap, cy, cx, smax);
    hipLaunchKernelGGL(obj_tree_decide_cover_kernel, dim3(1), dim3(1024), 0, st, comps, hdr, kcap, (double)H * (double)W, saliency_thr, consensus_thr);
    hipLaunchKernelGGL(obj_apply_kernel, gn, b, 0, st, alpha, H, W, hdr, lab, cid, comps, kcap);
    return check_launch("remove_invalid_objects kernels");
}
