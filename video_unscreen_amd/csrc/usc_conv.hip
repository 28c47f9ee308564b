// usc_conv.hip - fused NHWC float32 convolution on the f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// The matting network (reference unscreen/vmatting/model.py:435-568) is ~50 small convolutions on a
// 544 x 960 frame: dense contractions with M = output pixels (510 ... 522,240), N = Cout (32 ... 512),
// K = taps * Cin (72 ... 4608).  One implicit-GEMM kernel serves every shape:
//   * taps are a list of (dy, dx) offsets with an input stride, so 3x3 s1/s2, 1x1, the 2x2-s2 form of
//     avg-pool + 1x1, and the four 2x2 output phases of ConvTranspose2d(k4, s2, p1) are the same code;
//   * a workgroup of 4 waves owns 128 output pixels x (32 | 64) channels; every K step stages a
//     [128 px][32 k] slice of the im2col matrix and a [32 k][N] slice of the packed weights in LDS
//     (double buffered, one barrier per step) and each wave issues 16 x NT back-to-back MFMAs on its 32 rows;
//   * LDS images are padded / packed so every ds_read_b128 of a fragment is conflict-free
//     (A: 36-float row pitch; B: [k/4][n][4] so consecutive lanes read consecutive 16-byte words);
//   * the epilogue fuses bias, residual (optionally from a half-resolution tensor = nearest x2 upsample),
//     ReLU / LeakyReLU(0.2), a per-channel affine (the shortcut branches' BN-after-ReLU) and a second residual,
//     so no elementwise pass ever touches HBM again.
// f32 MFMA is an exact k-ordered fmaf chain; accumulation order is fixed, so results are run-to-run identical
// (the MIOpen kernels this replaces accumulate with atomics).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/unscreen_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BM = 128;        // output pixels per workgroup
constexpr int KC = 32;         // K slice per step
constexpr int APITCH = 36;     // floats per A row in LDS (32 + 4 pad: b128 reads hit 64 distinct banks)
constexpr int THREADS = 256;

template <int NT, bool IN_RELU>
__global__ __launch_bounds__(THREADS) void conv_mfma_kernel(usc_conv_desc d) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * APITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][8 * 32 * NT * 4];
    __shared__ int s_toff[16];
    __shared__ int s_outpix[BM], s_res1pix[BM], s_vbase[BM], s_vmask[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int M = d.OH * d.OW;
    // XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so with the natural
    // order the N tiles of one pixel tile, and the pixel tiles one image row apart that share its 3x3 halo, all sit on
    // different L2s and every one of them fetches the same activations from the fabric again.  The 1-D grid is re-labelled
    // (bijectively, any size) so that XCD k owns the k-th contiguous run of (pixel tile, N tile) pairs, N tile fastest.
    const int nblk = d.Cout / (32 * NT);
    const int nwg = (int)gridDim.x, xcd = (int)blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    const int mtile = wgid / nblk;
    const int m0 = mtile * BM;
    const int nb0 = (wgid - mtile * nblk) * (32 * NT);
    const int Ktot = d.ntaps * d.Cin;
    // blockIdx.z = phase * ksplit + k-slice.  Phases: the four (py, px) outputs of a transposed conv, each with
    // its own tap table and weight block.  K slices: contiguous chunk ranges whose partial sums go to a slab.
    const int S = d.ksplit;
    const int ph = blockIdx.z / S, ks = blockIdx.z - ph * S;
    const int nch_all = (Ktot + KC - 1) / KC;
    const int c_begin = (int)((long long)ks * nch_all / S), c_end = (int)((long long)(ks + 1) * nch_all / S);
    const float* __restrict__ wpk = d.wpk + (size_t)ph * d.wphase_stride;
    const int out_py = d.nphase > 1 ? (ph >> 1) : d.out_py, out_px = d.nphase > 1 ? (ph & 1) : d.out_px;


    // ---- per-tile tables, one output pixel per thread (tid < 128), everything that does not change from K step
    // to K step: byte offset of the pixel's row in the output / residual tensors, byte offset of its (0,0) tap in
    // the input, and a 16-bit mask of the taps that fall inside the image.  Tap offsets are read from the kernel
    // arguments (scalar registers), so the only long-latency operation here is one integer division.
    // Invalid rows / taps get offsets beyond their buffer: buffer loads then return 0 and buffer stores are dropped
    // by the hardware range check - no branches anywhere downstream.
    constexpr int INVALID = 0x7fff0000;
    uint32_t dyw[4], dxw[4];
    __builtin_memcpy(dyw, d.dy, 16);
    __builtin_memcpy(dxw, d.dx, 16);
    if (d.nphase > 1) {                                          // 4 phases x 4 taps: phase p owns bytes [4p, 4p+4)
        dyw[0] = ph == 0 ? dyw[0] : ph == 1 ? dyw[1] : ph == 2 ? dyw[2] : dyw[3];
        dxw[0] = ph == 0 ? dxw[0] : ph == 1 ? dxw[1] : ph == 2 ? dxw[2] : dxw[3];
    }
    if (tid < 16) {
        int dy = 0, dx = 0;
#pragma unroll
        for (int t = 0; t < 16; t++)
            if (t == tid) { dy = (int)(int8_t)(dyw[t >> 2] >> (8 * (t & 3))); dx = (int)(int8_t)(dxw[t >> 2] >> (8 * (t & 3))); }
        s_toff[tid] = (dy * d.IW + dx) * d.Cin * 4;
    }
    if (tid < BM) {
        const int m = m0 + tid;
        int opix = -1, rpb = INVALID, vb = 0;
        unsigned mk = 0;
        if (m < M) {
            const int oy = m / d.OW, ox = m - oy * d.OW;
            const int Y = oy * d.out_stride + out_py, X = ox * d.out_stride + out_px;
            opix = Y * d.OWf + X;
            rpb = ((Y >> d.res1_shift) * (d.OWf >> d.res1_shift) + (X >> d.res1_shift)) * d.Cout * 4;
            // batch: images are stacked along y on both sides (OH, OHf count all of them); only the input window is per image
            int img = 0, oyl = oy;
            if (d.batch > 1) { const int ohi = d.OH / d.batch; img = oy / ohi; oyl = oy - img * ohi; }
            const int iy0 = oyl * d.in_stride, ix0 = ox * d.in_stride;
            vb = ((img * d.IH + iy0) * d.IW + ix0) * d.Cin * 4;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                if (t < d.ntaps) {
                    const int iy = iy0 + (int)(int8_t)(dyw[t >> 2] >> (8 * (t & 3)));
                    const int ix = ix0 + (int)(int8_t)(dxw[t >> 2] >> (8 * (t & 3)));
                    mk |= (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) ? (1u << t) : 0u;
                }
            }
        }
        s_outpix[tid] = opix;
        s_res1pix[tid] = rpb;
        s_vbase[tid] = vb;
        s_vmask[tid] = (int)mk;
    }
    __syncthreads();

    // ---- staging assignment: thread -> (4 pixels 32 apart, one float4 of the 32-wide K slice)
    const int q = tid & 7;
    const bool cin_pow2 = (d.Cin & (d.Cin - 1)) == 0;
    const int log2cin = 31 - __builtin_clz(d.Cin);
    int vbase[4];
    unsigned vmask[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        vbase[j] = s_vbase[(tid >> 3) + 32 * j];
        vmask[j] = (unsigned)s_vmask[(tid >> 3) + 32 * j];
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.in), 0, (d.batch > 1 ? d.batch : 1) * d.IH * d.IW * d.Cin * 4, 0x00020000);
    const int wbytes = nch_all * 8 * d.Cout * 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, wbytes, 0x00020000);
    const int bkq = tid >> 5, bn = tid & 31;
    const int wbase = (bkq * d.Cout + nb0 + bn) * 16;          // bytes; + 32*16 per extra N tile, + 8*Cout*16 per chunk
    const int wstep = 8 * d.Cout * 16;

    // Two register sets: chunk c+2 is requested while chunk c is multiplied, so a global load has two MFMA
    // phases (~2 x 1024 cycles) to land before its ds_write.
    float4 ra0[4], rb0[NT], ra1[4], rb1[NT];
    // Loads are unconditional: a chunk index past this block's K range (c >= c_end) gets out-of-range offsets and
    // comes back as zeros, so the main loop has no branches and hipcc can count vmcnt exactly.
    auto load_chunk = [&](int c, float4 (&ra)[4], float4 (&rb)[NT]) {
        const bool live = c < c_end;
        const int k0 = c * KC + 4 * q;
        // tap of this thread's float4: a shift when Cin is a power of two (a 32-wide K slice may then span several
        // taps, Cin = 4 / 8 / 16); otherwise Cin is a multiple of 32 and the whole slice lies in one tap
        const int tap = cin_pow2 ? (k0 >> log2cin) : (c * KC) / d.Cin;      // >= ntaps in the zero-padded K tail: mask bit is 0
        const int koff = s_toff[tap & 15] + ((k0 - tap * d.Cin) << 2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = live && ((vmask[j] >> tap) & 1u);
            const int off = ok ? vbase[j] + koff : 0x7ffffff0;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, off, 0, 0);
            ra[j] = *reinterpret_cast<const float4*>(&v);
        }
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, live ? wbase + c * wstep + j * 512 : 0x7ffffff0, 0, 0);
            rb[j] = *reinterpret_cast<const float4*>(&v);
        }
    };
    auto store_chunk = [&](int buf, const float4 (&ra)[4], const float4 (&rb)[NT]) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float4 a = ra[j];
            if (IN_RELU) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }   // conv(relu(x))
            *reinterpret_cast<float4*>(&As[buf][((tid >> 3) + 32 * j) * APITCH + 4 * q]) = a;
        }
#pragma unroll
        for (int j = 0; j < NT; j++)
            *reinterpret_cast<float4*>(&Bs[buf][(bkq * (32 * NT) + bn + 32 * j) * 4]) = rb[j];
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    const int arow = (32 * wave + (lane & 31)) * APITCH + 4 * (lane >> 5);
    auto compute = [&](int buf) {
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const float4 a = *reinterpret_cast<const float4*>(&As[buf][arow + g * 8]);
#pragma unroll
            for (int j = 0; j < NT; j++) {
                const float4 b = *reinterpret_cast<const float4*>(&Bs[buf][((g * 2 + (lane >> 5)) * (32 * NT) + j * 32 + (lane & 31)) * 4]);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
            }
        }
    };

    load_chunk(c_begin, ra0, rb0);
    load_chunk(c_begin + 1, ra1, rb1);
    store_chunk(0, ra0, rb0);
    __syncthreads();
    int c = c_begin;
    for (; c + 1 < c_end; c += 2) {
        load_chunk(c + 2, ra0, rb0);
        compute(0);
        store_chunk(1, ra1, rb1);
        __syncthreads();
        load_chunk(c + 3, ra1, rb1);
        compute(1);
        store_chunk(0, ra0, rb0);
        __syncthreads();
    }
    if (c < c_end) compute(0);          // odd chunk count: the last chunk is already staged in buffer 0

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): a lane
    // holds ONE channel of 16 rows, which would mean 16 scattered dword stores per tile.  Each wave therefore bounces its
    // 32 x 32 tile through its own rows of LDS buffer 1 (free since the last barrier; only this wave touches them) and
    // re-reads it row-major, 4 channels per lane: every global access of the epilogue is then 16 bytes per lane and a
    // wave instruction covers eight 128-byte rows.  Offsets are 32-bit and go through buffer instructions: rows past the
    // end of the image carry an out-of-range offset and are dropped (stores) / read as zero (residuals).
    float* const T = &As[1][(32 * wave) * APITCH];
    const int out_ld = d.out_ld > 0 ? d.out_ld : d.Cout;
    const int out_bytes = d.OHf * d.OWf * d.Cout * 4;
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;            // this lane's row (per pass of 8 rows) and first channel
    int ops[4], oos[4], rps[4];                                   // dense [pixel][Cout] offset, output offset, residual offset
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
        const int p = 32 * wave + ps * 8 + erow;
        const int opix = s_outpix[p];
        ops[ps] = opix < 0 ? INVALID : opix * d.Cout * 4;
        oos[ps] = opix < 0 ? INVALID : (opix * out_ld + d.out_c0) * 4;
        rps[ps] = s_res1pix[p];
    }
    const bool has_r1 = d.res1 != nullptr, has_r2 = d.res2 != nullptr, has_aff = d.scale != nullptr;
    const __amdgpu_buffer_rsrc_t out_rsrc = S > 1
        ? __builtin_amdgcn_make_buffer_rsrc(d.slab + (size_t)ks * d.OHf * d.OWf * d.Cout, 0, out_bytes, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.OHf * d.OWf * out_ld * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.res1), 0, (has_r1 && S == 1) ? (d.OHf >> d.res1_shift) * (d.OWf >> d.res1_shift) * d.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.res2), 0, (has_r2 && S == 1) ? out_bytes : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; j++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const float a = acc[j][r];
            T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * APITCH + (lane & 31)] = a;
        }
        const int n = nb0 + j * 32 + ec4;
        const int nb4 = n * 4;
        if (S > 1) {                 // partial sums only; splitk_reduce_kernel finishes the layer
#pragma unroll
            for (int ps = 0; ps < 4; ps++) {
                const float4 v = *reinterpret_cast<const float4*>(&T[(ps * 8 + erow) * APITCH + ec4]);
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&v), out_rsrc, ops[ps] + nb4, 0, 0);
            }
            continue;
        }
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bias;
        if (d.bias) bias = *reinterpret_cast<const float4*>(d.bias + n);
        if (has_aff) { sc = *reinterpret_cast<const float4*>(d.scale + n); sh = *reinterpret_cast<const float4*>(d.shift + n); }
        // residual loads of all four passes are issued before their first use (one wait); a zero-sized descriptor
        // (no residual) turns them into zeros without touching memory
        u32x4 r1v[4], r2v[4];
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            r1v[ps] = __builtin_amdgcn_raw_buffer_load_b128(r1_rsrc, rps[ps] + nb4, 0, 0);
            r2v[ps] = __builtin_amdgcn_raw_buffer_load_b128(r2_rsrc, ops[ps] + nb4, 0, 0);
        }
#pragma unroll
        for (int ps = 0; ps < 4; ps++) {
            const float4 t = *reinterpret_cast<const float4*>(&T[(ps * 8 + erow) * APITCH + ec4]);
            const float4 r1 = *reinterpret_cast<const float4*>(&r1v[ps]), r2 = *reinterpret_cast<const float4*>(&r2v[ps]);
            float v[4] = {t.x + bias.x, t.y + bias.y, t.z + bias.z, t.w + bias.w};
            const float a1[4] = {r1.x, r1.y, r1.z, r1.w}, a2[4] = {r2.x, r2.y, r2.z, r2.w};
            const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float x = v[e] + a1[e];
                if (d.act == USC_ACT_RELU) x = fmaxf(x, 0.f);
                else if (d.act == USC_ACT_LEAKY) x = x > 0.f ? x : x * 0.2f;
                else if (d.act == USC_ACT_LEAKY01) x = x > 0.f ? x : x * 0.01f;
                if (has_aff) x = x * s4[e] + h4[e];
                v[e] = x + a2[e];
            }
            const float4 o = make_float4(v[0], v[1], v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&o), out_rsrc, oos[ps] + nb4, 0, 0);
        }
    }
}

// Second half of a split-K layer: sum the K slices in a fixed order (deterministic), then the fused epilogue.
// One lane per (output pixel, 4 channels).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(usc_conv_desc d) {
    const int npix = d.OHf * d.OWf, c4 = d.Cout / 4;
    const size_t plane = (size_t)npix * d.Cout;
    const bool has_r1 = d.res1 != nullptr, has_r2 = d.res2 != nullptr, has_aff = d.scale != nullptr;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < npix * c4; i += gridDim.x * blockDim.x) {
        const int op = i / c4, n = (i - op * c4) * 4;
        const size_t o = (size_t)op * d.Cout + n;
        float4 v = *reinterpret_cast<const float4*>(d.slab + o);
        for (int s = 1; s < d.ksplit; s++) {
            const float4 t = *reinterpret_cast<const float4*>(d.slab + s * plane + o);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float a[4] = {v.x, v.y, v.z, v.w};
        const int Y = op / d.OWf, X = op - Y * d.OWf;
        const size_t ro = ((size_t)(Y >> d.res1_shift) * (d.OWf >> d.res1_shift) + (X >> d.res1_shift)) * d.Cout + n;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            float x = a[k] + (d.bias ? d.bias[n + k] : 0.f);
            if (has_r1) x += d.res1[ro + k];
            if (d.act == USC_ACT_RELU) x = fmaxf(x, 0.f);
            else if (d.act == USC_ACT_LEAKY) x = x > 0.f ? x : x * 0.2f;
            else if (d.act == USC_ACT_LEAKY01) x = x > 0.f ? x : x * 0.01f;
            if (has_aff) x = x * d.scale[n + k] + d.shift[n + k];
            if (has_r2) x += d.res2[o + k];
            a[k] = x;
        }
        const int out_ld = d.out_ld > 0 ? d.out_ld : d.Cout;
        *reinterpret_cast<float4*>(d.out + (size_t)op * out_ld + d.out_c0 + n) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// Output head: Conv2d(32 -> 1, 3x3, pad 1, bias) followed by (tanh(x) + 1) / 2  (model.py:389, 566).
// N = 1 is no matrix product: memory-bound.  A workgroup stages an (8+2) x (32+2) pixel patch of the 32-channel
// input in LDS with coalesced 16-byte loads (pixel pitch 36 floats -> conflict-free b128 reads), then each lane
// owns one output pixel: 9 taps x 8 float4 reads against the 288 weights held in LDS.
constexpr int HT_H = 8, HT_W = 32, HPITCH = 36;
__global__ __launch_bounds__(256) void head_conv_kernel(usc_conv_desc d) {
    __shared__ __attribute__((aligned(16))) float patch[(HT_H + 2) * (HT_W + 2) * HPITCH];
    __shared__ __attribute__((aligned(16))) float w[9 * 32];
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * 32; i += 256) w[i] = d.wpk[i];
    const int y0 = blockIdx.y * HT_H, x0 = blockIdx.x * HT_W;
    // (10 x 34) pixels x 8 float4 each
    for (int i = tid; i < (HT_H + 2) * (HT_W + 2) * 8; i += 256) {
        const int px = i >> 3, q = i & 7;
        const int py = px / (HT_W + 2), pxx = px - py * (HT_W + 2);
        const int gy = y0 - 1 + py, gx = x0 - 1 + pxx;
        const bool ok = gy >= 0 && gy < d.IH && gx >= 0 && gx < d.IW;
        const int cy = min(max(gy, 0), d.IH - 1), cx = min(max(gx, 0), d.IW - 1);
        float4 v = *reinterpret_cast<const float4*>(d.in + ((size_t)(cy * d.IW + cx) * 32 + 4 * q));
        if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&patch[px * HPITCH + 4 * q]) = v;
    }
    __syncthreads();
    const int ty = tid >> 5, tx = tid & 31;
    const int gy = y0 + ty, gx = x0 + tx;
    float acc = 0.f;
#pragma unroll 1
    for (int t = 0; t < 9; t++) {
        const int ky = t / 3, kx = t - ky * 3;
        const float* src = &patch[((ty + ky) * (HT_W + 2) + tx + kx) * HPITCH];
        const float* wt = &w[t * 32];
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const float4 v = *reinterpret_cast<const float4*>(src + 4 * q);
            const float4 ww = *reinterpret_cast<const float4*>(wt + 4 * q);
            acc += v.x * ww.x;
            acc += v.y * ww.y;
            acc += v.z * ww.z;
            acc += v.w * ww.w;
        }
    }
    if (gy < d.OH && gx < d.OW) {
        acc += d.bias ? d.bias[0] : 0.f;
        d.out[(size_t)gy * d.OW + gx] = (tanhf(acc) + 1.0f) / 2.0f;
    }
}

thread_local char g_conv_err[256] = "";

}  // namespace

extern "C" const char* usc_conv_last_error(void) { return g_conv_err; }

static int conv_fail(int code, const char* msg) {
    snprintf(g_conv_err, sizeof(g_conv_err), "%s", msg);
    return code;
}

// One layer = one main launch (+ one reduce launch when K is split).
struct LayerPlan {
    usc_conv_desc d;
    void* fn;
    dim3 grid, block;
    bool reduce;
    dim3 rgrid;
};

static int plan_layer(const usc_conv_desc& d, LayerPlan& p) {
    if (!d.in || !d.out || !d.wpk || d.Cin <= 0 || d.Cin % 4 || d.ntaps < 1 || d.ntaps > 16 || d.OH <= 0 || d.OW <= 0)
        return conv_fail(USC_E_ARG, "usc_conv: malformed layer descriptor");
    const long long M = (long long)d.OH * d.OW;
    p.d = d;
    p.reduce = false;
    p.block = dim3(THREADS);
    if (d.kind == USC_CONV_HEAD) {
        if (d.Cout != 1 || d.Cin != 32 || d.ntaps != 9 || d.in_stride != 1 || d.IH != d.OH || d.IW != d.OW)
            return conv_fail(USC_E_UNSUPPORTED, "usc_conv: the head kernel is Conv2d(32, 1, 3, padding=1)");
        if (d.in_relu) return conv_fail(USC_E_UNSUPPORTED, "usc_conv: in_relu is not offered by the head kernel");
        p.fn = (void*)head_conv_kernel;
        p.grid = dim3((d.OW + HT_W - 1) / HT_W, (d.OH + HT_H - 1) / HT_H);
        return USC_OK;
    }
    if (d.Cout % 32) return conv_fail(USC_E_UNSUPPORTED, "usc_conv: Cout must be a multiple of 32");
    if ((d.Cin & (d.Cin - 1)) && d.Cin % 32) return conv_fail(USC_E_UNSUPPORTED, "usc_conv: Cin must be a power of two or a multiple of 32");
    if (d.out_ld && (d.out_ld < d.Cout + d.out_c0 || d.out_c0 < 0 || d.out_ld % 4 || d.out_c0 % 4))
        return conv_fail(USC_E_ARG, "usc_conv: out_ld / out_c0 must satisfy out_c0 + Cout <= out_ld, both multiples of 4");
    const int nbatch = d.batch > 1 ? d.batch : 1;
    if (nbatch > 1 && (d.OH % nbatch || d.nphase > 1 || d.kind == USC_CONV_HEAD))
        return conv_fail(USC_E_ARG, "usc_conv: batch needs OH = batch * per-image rows and a plain (single-phase) layer");
    if ((long long)nbatch * d.IH * d.IW * d.Cin * 4 >= 0x7fff0000ll || (long long)d.OHf * d.OWf * (d.out_ld > d.Cout ? d.out_ld : d.Cout) * 4 >= 0x7fff0000ll)
        return conv_fail(USC_E_UNSUPPORTED, "usc_conv: tensor >= 2 GiB");
    const int nphase = d.nphase > 1 ? d.nphase : 1;
    if (nphase != 1 && nphase != 4) return conv_fail(USC_E_ARG, "usc_conv: nphase must be 1 or 4");
    if (nphase * d.ntaps > 16) return conv_fail(USC_E_ARG, "usc_conv: more than 16 taps");
    const int mblocks = (int)((M + BM - 1) / BM);
    // two 32-column tiles per workgroup only while that still leaves >= 2 workgroups per CU
    // (tile shape and K split are decided per image, so that a batched layer sums K in exactly the order the single-image
    // layer does)
    const long long mblocks_img = (M / nbatch + BM - 1) / BM;
    const bool nt2 = (d.Cout % 64 == 0) && (mblocks_img * (d.Cout / 64) * nphase >= 512);
    const int nblocks = nt2 ? d.Cout / 64 : d.Cout / 32;
    // split K when the grid would leave most CUs idle (M = 510 ... 2040 layers)
    const long long blocks = mblocks_img * nblocks * nphase;
    const int nch = (d.ntaps * d.Cin + KC - 1) / KC;
    int S = 1;
    static const int split_below = getenv("USC_SPLIT_BELOW") ? atoi(getenv("USC_SPLIT_BELOW")) : 384;      // tuning knobs
    static const int split_target = getenv("USC_SPLIT_TARGET") ? atoi(getenv("USC_SPLIT_TARGET")) : 768;
    if (d.slab && blocks < split_below) {
        S = (int)((split_target + blocks - 1) / blocks);
        if (S > nch / 6) S = nch / 6;          // a slice needs >= 6 K steps to amortise its prologue / epilogue / reduce
        if (S > 16) S = 16;
        const size_t need = (size_t)S * d.OHf * d.OWf * d.Cout * sizeof(float);
        if (need > d.slab_bytes) S = (int)(d.slab_bytes / ((size_t)d.OHf * d.OWf * d.Cout * sizeof(float)));
        if (S < 1) S = 1;
    }
    p.d.ksplit = S;
    p.d.nphase = nphase;
    p.grid = dim3(mblocks * nblocks, 1, nphase * S);          // 1-D over (pixel tile, N tile): the kernel re-labels it per XCD
    if (d.in_relu)
        p.fn = nt2 ? (void*)conv_mfma_kernel<2, true> : (void*)conv_mfma_kernel<1, true>;
    else
        p.fn = nt2 ? (void*)conv_mfma_kernel<2, false> : (void*)conv_mfma_kernel<1, false>;
    if (S > 1) {
        const long long items = (long long)d.OHf * d.OWf * (d.Cout / 4);
        int rb = (int)((items + 255) / 256);
        if (rb > 2048) rb = 2048;
        p.reduce = true;
        p.rgrid = dim3(rb);
    }
    return USC_OK;
}

extern "C" size_t usc_conv_desc_bytes(void) { return sizeof(usc_conv_desc); }

extern "C" int usc_conv_sequence_f32(const usc_conv_desc* descs, int n, usc_stream_t stream) {
    if (!descs || n < 0) return conv_fail(USC_E_ARG, "usc_conv_sequence_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; i++) {
        LayerPlan p;
        int rc = plan_layer(descs[i], p);
        if (rc) return rc;
        void* args[] = {&p.d};
        hipError_t e = hipLaunchKernel(p.fn, p.grid, p.block, args, 0, st);
        if (e == hipSuccess && p.reduce) e = hipLaunchKernel((void*)splitk_reduce_kernel, p.rgrid, dim3(256), args, 0, st);
        if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
    }
    return USC_OK;
}

// ---- the same layers as a hipGraph: one node per launch, edges from the producers of in / res1 / res2 (and from the
// previous user of a shared split-K slab), so independent branches (shortcut convs, the 1x1 up-branch) overlap the
// ramp / tail of the main chain, and one hipGraphLaunch replaces ~85 kernel launches per frame.
struct ConvGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<LayerPlan> plans;          // kernel arguments must outlive the graph
};

extern "C" int usc_conv_graph_create(const usc_conv_desc* descs, int n, void** handle) {
    if (!descs || n <= 0 || !handle) return conv_fail(USC_E_ARG, "usc_conv_graph_create: bad argument");
    ConvGraph* g = new ConvGraph();
    g->plans.resize(n);
    for (int i = 0; i < n; i++) {
        int rc = plan_layer(descs[i], g->plans[i]);
        if (rc) { delete g; return rc; }
    }
    if (hipGraphCreate(&g->graph, 0) != hipSuccess) { delete g; return conv_fail(USC_E_LAUNCH, "hipGraphCreate failed"); }
    std::vector<hipGraphNode_t> last(n);                     // node that completes layer i
    std::vector<void*> kargs(n);
    for (int i = 0; i < n; i++) {
        LayerPlan& p = g->plans[i];
        std::vector<hipGraphNode_t> deps;
        for (int j = i - 1; j >= 0; j--) {
            const usc_conv_desc& q = g->plans[j].d;
            const bool raw = q.out == p.d.in || q.out == p.d.res1 || q.out == p.d.res2;
            const bool waw = q.out == p.d.out;                                   // phases / reuse of an output buffer
            const bool slab = p.reduce && g->plans[j].reduce && q.slab == p.d.slab;
            if (raw || waw || slab) deps.push_back(last[j]);
        }
        kargs[i] = &p.d;
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.func = p.fn; kp.gridDim = p.grid; kp.blockDim = p.block; kp.sharedMemBytes = 0;
        kp.kernelParams = &kargs[i]; kp.extra = nullptr;
        hipGraphNode_t node;
        if (hipGraphAddKernelNode(&node, g->graph, deps.data(), deps.size(), &kp) != hipSuccess) {
            (void)hipGraphDestroy(g->graph); delete g;
            return conv_fail(USC_E_LAUNCH, "hipGraphAddKernelNode failed");
        }
        last[i] = node;
        if (p.reduce) {
            kp.func = (void*)splitk_reduce_kernel; kp.gridDim = p.rgrid; kp.blockDim = dim3(256);
            hipGraphNode_t rnode;
            if (hipGraphAddKernelNode(&rnode, g->graph, &node, 1, &kp) != hipSuccess) {
                (void)hipGraphDestroy(g->graph); delete g;
                return conv_fail(USC_E_LAUNCH, "hipGraphAddKernelNode (reduce) failed");
            }
            last[i] = rnode;
        }
    }
    if (hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGraphDestroy(g->graph); delete g;
        return conv_fail(USC_E_LAUNCH, "hipGraphInstantiate failed");
    }
    *handle = g;
    return USC_OK;
}

extern "C" int usc_conv_graph_launch(void* handle, usc_stream_t stream) {
    if (!handle) return conv_fail(USC_E_ARG, "usc_conv_graph_launch: null handle");
    ConvGraph* g = (ConvGraph*)handle;
    hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
    if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
    return USC_OK;
}

extern "C" void usc_conv_graph_destroy(void* handle) {
    if (!handle) return;
    ConvGraph* g = (ConvGraph*)handle;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
}
