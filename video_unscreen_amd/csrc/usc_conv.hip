// usc_conv.hip - fused NHWC float32 convolution on the f32-input matrix cores (v_mfma_f32_32x32x2_f32).
//
// The networks of the path (matting UNet, human-parsing ResNet-101, DeepLabv3+, STM) are hundreds of small convolutions:
// dense contractions with M = output pixels (510 ... 522,240), N = Cout (32 ... 2048), K = taps * Cin (72 ... 36,864).
// One implicit-GEMM kernel serves every shape:
//   * taps are a list of (dy, dx) offsets with an input stride, so 3x3 s1/s2 (any dilation), 1x1, the 2x2-s2 form of
//     avg-pool + 1x1, and the four 2x2 output phases of ConvTranspose2d(k4, s2, p1) are the same code;
//   * a workgroup owns 128 output pixels x (32 | 64) channels; each of its waves owns 32 of those pixels.  A wave only ever
//     multiplies ITS OWN 32 pixel rows, so LDS is a layout transposer for A (coalesced 128-byte pixel rows -> the MFMA's
//     one-row-per-lane operand), not a sharing medium: every wave has a private 4.6 KB LDS image, which its A chunk reaches
//     straight from global memory (buffer_load ... lds, 16 bytes per lane, swizzled so that fragment reads are conflict-free; one K
//     step ahead: round 5 - rounds 1 - 4 went through two register sets), and reads its weight
//     fragments straight from global memory - the packed layout [k/4][n][4] IS the B-operand layout, a fragment load is two
//     contiguous 512-byte runs that all waves of the layer share through L1 / L2.  No __syncthreads in the K loop: waves
//     drift freely, 5 (4 for the 64-column tile) per SIMD;
//   * K parallelism for the many layers with only one or two tiles per CU: a workgroup may hold KW groups of 4 waves that
//     split the K range of the same tile (partial sums meet in LDS, added in slice order), and / or the grid splits K into
//     S slices whose partial sums go through a slab and splitk_reduce_kernel;
//   * the epilogue's transposition through the same image uses a 36-float row pitch (conflict-free b128 accesses);
//   * the epilogue fuses bias, residual (optionally from a half-resolution tensor = nearest x2 upsample),
//     ReLU / LeakyReLU, a per-channel affine (the shortcut branches' BN-after-ReLU), a second residual and a channel
//     slice of a wider output tensor (concat buffers), so no elementwise pass ever touches HBM again.
// f32 MFMA is an exact k-ordered fmaf chain; the K partition and the order partial sums are added in depend only on the
// layer's shape (and the caller's plan_batch), so results are run-to-run and batch-to-batch identical.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/unscreen_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KC = 32;         // K slice per step
constexpr int APITCH = 36;     // floats per A row in LDS (32 + 4 pad: b128 reads hit 64 distinct banks)

#ifdef USC_MFMA_TIMING           // dev instrumentation (tools/dev/mfma_phases.py): per wave 8 slots of {shader clock, 100 MHz real time} + its hardware id
#define MFMA_STAMP(k) do { if (d.slab && d.ksplit == 1 && lane == 0) { long long* ts_ = reinterpret_cast<long long*>(d.slab) + \
        (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * (RG * KW) + wave) * 16; \
        ts_[2 * (k)] = (long long)__builtin_readcyclecounter(); ts_[2 * (k) + 1] = (long long)__builtin_amdgcn_s_memrealtime(); \
        if ((k) == 0) ts_[15] = (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32); } } while (0)
#else
#define MFMA_STAMP(k) do { } while (0)
#endif

// NT: 32-column tiles per wave (1 | 2).  RG: row groups = waves side by side on 32 pixels each (4 | 2 | 1: a workgroup owns
// 128 | 64 | 32 pixels).  KW: K groups (RG x KW waves in all, <= 16).  IN_RELU: the layer computes conv(relu(x)).
template <int NT, int RG, int KW, bool IN_RELU>
__global__ __launch_bounds__(64 * RG * KW) __attribute__((amdgpu_waves_per_eu(NT == 1 ? 5 : 4))) void conv_mfma_kernel(usc_conv_desc d) {
    constexpr int BM = 32 * RG;                                  // output pixels per workgroup
    __shared__ __attribute__((aligned(16))) float As[RG * KW][32 * APITCH];
    __shared__ int s_toff[16];
    __shared__ int s_outpix[BM], s_res1pix[BM], s_vbase[BM], s_vmask[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifndef USC_MFMA_NO_PRIO
    // A wave that starts while older waves of its SIMD are inside their K loops is the youngest in the issue arbiter's order: its few
    // hundred scalar / vector instructions of table building waited behind their MFMA streams (tools/dev/mfma_phases.py: 9300 cycles on
    // average for the prologue of the parsing net's 1024 -> 256 layers, more than the 8192 MFMA cycles of the whole wave).  Prologue and
    // epilogue therefore run at a raised priority, the K loop at the default.
    __builtin_amdgcn_s_setprio(3);
#endif
    MFMA_STAMP(0);
    const int rg = wave % RG, kg = wave / RG;                    // row group (32 pixels) and K group of this wave
    const int M = d.OH * d.OW;
    const int nblk = d.Cout / (32 * NT);
    const int nwg = (int)gridDim.x, xcd = (int)blockIdx.x & 7, q8 = nwg >> 3, r8 = nwg & 7;      // XCD-aware order, see above
    const int wgid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + ((int)blockIdx.x >> 3);
    const int mtile = wgid / nblk;
    const int m0 = mtile * BM;
    const int nb0 = (wgid - mtile * nblk) * (32 * NT);
    const int Ktot = d.ntaps * d.Cin;
    const int S = d.ksplit;
    const int ph = blockIdx.z / S, ks = blockIdx.z - ph * S;
    const int nch_all = (Ktot + KC - 1) / KC;
    const int b_begin = (int)((long long)ks * nch_all / S), b_end = (int)((long long)(ks + 1) * nch_all / S);
    // this wave's slice of the block's chunk range
    const int c_begin = b_begin + (int)((long long)kg * (b_end - b_begin) / KW);
    const int c_end = b_begin + (int)((long long)(kg + 1) * (b_end - b_begin) / KW);
    const float* __restrict__ wpk = d.wpk + (size_t)ph * d.wphase_stride;
    const int out_py = d.nphase > 1 ? (ph >> 1) : d.out_py, out_px = d.nphase > 1 ? (ph & 1) : d.out_px;

    const int wbytes = nch_all * 8 * d.Cout * 16;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wpk), 0, wbytes, 0x00020000);
    // B fragment of MFMA group g (4 MFMAs = k rows 8g .. 8g+7), N tile j: float4 at ((8c + 2g + h) * Cout + nb0 + 32j + n) * 16 bytes
    const int wbase = ((lane >> 5) * d.Cout + nb0 + (lane & 31)) * 16;
    const int wstep = 8 * d.Cout * 16, wg2 = 2 * d.Cout * 16;
    // The weight fragments of the first K step do not depend on the per-pixel tables: request them before anything else.
    float4 bf[4 * NT];
    auto load_b = [&](int c, int g) {                      // the fragments of MFMA group g of chunk c
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, c < c_end ? wbase + c * wstep + g * wg2 + j * 512 : 0x7ffffff0, 0, 0);
            bf[g * NT + j] = *reinterpret_cast<const float4*>(&v);
        }
    };
#pragma unroll
    for (int g = 0; g < 4; g++) load_b(c_begin, g);

    constexpr int INVALID = 0x7fff0000;
    uint32_t dyw[4], dxw[4];
    __builtin_memcpy(dyw, d.dy, 16);
    __builtin_memcpy(dxw, d.dx, 16);
    if (d.nphase > 1) {
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
    MFMA_STAMP(1);

    // ---- staging assignment: lane -> (4 of this wave's rows, 8 apart; one float4 of the 32-wide K slice)
    const bool cin_pow2 = (d.Cin & (d.Cin - 1)) == 0;
    const int log2cin = 31 - __builtin_clz(d.Cin);
    int vbase[4];
    unsigned vmask[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        vbase[j] = s_vbase[32 * rg + (lane >> 3) + 8 * j];
        vmask[j] = (unsigned)s_vmask[32 * rg + (lane >> 3) + 8 * j];
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.in), 0, (d.batch > 1 ? d.batch : 1) * d.IH * d.IW * d.Cin * 4, 0x00020000);
    float* const Aw = As[wave];

    // A goes global -> LDS without passing through registers (buffer_load ... lds, 16 bytes per lane, lane l of request j lands at byte
    // 1024 j + 16 l of the wave's image): request j holds rows 8 j .. 8 j + 7, lane (r, p) of it fetches k-quad p ^ r of row 8 j + r, so
    // that a fragment read (32 rows x one k-quad) finds its 16-byte pieces in eight distinct columns of the 128-byte rows - conflict-free
    // without a padded pitch.  One image, requested one K step ahead (the fragments of step c are in registers before the requests of
    // step c + 1 overwrite it).  Rounds 1 - 4 staged A through registers (global -> two register sets -> ds_write_b128, two steps ahead):
    // 32 VGPRs more, four waves per SIMD (three at NT = 2) instead of five (four) - profiles/r05/mfma_phases.md, mfma_dma.md.
    const int qsw = (lane & 7) ^ ((lane >> 3) & 7);
    __attribute__((address_space(3))) float* const Al = (__attribute__((address_space(3))) float*)Aw;
    auto dma_a = [&](int c) {
        // (always four requests, also for the chunk past the last one, whose offsets are out of range and only write zeros: under a
        // condition the compiler loses count of the loads in flight and waits for more than the weight fragment it needs)
        const bool live = c < c_end;
        const int k0 = c * KC + 4 * qsw;
        const int tap = cin_pow2 ? (k0 >> log2cin) : (c * KC) / d.Cin;
        const int koff = s_toff[tap & 15] + ((k0 - tap * d.Cin) << 2);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool ok = live && ((vmask[j] >> tap) & 1u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, Al + 256 * j, 16, ok ? vbase[j] + koff : 0x7ffffff0, 0, 0, 0);
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;

    // Fragment reads are inline assembly with their own waits: the compiler cannot tell the image's LDS addresses from the requests'
    // destinations and would put `s_waitcnt vmcnt(0)` - a wait for the weight fragments requested a moment ago - in front of every
    // fragment read it can see.  The load counter is in order and every vector-memory instruction of the loop is known to the compiler's
    // own count (requests and weight loads are builtins), so its waits for the weight fragments stay exact; what the image needs is
    // written out here: the 4 NT weight loads of a step are the only loads younger than that step's requests.
    const int lds_img = (int)(unsigned)(unsigned long long)((__attribute__((address_space(3))) float*)Aw);
    const int asw = lane & 7, ah = lane >> 5;
    int fa[4];
#pragma unroll
    for (int g = 0; g < 4; g++) fa[g] = lds_img + ((lane & 31) * 32 + (((2 * g + ah) ^ asw) << 2)) * 4;
    f32x4 a0, a1, a2, a3;
    auto read_frags = [&](auto vm) {                      // vm: loads that may still be in flight behind the image's requests
        asm volatile("s_waitcnt vmcnt(%8)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(fa[0]), "v"(fa[1]), "v"(fa[2]), "v"(fa[3]), "n"(decltype(vm)::value) : "memory");
        if (IN_RELU) {
#pragma unroll
            for (int e = 0; e < 4; e++) { a0[e] = fmaxf(a0[e], 0.f); a1[e] = fmaxf(a1[e], 0.f); a2[e] = fmaxf(a2[e], 0.f); a3[e] = fmaxf(a3[e], 0.f); }
        }
    };
    auto mfma_group = [&](const f32x4& a, int g) {
#pragma unroll
        for (int j = 0; j < NT; j++) {
            const float4 b = bf[g * NT + j];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b.x, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b.y, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b.z, acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b.w, acc[j], 0, 0, 0);
        }
    };
    dma_a(c_begin);
    read_frags(std::integral_constant<int, 0>());           // (the first chunk: nothing behind it may be left in flight)
    MFMA_STAMP(2);
#ifndef USC_MFMA_NO_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    for (int c = c_begin; c < c_end; c++) {
        dma_a(c + 1);                                           // the fragments of chunk c are in registers: the image is free
        __builtin_amdgcn_sched_barrier(0);                      // (the scheduler otherwise sinks the requests and the weight loads to the end of the step)
        mfma_group(a0, 0); load_b(c + 1, 0); __builtin_amdgcn_sched_barrier(0);
        mfma_group(a1, 1); load_b(c + 1, 1); __builtin_amdgcn_sched_barrier(0);
        mfma_group(a2, 2); load_b(c + 1, 2); __builtin_amdgcn_sched_barrier(0);
        mfma_group(a3, 3); load_b(c + 1, 3); __builtin_amdgcn_sched_barrier(0);
        read_frags(std::integral_constant<int, 4 * NT>());      // (after the last step: zeros, unused; the epilogue's own LDS accesses are the compiler's to order)
    }
    MFMA_STAMP(3);
#ifndef USC_MFMA_NO_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif

    // ---- epilogue: as above, through this wave's own LDS image (free now).  With K groups every wave deposits its partial tile and then
    // FINISHES a share of the tile (round 5; rounds 1 - 4 left the whole sum and the fused epilogue to K group 0 - sixteen partial tiles read
    // and added by one wave while fifteen waited): the tile's four passes of eight rows are dealt out to the KW waves of a row group - KW = 2:
    // two passes each, 4: one, 8: half a pass (four rows), 16: a quarter (two rows; the other lanes' offsets are out of range: their loads
    // return zeros, their stores are dropped).  The partial sums are added in the same slice order ((p0 + p1) + p2) + ... : same bits.
    float* const T = Aw;
    // two-output layers (usc_conv_desc.out2): this workgroup's channel tile lies on one side of split_c - which tensor, row pitch,
    // channel origin and activation it uses is workgroup-uniform
    const bool second = d.out2 != nullptr && nb0 >= d.split_c;
    const int out_ld = second ? d.Cout - d.split_c : d.out_ld > 0 ? d.out_ld : d.out2 ? d.split_c : d.Cout;
    const int out_c0 = second ? -d.split_c : d.out_c0;
    const int act = second ? d.act2 : d.act;
    const int out_bytes = d.OHf * d.OWf * d.Cout * 4;
    const int erow = lane >> 3, ec4 = (lane & 7) * 4;
    constexpr int NP = KW >= 4 ? 1 : 4 / KW;                     // passes this wave finishes
    const int ps0 = KW == 1 ? 0 : KW == 2 ? 2 * kg : KW == 4 ? kg : KW == 8 ? (kg >> 1) : (kg >> 2);
    const bool row_on = KW <= 4 ? true : KW == 8 ? ((erow >> 2) == (kg & 1)) : ((erow >> 1) == (kg & 3));
    int ops[NP], oos[NP], rps[NP];
#pragma unroll
    for (int i = 0; i < NP; i++) {
        const int p = 32 * rg + (ps0 + i) * 8 + erow;
        const int opix = row_on ? s_outpix[p] : -1;
        ops[i] = opix < 0 ? INVALID : opix * d.Cout * 4;
        oos[i] = opix < 0 ? INVALID : (opix * out_ld + out_c0) * 4;
        rps[i] = row_on ? s_res1pix[p] : INVALID;
    }
    const bool has_r1 = d.res1 != nullptr, has_r2 = d.res2 != nullptr, has_aff = d.scale != nullptr;
    const __amdgpu_buffer_rsrc_t out_rsrc = S > 1
        ? __builtin_amdgcn_make_buffer_rsrc(d.slab + (size_t)ks * d.OHf * d.OWf * d.Cout, 0, out_bytes, 0x00020000)
        : __builtin_amdgcn_make_buffer_rsrc(second ? d.out2 : d.out, 0, d.OHf * d.OWf * out_ld * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.res1), 0, (has_r1 && S == 1) ? (d.OHf >> d.res1_shift) * (d.OWf >> d.res1_shift) * d.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.res2), 0, (has_r2 && S == 1) ? out_bytes : 0, 0x00020000);
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const int n = nb0 + j * 32 + ec4;
        const int nb4 = n * 4;
        // residuals of the wave's passes are requested before the accumulators go through LDS (one wait, hidden behind the
        // transposition); a zero-sized descriptor (no residual, split layer) turns them into zeros without touching memory
        u32x4 r1v[NP], r2v[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            r1v[i] = __builtin_amdgcn_raw_buffer_load_b128(r1_rsrc, rps[i] + nb4, 0, 0);
            r2v[i] = __builtin_amdgcn_raw_buffer_load_b128(r2_rsrc, ops[i] + nb4, 0, 0);
        }
        if (KW > 1 && j > 0) __syncthreads();                     // the previous tile's partial sums have been read by every wave
#pragma unroll
        for (int r = 0; r < 16; r++) T[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * APITCH + (lane & 31)] = acc[j][r];
        if (KW > 1) __syncthreads();
        float4 tv[NP];
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const int o = ((ps0 + i) * 8 + erow) * APITCH + ec4;
            float4 t = *reinterpret_cast<const float4*>(&As[rg][o]);
#pragma unroll
            for (int k = 1; k < KW; k++) {                        // slice order: ((p0 + p1) + p2) + p3
                const float4 u = *reinterpret_cast<const float4*>(&As[RG * k + rg][o]);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            tv[i] = t;
        }
        if (S > 1) {
#pragma unroll
            for (int i = 0; i < NP; i++)
                __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&tv[i]), out_rsrc, ops[i] + nb4, 0, 0);
            continue;
        }
        float4 bias = make_float4(0.f, 0.f, 0.f, 0.f), sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = bias;
        if (d.bias) bias = *reinterpret_cast<const float4*>(d.bias + n);
        if (has_aff) { sc = *reinterpret_cast<const float4*>(d.scale + n); sh = *reinterpret_cast<const float4*>(d.shift + n); }
#pragma unroll
        for (int i = 0; i < NP; i++) {
            const float4 t = tv[i];
            const float4 r1 = *reinterpret_cast<const float4*>(&r1v[i]), r2 = *reinterpret_cast<const float4*>(&r2v[i]);
            float v[4] = {t.x + bias.x, t.y + bias.y, t.z + bias.z, t.w + bias.w};
            const float a1[4] = {r1.x, r1.y, r1.z, r1.w}, a2[4] = {r2.x, r2.y, r2.z, r2.w};
            const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float x = v[e] + a1[e];
                if (act == USC_ACT_RELU) x = fmaxf(x, 0.f);
                else if (act == USC_ACT_LEAKY) x = x > 0.f ? x : x * 0.2f;
                else if (act == USC_ACT_LEAKY01) x = x > 0.f ? x : x * 0.01f;
                if (has_aff) x = x * s4[e] + h4[e];
                v[e] = x + a2[e];
            }
            const float4 o = make_float4(v[0], v[1], v[2], v[3]);
            __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const u32x4*>(&o), out_rsrc, oos[i] + nb4, 0, 0);
        }
    }
    MFMA_STAMP(4);
}


// ---- Winograd F(2x2, 3x3) for the 3x3 / stride-1 / pad-1 layers (78 % of the matting net's flops) ---------------------------
// Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A : 16 multiplications per 2x2 output tile and (cin, cout) pair instead of 36, so
// the matrix cores do 2.25x less work; the transforms are additions on the vector ALU.  As an implicit GEMM: for each of the 16
// positions (a, b) of the transformed 4x4 patch, M_ab[tile][cout] = sum_cin V_ab[tile][cin] U_ab[cin][cout].
//   * a UNIT is a block of 8 x 4 tiles (16 x 8 output pixels) x 32 output channels x one K group of the input channels.  Two
//     waves share a unit: wave h owns the positions of rows a = 2h, 2h + 1 - eight accumulators of 32 x 32 (128 registers), so a
//     wave fits in half a SIMD's register file and every SIMD holds TWO waves.  A workgroup is 8 waves = 4 units: waves 0..3 are
//     the h = 0 halves, waves 4..7 the h = 1 halves (a unit's two waves land on the same SIMD);
//   * the two waves of a SIMD work in opposite phases, half a K step apart: while one issues the 32 MFMAs of a step (2048 cycles
//     of matrix-pipe time), the other stages the next activations, reads its fragments of the patch and transforms them - the
//     vector work, the LDS traffic and every wait on memory of one wave sit in the shadow of the other wave's MFMAs (a single
//     wave per SIMD that did both measured 6100 - 6900 cycles per 4096 cycles of MFMA work: profiles/r02/wino_phase_timing.md);
//     a workgroup barrier separates the phases (the pair shares the staged patch);
//   * A side: the unit's 10 x 18 input patch (8 channels = 32 bytes per pixel) goes global -> registers (one step ahead) -> one of
//     two LDS images per pair, each wave staging half of it.  The image is stored as four parity sub-images (pixel row / column
//     parity) with a 19-slot row pitch: the 16 lanes of a ds_read_b128 phase (8 tiles of one tile row, two tile rows) then hit 64
//     distinct banks.  Lane (tile, half) reads three rows of its tile's 4 x 4 patch (12 ds_read_b128), column-transforms them to
//     its two rows of T = B^T d and row-transforms on the fly in the MFMA phase; the four components of a float4 feed four MFMAs;
//   * B side: U = G g G^T is prepared on the host (vmatting/net_hip.py pack_wino) in the fragment order [cin/8][position][half]
//     [cout][4], read straight from global memory, one float4 per position and step, re-requested (for the next step) into the
//     registers a row of positions has just consumed;
//   * epilogue: each wave output-transforms its two rows into partial sums of the four 2x2 output phases; per phase the partial
//     tiles of all waves of the unit's group (2 position halves x KW K groups) meet in LDS laid out [wave][register][lane]; read
//     back 16 bytes per lane every lane holds four consecutive channels of a pixel: the group's waves sum (fixed order) and
//     finish 16 / (2 KW) of the 16 accumulator registers each - bias, residuals, activation, per-channel affine - with one
//     16-byte load per residual tensor and one 16-byte store per task (epilogue4; KW = 1 layers without residual tensors keep a
//     one-barrier pair exchange with dword stores, finish_pair).
// Summation order depends on the shape (and KW) only: run-to-run and batch-to-batch identical results.  Results differ from
// the direct kernel's by float rounding (relative 2e-7 per layer): the planner's choice is a function of the shape alone.
constexpr int WSUBROW = 19, WSUB = 5 * WSUBROW;     // parity sub-image: 5 rows of 9 pixels x 2 halves, row pitch 19 slots
constexpr int WIMG = 384 * 4;               // floats of one LDS image: 4 x 95 = 380 slots, padded to 6 requests of 64 (6144 bytes)
constexpr int WREGION = 8 * 16 * 64;        // floats of one epilogue exchange region: 8 waves x 16 registers x 64 lanes (32 KB)
constexpr int WLDS = 2 * WREGION;           // 64 KB: two exchange regions; the K loop uses 4 pairs x 2 images = 48 KB of it
static_assert(WLDS >= 4 * 2 * WIMG, "the patch images must fit");

#ifdef USC_WINO_TIMING2          // per-phase stamps of the first three K steps: 16 shader-clock slots per wave
#define WINO_STAMP(k) do { } while (0)
#define WINO_PH(k) do { if (d.slab && lane == 0 && (k) < 16) reinterpret_cast<long long*>(d.slab)[((size_t)bid * 8 + wave) * 16 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
#elif defined(USC_WINO_TIMING)
#define WINO_PH(k) do { } while (0)
#define WINO_STAMP(k) do { if (d.slab && lane == 0) { long long* ts_ = reinterpret_cast<long long*>(d.slab) + ((size_t)bid * 8 + wave) * 16; \
        ts_[2 * (k)] = (long long)__builtin_readcyclecounter(); ts_[2 * (k) + 1] = (long long)__builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define WINO_STAMP(k) do { } while (0)
#define WINO_PH(k) do { } while (0)
#endif

// LIN: units are runs of 32 CONSECUTIVE tiles in raster order (tiles per image / 32 blocks, rounded up once per image) instead of
// 8 x 4 rectangles - the 34 x 60 tiles of a 68 x 120 map are 64 blocks instead of 72, so that 4 channel groups make 256 units:
// one workgroup per CU.  A block is then up to four row segments; every segment stages its own four pixel rows (no vertical
// sharing between tile rows), stored per (patch row, column parity, channel half) as arrays over the column pairs - consecutive
// tiles read consecutive 16-byte slots.  Needs TW >= 11 (at most four segments: 16 x (32 + 4) slots = 9 requests of 64).
constexpr int WLIN_IMG = 576 * 4;           // floats of one LDS image of the linear form
// T3: the four output phases of a ConvTranspose2d(k 4, s 2, p 1) are 2x2 convolutions of the input - Winograd F(3x3, 2x2) on
// 4x4 patches at a stride of THREE pixels: 16 multiplications per 9 outputs instead of 36.  With G's last row negated on the
// host the input transform B^T is F(2x2, 3x3)'s own, so the K loop is the same code; what changes is the geometry, the output
// transform (A^T = [1 1 1 0; 0 1 -1 0; 0 1 1 1]: nine output phases) and the output addressing (pixel (2 (3 ty + i) + py,
// 2 (3 tx + j) + px) of the doubled map).  The four units of a workgroup are the FOUR PHASES of one block x channel group (no K
// groups): their 4x4 patches are windows of the same 5 x 5 pixels, so the workgroup stages ONE image per step - five pixel rows
// x three column residues mod 3 as planes over the column triples (30 slots per triple: at most 30 x 36 = 17 requests of 64,
// spread over the eight waves) - instead of four: a first version with an image per phase spent 7960 cycles per K step on its
// 14 staging requests per pair (the same dependence on the request count as between the 8 x 4 and the linear form).
constexpr int WT3_IMG = 24 * 256;           // floats of one LDS image of the transposed form: three requests per wave (24 KB)
constexpr int WLDS_LIN = 3 * WT3_IMG + 8 * 320 + 8 * 128;      // floats of LDS per workgroup (86 KB): the transposed form's three images + tables (linear form without K groups: 8 x 320 request offsets + 8 x 128 fragment addresses)
constexpr int WT3_GOFF = 3 * WT3_IMG;       // (int index) the transposed form's request offsets: 8 waves x 3 x 64, past the images
constexpr int WT3_FA = WT3_GOFF + 8 * 192;  // (int index) the transposed form's fragment addresses: 8 waves x 2 x 64
static_assert(WLDS_LIN >= 8 * WLIN_IMG && WLDS_LIN >= WLDS && WLDS_LIN >= WT3_FA + 8 * 128, "every form's images and tables must fit");
constexpr int WLIN_FA = WT3_GOFF + 8 * 320;  // (int index) fragment addresses of the linear form without K groups: 8 waves x 2 x 64
static_assert(WT3_GOFF >= 8 * WLIN_IMG && WLDS_LIN >= WLIN_FA + 8 * 128, "the linear form's request offsets (8 waves x 5 x 64) and fragment addresses sit past its images");
constexpr int WSLAB_TAIL = 65536;      // the slab's last bytes: arrival counters of layers that split K over two workgroups

struct VA { int vo, so; bool ok; };          // where a finished value goes: lane offset, scalar offset, exists

// ---- Run mode (RUN = true): ONE launch spans a run of consecutive layers of the same geometry and kernel variant (the matting net's
// 64 -> 64 layers at 136 x 240, ...).  A workgroup pulls (layer, workgroup index) items in layer-major order from an atomic counter;
// before it requests its first patch it waits until the (up to nine) blocks of the previous layer that its patch overlaps have been
// stored - per-block arrival counters, polled with a bound (a broken hand-over becomes a counted error, never a hang).  No grid
// barrier, and no workgroup ever waits for work that is not already owned by a running workgroup (items are handed out in order, an
// item's producers have lower indices): the scheme cannot deadlock however many CUs other kernels hold.  Visibility across the eight
// non-coherent L2s (MI355X_MICROARCH.md, inter-workgroup visibility, first row of the sc1 hand-off table): every store of a layer's
// output is a 16-byte write-through store (sc0 sc1), every storing wave waits for its stores (vmcnt(0)), a workgroup barrier, then ONE
// lane per block x channel group adds to the block's counter (agent scope); consumers poll with sc1 loads and read the activations
// (patches and residuals) with sc0 sc1 loads only after their own poll has matched.  Tensors are written once per launch and kernel
// boundaries invalidate the L2s, so no stale line can sit in a consumer's L2.
constexpr int USC_RUN_MAX = 8;
#ifndef USC_RUN_EXP
#define USC_RUN_EXP 0            // timing experiments of round 4 (profiles/r04/runs_mode.md; NOT correct with a bit set): 1 plain patch / residual
#endif                           // loads, 2 plain stores, 4 no poll, 8 items by workgroup index instead of the counter, 16 no arrival counts
struct usc_conv_run {
    usc_conv_desc d[USC_RUN_MAX];
    int nlayers, wgs, nblocks;      // layers of the run, workgroups per layer, blocks per layer (= arrival counters per layer)
    unsigned* ctl;                  // [0] next item, [1] workgroups that found the queue empty, [2] launch sequence number, [3] polls that ran out
    unsigned* flags;                // [layer][block] arrival counters, never reset: after launch q a counter shows q x channel groups
};
template <bool RUN> struct WinoArg { typedef usc_conv_desc type; };
template <> struct WinoArg<true> { typedef usc_conv_run type; };

template <int KW, bool IN_RELU, int MODE, bool RUN = false>         // MODE 0: 8 x 4 blocks, 1: linear blocks, 2: transposed (T3, linear)
__global__ __launch_bounds__(512) void conv_wino_kernel(typename WinoArg<RUN>::type arg) {
    __shared__ __attribute__((aligned(16))) float Ls[WLDS_LIN];     // (a template-dependent bound makes the host pass drop the kernel's stub without a word)
    __shared__ int s_item;
    unsigned run_seq = 0;
    if constexpr (RUN) run_seq = __hip_atomic_load(&arg.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if constexpr (RUN) {                                             // the first item; every later one is fetched while the stores of the one before drain
        if (threadIdx.x == 0) s_item = (USC_RUN_EXP & 8) ? (int)blockIdx.x : (int)__hip_atomic_fetch_add(&arg.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    int run_pass = 0;
    do {                                                             // (one pass unless RUN)
    int run_layer = 0, bid = (int)blockIdx.x, nwg_ = (int)gridDim.x;
    if constexpr (RUN) {
        const int item = __builtin_amdgcn_readfirstlane(s_item);     // (written before the barrier that ended the previous pass)
        run_pass++;
        if (item >= arg.nlayers * arg.wgs) {
            // the queue is empty: the last workgroup to see that puts the queue back and advances the sequence number for the next launch
            if (threadIdx.x == 0 && __hip_atomic_fetch_add(&arg.ctl[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(&arg.ctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&arg.ctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&arg.ctl[2], run_seq + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            break;
        }
        run_layer = item / arg.wgs;
        bid = item - run_layer * arg.wgs;
        nwg_ = arg.wgs;
    }
    const usc_conv_desc& d = *[&]() { if constexpr (RUN) return &arg.d[run_layer]; else return &arg; }();
    constexpr bool LIN = MODE != 0, T3 = MODE == 2;
    constexpr int IMGF = T3 ? WT3_IMG : LIN ? WLIN_IMG : WIMG;      // floats per LDS image
    constexpr int NG = LIN && !T3 ? 5 : 3;                          // global -> LDS requests of a wave per image
    static_assert(!T3 || KW == 1, "the transposed form has no K groups: the four pairs are the four output phases");
    static_assert(!RUN || (MODE == 0 && KW >= 2), "run mode: 8 x 4 blocks with K groups (the epilogue in 16-byte rounds)");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, sub = wave & 3;                        // position half (rows 2h, 2h + 1) and pair of this wave
    WINO_STAMP(0);
    WINO_PH(0);
    const int nnb = d.Cout >> 5;
    // Dilation D (taps at +-D): the layer is D x D independent standard convolutions, one per pixel-parity class of the image -
    // every image counts as D x D virtual images of (IH / D) x (IW / D) pixels whose pixel (y, x) is the real pixel
    // (y D + py, x D + px).  All geometry below is in virtual-image coordinates; pix() maps to the real pixel index.
    const int D = T3 ? 1 : d.dx[5];
    const int NV = D * D;                                           // virtual images per image
    const int nimg = (d.batch > 1 ? d.batch : 1) * NV;
    const int H = d.IH / D, W = d.IW / D;                           // per (virtual) image; output has the same size
    const int WR = d.IW;
    const int TH = T3 ? (H + 2) / 3 : (H + 1) >> 1, TW = T3 ? (W + 2) / 3 : (W + 1) >> 1;
    const int BH = (TH + 3) >> 2, BW = (TW + 7) >> 3;
    // tiles per linear block: 32, or (T3, narrow maps) as many whole tile rows as fit in 32 tiles and four segments
    const int nbt = T3 && TW < 11 ? min(4, 32 / TW) * TW : 32;
    const int ntile_img = TH * TW, nblk_img = T3 ? (ntile_img + nbt - 1) / nbt : LIN ? (ntile_img + 31) >> 5 : BH * BW;
    const int ntiles = nimg * nblk_img * nnb;                       // blocks x channel groups
    const int nwg = nwg_, xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;      // XCD-aware order (see above)
    // (run mode: items are handed out in order and the activations are read past the L2 anyway)
    const int wid = RUN ? bid : (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // K split across TWO workgroups (d.ksplit = 2, K groups only): neighbours in the XCD-local order take the two halves of the
    // channel range of the same units; their partial output tiles meet in the slab, the workgroup that arrives second finishes.
    const int S = d.ksplit == 2 ? 2 : 1;
    const int ks = S == 2 ? (wid & 1) : 0, wgid = S == 2 ? (wid >> 1) : wid;
    constexpr int TPW = 4 / KW;                                     // blocks x channel groups per workgroup
    const int wt = T3 ? wgid : wgid * TPW + sub / KW, kg = sub % KW;  // (T3: the pairs are the phases of ONE block x channel group)
    const bool live = wt < ntiles;
    // (integer division runs on the vector ALU: readfirstlane brings the wave-uniform results back to scalar registers, which
    // the scalar-offset addressing of the epilogue needs)
    const int nb = live ? wt % nnb : 0, sb = live ? wt / nnb : 0;
    const int img = __builtin_amdgcn_readfirstlane(sb / nblk_img), brem = sb - img * nblk_img;
    const int by = LIN ? 0 : __builtin_amdgcn_readfirstlane(brem / BW), bx = LIN ? 0 : brem - by * BW;
    // linear form: the block's (up to four) row segments - tile row, first tile column, length, first tile index inside the
    // block, slots per plane (length + 1 column pairs), first LDS slot.  All wave-uniform.
    int seg_row[4] = {0, 0, 0, 0}, seg_a[4] = {0, 0, 0, 0}, seg_len[4] = {0, 0, 0, 0}, seg_start[4] = {0, 0, 0, 0}, seg_L[4] = {0, 0, 0, 0}, seg_slot[4] = {0, 0, 0, 0};
    float seg_rL[4] = {0.f, 0.f, 0.f, 0.f};                         // 1 / seg_L (one v_rcp_f32 each: the decoding below is exact with 1 ulp)
    int cnt = 0, nslots = 0;
    if (LIN) {
        const int T0 = brem * nbt;
        cnt = live ? min(nbt, ntile_img - T0) : 0;
        const int ty0 = __builtin_amdgcn_readfirstlane(T0 / TW), a0 = T0 - ty0 * TW;
        int used = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int a = k == 0 ? a0 : 0;
            const int len = max(0, min(TW - a, cnt - used));
            seg_row[k] = ty0 + k; seg_a[k] = __builtin_amdgcn_readfirstlane(a); seg_len[k] = __builtin_amdgcn_readfirstlane(len);
            seg_start[k] = __builtin_amdgcn_readfirstlane(used); seg_L[k] = len > 0 ? seg_len[k] + 1 : 0; seg_slot[k] = __builtin_amdgcn_readfirstlane(nslots);
            seg_rL[k] = len > 0 ? __builtin_amdgcn_rcpf((float)seg_L[k]) : 0.f;
            used += len;
            nslots += (T3 ? 30 : 16) * seg_L[k];
        }
    }
    // tile t of the block (t = the MFMA row) -> its segment and position (dead tiles of a partial block -> segment 0, position 0)
    auto lin_seg = [&](int t, int& row, int& a, int& L, int& slot0, int& p) {
        int k_start = seg_start[0];
        row = seg_row[0]; a = seg_a[0]; L = seg_L[0]; slot0 = seg_slot[0];
#pragma unroll
        for (int k = 1; k < 4; k++) {
            const bool in = seg_len[k] > 0 && t >= seg_start[k];
            row = in ? seg_row[k] : row; a = in ? seg_a[k] : a; L = in ? seg_L[k] : L; slot0 = in ? seg_slot[k] : slot0; k_start = in ? seg_start[k] : k_start;
        }
        p = t < cnt ? t - k_start : 0;
        if (t >= cnt) { row = seg_row[0]; a = seg_a[0]; L = seg_L[0]; slot0 = seg_slot[0]; }
    };
    const int nb0 = __builtin_amdgcn_readfirstlane(nb * 32);
    const int imr = img / NV, par = T3 ? sub : img - imr * NV, pary = T3 ? par >> 1 : par / D, parx = par - pary * (T3 ? 2 : D);
    const int pix0 = __builtin_amdgcn_readfirstlane(T3 ? imr * d.IH * WR : (imr * d.IH + pary) * WR + parx);      // real pixel index of virtual (0, 0)
    auto pix = [&](int y, int x) { return pix0 + (y * WR + x) * D; };
    const int nch = d.Cin >> 3;                                     // K steps of 8 input channels
    const int nmax = nch / (KW * S);                                // steps per wave: the planner splits K evenly, nch = S * KW * nmax
    const int c_begin = live ? (ks * KW + kg) * nmax : 0;

    // ---- weight fragments: [c][p][k half][n][4] floats; this wave's positions are p = (2h + ar) * 4 + b.  The lane part of the
    // address is a constant, the (step, position) part goes through the scalar offset: no address registers per load.
    const int wbytes = nch * 16 * 2 * d.Cout * 16 * (T3 ? 4 : 1);      // (T3: one operand per output phase)
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.wino), 0, wbytes, 0x00020000);
    const int wpos = 2 * d.Cout * 16, wstep = 16 * wpos;
    const int wph0 = T3 ? __builtin_amdgcn_readfirstlane(par * nch * wstep) : 0;
    const int wbase = ((lane >> 5) * d.Cout + nb0 + (lane & 31)) * 16;
    float4 bf[2][4];
    auto load_b = [&](int c, int ar) {                              // (past the last step: the next K group's, or zeros - unused)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, wbase, wph0 + c * wstep + (8 * h + ar * 4 + b) * wpos, 0);
            bf[ar][b] = *reinterpret_cast<const float4*>(&v);
        }
    };
    load_b(c_begin, 0);
    load_b(c_begin, 1);

    // ---- patch staging, global memory -> LDS without passing through registers (buffer_load ... lds: lane l of a request
    // writes 16 bytes at LDS base + 16 l): each LDS slot (16 bytes = 4 channels of one patch pixel) of the image is assigned the
    // global address of the pixel half it holds, out-of-image pixels and the layout's holes an out-of-range offset (= zeros).
    // An image is 384 slots = 6 requests of 64 lanes; wave h issues requests 3h .. 3h + 2.  Slot s = sub-image (s / 95: pixel
    // row parity x 2 + column parity), row (s % 95) / 19, pixel pair ((s % 19) >> 1), half (s & ... & 1).
    constexpr int INVALID = 0x7fff0000;
    const int y0 = 8 * by - 1, x0 = 16 * bx - 1;
    int goff[5];                                                    // (NG used; a template-dependent bound here makes the host pass drop the kernel stub silently)
#pragma unroll
    for (int r = 0; r < NG; r++) {
        const int sl = (T3 ? wave + 8 * r : NG * h + r) * 64 + lane;   // (T3: request w + 8 r of the workgroup's one image)
        if (!LIN) {
            const int sp = sl / WSUB, rem = sl - sp * WSUB;
            const int row = rem / WSUBROW, q = rem - row * WSUBROW;
            const int py = 2 * row + (sp >> 1), px = 2 * (q >> 1) + (sp & 1), half = q & 1;
            const int y = y0 + py, x = x0 + px;
            const bool ok = live && sl < 4 * WSUB && q < 18 && y >= 0 && y < H && x >= 0 && x < W;
            goff[r] = ok ? pix(y, x) * d.Cin * 4 + half * 16 : INVALID;
        } else {
            // slot -> segment k (slots of segment k: [seg_slot[k], + 16 L_k)); inside: [plane = (patch row i, column parity cp)]
            // [column pair][channel half]: pixel (2 row_k - 1 + i, 2 (a_k + pair) - 1 + cp)
            int row = seg_row[0], a0s = seg_a[0], L = seg_L[0], s0 = seg_slot[0];
            float rL = seg_rL[0];
#pragma unroll
            for (int k = 1; k < 4; k++) {
                const bool in = seg_L[k] > 0 && sl >= seg_slot[k];
                row = in ? seg_row[k] : row; a0s = in ? seg_a[k] : a0s; L = in ? seg_L[k] : L; s0 = in ? seg_slot[k] : s0;
                rL = in ? seg_rL[k] : rL;
            }
            const int q = (sl - s0) >> 1, hf = (sl - s0) & 1;                       // the two halves of a pixel are neighbouring slots: lanes 2k, 2k + 1 fetch 32 contiguous bytes
            const int plane = (int)(((float)q + 0.5f) * rL);                        // exact: q < 288, L <= 33 (the fraction is >= 0.5 / L away from an integer)
            const int pair = q - plane * L;
            // (T3: plane = (pixel row i of five, column residue cr), pixel (3 row - 1 + i, 3 (a_k + pair) - 1 + cr))
            const int y = T3 ? 3 * row - 1 + plane / 3 : 2 * row - 1 + (plane >> 1);
            const int x = T3 ? 3 * (a0s + pair) - 1 + plane % 3 : 2 * (a0s + pair) - 1 + (plane & 1);
            const bool ok = live && sl < nslots && y >= 0 && y < H && x >= 0 && x < W;
            goff[r] = ok ? pix(y, x) * d.Cin * 4 + hf * 16 : INVALID;
        }
    }
    if (T3) {
#pragma unroll
        for (int r = 0; r < 3; r++) reinterpret_cast<int*>(Ls)[WT3_GOFF + wave * 192 + r * 64 + lane] = goff[r];
    }
    // Linear form without K groups: its epilogue (the pair exchange) keeps more values alive through the K loop than the forms with K
    // groups do, and the compiler spilled four of the five request offsets - reloaded from scratch in front of every request, each
    // reload followed by a wait for EVERY load in flight (round 4: found by tests/test_isa_cpu.py's rule "no scratch instruction in a
    // loop that holds MFMAs").  They live in the table region past the images instead (free in this form: 8 waves x 5 x 64 ints), as
    // the transposed form's do.
    constexpr bool GOFF_LDS = LIN && !T3 && KW == 1;
    if (GOFF_LDS) {
#pragma unroll
        for (int r = 0; r < NG; r++) reinterpret_cast<int*>(Ls)[WT3_GOFF + wave * 320 + r * 64 + lane] = goff[r];
    }
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.in), 0, (d.batch > 1 ? d.batch : 1) * d.IH * d.IW * d.Cin * 4, 0x00020000);
    typedef int v4i_t __attribute__((ext_vector_type(4)));
    const unsigned long long in_addr = (unsigned long long)d.in;
    const v4i_t in_v4 = {(int)(unsigned)in_addr, (int)((in_addr >> 32) & 0xffff), (d.batch > 1 ? d.batch : 1) * d.IH * d.IW * d.Cin * 4, 0x00020000};     // the same descriptor as in_rsrc, as four scalar registers
    const int lds_base = (int)(unsigned)(unsigned long long)((__attribute__((address_space(3))) float*)Ls);
    const int img0 = T3 ? 0 : sub * 2 * IMGF;                       // the pair's two images (float index into Ls)
    auto issue_a = [&](int c) {                                     // the patch of step c -> image (c - c_begin) & 1
        if (T3) {
            // requests w, w + 8, w + 16 of the image's (at most) 17.  All three are issued whether they hold slots or not (an
            // empty one only writes zeros): under a condition the compiler loses count of the requests in flight and waits for
            // EVERY outstanding load - the weight fragments just requested - in front of the next fragment read (measured: fragment
            // phases of 1200 - 2700 cycles instead of 400).
            // Waits: the image is shared by all pairs and the compiler places no wait between these requests and the fragment
            // reads of OTHER waves, so images are requested two steps ahead (three images): every MFMA phase waits for weight
            // fragments that were requested AFTER the image requests of the step before, the load counter is in order, so an image
            // has landed one whole step before its first reader starts - behind two barriers - with a step and a half of latency
            // budget.  (An explicit wait one phase after the request, with two images, cost 9900 cycles per step: the requests take
            // about 4500 cycles to land.)
            // The request offsets live in LDS, not in registers: the K loop has none to spare, and a spilled offset comes back through
            // a scratch load - whose wait, the counter being in order, is a wait for the staging request issued just before it.
            const int dst = ((c - c_begin) % 3) * IMGF + wave * 256;
            typedef __attribute__((address_space(3))) volatile int lds_vint;      // (an LDS pointer: 32-bit addressing, ds_read_b32)
            lds_vint* const go = (lds_vint*)Ls + WT3_GOFF + wave * 192 + lane;
            const int g0 = go[0], g1 = go[64], g2 = go[128];
            // (issued as inline assembly: the compiler, seeing LDS-writing loads it cannot tell apart from the fragment reads'
            // addresses, would put a wait for the newest of them in front of every fragment phase.  Its own waits for ordinary
            // loads stay correct with loads it does not know of in flight: the counter is in order, it can only wait longer.)
            const int m0v = lds_base + dst * 4, so = c * 32;
            int m0_keep;                                          // m0 is saved and put back inside the statement: whatever the compiler keeps in it survives
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %5, %6 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
                         "s_add_u32 m0, m0, 0x2000\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(m0_keep) : "s"(m0v), "v"(g0), "v"(g1), "v"(g2), "s"(in_v4), "s"(so) : "memory", "scc");
            return;
        }
        const int dst = img0 + ((c - c_begin) & 1) * IMGF + NG * h * 256;
        if (GOFF_LDS) {
            typedef __attribute__((address_space(3))) volatile int lds_vint;
            lds_vint* const go = (lds_vint*)Ls + WT3_GOFF + wave * 320 + lane;
            const int g0 = go[0], g1 = go[64], g2 = go[128], g3 = go[192];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst], 16, g0, c * 32, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 256], 16, g1, c * 32, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 512], 16, g2, c * 32, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 768], 16, g3, c * 32, 0, 0);
            if (h == 0) { const int g4 = go[256]; __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 1024], 16, g4, c * 32, 0, 0); }
            return;
        }
        constexpr int AUX = RUN && !(USC_RUN_EXP & 1) ? 0x11 : 0;    // run mode: sc0 sc1 - the activations were stored by other workgroups of this launch
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst], 16, goff[0], c * 32, 0, AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 256], 16, goff[1], c * 32, 0, AUX);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 512], 16, goff[2], c * 32, 0, AUX);
        if (LIN) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 768], 16, goff[NG - 2], c * 32, 0, AUX);
            if (h == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, &Ls[dst + 1024], 16, goff[NG - 1], c * 32, 0, AUX);   // 9 requests: 5 + 4
        }
    };
    auto wait_vm0 = [] { __builtin_amdgcn_s_waitcnt(0x0F70); };     // vmcnt(0): every load of this wave has landed (LDS ones too)
    // Phase barrier of the K loop: LDS accesses of this wave are complete, nothing else is implied.  (__syncthreads() carries a
    // workgroup release fence, which the compiler turns into a wait for the global -> LDS requests in flight - exactly the latency
    // the phase structure is there to hide; it still waits for them in front of the next ds_read, where they are needed.)
    auto phase_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.f;

    // fragment base of this lane: tile t = lane & 31 -> (ty, tx) = (t >> 3, t & 7), channel half = lane >> 5.  Patch pixel
    // (2 ty + i, 2 tx + j) lives in sub-image (i & 1, j & 1) at row ty + (i >> 1), pixel tx + (j >> 1).  This wave reads the patch
    // rows i = h, h + 1, h + 2.
    const int fb = ((((lane & 31) >> 3) * WSUBROW + 2 * (lane & 7) + (lane >> 5))) * 4;
    // linear form: patch pixel (i, j) of tile (segment k, position p) is slot seg_slot[k] + 2 ((i 2 + (j & 1)) L_k + p + (j >> 1)) + half
    // (consecutive tiles are 32 bytes apart: a two-way bank conflict on the fragment reads, which sit in the partner's MFMA shadow,
    // bought for staging requests whose lane pairs fetch whole 32-byte sectors)
    // transposed form: patch pixel (i, j) of phase (py, px) is pixel (py + i, px + j) of the staged five rows: plane (py + i,
    // (px + j) mod 3), column triple p + (px + j) / 3
    // (as few registers as possible live through the K loop: ONE spilled value reloaded in front of the fragment reads makes the
    // wait for it a wait for every load in flight, the weight fragments just requested included - 2000 cycles per step.  T3: the
    // row addresses are derived from the first, the column offsets are compile-time per column phase px)
    int fa[3] = {0, 0, 0}, fs2 = 0;
    if (LIN) {
        int row, a, L, s0, pp;
        lin_seg(lane & 31, row, a, L, s0, pp);
#pragma unroll
        for (int k = 0; k < 3; k++) fa[k] = (s0 + 2 * ((T3 ? (pary + h + k) * 3 : (h + k) * 2) * L + pp) + (lane >> 5)) * 4;
        fs2 = 2 * L * 4;
        if (T3 || GOFF_LDS) {                                       // (read back at the start of every fragment phase)
            reinterpret_cast<int*>(Ls)[(T3 ? WT3_FA : WLIN_FA) + wave * 128 + lane] = fa[0];
            reinterpret_cast<int*>(Ls)[(T3 ? WT3_FA : WLIN_FA) + wave * 128 + 64 + lane] = fs2;
        }
    }
    // (vector types, not float4 structs: additions of <4 x float> become two v_pk_add_f32 - half the vector instructions that
    // compete with the partner wave's MFMAs for the register ports)
    auto relu4 = [](const f32x4& a) { return f32x4{fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)}; };
    auto vrow = [&](const f32x4 (&T)[4], int b) {                   // one element of a row of V = T B
        return b == 0 ? T[0] - T[2] : b == 1 ? T[1] + T[2] : b == 2 ? T[2] - T[1] : T[1] - T[3];
    };

    f32x4 V0[4], V1[4];                                             // handed from the fragment phase to the MFMA phase
    // Fragment phase of step c: this wave's two rows of T = B^T d from the staged image, then its two rows of V = T B.  (The MFMA
    // phase carries no vector arithmetic at all: next to f32 32x32 MFMAs, whose accumulators travel through the same register
    // ports, every vector instruction of the same wave measurably delays the matrix pipe.)
    auto frag_phase = [&](int c, auto hh, auto pxc) {
        constexpr int HH = decltype(hh)::value, PX = decltype(pxc)::value;
        const int cur = T3 ? ((c - c_begin) % 3) * IMGF : img0 + ((c - c_begin) & 1) * IMGF + (LIN ? 0 : fb);
        int t3_fa = 0, t3_fs = 0;
        if (T3 || GOFF_LDS) {
            typedef __attribute__((address_space(3))) volatile int lds_vint;
            lds_vint* const fp = (lds_vint*)Ls + (T3 ? WT3_FA : WLIN_FA) + wave * 128 + lane;
            t3_fa = fp[0]; t3_fs = fp[64];
        }
        f32x4 Ta[4], Tb[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int co = ((j & 1) * WSUB + (j >> 1) * 2) * 4;
            f32x4 e[3];
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (T3) e[k] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&Ls[cur + t3_fa + (k * 3 + (PX + j) % 3) * t3_fs + ((PX + j) / 3) * 8], 16));
                else if (GOFF_LDS) e[k] = *reinterpret_cast<const f32x4*>(__builtin_assume_aligned(&Ls[cur + t3_fa + (2 * k + (j & 1)) * t3_fs + (j >> 1) * 8], 16));   // fa[k] = fa[0] + 2 k fs2
                else if (LIN) e[k] = *reinterpret_cast<const f32x4*>(&Ls[cur + fa[k] + (j & 1) * fs2 + (j >> 1) * 8]);
                else e[k] = *reinterpret_cast<const f32x4*>(&Ls[cur + ((((HH + k) & 1) * 2) * WSUB + ((HH + k) >> 1) * WSUBROW) * 4 + co]);
                if (IN_RELU) e[k] = relu4(e[k]);
            }
            if (HH == 0) { Ta[j] = e[0] - e[2]; Tb[j] = e[1] + e[2]; }                 // rows 0, 1: d0 - d2, d1 + d2
            else { Ta[j] = e[1] - e[0]; Tb[j] = e[0] - e[2]; }                         // rows 2, 3: d2 - d1, d1 - d3
        }
#pragma unroll
        for (int b = 0; b < 4; b++) { V0[b] = vrow(Ta, b); V1[b] = vrow(Tb, b); }
    };
    // MFMA phase of step c: 2 rows x 4 positions x 4 components, component-major within a row (consecutive MFMAs into different
    // accumulators); each row's weight fragments are re-requested for step c + 1 as soon as the row is done.
#define USC_WROW(AR, V, COMP)                                                                                                    \
        acc[AR * 4 + 0] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[0].COMP, bf[AR][0].COMP, acc[AR * 4 + 0], 0, 0, 0);             \
        acc[AR * 4 + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[1].COMP, bf[AR][1].COMP, acc[AR * 4 + 1], 0, 0, 0);             \
        acc[AR * 4 + 2] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[2].COMP, bf[AR][2].COMP, acc[AR * 4 + 2], 0, 0, 0);             \
        acc[AR * 4 + 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[3].COMP, bf[AR][3].COMP, acc[AR * 4 + 3], 0, 0, 0);
    auto mfma_phase = [&](int c) {
        __builtin_amdgcn_s_setprio(1);
        USC_WROW(0, V0, x) USC_WROW(0, V0, y) USC_WROW(0, V0, z) USC_WROW(0, V0, w)
        load_b(c + 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        USC_WROW(1, V1, x) USC_WROW(1, V1, y) USC_WROW(1, V1, z) USC_WROW(1, V1, w)
        // Every load older than the four weight requests of row 0 has landed - THIS is what publishes a staged image: the image
        // requests of this step were issued before this phase, the counter is in order, and the phase barrier that follows hands the
        // image to the partner wave (which never waits for requests it did not issue).  The compiler emits the same literal in front
        // of row 1's last fragment - as long as it keeps merging the paths with and without image requests the way it does today; said
        // explicitly the hand-over does not depend on that.  (No cost: the wait is already satisfied when it is reached.)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        load_b(c + 1, 1);
        __builtin_amdgcn_s_setprio(0);
    };

    if constexpr (RUN) {
        static_assert(!RUN || MODE == 0, "run mode: 8 x 4 blocks only (the neighbours of a block are its eight neighbours in the block grid)");
        // wait for the blocks of the previous layer this block's patch overlaps: lane i < 9 polls neighbour (i / 3 - 1, i % 3 - 1); every
        // lane iterates the same number of times (`__all` is the only exit: a divergent poll loop can hang - profiles/r03/dataflow_probe.md)
        if (run_layer > 0 && !(USC_RUN_EXP & 4)) {
            const unsigned* const fp = arg.flags + (run_layer - 1) * arg.nblocks;
            const unsigned target = run_seq * (unsigned)(d.Cin >> 5);     // the previous layer's channel groups (its Cout = this layer's Cin)
            const int ny = by + lane / 3 - 1, nx = bx + lane % 3 - 1;
            bool ready = !(live && lane < 9 && ny >= 0 && ny < BH && nx >= 0 && nx < BW);
            const unsigned* const mine = fp + (img * nblk_img + (ready ? 0 : ny * BW + nx));
            int polls = 0;
            for (;;) {
                if (!ready) ready = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == target;
                if (__all(ready) || ++polls >= (1 << 17)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (polls >= (1 << 17) && lane == 0) __hip_atomic_fetch_add(&arg.ctl[3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    issue_a(c_begin);
    if (T3 && nmax > 1) issue_a(c_begin + 1);                       // (transposed form: images are requested TWO steps ahead)
    wait_vm0();
    __syncthreads();
    WINO_STAMP(1);
    WINO_PH(1);
    // Phase ph: the h = 0 wave runs half-phase k = ph, the h = 1 wave k = ph - 1; even k = fragment phase, odd k = MFMA phase of
    // step k >> 1.  The image of local step s is requested during phase 2s - 2 by both waves (its LDS image was last read in phase
    // 2s - 3): by the h = 0 wave when its fragment phase has read image s - 1, by the h = 1 wave at the start of its MFMA phase;
    // it is read in phases 2s (h = 0) and 2s + 1 (h = 1).  Who waits for the requests: a wave can only wait for its OWN (the load
    // counter is per wave), and the compiler places no wait in front of the fragment reads (ISA of the loop: ds_reads straight
    // after the barrier) - so each wave makes sure its requests have landed BEFORE the barrier that precedes the partner's reads:
    // the explicit vmcnt(4) at the end of every MFMA phase (mfma_phase: in-order counter, the image requests are older than row
    // 0's weight requests), and for the one request the h = 1 wave issues before its first MFMA phase an explicit wait.  Every wave of the workgroup
    // runs nmax steps (the planner only splits K evenly; a wave without a unit multiplies zeros): no state of the loop is
    // modified under a condition, which is what lets the compiler keep the accumulators in place.
    auto k_loop = [&](auto pxc) {
    if (h == 0) {
        const std::integral_constant<int, 0> H0;
        for (int s = 0; s < nmax; s++) {
            const int c = c_begin + s;
            WINO_PH(2 + 4 * s);
            frag_phase(c, H0, pxc);
            if (s + 1 + T3 < nmax) issue_a(c + 1 + T3);
            WINO_PH(3 + 4 * s);
            phase_barrier();
            WINO_PH(4 + 4 * s);
            mfma_phase(c);
            WINO_PH(5 + 4 * s);
            phase_barrier();
        }
        WINO_PH(14);
    } else {
        const std::integral_constant<int, 1> H1;
        if (nmax > 1 + T3) issue_a(c_begin + 1 + T3);
        phase_barrier();
        WINO_PH(2);
        frag_phase(c_begin, H1, pxc);
        WINO_PH(3);
        // The image requested above is read by the h = 0 wave right after the next barrier, and this wave has no MFMA phase in
        // between whose weight waits would cover it (the in-order argument of mfma_phase starts with the first MFMA phase): without
        // this wait the hand-over of the pair's SECOND image rested on its requests landing within a phase and a half - true on an
        // idle GPU (the partner waits about as long for its own half), false now and then beside another stream's kernels: a unit
        // then multiplied one K step of stale patch values (found in round 3: tools/dev/concurrent_conv_bisect.py).
        wait_vm0();
        phase_barrier();
        for (int s = 1; s < nmax; s++) {
            const int c = c_begin + s;
            if (s + 1 + T3 < nmax) issue_a(c + 1 + T3);
            WINO_PH(4 * s);
            mfma_phase(c - 1);
            WINO_PH(1 + 4 * s);
            phase_barrier();
            WINO_PH(2 + 4 * s);
            frag_phase(c, H1, pxc);
            WINO_PH(3 + 4 * s);
            phase_barrier();
        }
        WINO_PH(4 * nmax);
        mfma_phase(c_begin + nmax - 1);
        WINO_PH(14);
    }
    };
    if constexpr (T3) {
        if (parx == 0) k_loop(std::integral_constant<int, 0>()); else k_loop(std::integral_constant<int, 1>());
    } else {
        k_loop(std::integral_constant<int, 0>());
    }
#undef USC_WROW
    WINO_STAMP(2);

    // ---- output transform of this wave's rows: partial sums of the phases (i, j) -> pixel (2 ty + i, 2 tx + j).
    // A^T = [1 1 1 0; 0 1 -1 -1]: rows a = 0, 1 give Y0 += s0 + s1, Y1 += s1; rows a = 2, 3 give Y0 += s2, Y1 -= s2 + s3.
    f32x16 Yp[4];
    if constexpr (!T3) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const f32x16 s0 = j == 0 ? acc[0] + acc[1] + acc[2] : acc[1] - acc[2] - acc[3];
        const f32x16 s1 = j == 0 ? acc[4] + acc[5] + acc[6] : acc[5] - acc[6] - acc[7];
        const f32x16 t = s0 + s1;
        if (h == 0) { Yp[j] = t; Yp[2 + j] = s1; }
        else { Yp[j] = s0; Yp[2 + j] = -t; }
    }
    }
    WINO_STAMP(3);
    // ---- the group of this block x channel group: G = 2 KW waves (member m = 2 kg + h); member m finishes registers
    // [m NR, (m + 1) NR) of every phase.  Register r of lane l is tile (ty, tx) = (r >> 2, (r & 3) + 4 (l >> 5)), channel l & 31:
    // everything about a register's pixel is wave-uniform except the 8-pixel step of the upper half-wave, so addresses are one
    // lane offset per tensor plus a scalar offset per (register, phase).
    constexpr int G = 2 * KW, NR = 16 / G;
    const int me = 2 * kg + h;
    const int ubase = (sub / KW) * KW;                              // first pair of the group
    const int out_ld = d.out_ld > 0 ? d.out_ld : d.Cout;
    const int n = nb0 + (lane & 31), lh = lane >> 5;
    const bool has_r1 = d.res1 != nullptr, has_r2 = d.res2 != nullptr, has_aff = d.scale != nullptr;
    const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.out, 0, d.OHf * d.OWf * out_ld * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.res1), 0, has_r1 ? (d.OHf >> d.res1_shift) * (d.OWf >> d.res1_shift) * d.Cout * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.res2), 0, has_r2 ? d.OHf * d.OWf * d.Cout * 4 : 0, 0x00020000);
    const float bias = d.bias ? d.bias[n] : 0.f;
    const float sc = has_aff ? d.scale[n] : 1.f, sh = has_aff ? d.shift[n] : 0.f;
    // activation without branches: max(x, slope x) with slope = 1 (none), 0.2 / 0.01 (leaky), 0 (ReLU)
    const float slope = d.act == USC_ACT_LEAKY ? 0.2f : d.act == USC_ACT_LEAKY01 ? 0.01f : d.act == USC_ACT_RELU ? 0.f : 1.f;
    const int vo_out = (8 * lh * D * out_ld + n) * 4;                   // lane parts of the byte offsets
    const int vo_r2 = (8 * lh * D * d.Cout + n) * 4;
    const int vo_r1 = ((8 >> d.res1_shift) * lh * D * d.Cout + n) * 4;
    WINO_STAMP(4);
    const bool interior = T3 ? (cnt == 32 && H % 3 == 0 && W % 3 == 0) : LIN ? (cnt == 32 && !(H & 1) && !(W & 1)) : (live && 8 * by + 8 <= H && 16 * bx + 16 <= W);
    const bool plain = !has_r1 && !has_r2 && !has_aff;
    const std::true_type yes;
    const std::false_type no;
    // Where the value of accumulator register r goes.  Rectangular blocks: register r of lane l is tile (r >> 2, (r & 3) + 4 (l >>
    // 5)) of the block, so everything but the 8-pixel step of the upper half-wave is wave-uniform: one lane offset per tensor
    // (vo_*) + a scalar offset per value.  Linear blocks: tile (r & 3) + 8 (r >> 2) + 4 (l >> 5) of the run lies in some segment:
    // per-lane pixel index `pb` of its output pixel (0, 0) + flags (bit 0 tile exists, bit 1 row 1 exists, bit 2 column 1 exists),
    // computed once per register (lin_geom), + a scalar offset per output phase.
    auto lin_geom = [&](int r, int& pb, int& fl) {
        int row, a, L, s0, pp;
        const int t = (r & 3) + 8 * (r >> 2) + 4 * lh;
        lin_seg(t, row, a, L, s0, pp);
        const int oy = 2 * row, ox = 2 * (a + pp);
        pb = pix(oy, ox);
        fl = (t < cnt ? 1 : 0) | (oy + 1 < H ? 2 : 0) | (ox + 1 < W ? 4 : 0);
    };
    auto lin_ok = [](int fl, int i, int j) { return (fl & 1) && (i == 0 || (fl & 2)) && (j == 0 || (fl & 4)); };
    auto a_out = [&](int r, int pb, int fl, int i, int j) {
        VA v;
        if (LIN) {
            v.vo = (pb * out_ld + n) * 4; v.so = __builtin_amdgcn_readfirstlane(((i * WR + j) * D * out_ld + d.out_c0) * 4); v.ok = lin_ok(fl, i, j);
        } else {
            const int oy = __builtin_amdgcn_readfirstlane(2 * (4 * by + (r >> 2)) + i), ox = __builtin_amdgcn_readfirstlane(16 * bx + 2 * (r & 3) + j);
            v.vo = vo_out; v.so = __builtin_amdgcn_readfirstlane((pix(oy, ox) * out_ld + d.out_c0) * 4); v.ok = live && oy < H && ox + 8 * lh < W;
        }
        return v;
    };
    auto a_r2 = [&](int r, int pb, int fl, int i, int j) {
        VA v;
        if (LIN) {
            v.vo = (pb * d.Cout + n) * 4; v.so = __builtin_amdgcn_readfirstlane((i * WR + j) * D * d.Cout * 4); v.ok = lin_ok(fl, i, j);
        } else {
            const int oy = __builtin_amdgcn_readfirstlane(2 * (4 * by + (r >> 2)) + i), ox = __builtin_amdgcn_readfirstlane(16 * bx + 2 * (r & 3) + j);
            v.vo = vo_r2; v.so = __builtin_amdgcn_readfirstlane(pix(oy, ox) * d.Cout * 4); v.ok = live && oy < H && ox + 8 * lh < W;
        }
        return v;
    };
    auto a_r1 = [&](int r, int pb, int fl, int i, int j) {            // (a half-resolution residual only exists for D = 1: planner)
        VA v;
        if (LIN) {
            if (d.res1_shift) {                                        // D = 1: pb = (img H + oy) W + ox
                const int Y = pb / W, X = pb - Y * W;
                v.vo = ((((Y + i) >> d.res1_shift) * (d.OWf >> d.res1_shift) + ((X + j) >> d.res1_shift)) * d.Cout + n) * 4; v.so = 0;
            } else {
                v.vo = (pb * d.Cout + n) * 4; v.so = __builtin_amdgcn_readfirstlane((i * WR + j) * D * d.Cout * 4);
            }
            v.ok = lin_ok(fl, i, j);
        } else {
            const int oy = __builtin_amdgcn_readfirstlane(2 * (4 * by + (r >> 2)) + i), ox = __builtin_amdgcn_readfirstlane(16 * bx + 2 * (r & 3) + j);
            const int Y = img * H + oy;
            v.vo = vo_r1;
            v.so = __builtin_amdgcn_readfirstlane(d.res1_shift ? ((Y >> d.res1_shift) * (d.OWf >> d.res1_shift) + (ox >> d.res1_shift)) * d.Cout * 4 : pix(oy, ox) * d.Cout * 4);
            v.ok = live && oy < H && ox + 8 * lh < W;
        }
        return v;
    };
    // ---- Epilogue in four rounds with 16-byte accesses (all K-group layers; KW = 1 layers with residual tensors).  The group of a
    // block x channel group is G = 2 KW waves (member m = 2 kg + h); member m finishes registers [m NR, (m + 1) NR) of every output
    // phase, one phase per round, through two alternating exchange regions.  What bounds an epilogue is the NUMBER of loads and
    // stores (a dword one costs the memory pipeline as much as a dwordx4: see the transposed form below), and the accumulator
    // layout puts one channel in a lane - so the exchange region is laid out [wave][register][lane]: read back linearly, 16 bytes
    // per lane, it hands lane l four consecutive channels 4 (l & 7) .. + 3 of the pixel of register me NR + (l >> 4) + 4 u,
    // half-wave (l >> 3) & 1 (u < NR / 4 tasks per lane; four K groups: NR = 2 registers are 32 tasks, the upper half-wave idles):
    // one load per residual tensor and ONE store per task and round.  Summation order: K group major, position half minor (fixed:
    // run-to-run and batch-to-batch identical results).  ALL residual values of the four rounds are requested before the first
    // store is issued: loads and stores share one in-order counter, a load waited for after a store would wait for that store.
    auto epilogue4 = [&]() {
    constexpr int NTK = NR >= 4 ? NR / 4 : 1;
    const bool task = NR >= 4 || lane < 32;
    const int hf4 = (lane >> 3) & 1, ch = nb0 + 4 * (lane & 7);
    int oy0[NTK], ox0[NTK];
    bool tile_ok[NTK];
#pragma unroll
    for (int u = 0; u < NTK; u++) {
        const int r4 = me * NR + (task ? lane >> 4 : 0) + 4 * u;
        if (LIN) {
            int row, a, L, s0, pp;
            const int t = (r4 & 3) + 8 * (r4 >> 2) + 4 * hf4;
            lin_seg(t, row, a, L, s0, pp);
            oy0[u] = 2 * row; ox0[u] = 2 * (a + pp); tile_ok[u] = task && t < cnt;
        } else {
            oy0[u] = 2 * (4 * by + (r4 >> 2)); ox0[u] = 16 * bx + 2 * ((r4 & 3) + 4 * hf4); tile_ok[u] = task && live;
        }
    }
    const int rs = d.res1_shift;
    auto pix4 = [&](int u, int i, int j) { return pix(oy0[u] + i, ox0[u] + j); };              // real pixel index of output phase (i, j)
    auto ok4 = [&](int u, int i, int j) { return tile_ok[u] && oy0[u] + i < H && ox0[u] + j < W; };
    auto r1_off = [&](int u, int i, int j) {                                                   // (a half-resolution residual only exists for D = 1)
        const int Yg = imr * d.IH + oy0[u] + i, Xg = ox0[u] + j;
        return rs ? (((Yg >> rs) * (d.OWf >> rs) + (Xg >> rs)) * d.Cout + ch) * 4 : (pix4(u, i, j) * d.Cout + ch) * 4;
    };
    auto ld4 = [&](const float* p) { return *reinterpret_cast<const float4*>(p + ch); };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f), one4 = make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 bias4 = d.bias ? ld4(d.bias) : zero4, sc4 = has_aff ? ld4(d.scale) : one4, sh4 = has_aff ? ld4(d.shift) : zero4;
    float4 r1q[4][NTK], r2q[4][NTK];
#pragma unroll
    for (int rd = 0; rd < 4; rd++)
#pragma unroll
        for (int u = 0; u < NTK; u++) { r1q[rd][u] = zero4; r2q[rd][u] = zero4; }
    if (has_r1) {
#pragma unroll
        for (int rd = 0; rd < 4; rd++)
#pragma unroll
            for (int u = 0; u < NTK; u++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r1_rsrc, ok4(u, rd >> 1, rd & 1) ? r1_off(u, rd >> 1, rd & 1) : INVALID, 0, RUN && !(USC_RUN_EXP & 1) ? 0x11 : 0);
                r1q[rd][u] = *reinterpret_cast<const float4*>(&v);
            }
    }
    if (has_r2) {
#pragma unroll
        for (int rd = 0; rd < 4; rd++)
#pragma unroll
            for (int u = 0; u < NTK; u++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r2_rsrc, ok4(u, rd >> 1, rd & 1) ? (pix4(u, rd >> 1, rd & 1) * d.Cout + ch) * 4 : INVALID, 0, RUN && !(USC_RUN_EXP & 1) ? 0x11 : 0);
                r2q[rd][u] = *reinterpret_cast<const float4*>(&v);
            }
    }
    auto rounds4 = [&](auto full_c, auto edge_c) {
        constexpr bool FULL = decltype(full_c)::value, EDGE = decltype(edge_c)::value;
        // bias, residual, activation, affine, second residual, one 16-byte store.  (The interior variant still needs the lane
        // mask when the upper half-wave has no task.)
        auto finish4 = [&](const float (&v)[4], int rd, int u) {
            const float b4[4] = {bias4.x, bias4.y, bias4.z, bias4.w}, s4[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, t4[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
            const float p1[4] = {r1q[rd][u].x, r1q[rd][u].y, r1q[rd][u].z, r1q[rd][u].w}, p2[4] = {r2q[rd][u].x, r2q[rd][u].y, r2q[rd][u].z, r2q[rd][u].w};
            float x[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                x[k] = v[k] + b4[k];
                if (FULL) x[k] += p1[k];
                x[k] = fmaxf(x[k], x[k] * slope);
                if (FULL) { x[k] = x[k] * s4[k] + t4[k]; x[k] += p2[k]; }
            }
            const u32x4 xs = {__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1]), __builtin_bit_cast(unsigned, x[2]), __builtin_bit_cast(unsigned, x[3])};
            // (scalar offset = the constant 0: with a register there the next instruction may overwrite the data registers of a
            // 16-byte store before it has read them - see the transposed form)
            const bool st_ok = EDGE ? ok4(u, rd >> 1, rd & 1) : (NR >= 4 || task);
            __builtin_amdgcn_raw_buffer_store_b128(xs, out_rsrc, st_ok ? (pix4(u, rd >> 1, rd & 1) * out_ld + d.out_c0 + ch) * 4 : INVALID, 0, RUN && !(USC_RUN_EXP & 2) ? 0x11 : 0);   // (run mode: write-through)
        };
        float vs[4][NTK][4];
        constexpr int SC = 0x11;                                  // sc0 | sc1 (two workgroups per unit, below)
        const __amdgpu_buffer_rsrc_t sl_rsrc = __builtin_amdgcn_make_buffer_rsrc(d.slab, 0, S == 2 ? (int)d.slab_bytes : 0, 0x00020000);
        // slab: [unit][half][phase][member wave][task][4] floats = 16 KB per unit and half (16 NR tasks per member and phase)
        constexpr int TL = NR >= 4 ? 16 * NR : 32;
        auto slab_off = [&](int half, int rd, int u) { return (((wt * 2 + half) * 4 + rd) * G + me) * TL * 16 + (u * 64 + lane) * 16; };
#pragma unroll
        for (int rd = 0; rd < 4; rd++) {
            float* const reg = Ls + (rd & 1) * WREGION;
#pragma unroll
            for (int r = 0; r < 16; r++) reg[(wave * 16 + r) * 64 + lane] = Yp[rd][r];
            __syncthreads();
#pragma unroll
            for (int u = 0; u < NTK; u++) {
                float v[4];
#pragma unroll
                for (int m = 0; m < G; m++) {                     // fixed order: K group major, position half minor
                    const int w2 = (m & 1) * 4 + ubase + (m >> 1);
                    const float4 e = *reinterpret_cast<const float4*>(&reg[(w2 * 16 + me * NR + 4 * u) * 64 + 4 * lane]);
                    if (m == 0) { v[0] = 0.f + e.x; v[1] = 0.f + e.y; v[2] = 0.f + e.z; v[3] = 0.f + e.w; }
                    else { v[0] += e.x; v[1] += e.y; v[2] += e.z; v[3] += e.w; }
                }
#pragma unroll
                for (int k = 0; k < 4; k++) vs[rd][u][k] = v[k];
                if (S == 1) finish4(v, rd, u);
                else {
                    const u32x4 raw = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(raw, sl_rsrc, task && live ? slab_off(ks, rd, u) : INVALID, 0, SC);
                }
            }
            if (rd == 0) WINO_STAMP(5);
        }
        if (S == 1) return;
        // ---- two workgroups per unit: this one's sums went to the slab with system-scope write-through stores (the partner may sit
        // on another XCD, whose L2 is not coherent with ours: sc0 sc1 on both sides instead of a release that writes the whole L2
        // back), then one relaxed atomic per workgroup counts arrivals; the second to arrive adds the first one's sums to its own
        // (half 0 + half 1: addition commutes) and finishes.  The counters (last WSLAB_TAIL bytes of the slab) are zeroed when a plan is
        // made and put back to zero by every second arriver: between launches they are always zero.
        int* const counter = reinterpret_cast<int*>(reinterpret_cast<char*>(d.slab) + d.slab_bytes - WSLAB_TAIL) + wt;
        __builtin_amdgcn_s_waitcnt(0x0F70);                       // vmcnt(0): the stores have been acknowledged
        __syncthreads();
        int* const s_flag = reinterpret_cast<int*>(Ls);           // (the exchange regions are free: every wave is past its last read)
        if ((wave & 3) == ubase && h == 0 && kg == 0 && lane == 0)
            s_flag[sub / KW] = live ? (__hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) : 0;
        __syncthreads();
        if (s_flag[sub / KW] == 0) return;                        // first to arrive (even count): the partner finishes
        // second to arrive: put the counter back to zero (both arrivals have happened, nobody else touches it in this launch), so the
        // protocol does not depend on the parity a previous - possibly aborted - launch left behind
        if ((wave & 3) == ubase && h == 0 && kg == 0 && lane == 0 && live) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int rd = 0; rd < 4; rd++)
#pragma unroll
            for (int u = 0; u < NTK; u++) {
                const u32x4 o = __builtin_amdgcn_raw_buffer_load_b128(sl_rsrc, task ? slab_off(1 - ks, rd, u) : INVALID, 0, SC);
                const float4 of4 = *reinterpret_cast<const float4*>(&o);
                const float of[4] = {of4.x, of4.y, of4.z, of4.w};
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) v[k] = ks == 0 ? vs[rd][u][k] + of[k] : of[k] + vs[rd][u][k];
                finish4(v, rd, u);
            }
    };
    if (plain) { if (interior) rounds4(no, no); else rounds4(no, yes); }
    else { if (interior) rounds4(yes, no); else rounds4(yes, yes); }
    };
    if constexpr (T3) {
    // ---- transposed form: nine output phases (i, j) -> pixel (2 (3 ty + i) + py, 2 (3 tx + j) + px) of the doubled map.  With
    // C_j(a) = sum_b A^T[j][b] M[a][b] (C_0 = M0 + M1 + M2, C_1 = M1 - M2, C_2 = M1 + M2 + M3) this wave's partial sums are
    // h = 0 (rows a = 0, 1): Y_0j = C_j(0) + C_j(1), Y_1j = Y_2j = C_j(1); h = 1 (rows 2, 3): Y_0j = C_j(2), Y_1j = -C_j(2), Y_2j =
    // C_j(2) + C_j(3).  Nine exchange rounds as in the general scheme below (member m = 2 kg + h of the unit's 2 KW waves finishes
    // registers [m NR, (m + 1) NR) of every phase); the second residual of round rd + 1 is requested BEFORE the stores of round rd
    // (loads and stores share one in-order counter: waiting for a load issued after a store would wait for the store).
    // The planner admits layers with bias / activation / second residual only.
    // Stores are 16 bytes per lane: a `buffer_store_dword` of 64 lanes costs the memory pipeline as many cycles (~20) as a
    // `buffer_store_dwordx4`, and with 72 dword stores per wave the nine rounds were bound by exactly that (11.5 k cycles, the same
    // with the stores sent out of range).  The exchange region is laid out [wave][register][lane]: read linearly, 16 bytes per
    // lane, it hands lane t (+ 64 u) FOUR CONSECUTIVE CHANNELS 4 (t & 7) .. + 3 of the pixel of register me NR + (t >> 4) + 4 u,
    // half-wave (t >> 3) & 1 - the transposition costs nothing but 16 ds_write_b32 instead of 4 ds_write_b128.
    const int OW2 = d.OWf;
    constexpr int NTK = NR / 4;                                    // 16-byte tasks per lane and round (KW = 1: 2)
    static_assert(NTK >= 1, "the transposed form runs with KW = 1");
    const int ch = nb0 + 4 * (lane & 7);
    int tk_pb[NTK], tk_fl[NTK];
#pragma unroll
    for (int u = 0; u < NTK; u++) {
        int row, a, L, s0, pp;
        const int r = me * NR + (lane >> 4) + 4 * u, t = (r & 3) + 8 * (r >> 2) + 4 * ((lane >> 3) & 1);
        lin_seg(t, row, a, L, s0, pp);
        const int ty3 = 3 * row, tx3 = 3 * (a + pp);
        tk_pb[u] = (imr * 2 * H + 2 * ty3 + pary) * OW2 + 2 * tx3 + parx;
        tk_fl[u] = (t < cnt ? 1 : 0) | (ty3 + 1 < H ? 2 : 0) | (ty3 + 2 < H ? 4 : 0) | (tx3 + 1 < W ? 8 : 0) | (tx3 + 2 < W ? 16 : 0);
    }
    const f32x4 bias4 = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
    auto t3_ok = [](int fl, int i, int j) { return (fl & 1) && (i == 0 || (fl & (1 << i))) && (j == 0 || (fl & (4 << j))); };
    // (two variants, with and without a second residual: in a common one the waits for the residual loads sit on the path
    // without loads too, where all they find in flight are the stores of the round before)
    auto t3_rounds = [&](auto edge_c, auto r2_c) {
        constexpr bool EDGE = decltype(edge_c)::value, R2 = decltype(r2_c)::value;
        auto r2_request = [&](int rd, f32x4 (&dst)[NTK]) {
            const int i = rd % 3, j = rd / 3;
            const int so = __builtin_amdgcn_readfirstlane((2 * i * OW2 + 2 * j) * d.Cout * 4);
#pragma unroll
            for (int u = 0; u < NTK; u++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r2_rsrc, t3_ok(tk_fl[u], i, j) ? (tk_pb[u] * d.Cout + ch) * 4 : INVALID, so, 0);
                dst[u] = *reinterpret_cast<const f32x4*>(&v);
            }
        };
        f32x4 r2c[NTK], r2n[NTK];
#pragma unroll
        for (int u = 0; u < NTK; u++) { r2c[u] = f32x4{0.f, 0.f, 0.f, 0.f}; r2n[u] = r2c[u]; }
        if (R2) r2_request(0, r2c);
#pragma unroll
        for (int j = 0; j < 3; j++) {
            const f32x16 s0 = j == 0 ? acc[0] + acc[1] + acc[2] : j == 1 ? acc[1] - acc[2] : acc[1] + acc[2] + acc[3];
            const f32x16 s1 = j == 0 ? acc[4] + acc[5] + acc[6] : j == 1 ? acc[5] - acc[6] : acc[5] + acc[6] + acc[7];
            const f32x16 t01 = s0 + s1;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const int rd = j * 3 + i;
                f32x16 y;
                if (i == 0) y = h == 0 ? t01 : s0;
                else if (i == 1) y = h == 0 ? s1 : -s0;
                else y = h == 0 ? s1 : t01;
                float* const reg = Ls + (rd & 1) * WREGION;
#pragma unroll
                for (int r = 0; r < 16; r++) reg[(wave * 16 + r) * 64 + lane] = y[r];
                __syncthreads();
                float v[NTK][4];
#pragma unroll
                for (int u = 0; u < NTK; u++) {
#pragma unroll
                    for (int m = 0; m < G; m++) {                 // fixed order: K group major, position half minor
                        const int w2 = (m & 1) * 4 + ubase + (m >> 1);
                        const float4 e = *reinterpret_cast<const float4*>(&reg[(w2 * 16 + me * NR + 4 * u) * 64 + 4 * lane]);
                        if (m == 0) { v[u][0] = e.x; v[u][1] = e.y; v[u][2] = e.z; v[u][3] = e.w; }
                        else { v[u][0] += e.x; v[u][1] += e.y; v[u][2] += e.z; v[u][3] += e.w; }
                    }
                }
                if (R2 && rd + 1 < 9) r2_request(rd + 1, r2n);
                const int so = __builtin_amdgcn_readfirstlane(((2 * i * OW2 + 2 * j) * out_ld + d.out_c0) * 4);
#pragma unroll
                for (int u = 0; u < NTK; u++) {
                    float x[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        x[k] = v[u][k] + bias4[k];
                        x[k] = fmaxf(x[k], x[k] * slope);
                        if (R2) x[k] += r2c[u][k];
                    }
                    const u32x4 xs = {__builtin_bit_cast(unsigned, x[0]), __builtin_bit_cast(unsigned, x[1]), __builtin_bit_cast(unsigned, x[2]), __builtin_bit_cast(unsigned, x[3])};
                    // (the round's offset goes into the VECTOR offset, the scalar one stays the constant 0: with a register there the
                    // compiler assumes a 16-byte store has read its data when it issues and lets the next instruction overwrite the
                    // registers - on this part it has not: single components of some lanes then carry the next value, found with a
                    // pattern pushed through the exchange, tools/dev/t3_debug.py.  With a constant it inserts the wait state itself.)
                    __builtin_amdgcn_raw_buffer_store_b128(xs, out_rsrc, (EDGE ? t3_ok(tk_fl[u], i, j) : true) ? (tk_pb[u] * out_ld + ch) * 4 + so : INVALID, 0, 0);
                }
                if (R2) {
#pragma unroll
                    for (int u = 0; u < NTK; u++) r2c[u] = r2n[u];
                }
                if (rd == 0) WINO_STAMP(5);
            }
        }
    };
    if (has_r2) { if (interior) t3_rounds(no, yes); else t3_rounds(yes, yes); }
    else { if (interior) t3_rounds(no, no); else t3_rounds(yes, no); }
#ifdef USC_WINO_NO_PAIR_EPILOGUE
    } else if constexpr (false) {
#else
    } else if constexpr (KW == 1) {
#endif
        // The group is the pair: wave h finishes the two phases of output row parity i = h in full (32 values per lane), its
        // partials of the other two phases go to the partner through LDS ([wave][8 quads of registers][lane][4] floats, 8 KB per
        // wave) - one exchange, one barrier.  h0's partial + h1's partial, as in the general scheme (addition commutes).
        auto finish_pair = [&](auto hh, auto r1_c, auto r2a_c, auto edge_c) {
            constexpr int HH = decltype(hh)::value;
            constexpr bool R1 = decltype(r1_c)::value, R2A = decltype(r2a_c)::value, EDGE = decltype(edge_c)::value;
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const f32x16& y = Yp[2 * (1 - HH) + jj];
                    *reinterpret_cast<float4*>(&Ls[((wave * 8 + jj * 4 + qd) * 64 + lane) * 4]) = make_float4(y[4 * qd], y[4 * qd + 1], y[4 * qd + 2], y[4 * qd + 3]);
                }
            int g_pb[16], g_fl[16];
#pragma unroll
            for (int r = 0; r < 16; r++) { g_pb[r] = 0; g_fl[r] = 0; if (LIN) lin_geom(r, g_pb[r], g_fl[r]); }
            // residual values: requested before the barrier, all of them before the first store, and only where a residual exists
            // (R1: variants with a first residual; R2A: with a per-channel affine and / or a second residual).  A half-resolution
            // residual (nearest x2 upsampling: res1_shift = 1) is the same value for both column phases of a tile: requested once.
            float r1p[2][16], r2p[2][16];
            if (R1 || R2A) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int r = 0; r < 16; r++) { r1p[jj][r] = 0.f; r2p[jj][r] = 0.f; }
            }
            if (R1 && has_r1) {
                const bool r1_shared = d.res1_shift == 1;
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        if (jj == 0 || !r1_shared) {
                            const VA v = a_r1(r, g_pb[r], g_fl[r], HH, jj);
                            r1p[jj][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1_rsrc, v.ok ? v.vo : INVALID, v.so, 0));
                        } else {
                            r1p[jj][r] = r1p[0][r];
                        }
                    }
            }
            if (R2A && has_r2) {
#pragma unroll
                for (int jj = 0; jj < 2; jj++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const VA v = a_r2(r, g_pb[r], g_fl[r], HH, jj);
                        r2p[jj][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2_rsrc, v.ok ? v.vo : INVALID, v.so, 0));
                    }
            }
            __syncthreads();
            WINO_STAMP(5);
#pragma unroll
            for (int jj = 0; jj < 2; jj++)
#pragma unroll
                for (int qd = 0; qd < 4; qd++) {
                    const float4 u = *reinterpret_cast<const float4*>(&Ls[(((wave ^ 4) * 8 + jj * 4 + qd) * 64 + lane) * 4]);
                    const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int r = 4 * qd + e;
                        float x = Yp[2 * HH + jj][r] + uu[e] + bias;
                        if (R1) x += r1p[jj][r];
                        x = fmaxf(x, x * slope);
                        if (R2A) { x = x * sc + sh; x += r2p[jj][r]; }
                        const VA v = a_out(r, g_pb[r], g_fl[r], HH, jj);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), out_rsrc, (!EDGE || v.ok) ? v.vo : INVALID, v.so, 0);
                    }
                }
        };
        const bool r2a = has_r2 || has_aff;
        auto dispatch = [&](auto hh) {
            if (plain) { if (interior) finish_pair(hh, no, no, no); else finish_pair(hh, no, no, yes); }
            else if (!r2a) { if (interior) finish_pair(hh, yes, no, no); else finish_pair(hh, yes, no, yes); }
            else if (!has_r1) { if (interior) finish_pair(hh, no, yes, no); else finish_pair(hh, no, yes, yes); }
            else { if (interior) finish_pair(hh, yes, yes, no); else finish_pair(hh, yes, yes, yes); }
        };
        // (with residual tensors the 16-byte rounds win: the pair path loads them dword by dword, 32 per wave and tensor)
        if (has_r1 || has_r2) epilogue4();
        else if (h == 0) dispatch(std::integral_constant<int, 0>());
        else dispatch(std::integral_constant<int, 1>());
    } else {
    epilogue4();
    }
    WINO_STAMP(6);
    WINO_PH(15);
    if constexpr (RUN) {
        // the next item is requested now: the counter's round trip (and the queue of 255 workgroups that ask at about the same time)
        // overlaps the drain of this item's write-through stores, which has to be waited for anyway
        unsigned nxt = 0;
        if (threadIdx.x == 0) nxt = (USC_RUN_EXP & 8) ? (unsigned)(run_pass * (int)gridDim.x + (int)blockIdx.x) : __hip_atomic_fetch_add(&arg.ctl[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // publish this item: every wave's write-through stores have been acknowledged, then one lane per block x channel group counts
        __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (said twice: the compiler may drop a wait it believes satisfied - MI355X_MICROARCH.md, compiler hazard)
        if (threadIdx.x == 0) s_item = (int)nxt;
        __syncthreads();
        if (h == 0 && kg == 0 && lane == 0 && live && !(USC_RUN_EXP & 16))
            __hip_atomic_fetch_add(&arg.flags[run_layer * arg.nblocks + sb], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    } while (RUN);
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
// N = 1 is no matrix product: the layer is bound by reading its 32-channel input once (66.8 MB at 544 x 960).  Round 4 form: one WAVE
// per workgroup streams a strip of 32 pixels x 8 output rows: the 10 input rows (34 pixels x 128 bytes each) pass through a ring of three
// row buffers in LDS, requested two rows ahead with global -> LDS loads (five requests of 64 x 16 bytes per row; pixel pitch 36 floats:
// conflict-free b128 reads; padding pixels and the pad slot are out-of-range offsets = zeros).  The ring is private to the wave - the
// only synchronisation is the wave's own `s_waitcnt vmcnt`.  Lane (pixel, channel half) forms, per INPUT row, the three partial sums
// S_ky = sum over kx and its 16 channels (12 ds_read_b128 and 144 fused multiply-adds with the 144 weights it keeps in registers);
// an output row is S_0(row - 1) + S_1(row) + S_2(row + 1), carried in two rolling registers, the two channel halves meet in one
// cross-lane add.  Halo re-reads 34 / 32 x 10 / 8 = 1.33 as with the 8 x 32 tiles of rounds 1 - 3 (taller strips re-read less and were slower, below), LDS reads per output 13.5 instead of
// 144 (the old kernel also read its weights from LDS per tap and summed in ONE dependent chain of 288 multiply + add pairs).
// (strip height and ring depth timed on the device, tools/dev/head_ab.py, us for the 544 x 960 layer: 8 rows / ring 3: 17.4, 8 / 4: 17.8 - 18.2,
// 12 / 4: 18.6, 16 / 4: 20.5, 4 / 4: 21.4, 8 / 6: 24.0, 32 / 4: 32.2 - two waves per SIMD matter more than the halo rows saved)
constexpr int HS_W = 32, HS_R = 8, HS_RING = 3, HS_PITCH = 36, HS_REQ = 5, HS_ROWF = HS_REQ * 64 * 4;
static_assert(HS_RING >= 2 && HS_RING <= 9 && (HS_W + 2) * 9 <= HS_REQ * 64, "a row's slots (34 pixels x 9 slots of 16 bytes) fit five requests");
__global__ __launch_bounds__(64) void head_conv_kernel(usc_conv_desc d) {
    __shared__ __attribute__((aligned(16))) float ring[HS_RING * HS_ROWF];
    constexpr int INVALID = 0x7fff0000;
    const int lane = threadIdx.x, px = lane & 31, half = lane >> 5;
    const int x0 = blockIdx.x * HS_W, y0 = blockIdx.y * HS_R;
    const int H = d.IH, W = d.IW;
    const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.in), 0, H * W * 128, 0x00020000);
    int goff[HS_REQ];
#pragma unroll
    for (int r = 0; r < HS_REQ; r++) {
        const int s = r * 64 + lane, pxs = s / 9, q = s - pxs * 9, gx = x0 - 1 + pxs;
        goff[r] = (pxs < HS_W + 2 && q < 8 && gx >= 0 && gx < W) ? (gx * 32 + q * 4) * 4 : INVALID;
    }
    auto request = [&](int rr) {                                       // input row y0 - 1 + rr -> ring slot rr & 3
        const int y = y0 - 1 + rr;
        const bool ok = y >= 0 && y < H;
        const int so = ok ? y * W * 128 : 0;
        float* const dst = ring + (rr % HS_RING) * HS_ROWF;
#pragma unroll
        for (int r = 0; r < HS_REQ; r++)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(in_rsrc, dst + r * 256, 16, ok ? goff[r] : INVALID, so, 0, 0);
    };
#pragma unroll
    for (int rr = 0; rr < HS_RING - 1; rr++) request(rr);
    // this lane's 9 x 16 weights (wpk is [tap][32 channels]; a tap's 16 channels of a half are 64 contiguous bytes)
    float4 wr[9][4];
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
        for (int k = 0; k < 4; k++) wr[t][k] = *reinterpret_cast<const float4*>(d.wpk + t * 32 + half * 16 + 4 * k);
    const float bias = d.bias ? d.bias[0] : 0.f;
    float A = 0.f, B = 0.f;                                             // partial sums of output rows rr - 2 and rr - 1
    constexpr int NROWS = HS_R + 2;
#pragma unroll 1
    for (int rr = 0; rr < NROWS; rr++) {
        // (the ring slot row rr + 3 goes to was read in the pass before: its reads are complete before the request may write)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (rr + HS_RING - 1 < NROWS) request(rr + HS_RING - 1);
        // requests are answered in order: row rr has landed when at most min(HS_RING - 1, NROWS - 1 - rr) rows' requests are outstanding
        const int ahead = NROWS - 1 - rr;
        if (ahead >= HS_RING - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((HS_RING - 1) * HS_REQ) : "memory");
        else if (ahead >= 7) asm volatile("s_waitcnt vmcnt(35)" ::: "memory");      // (the tail of the strip: fewer rows outstanding)
        else if (ahead == 6) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
        else if (ahead == 5) asm volatile("s_waitcnt vmcnt(25)" ::: "memory");
        else if (ahead == 4) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* const rowp = ring + (rr % HS_RING) * HS_ROWF + half * 16;
        float S[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};           // two chains per kernel row
#pragma unroll
        for (int dx = 0; dx < 3; dx++) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) v[k] = *reinterpret_cast<const float4*>(rowp + (px + dx) * HS_PITCH + 4 * k);
#pragma unroll
            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float4 ww = wr[ky * 3 + dx][k];
                    S[ky][k & 1] = fmaf(v[k].x, ww.x, S[ky][k & 1]);
                    S[ky][k & 1] = fmaf(v[k].y, ww.y, S[ky][k & 1]);
                    S[ky][k & 1] = fmaf(v[k].z, ww.z, S[ky][k & 1]);
                    S[ky][k & 1] = fmaf(v[k].w, ww.w, S[ky][k & 1]);
                }
        }
        const float done = A + (S[2][0] + S[2][1]);
        A = B + (S[1][0] + S[1][1]);
        B = S[0][0] + S[0][1];
        if (rr >= 2) {                                                  // output row y0 + rr - 2 is complete: both channel halves, bias, activation
            const float tot = done + __shfl_xor(done, 32, 64) + bias;
            const int gy = y0 + rr - 2, gx = x0 + px;
            if (half == 0 && gy < d.OH && gx < d.OW) d.out[(size_t)gy * d.OW + gx] = (tanhf(tot) + 1.0f) / 2.0f;
        }
    }
}

thread_local char g_conv_err[256] = "";

}  // namespace

extern "C" const char* usc_conv_last_error(void) { return g_conv_err; }

static int conv_fail(int code, const char* msg) {
    snprintf(g_conv_err, sizeof(g_conv_err), "%s", msg);
    return code;
}

// Tuning hook: force (tile width, row groups, grid K split, K groups) for every layer planned from this thread (0 = let the
// planner decide).  Used by tools/dev/tune_convs.py to time all admissible plans of a layer shape on the device.
static thread_local int g_force_nt = 0, g_force_rg = 0, g_force_s = 0, g_force_kw = 0;
extern "C" void usc_conv_force_plan(int nt, int rowgroups, int ksplit, int kgroups) {
    g_force_nt = nt; g_force_rg = rowgroups; g_force_s = ksplit; g_force_kw = kgroups;
}
static thread_local int g_force_wino = -1;
extern "C" void usc_conv_force_winograd(int mode) { g_force_wino = mode; }
// which Winograd form the planner may / must take (tests and A/B timing; thread-local like the hooks above, no environment):
// t3: transposed form -1 cost model | 0 never | 1 wherever admissible; lin: blocks of 32 consecutive tiles, same coding;
// split: K halves on two workgroups 0 cost model | 1 never | 2 wherever admissible
static thread_local int g_force_t3 = -1, g_force_lin = -1, g_force_ws = 0;
extern "C" void usc_conv_force_winograd_form(int t3, int lin, int split) { g_force_t3 = t3; g_force_lin = lin; g_force_ws = split; }
// cost-model constants of the direct kernel's planner (microseconds; fitted on the device with tools/dev/tune_convs.py)
// (round 5: the LDS-DMA kernel admits 5 / 4 waves per SIMD; the constants and the model's 4 / 3 slots were kept - tools/dev/tune_convs.py times every
// admissible plan of every layer shape, tools/dev/plan_model_fit.py replays this model on that table: a re-fit is worth 1.3 % of the direct layers on paper
// (16.06 -> 15.85 ms over all networks; best plan per shape 15.52) and measured as a wash on the two networks of the headline loop (matting -5.6 us, parsing
// +6.4 us per frame); plans - and with them every summation order - are those of rounds 2 - 4)
static constexpr int kNt2Min = 512, kMinSteps = 9, kKwSteps = 4, kRgMin = 1;
static constexpr double kTStep = 0.43, kTLone = 1.5, kTFixed = 3.0, kTReduce = 8.0;

// One layer = one main launch (+ one reduce launch when K is split).
struct LayerPlan {
    usc_conv_desc d;
    void* fn;
    dim3 grid, block;
    bool reduce;
    dim3 rgrid;
    int nt = 0, rg = 0, kw = 0;            // for usc_conv_plan_query
    bool wino = false, lin = false;
    bool t3 = false;                       // transposed Winograd form
    bool wsplit = false;                   // Winograd layer with two workgroups per unit: uses the arrival counters in the slab's tail
};

static inline void* wslab_counters(const usc_conv_desc& d) { return reinterpret_cast<char*>(d.slab) + d.slab_bytes - WSLAB_TAIL; }

static int plan_layer(const usc_conv_desc& d, LayerPlan& p) {
    if (!d.in || !d.out || !d.wpk || d.Cin <= 0 || d.Cin % 4 || d.ntaps < 1 || d.ntaps > 16 || d.OH <= 0 || d.OW <= 0)
        return conv_fail(USC_E_ARG, "usc_conv: malformed layer descriptor");
    const long long M = (long long)d.OH * d.OW;
    p.d = d;
    p.reduce = false;
    p.block = dim3(256);
    if (d.kind == USC_CONV_HEAD) {
        if (d.Cout != 1 || d.Cin != 32 || d.ntaps != 9 || d.in_stride != 1 || d.IH != d.OH || d.IW != d.OW)
            return conv_fail(USC_E_UNSUPPORTED, "usc_conv: the head kernel is Conv2d(32, 1, 3, padding=1)");
        if (d.in_relu) return conv_fail(USC_E_UNSUPPORTED, "usc_conv: in_relu is not offered by the head kernel");
        p.fn = (void*)head_conv_kernel;
        p.grid = dim3((d.OW + HS_W - 1) / HS_W, (d.OH + HS_R - 1) / HS_R);
        p.block = dim3(64);                                    // one wave streams a 32-pixel x 8-row strip
        return USC_OK;
    }
    if (d.Cout % 32) return conv_fail(USC_E_UNSUPPORTED, "usc_conv: Cout must be a multiple of 32");
    if (d.out2 && (d.split_c <= 0 || d.split_c >= d.Cout || d.split_c % 32 || d.nphase > 1 || d.res1 || d.res2 || d.scale || d.out_c0 ||
                   (d.out_ld && d.out_ld < d.split_c)))
        return conv_fail(USC_E_ARG, "usc_conv: out2 needs 0 < split_c < Cout (a multiple of 32), one phase, no residual / affine tensors");
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
    // Tile shape and K partition are planned for `plan_batch` images of this layer's per-image size, whatever `batch` is:
    // a batched layer then sums K in exactly the order the single-image layer of the same plan does.
    // Two 32-column tiles per wave only while that still leaves >= 2 workgroups per CU.
    const int plan_batch = d.plan_batch > 1 ? d.plan_batch : 1;
    const long long m_plan = M / nbatch * plan_batch;
    // Transposed Winograd form (conv_wino_kernel, MODE 2): the four-phase ConvTranspose2d(k 4, s 2, p 1) layer (phase (py, px), tap
    // (iy, ix) reads the input at (py - iy, px - ix)) whose caller supplied the F(3x3, 2x2) operands; bias / activation / second
    // residual only.  Cost model as below: a round of workgroups (each the four phases of a block x channel group, all of K)
    // costs 14.5 (nine exchange rounds instead of four) + 2.55 per K step (profiles/r02/unet_layer_table.md: 34.1 us for 230
    // workgroups x 8 steps, 51.7 us for 455 x 4).
    {
        bool t3 = d.wino && nphase == 4 && d.ntaps == 4 && d.in_stride == 1 && d.out_stride == 2 && nbatch == 1 && d.Cin % 8 == 0 &&
                  !d.res1 && !d.scale && d.OH == d.IH && d.OW == d.IW && d.OHf == 2 * d.IH && d.OWf == 2 * d.IW && d.kind != USC_CONV_HEAD;
        for (int t = 0; t3 && t < 16; t++) t3 = d.dy[t] == (t >> 3) - ((t >> 1) & 1) && d.dx[t] == ((t >> 2) & 1) - (t & 1);
        const int force_t3 = g_force_t3;
        if (t3 && g_force_wino != 0 && force_t3 != 0) {
            const int TH = (d.IH + 2) / 3, TW = (d.IW + 2) / 3;
            const int nbt = TW < 11 ? (32 / TW < 4 ? 32 / TW : 4) * TW : 32;
            const long long blk = ((long long)TH * TW + nbt - 1) / nbt;
            const long long wgs = blk * (d.Cout / 32);             // a workgroup = the four phases of a block x channel group
            const int nsteps = d.Cin / 8;
            const double wino_us = (double)((wgs + 255) / 256) * (14.5 + nsteps * 2.55);
            const double direct_us = 6.0 + 2.0 * (double)M * 4 * d.Cout * 4.0 * d.Cin / 118e6 + (double)M * 4 * d.Cout * 4.0 / 3e6;
            if (wino_us < direct_us || g_force_wino == 1 || force_t3 == 1) {
                p.wino = true;
                p.lin = true;
                p.t3 = true;
                p.kw = 1;
                p.d.ksplit = 1;
                p.grid = dim3((unsigned)wgs);
                p.block = dim3(512);
                p.fn = d.in_relu ? (void*)conv_wino_kernel<1, true, 2> : (void*)conv_wino_kernel<1, false, 2>;
                return USC_OK;
            }
        }
    }
    // Winograd form (conv_wino_kernel): the standard 3x3 / stride 1 / pad 1 layer whose caller supplied the transformed weights.
    // The choice and the K groups are functions of the per-image shape and plan_batch only.
    {
        static const signed char sdy[9] = {-1, -1, -1, 0, 0, 0, 1, 1, 1}, sdx[9] = {-1, 0, 1, -1, 0, 1, -1, 0, 1};
        const int D = d.ntaps == 9 ? d.dx[5] : 0;                    // dilation: a dilated layer is D x D standard ones on the parity sub-images
        bool std3 = d.wino && !d.out2 && d.ntaps == 9 && nphase == 1 && d.in_stride == 1 && d.out_stride == 1 && d.Cin % 8 == 0 && D >= 1 && D <= 4 &&
                    d.IH % D == 0 && d.IW % D == 0 && (D == 1 || (d.res1_shift == 0)) &&
                    d.res1_shift >= 0 && d.IW == d.OW && (long long)d.IH * nbatch == d.OH && d.OHf == d.OH && d.OWf == d.OW;
        for (int t = 0; std3 && t < 9; t++) std3 = d.dy[t] == sdy[t] * D && d.dx[t] == sdx[t] * D;
        if (std3 && g_force_wino != 0) {
            const int TH = (d.IH / D + 1) / 2, TW = (d.IW / D + 1) / 2;
            // blocks per (virtual) image: 8 x 4 rectangles, or runs of 32 consecutive tiles (at most four row segments: TW >= 11).
            // K groups: a unit (block x channel group x K group) is two waves, a workgroup 4 units, a CU holds one workgroup.
            // K is only split evenly (every wave of a workgroup runs the same number of steps).  Cost model in microseconds,
            // fitted to profiles/r02/unet_layer_table.md and schp_layers_b12.txt: a round of workgroups costs 6.5 (launch ramp, first
            // loads, output transform + exchange + stores) + 2.45 per K step (two MFMA phases of 32 on every SIMD; 2.8 in the linear
            // form, which stages 9 instead of 6 requests per step and shares no pixel rows between tile rows); the direct kernel
            // 6 + flops / 118 TFLOP/s + output bytes / 3 TB/s (its epilogue is exposed when K is short).  All inputs are plan
            // quantities.
            const long long blk_rect = (long long)((TH + 3) / 4) * ((TW + 7) / 8), blk_lin = ((long long)TH * TW + 31) / 32;
            const int force_lin = g_force_lin;
            const int nsteps = d.Cin / 8;
            int kw = 1, ws = 1;
            bool lin = false;
            double wino_us = 1e30;
            const int force_ws = g_force_ws;
            for (int l = 0; l < 2; l++) {
                if (l == 1 && (TW < 11 || force_lin == 0)) continue;
                if (l == 0 && force_lin == 1 && TW >= 11) continue;
                const long long units = (long long)plan_batch * D * D * (l ? blk_lin : blk_rect) * (d.Cout / 32);
                for (int k = 1; k <= 4; k *= 2) {
                    if (nsteps % k) continue;
                    if ((g_force_kw == 1 || g_force_kw == 2 || g_force_kw == 4) && nsteps % g_force_kw == 0 && k != g_force_kw) continue;
                    // a second workgroup per unit (K halves meet in the slab, the later one finishes): K groups only
                    const long long units_cap = (long long)(nbatch > plan_batch ? nbatch : plan_batch) * D * D * (l ? blk_lin : blk_rect) * (d.Cout / 32);
                    const bool ok2 = k > 1 && nsteps % (2 * k) == 0 && d.slab && units_cap * 4 <= WSLAB_TAIL &&
                                     (size_t)units_cap * 2 * 16384 + WSLAB_TAIL <= d.slab_bytes && force_ws != 1;
                    for (int sp = 1; sp <= 2; sp++) {
                        if ((sp == 2 && !ok2) || (sp == 1 && ok2 && force_ws == 2)) continue;
                        const long long wgs = (units * k + 3) / 4 * sp;
                        const double t = (double)((wgs + 255) / 256) * (6.5 + (sp > 1 ? 3.5 : 0.0) + (nsteps / (k * sp)) * (l ? 2.8 : 2.45));
                        if (t < wino_us - 1e-9) { wino_us = t; kw = k; lin = l == 1; ws = sp; }
                    }
                }
            }
            const long long blk = lin ? blk_lin : blk_rect;
            const long long wt = (long long)nbatch * D * D * blk * (d.Cout / 32);
            const double direct_us = 6.0 + 2.0 * (double)m_plan * d.Cout * 9.0 * d.Cin / 118e6 + (double)m_plan * d.Cout * 4.0 / 3e6;
            const bool worth = wino_us < direct_us;
            if (worth || g_force_wino == 1) {
                p.wino = true;
                p.lin = lin;
                p.kw = kw;
                p.d.ksplit = ws;
                p.wsplit = ws == 2;
                p.d.nphase = 1;
                const long long waves = wt * kw;
                p.grid = dim3((unsigned)((waves + 3) / 4 * ws));
                p.block = dim3(512);                              // 4 units x 2 position halves
#define USC_PICKW(KWV) (lin ? (d.in_relu ? (void*)conv_wino_kernel<KWV, true, 1> : (void*)conv_wino_kernel<KWV, false, 1>) \
                           : (d.in_relu ? (void*)conv_wino_kernel<KWV, true, 0> : (void*)conv_wino_kernel<KWV, false, 0>))
                p.fn = kw == 4 ? USC_PICKW(4) : kw == 2 ? USC_PICKW(2) : USC_PICKW(1);
#undef USC_PICKW
                return USC_OK;
            }
        }
    }
    constexpr int nt2_min = kNt2Min;
    bool nt2 = (d.Cout % 64 == 0) && ((m_plan + 127) / 128 * (d.Cout / 64) * nphase >= nt2_min);
    if (g_force_nt) nt2 = g_force_nt == 2 && d.Cout % 64 == 0;
    if (d.out2 && d.split_c % 64) nt2 = false;                   // a 64-channel tile must not straddle the two outputs
    const int nblocks = nt2 ? d.Cout / 64 : d.Cout / 32;
    const int nch = (d.ntaps * d.Cin + KC - 1) / KC;
    // Parallelism for layers with few pixels: a workgroup's RG row groups cover 128 | 64 | 32 pixels, a workgroup may hold
    // KW groups of RG waves that split its K range (partial sums meet in LDS, no HBM traffic), and the grid may split K into
    // S slices (partial sums through the slab and splitk_reduce_kernel).  Chosen by a cost model; tools/dev/tune_convs.py
    // times every admissible plan of every layer shape of the path's networks on the device.
    constexpr int min_steps = kMinSteps, kw_steps = kKwSteps, rg_min = kRgMin;
    int RG = 4, KW = 1, S = 1;
    // slab capacity is checked for the planned batch (or the actual one if larger): the same S at every batch <= plan_batch
    const size_t plane_actual = (size_t)d.OHf * d.OWf * d.Cout * sizeof(float);
    const size_t plane_plan = (size_t)d.OHf / nbatch * plan_batch * d.OWf * d.Cout * sizeof(float);
    const size_t plane_bytes = plane_actual > plane_plan ? plane_actual : plane_plan;
    // kernel variants that exist: 64-column tiles with 4 row groups and KW <= 2 (register file); 32-column tiles with
    // RG x KW <= 16 waves, and at least 4 waves unless RG = 4
    auto exists = [&](int rg, int kw) {
        if (nt2) return rg == 4 && kw <= 2;
        if (rg == 4) return kw <= 4;
        return rg * kw >= 4 && rg * kw <= 16;
    };
    if (g_force_s || g_force_kw || g_force_rg) {
        S = g_force_s && !d.out2 ? g_force_s : 1;
        KW = g_force_kw ? g_force_kw : 1;
        RG = g_force_rg ? g_force_rg : 4;
        if (S > 1 && (!d.slab || (size_t)S * plane_bytes + WSLAB_TAIL > d.slab_bytes)) S = 1;
        if (S * KW > nch) { S = 1; KW = 1; }
        if (RG != 1 && RG != 2 && RG != 4) RG = 4;
        if (!exists(RG, KW)) { RG = 4; KW = nt2 ? (KW >= 2 ? 2 : 1) : (KW >= 4 ? 4 : KW >= 2 ? 2 : 1); }
    } else {
        // Cost model (microseconds) of one layer for every admissible (RG, S, KW); all inputs are plan quantities.
        //   a wave needs n = nch / (S KW) K steps; a step costs 1024 MFMA cycles (0.43 us) times the W waves that share the
        //   SIMD, but never less than the ~1.5 us a lone wave needs per step (load latency it cannot hide on its own);
        //   a workgroup puts RG KW / 4 waves on each SIMD; W = that x the workgroups on the busiest CU, in rounds when they
        //   exceed the CU's wave slots; a grid split adds the reduce launch and the slab round trip.
        constexpr double t_step = kTStep, t_lone = kTLone, t_fixed = kTFixed, t_reduce = kTReduce;
        const double plane_mb = (double)m_plan * d.Cout * 4e-6 * nphase;
        const double slots = nt2 ? 3. : 4.;                          // waves per SIMD the model counts with (see the note at the constants)
        double best = 1e30;
        for (int rg = 4; rg >= rg_min; rg >>= 1) {
            const long long blocks = (m_plan + 32 * rg - 1) / (32 * rg) * nblocks * nphase;
            for (int s = 1; s <= 16; s++) {
                if (s > 1 && (!d.slab || d.out2 || (size_t)s * plane_bytes + WSLAB_TAIL > d.slab_bytes)) break;      // (the reduce kernel knows one output)
                for (int kw = 1; kw <= 16; kw *= 2) {
                    if (!exists(rg, kw)) continue;
                    if ((s > 1 && nch / (s * kw) < min_steps) || (kw > 1 && nch / (s * kw) < kw_steps)) continue;
                    const int n = (nch + s * kw - 1) / (s * kw);
                    const double wps = rg * kw / 4.;                  // waves a workgroup puts on each SIMD
                    const long long wgs = blocks * s;
                    const int per_cu = (int)((wgs + 255) / 256);      // workgroups on the busiest CU
                    int fit = (int)(slots / wps);                     // workgroups a CU holds at once
                    if (fit < 1) fit = 1;
                    double t = 0.;
                    for (int left = per_cu; left > 0; left -= fit) {
                        const double W = (left < fit ? left : fit) * wps;
                        const double per_step = t_step * W > t_lone ? t_step * W : t_lone;
                        t += n * per_step + t_fixed;
                    }
                    if (s > 1) t += t_reduce + plane_mb * (s + 1.0) / 3.0;    // slab write + read at ~3 TB/s effective
                    t += 0.3 * (kw - 1) / (kw >= 8 ? 2. : 1.) + (rg < 4 ? 0.2 : 0.);   // partial tiles meet in LDS; weights fetched by more workgroups
                    if (t < best - 1e-9) { best = t; RG = rg; S = s; KW = kw; }
                }
            }
        }
    }
    const int mblocks = (int)((M + 32 * RG - 1) / (32 * RG));
    p.d.ksplit = S;
    p.d.nphase = nphase;
    p.grid = dim3(mblocks * nblocks, 1, nphase * S);          // 1-D over (pixel tile, N tile): the kernel re-labels it per XCD
    p.block = dim3(64 * RG * KW);
    p.nt = nt2 ? 2 : 1; p.rg = nt2 ? 4 : RG; p.kw = KW;
#define USC_PICK(NTV, RGV, KWV) (d.in_relu ? (void*)conv_mfma_kernel<NTV, RGV, KWV, true> : (void*)conv_mfma_kernel<NTV, RGV, KWV, false>)
    if (nt2) p.fn = KW >= 2 ? USC_PICK(2, 4, 2) : USC_PICK(2, 4, 1);
    else if (RG == 4) p.fn = KW == 4 ? USC_PICK(1, 4, 4) : KW == 2 ? USC_PICK(1, 4, 2) : USC_PICK(1, 4, 1);
    else if (RG == 2) p.fn = KW == 8 ? USC_PICK(1, 2, 8) : KW == 4 ? USC_PICK(1, 2, 4) : USC_PICK(1, 2, 2);
    else p.fn = KW == 16 ? USC_PICK(1, 1, 16) : KW == 8 ? USC_PICK(1, 1, 8) : USC_PICK(1, 1, 4);
#undef USC_PICK
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

extern "C" int usc_conv_plan_query(const usc_conv_desc* desc, usc_conv_plan_info* out) {
    if (!desc || !out) return conv_fail(USC_E_ARG, "usc_conv_plan_query: bad argument");
    LayerPlan p;
    int rc = plan_layer(*desc, p);
    if (rc) return rc;
    out->nt = p.nt; out->rowgroups = p.rg; out->kgroups = p.kw; out->ksplit = p.d.kind == USC_CONV_HEAD ? 1 : p.d.ksplit;
    out->winograd = p.wino ? (p.t3 ? 3 : p.lin ? 2 : 1) : 0;      // 2: blocks of 32 consecutive tiles, 3: transposed form
    out->workgroups = (int)(p.grid.x * p.grid.y * p.grid.z);
    out->threads = (int)p.block.x;
    out->launches = p.reduce ? 2 : 1;
    return USC_OK;
}

extern "C" int usc_conv_sequence_f32(const usc_conv_desc* descs, int n, usc_stream_t stream) {
    if (!descs || n < 0) return conv_fail(USC_E_ARG, "usc_conv_sequence_f32: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < n; i++) {
        LayerPlan p;
        int rc = plan_layer(descs[i], p);
        if (rc) return rc;
        if (p.wsplit && hipMemsetAsync(wslab_counters(p.d), 0, WSLAB_TAIL, st) != hipSuccess) return conv_fail(USC_E_LAUNCH, "hipMemsetAsync (arrival counters) failed");
        void* args[] = {&p.d};
        hipError_t e = hipLaunchKernel(p.fn, p.grid, p.block, args, 0, st);
        if (e == hipSuccess && p.reduce) e = hipLaunchKernel((void*)splitk_reduce_kernel, p.rgrid, dim3(256), args, 0, st);
        if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
    }
    return USC_OK;
}

// ---- the same layers as a hipGraph: one node per launch, chained in list order - one hipGraphLaunch replaces the ~85 kernel
// launches of a frame.  The graph is a LINEAR chain on purpose.  Round 1 gave it edges that followed the buffers (independent
// branches could overlap) and hit a host segfault inside hipGraphLaunch, worked around by never destroying a graph.  Root cause
// (profiles/r02/graph_segfault_root_cause.md: native backtrace + disassembly of the runtime torch bundles): an executable graph
// with parallel branches owns internal streams; at launch the runtime walks them, SKIPS each one that shares the launch stream's
// hardware queue and reads the next WITHOUT a bounds check - which internal stream collides depends on the process's whole
// history of stream creation, so the fault came and went with test order and with destroy / no destroy.  A chain needs no
// internal stream, the loop has nothing to walk.  Measured cost of giving up branch overlap: none (matting forward 1.59 ms as a
// branched graph, 1.56 ms as serial launches).
struct ConvGraph {
    hipGraph_t graph = nullptr;
    // One instantiation per stream the graph is launched on (kept from round 1; with chain graphs an exec could be shared).
    std::vector<std::pair<hipStream_t, hipGraphExec_t>> execs;
    std::vector<LayerPlan> plans;          // kernel arguments must outlive the graph
    int device = -1;                       // the device the graph was created (and is launched) on
    int mode = USC_CONV_LAUNCH_STREAM;     // how usc_conv_graph_launch enqueues the layers
    void* counters = nullptr;              // arrival counters (slab tail) of the layers that split K over two workgroups, if any
    // runs of same-shape Winograd layers that USC_CONV_LAUNCH_RUNS enqueues as ONE launch each (conv_wino_kernel<.., RUN = true>)
    struct Run { int first, count; usc_conv_run arg; void* fn; dim3 grid; };
    std::vector<Run> runs;
    unsigned* run_state = nullptr;         // per run: 4 control words + count x blocks arrival counters (device memory, owned)
};

// Layers [i, i + n) can go out as one run: the same kernel variant (8 x 4 blocks, K groups, no K split over workgroups), the same
// geometry, one image, dense outputs, each layer's input the output of the one before; residual tensors come from an earlier layer of the
// run (their blocks are finished transitively: a block waits for the same block one layer back) or from outside it.
static int find_run(const std::vector<LayerPlan>& pl, int i) {
    const LayerPlan& a = pl[i];
    if (a.fn != (void*)conv_wino_kernel<2, false, 0, false> || a.d.ksplit == 2 || (a.d.batch > 1) || a.d.dx[5] != 1) return 1;
    int n = 1;
    while (i + n < (int)pl.size() && n < USC_RUN_MAX) {
        const LayerPlan& b = pl[i + n];
        const LayerPlan& prev = pl[i + n - 1];
        if (b.fn != a.fn || b.d.ksplit == 2 || b.d.batch > 1 || b.d.IH != a.d.IH || b.d.IW != a.d.IW || b.d.Cin != a.d.Cin || b.d.Cout != a.d.Cout ||
            b.d.in != prev.d.out || (prev.d.out_ld && prev.d.out_ld != prev.d.Cout) || prev.d.out_c0 != 0 || a.d.Cout != a.d.Cin) break;
        bool ok = true;
        for (const float* r : {b.d.res1, b.d.res2}) {
            if (!r) continue;
            for (int k = i + n; k < (int)pl.size() && k < i + USC_RUN_MAX; k++) ok = ok && r != pl[k].d.out;     // never a tensor a later layer writes
        }
        if (b.d.res1 && b.d.res1_shift) ok = false;
        if (!ok) break;
        n++;
    }
    return n;
}

extern "C" int usc_conv_graph_create(const usc_conv_desc* descs, int n, void** handle) {
    if (!descs || n <= 0 || !handle) return conv_fail(USC_E_ARG, "usc_conv_graph_create: bad argument");
    ConvGraph* g = new ConvGraph();
    (void)hipGetDevice(&g->device);
    g->plans.resize(n);
    for (int i = 0; i < n; i++) {
        int rc = plan_layer(descs[i], g->plans[i]);
        if (rc) { delete g; return rc; }
        // arrival counters of Winograd layers split over two workgroups: zeroed here once; every launch leaves them zero (the second
        // arriver of a pair resets its counter), so no launch inherits a counter left odd and no memset sits in the launch path.
        // Handles that share a slab (the space-time memory network's parts) are created before any of them runs.
        if (g->plans[i].wsplit) {
            void* c = wslab_counters(g->plans[i].d);
            bool seen = false;                               // (layers may name different slabs: every distinct tail is zeroed once)
            for (int k = 0; k < i; k++) seen = seen || (g->plans[k].wsplit && wslab_counters(g->plans[k].d) == c);
            if (!g->counters) g->counters = c;
            if (!seen && hipMemset(c, 0, WSLAB_TAIL) != hipSuccess) { delete g; return conv_fail(USC_E_LAUNCH, "hipMemset (arrival counters) failed"); }
        }
    }
    // runs (USC_CONV_LAUNCH_RUNS): found once, their control words and arrival counters allocated and zeroed here (sequence number 1)
    {
        size_t words = 0;
        for (int i = 0; i < n;) {
            const int cnt = find_run(g->plans, i);
            if (cnt >= 2) {
                ConvGraph::Run r;
                r.first = i; r.count = cnt; r.fn = (void*)conv_wino_kernel<2, false, 0, true>;
                memset(&r.arg, 0, sizeof(r.arg));
                for (int k = 0; k < cnt; k++) r.arg.d[k] = g->plans[i + k].d;
                const usc_conv_desc& d0 = r.arg.d[0];
                const int TH = (d0.IH + 1) / 2, TW = (d0.IW + 1) / 2;
                r.arg.nlayers = cnt; r.arg.wgs = (int)g->plans[i].grid.x; r.arg.nblocks = ((TH + 3) / 4) * ((TW + 7) / 8);
                r.arg.ctl = reinterpret_cast<unsigned*>(words);              // offsets for now
                words += 4 + (size_t)cnt * r.arg.nblocks;
                g->runs.push_back(r);
            }
            i += cnt;
        }
        if (words) {
            if (hipMalloc(&g->run_state, words * 4) != hipSuccess || hipMemset(g->run_state, 0, words * 4) != hipSuccess) {
                delete g; return conv_fail(USC_E_LAUNCH, "hipMalloc (run state) failed");
            }
            int cus = 256;
            hipDeviceProp_t prop;
            if (hipGetDeviceProperties(&prop, g->device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
            for (auto& r : g->runs) {
                r.arg.ctl = g->run_state + (size_t)r.arg.ctl;
                r.arg.flags = r.arg.ctl + 4;
                const unsigned one = 1;
                if (hipMemcpy(r.arg.ctl + 2, &one, 4, hipMemcpyHostToDevice) != hipSuccess) { delete g; return conv_fail(USC_E_LAUNCH, "hipMemcpy (run state) failed"); }
                r.grid = dim3((unsigned)(r.arg.wgs < cus ? r.arg.wgs : cus));     // one workgroup per CU pulls items until the queue is empty
            }
        }
    }
    if (hipGraphCreate(&g->graph, 0) != hipSuccess) { delete g; return conv_fail(USC_E_LAUNCH, "hipGraphCreate failed"); }
    std::vector<hipGraphNode_t> last(n);                     // node that completes layer i
    std::vector<void*> kargs(n);
    for (int i = 0; i < n; i++) {
        LayerPlan& p = g->plans[i];
        std::vector<hipGraphNode_t> deps;
        if (i > 0) deps.push_back(last[i - 1]);              // linear chain (see above)
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
    *handle = g;
    return USC_OK;
}

extern "C" void usc_debug_arm_segv_backtrace(void);      // usc_debug.cpp (no-op unless USC_SEGV_BACKTRACE is set)

extern "C" int usc_conv_graph_launch(void* handle, usc_stream_t stream) {
    if (!handle) return conv_fail(USC_E_ARG, "usc_conv_graph_launch: null handle");
    usc_debug_arm_segv_backtrace();
    ConvGraph* g = (ConvGraph*)handle;
    hipStream_t st = (hipStream_t)stream;
    if (g->mode == USC_CONV_LAUNCH_STREAM || g->mode == USC_CONV_LAUNCH_RUNS) {    // the planned launches, one after the other on the stream
        size_t next_run = 0;
        for (int i = 0; i < (int)g->plans.size(); i++) {
            if (g->mode == USC_CONV_LAUNCH_RUNS && next_run < g->runs.size() && g->runs[next_run].first == i) {
                ConvGraph::Run& r = g->runs[next_run++];
                void* rargs[] = {&r.arg};
                hipError_t e = hipLaunchKernel(r.fn, r.grid, dim3(512), rargs, 0, st);
                if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
                i += r.count - 1;
                continue;
            }
            LayerPlan& p = g->plans[i];
            void* args[] = {&p.d};
            hipError_t e = hipLaunchKernel(p.fn, p.grid, p.block, args, 0, st);
            if (e == hipSuccess && p.reduce) e = hipLaunchKernel((void*)splitk_reduce_kernel, p.rgrid, dim3(256), args, 0, st);
            if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
        }
        return USC_OK;
    }
    hipGraphExec_t exec = nullptr;
    for (auto& se : g->execs)
        if (se.first == st) exec = se.second;
    if (!exec) {
        if (hipGraphInstantiate(&exec, g->graph, nullptr, nullptr, 0) != hipSuccess) return conv_fail(USC_E_LAUNCH, "hipGraphInstantiate failed");
        g->execs.emplace_back(st, exec);
    }
    hipError_t e = hipGraphLaunch(exec, st);
    if (e != hipSuccess) return conv_fail(USC_E_LAUNCH, hipGetErrorString(e));
    return USC_OK;
}

extern "C" int usc_conv_graph_set_mode(void* handle, int mode) {
    if (!handle || (mode != USC_CONV_LAUNCH_STREAM && mode != USC_CONV_LAUNCH_GRAPH && mode != USC_CONV_LAUNCH_RUNS)) return conv_fail(USC_E_ARG, "usc_conv_graph_set_mode: bad argument");
    ((ConvGraph*)handle)->mode = mode;
    return USC_OK;
}

extern "C" int usc_conv_graph_run_info(void* handle, int* nruns, int* layers_in_runs, unsigned* poll_timeouts) {
    if (!handle) return conv_fail(USC_E_ARG, "usc_conv_graph_run_info: null handle");
    ConvGraph* g = (ConvGraph*)handle;
    int layers = 0;
    unsigned bad = 0;
    for (auto& r : g->runs) {
        layers += r.count;
        unsigned v = 0;
        if (hipMemcpy(&v, r.arg.ctl + 3, 4, hipMemcpyDeviceToHost) != hipSuccess) return conv_fail(USC_E_LAUNCH, "usc_conv_graph_run_info: hipMemcpy failed");
        bad += v;
    }
    if (nruns) *nruns = (int)g->runs.size();
    if (layers_in_runs) *layers_in_runs = layers;
    if (poll_timeouts) *poll_timeouts = bad;
    return USC_OK;
}

extern "C" void usc_conv_graph_destroy(void* handle) {
    // An executable graph must not be destroyed while one of its launches is still queued or running: the runtime frees the
    // exec's kernel-argument blocks and its internal branch streams at once, and a later hipGraphLaunch on the same stream
    // then walks freed command objects (round 1 saw exactly that as a host segfault: ConvGraphNet.__del__ destroyed graphs
    // at garbage-collection time with launches in flight).  So: drain every stream the graph was launched on, then destroy.
    if (!handle) return;
    ConvGraph* g = (ConvGraph*)handle;
    int cur = -1;
    (void)hipGetDevice(&cur);
    if (g->device >= 0 && g->device != cur) (void)hipSetDevice(g->device);
    (void)hipDeviceSynchronize();          // both launch modes: stream launches of this handle may still be queued on any stream, and the
                                           // caller releases the layers' buffers right after this call (ConvGraphNet.close())
    for (auto& se : g->execs) (void)hipGraphExecDestroy(se.second);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    if (g->run_state) (void)hipFree(g->run_state);
    if (g->device >= 0 && g->device != cur && cur >= 0) (void)hipSetDevice(cur);
    delete g;
}
