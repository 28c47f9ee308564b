// Microbenchmark for DESIGN section 8, lead 1 ("a dataflow launch over runs of same-shape layers"): what does a LAYER BOUNDARY cost as a
// kernel boundary, and what as a flag hand-over inside one persistent launch?  A chain of L "layers" of NB = 256 workgroups (one per CU, 512
// threads): a workgroup reads the 16 KB outputs of three workgroups of the previous layer (its own index and both neighbours), computes for
// about T microseconds (a dependent FMA chain, standing in for the K loop), writes 16 KB.
//   (a) L launches on one stream - what the conv stack does today;
//   (b) ONE launch: (layer, block) items are handed out in layer-major order by an atomic counter; a consumer polls the ready flags of its
//       three producers (bounded polling: a time-out sets an error flag and goes on - the probe cannot hang the GPU); producers write their
//       tile with write-through stores (sc0 sc1), wait for them, then publish the flag; consumers read with sc0 sc1 loads (another XCD's L2
//       is not coherent with ours).  Items are handed out in order and an item's producers have lower indices, so no workgroup ever waits on
//       work that is not already owned by a running workgroup: no co-residency assumption, no grid barrier.
// Both forms must produce the same numbers (checked).   hipcc --offload-arch=gfx950 -O3 tools/dev/dataflow_probe.hip -o tools/dev/dataflow_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 256, THREADS = 512, TILE = 4096;          // floats per workgroup and layer (16 KB)
constexpr int SC = 0x11;                                       // sc0 | sc1

__device__ __forceinline__ float work(float v, int iters) {    // a dependent chain: ~4 cycles per FMA
    float a = v, b = v * 0.5f, c = v * 0.25f, d = v * 0.125f;
    for (int i = 0; i < iters; i++) { a = a * 0.999f + 0.001f; b = b * 0.998f + 0.002f; c = c * 0.997f + 0.003f; d = d * 0.996f + 0.004f; }
    return ((a + b) + (c + d)) * 0.25f;
}

template <bool COHERENT>
__device__ __forceinline__ void layer_tile(const float* __restrict__ in, float* __restrict__ out, int b, int iters) {
    const __amdgpu_buffer_rsrc_t ir = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, NB * TILE * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(out, 0, NB * TILE * 4, 0x00020000);
    const int bl = (b + NB - 1) % NB, br = (b + 1) % NB;
    for (int q = threadIdx.x; q < TILE / 4; q += THREADS) {
        const u32x4 m = __builtin_amdgcn_raw_buffer_load_b128(ir, (b * TILE + 4 * q) * 4, 0, COHERENT ? SC : 0);
        const u32x4 l = __builtin_amdgcn_raw_buffer_load_b128(ir, (bl * TILE + 4 * q) * 4, 0, COHERENT ? SC : 0);
        const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(ir, (br * TILE + 4 * q) * 4, 0, COHERENT ? SC : 0);
        u32x4 o;
        for (int k = 0; k < 4; k++) {
            const float v = 0.5f * __builtin_bit_cast(float, m[k]) + 0.25f * (__builtin_bit_cast(float, l[k]) + __builtin_bit_cast(float, r[k]));
            o[k] = __builtin_bit_cast(unsigned, work(v, iters));
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, orr, (b * TILE + 4 * q) * 4, 0, COHERENT ? SC : 0);
    }
}

__global__ __launch_bounds__(THREADS) void layer_kernel(const float* in, float* out, int iters) { layer_tile<false>(in, out, blockIdx.x, iters); }

// bufs: L + 1 tensors of NB * TILE floats, tensor 0 = input.  flags: L * NB ints (zero), ctl: {next item, error}
// mode 3 (control): the same kernel code launched once per layer - `base` is the layer, one item per workgroup, no waiting
__global__ __launch_bounds__(THREADS) void dataflow_kernel(float* bufs, int L, int iters, int* flags, int* ctl, int mode, int base) {
    __shared__ int s_item;
    for (int guard = 0; guard < (mode == 3 ? 1 : 4 * L + 8); guard++) {           // (a workgroup takes ~L items: the guard only bounds a broken run)
        // mode 2: static assignment - workgroup b owns block b of every layer (no imbalance, but every workgroup must be resident: the
        // probe's grid fits the GPU and nothing else runs; a product kernel could not assume that)
        if (mode != 2 && threadIdx.x == 0) s_item = atomicAdd(&ctl[0], 1);
        __syncthreads();
        const int item = mode == 3 ? base * NB + (int)blockIdx.x : mode == 2 ? guard * NB + (int)blockIdx.x : s_item;
        __syncthreads();
        if (item >= L * NB) return;
        const int l = item / NB, b = item - l * NB;
        if (mode == 4 && l > 0 && threadIdx.x < 64) {            // wave 0 polls all flags at once, in a loop every lane iterates the same number of times
            const int k = (int)threadIdx.x;
            const int p = (b + NB - 1 + (k < 3 ? k : 0)) % NB;
            bool ready = k >= 3;
            int polls = 0;
            for (;;) {
                if (!ready) ready = __hip_atomic_load(&flags[(l - 1) * NB + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (__all(ready) || ++polls >= (1 << 12)) break;
                __builtin_amdgcn_s_sleep(2);
            }
            if (polls >= (1 << 12) && k == 0) atomicExch(&ctl[1], 1);
        }
        if (mode > 0 && mode < 3 && l > 0 && threadIdx.x == 0) {             // wait for the three producers of the previous layer (one lane, one flag after the other)
            for (int k = 0; k < 3; k++) {
                const int p = (b + NB - 1 + k) % NB;
                int polls = 0;
                while (polls < (1 << 12) && __hip_atomic_load(&flags[(l - 1) * NB + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
                    __builtin_amdgcn_s_sleep(2);
                    polls++;
                }
                if (polls == (1 << 12)) atomicExch(&ctl[1], 1);     // (bounded: a broken hand-over shows as a count, not as a hang)
            }
        }
        __syncthreads();
        layer_tile<true>(bufs + (size_t)l * NB * TILE, bufs + (size_t)(l + 1) * NB * TILE, b, iters);
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): the write-through stores have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[l * NB + b], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int L = argc > 1 ? atoi(argv[1]) : 16;
    float *bufs, *bufs2; int *flags, *ctl;
    const size_t tensor = (size_t)NB * TILE;
    CK(hipMalloc(&bufs, (L + 1) * tensor * 4)); CK(hipMalloc(&bufs2, (L + 1) * tensor * 4));
    CK(hipMalloc(&flags, L * NB * 4)); CK(hipMalloc(&ctl, 8));
    std::vector<float> h(tensor);
    for (size_t i = 0; i < tensor; i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f;
    CK(hipMemcpy(bufs, h.data(), tensor * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bufs2, h.data(), tensor * 4, hipMemcpyHostToDevice));
    hipStream_t st; CK(hipStreamCreate(&st));
    const int reps = 30;
    printf("%d layers of %d workgroups x %d threads, 16 KB in / out per workgroup and layer; us per layer\n", L, NB, THREADS);
    printf("FMA chain | tiny kernel per layer | SAME code, one launch per layer | one launch, items from a counter | one launch, static items | hand-over saves | values differing / poll time-outs\n");
    for (int iters : {0, 200, 500, 1000}) {
        auto timed = [&](auto&& fn) -> double {
            fn();
            if (hipStreamSynchronize(st) != hipSuccess) return -1.;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) fn();
            if (hipStreamSynchronize(st) != hipSuccess) return -1.;
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps / L;
        };
        auto tiny = [&]() { for (int l = 0; l < L; l++) hipLaunchKernelGGL(layer_kernel, dim3(NB), dim3(THREADS), 0, st, bufs + l * tensor, bufs + (l + 1) * tensor, iters); };
        auto df = [&](int mode) {
            return [&, mode]() {
                hipMemsetAsync(flags, 0, L * NB * 4, st); hipMemsetAsync(ctl, 0, 8, st);
                if (mode == 3) { for (int l = 0; l < L; l++) hipLaunchKernelGGL(dataflow_kernel, dim3(NB), dim3(THREADS), 0, st, bufs2, L, iters, flags, ctl, mode, l); }
                else hipLaunchKernelGGL(dataflow_kernel, dim3(NB), dim3(THREADS), 0, st, bufs2, L, iters, flags, ctl, mode, 0);
            };
        };
        const double t_tiny = timed(tiny), t_same = timed(df(3)), t_static = timed(df(2)), t_par = argc > 2 ? timed(df(4)) : 0., t_queue = timed(df(1));   // (the queue form runs last: its result is compared)
        if (argc > 2) printf("          (items from a counter, wave 0 polling its flags in parallel with a uniform loop: %.2f)\n", t_par);
        int hc[2]; CK(hipMemcpy(hc, ctl, 8, hipMemcpyDeviceToHost));
        std::vector<float> ra(tensor), rb(tensor);
        CK(hipMemcpy(ra.data(), bufs + (size_t)L * tensor, tensor * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(rb.data(), bufs2 + (size_t)L * tensor, tensor * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < tensor; i++) bad += memcmp(&ra[i], &rb[i], 4) != 0;
        printf("%9d | %8.2f | %8.2f | %8.2f | %8.2f | %6.2f | %zu / %d\n", iters, t_tiny, t_same, t_queue, t_static, t_same - t_queue, bad, hc[1]);
    }
    return 0;
}
