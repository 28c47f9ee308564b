// Microbenchmark: sustained v_mfma_f32_32x32x2_f32 rate for dependent chains (1 accumulator) vs interleaved accumulators,
// at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_chain.hip -o tools/dev/mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; u++)
#pragma unroll
            for (int j = 0; j < NACC; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; j++)
        for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(int wg_per_cu, float* out) {
    const int iters = 4000;
    const int grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(chain<NACC>, dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(chain<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 16 * 4096.0;
    printf("acc=%d waves/SIMD=%d : %.3f ms  %.1f TF/s\n", NACC, wg_per_cu, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    for (int w = 1; w <= 8; w *= 2) { run<1>(w, out); run<2>(w, out); run<4>(w, out); }
    return 0;
}
