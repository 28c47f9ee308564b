// Microbenchmarks behind the conv kernel's inner loop (build + run on the GPU box):
//   hipcc -O3 --offload-arch=gfx950 tools/dev/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int APITCH = 36;

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][128 * APITCH];
    __shared__ __attribute__((aligned(16))) float Bs[2][8 * 32 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 128 * APITCH; i += 256) (&As[0][0])[i] = (float)(i % 7) * 0.01f;
    for (int i = tid; i < 2 * 8 * 32 * 4; i += 256) (&Bs[0][0])[i] = (float)(i % 5) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    const int arow = (32 * wave + (lane & 31)) * APITCH + 4 * (lane >> 5);
    float4 st = make_float4(1.f, 2.f, 3.f, 4.f);
    const int q = tid & 7;
    for (int it = 0; it < iters; it++) {
        const int buf = it & 1;
#pragma unroll
        for (int g = 0; g < 4; g++) {
            float4 a, b;
            if (MODE >= 1) {
                a = *reinterpret_cast<const float4*>(&As[buf][arow + g * 8]);
                b = *reinterpret_cast<const float4*>(&Bs[buf][((g * 2 + (lane >> 5)) * 32 + (lane & 31)) * 4]);
            } else { a = st; b = st; }
#pragma unroll
            for (int j = 0; j < NACC; j++) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[j], 0, 0, 0);
            }
        }
        if (MODE >= 3) {
#pragma unroll
            for (int j = 0; j < 4; j++)
                *reinterpret_cast<float4*>(&As[buf ^ 1][((tid >> 3) + 32 * j) * APITCH + 4 * q]) = st;
            *reinterpret_cast<float4*>(&Bs[buf ^ 1][tid * 4]) = st;
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int j = 0; j < NACC; j++) for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NACC>
void run(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE, NACC><<<blocks, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) probe<MODE, NACC><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    double flop = (double)blocks * 4 * iters * 16 * NACC * 4096.0;
    printf("%-44s blocks %5d iters %4d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks, iters, ms, flop / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 512, 768, 1536}) {
        run<0, 1>("mfma only, 1 acc", blocks, 2000);
        run<0, 2>("mfma only, 2 acc", blocks, 1000);
        run<1, 1>("mfma + ds_read_b128 frags", blocks, 2000);
        run<2, 1>("mfma + reads + barrier/16 mfma", blocks, 2000);
        run<3, 1>("mfma + reads + 5 ds_write_b128 + barrier", blocks, 2000);
        run<3, 2>("same, 2 acc (32 mfma per barrier)", blocks, 1000);
    }
    return 0;
}
