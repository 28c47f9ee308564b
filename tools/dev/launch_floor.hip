// Microbenchmark: cost of a dependent kernel launch - stream back-to-back vs hipGraph chain; empty kernel vs 256-WG kernel
// that does one global load + store per thread.  Build: hipcc --offload-arch=gfx950 -O3 tools/dev/launch_floor.hip -o tools/dev/launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void empty_kernel(float* p) { if (p == nullptr) p[0] = 0.f; }
__global__ void touch_kernel(float* p) { const int i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = p[i] + 1.f; }

int main() {
    float* buf;
    CK(hipMalloc(&buf, 1024 * 256 * sizeof(float)));
    CK(hipMemset(buf, 0, 1024 * 256 * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const int N = 2000;
    for (int variant = 0; variant < 3; variant++) {
        const dim3 grid(variant == 0 ? 1 : variant == 1 ? 256 : 1024), block(256);
        void* fn = variant == 0 ? (void*)empty_kernel : (void*)touch_kernel;
        void* args[] = {&buf};
        for (int i = 0; i < 10; i++) CK(hipLaunchKernel(fn, grid, block, args, 0, st));
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; i++) CK(hipLaunchKernel(fn, grid, block, args, 0, st));
        CK(hipStreamSynchronize(st));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        // graph chain of 100 nodes
        hipGraph_t g; CK(hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        for (int i = 0; i < 100; i++) {
            hipKernelNodeParams kp = {};
            kp.func = fn; kp.gridDim = grid; kp.blockDim = block; kp.kernelParams = args;
            hipGraphNode_t node;
            CK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
        hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; i++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < 20; i++) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        double gus = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000;
        printf("%-28s stream %.2f us / launch, graph chain %.2f us / node\n",
               variant == 0 ? "empty kernel, 1 WG" : variant == 1 ? "load+store, 256 WGs" : "load+store, 1024 WGs", us, gus);
    }
    return 0;
}
