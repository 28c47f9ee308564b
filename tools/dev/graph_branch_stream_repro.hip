// Standalone probe (no torch, no library) for the fault behind round 1's "graphs are retired, not destroyed" workaround.
// Evidence from the failing process (profiles/r02/graph_segfault_root_cause.md): SIGSEGV inside hipGraphLaunch, in the loop that
// assigns the executable graph's internal branch streams - it skips every internal stream that shares the launch stream's
// hardware queue and reads the next one WITHOUT a bounds check.  So: one graph with several parallel branches, instantiated once,
// then launched on many different user streams; a launch stream that maps to the same hardware queue as one of the exec's
// internal streams runs the loop past the end of the vector.
//   hipcc --offload-arch=gfx950 -O2 -o graph_branch_stream_repro graph_branch_stream_repro.hip && ./graph_branch_stream_repro [branches] [streams]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
__global__ void k(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main(int argc, char** argv) {
    const int branches = argc > 1 ? atoi(argv[1]) : 4, nstreams = argc > 2 ? atoi(argv[2]) : 16;
    float* buf; CK(hipMalloc(&buf, 64 * 4096 * sizeof(float))); CK(hipMemset(buf, 0, 64 * 4096 * sizeof(float)));
    std::vector<hipStream_t> st(nstreams);
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (auto& s : st) { hipLaunchKernelGGL(k, dim3(16), dim3(256), 0, s, buf, 4096); }      // every stream has a hardware queue now
    CK(hipDeviceSynchronize());
    hipGraph_t g; CK(hipGraphCreate(&g, 0));
    hipGraphNode_t root, join; std::vector<hipGraphNode_t> mid(branches);
    auto add = [&](hipGraphNode_t* n, const hipGraphNode_t* deps, size_t nd, int slot) {
        float* p = buf + slot * 4096; int cnt = 4096; void* args[] = {&p, &cnt};
        hipKernelNodeParams kp = {}; kp.func = (void*)k; kp.gridDim = dim3(16); kp.blockDim = dim3(256); kp.kernelParams = args;
        CK(hipGraphAddKernelNode(n, g, deps, nd, &kp));
    };
    add(&root, nullptr, 0, 0);
    for (int b = 0; b < branches; b++) add(&mid[b], &root, 1, 1 + b);                         // `branches` nodes that may run side by side
    add(&join, mid.data(), mid.size(), 0);
    hipGraphExec_t e; CK(hipGraphInstantiate(&e, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 3; rep++)
        for (int i = 0; i < nstreams; i++) { printf("launch on stream %d\n", i); fflush(stdout); CK(hipGraphLaunch(e, st[i])); CK(hipStreamSynchronize(st[i])); }
    CK(hipGraphLaunch(e, nullptr)); CK(hipDeviceSynchronize());
    printf("ok: %d branches, %d launch streams\n", branches, nstreams); return 0;
}
