// Standalone probe (no torch, no library): does destroying executable graphs disturb later graph launches on ROCm 7.2?
// The call sequence of ConvGraphNet's lifetime: graphs of ~50 kernel nodes with branch edges, one instantiation per stream,
// launches on the null stream and two created streams, device drained, execs and graph destroyed, new graphs created while a
// few long-lived ones keep being launched.  Exit code 0 and "ok" = no fault in `rounds` rounds.
//   hipcc --offload-arch=gfx950 -O2 -o graph_destroy_repro graph_destroy_repro.hip && ./graph_destroy_repro [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
struct Desc { float* p; int n; float add; char pad[200]; };                    // a by-value struct argument, like usc_conv_desc
__global__ void k(Desc d) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < d.n) d.p[i] += d.add; }
struct G { hipGraph_t g = nullptr; std::vector<std::pair<hipStream_t, hipGraphExec_t>> ex; };
static G* make(float* buf, int nodes) {
    G* g = new G(); CK(hipGraphCreate(&g->g, 0));
    std::vector<hipGraphNode_t> nd(nodes);
    for (int i = 0; i < nodes; i++) {
        Desc d{buf + (i % 4) * 4096, 4096, 1.f, {0}}; void* args[] = {&d};
        hipKernelNodeParams kp = {}; kp.func = (void*)k; kp.gridDim = dim3(16); kp.blockDim = dim3(256); kp.kernelParams = args;
        std::vector<hipGraphNode_t> deps; if (i > 0) deps.push_back(nd[i - 1]); if (i > 4 && i % 3 == 0) deps.push_back(nd[i - 4]);
        if (i % 5 == 2) deps.clear();                                             // a fresh branch now and then
        CK(hipGraphAddKernelNode(&nd[i], g->g, deps.data(), deps.size(), &kp));
    }
    return g;
}
static void launch(G* g, hipStream_t s) {
    hipGraphExec_t e = nullptr; for (auto& p : g->ex) if (p.first == s) e = p.second;
    if (!e) { CK(hipGraphInstantiate(&e, g->g, nullptr, nullptr, 0)); g->ex.emplace_back(s, e); }
    CK(hipGraphLaunch(e, s));
}
static void destroy(G* g) { CK(hipDeviceSynchronize()); for (auto& p : g->ex) CK(hipGraphExecDestroy(p.second)); CK(hipGraphDestroy(g->g)); delete g; }
int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 200;
    float* buf; CK(hipMalloc(&buf, 4 * 4096 * sizeof(float))); CK(hipMemset(buf, 0, 4 * 4096 * sizeof(float)));
    hipStream_t s[3] = {nullptr, nullptr, nullptr}; CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[2], hipStreamNonBlocking));
    std::vector<G*> keep; for (int i = 0; i < 3; i++) keep.push_back(make(buf, 54));
    for (int r = 0; r < rounds; r++) {
        std::vector<G*> tmp; for (int i = 0; i < 1 + r % 7; i++) tmp.push_back(make(buf, 20 + (r * 7 + i * 13) % 100));
        for (int rep = 0; rep < 2; rep++) for (G* g : tmp) { launch(g, s[(r + rep) % 3]); CK(hipStreamSynchronize(s[(r + rep) % 3])); }
        for (G* g : keep) { launch(g, s[r % 3]); CK(hipStreamSynchronize(s[r % 3])); }
        if (r % 2 == 0) { G* never = make(buf, 30); destroy(never); }                // a graph that was never instantiated
        for (G* g : tmp) destroy(g);
        if (r % 10 == 9) { destroy(keep[r / 10 % 3]); keep[r / 10 % 3] = make(buf, 54); }
    }
    CK(hipDeviceSynchronize()); printf("ok: %d rounds\n", rounds); return 0;
}
