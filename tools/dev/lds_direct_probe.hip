#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// each lane fetches 16 bytes from an arbitrary global offset straight into LDS (lane-linear: base + lane * 16)
__global__ __launch_bounds__(64) void probe(const float* src, int nbytes, const int* offs, float* out) {
    __shared__ __attribute__((aligned(16))) float L[64 * 4 * 2];
    const int lane = threadIdx.x;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, L + 256, 16, offs[lane], 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int k = 0; k < 4; k++) out[lane * 4 + k] = L[256 + lane * 4 + k];
}
int main() {
    const int N = 4096;
    std::vector<float> h(N); for (int i = 0; i < N; i++) h[i] = (float)i;
    std::vector<int> o(64); for (int l = 0; l < 64; l++) o[l] = ((l * 37) % 1000) * 16; o[5] = 0x7fff0000;
    float *ds, *dout; int* dof;
    hipMalloc(&ds, N * 4); hipMalloc(&dout, 256 * 4); hipMalloc(&dof, 64 * 4);
    hipMemcpy(ds, h.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(dof, o.data(), 256, hipMemcpyHostToDevice);
    hipMemset(dout, 0xff, 1024);
    probe<<<1, 64>>>(ds, N * 4, dof, dout);
    std::vector<float> r(256); hipMemcpy(r.data(), dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 4; k++) { float want = l == 5 ? 0.f : (float)(o[l] / 4 + k); if (r[l * 4 + k] != want) { if (bad < 5) printf("lane %d k %d got %f want %f\n", l, k, r[l*4+k], want); bad++; } }
    printf("lds direct b128: %s (%d bad)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
