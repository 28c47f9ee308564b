// buffer_store_dwordx4 with a REGISTER scalar offset, followed directly by a vector write to its data registers: does the store
// still write the old values?  (gfx950; found in the Winograd epilogue, profiles/r02/wino_phase_timing.md.)  Three variants per
// launch: scalar offset in an SGPR + overwrite in the next instruction; the same with one s_nop in between; scalar offset = 0.
//   hipcc --offload-arch=gfx950 -O2 -o store_x4_hazard store_x4_hazard.hip && ./store_x4_hazard
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void k(unsigned* out, int bytes, int soff) {
    const unsigned long long a = (unsigned long long)out;
    const v4i rsrc = {(int)(unsigned)a, (int)((a >> 32) & 0xffff), bytes, 0x00020000};
    const int lane = threadIdx.x, voff = (blockIdx.x * 64 + lane) * 16;
    u32x4 v = {0x11110000u + lane, 0x22220000u + lane, 0x33330000u + lane, 0x44440000u + lane};
    // (the data registers are overwritten by the compiler-generated instructions right behind the asm statement: v = ~v)
    if (MODE == 0) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (MODE == 1) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 0" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    if (MODE == 2) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" :: "v"(v), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    v = ~v;
    asm volatile("s_waitcnt vmcnt(0)" :: "v"(v) : "memory");
}

int main() {
    const int nblk = 4096, n = nblk * 64 * 4, soff = 0;
    unsigned* d;
    (void)hipMalloc(&d, (size_t)n * 4);
    std::vector<unsigned> h(n);
    for (int mode = 0; mode < 3; mode++) {
        int bad_total = 0;
        for (int rep = 0; rep < 20; rep++) {
            (void)hipMemset(d, 0, (size_t)n * 4);
            if (mode == 0) k<0><<<nblk, 64>>>(d, n * 4, soff);
            if (mode == 1) k<1><<<nblk, 64>>>(d, n * 4, soff);
            if (mode == 2) k<2><<<nblk, 64>>>(d, n * 4, soff);
            (void)hipMemcpy(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < n; i++) {
                const unsigned lane = (i / 4) % 64, want = (0x11110000u * (i % 4 + 1)) + lane;
                if (h[i] != want) { if (bad_total < 4) printf("  mode %d: dword %d of lane %u holds %08x, wanted %08x\n", mode, i % 4, lane, h[i], want); bad_total++; }
            }
        }
        printf("mode %d (%s): %d wrong dwords in 20 launches of %d stores\n", mode,
               mode == 0 ? "register offset, overwrite at once" : mode == 1 ? "s_nop 0 in between" : "s_nop 1 in between", bad_total, nblk * 64);
    }
    return 0;
}
