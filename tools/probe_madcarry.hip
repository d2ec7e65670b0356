// Does the carry-out SGPR pair of v_mad_u64_u32 serialise a wave's multiply-adds?  r01_ubench.txt: one wave issues one every 12 cycles
// whatever the occupancy.  The compiler gives EVERY v_mad_u64_u32 of a kernel the same scalar pair for its (unused) carry-out.  Three forms
// of the same 8 independent chains per wave: carry-out always to vcc, always to ONE sgpr pair, to EIGHT different pairs round-robin.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_madcarry.hip -o tools/bin/probe_madcarry
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    uint64_t r0 = a, r1 = a + 1, r2 = a + 2, r3 = a + 3, r4 = a + 4, r5 = a + 5, r6 = a + 6, r7 = a + 7;
    const uint64_t t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                         "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                         "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");
        } else if (MODE == 1) {
            asm volatile("v_mad_u64_u32 %0, s[40:41], %8, %9, %0\n\tv_mad_u64_u32 %1, s[40:41], %8, %9, %1\n\tv_mad_u64_u32 %2, s[40:41], %8, %9, %2\n\t"
                         "v_mad_u64_u32 %3, s[40:41], %8, %9, %3\n\tv_mad_u64_u32 %4, s[40:41], %8, %9, %4\n\tv_mad_u64_u32 %5, s[40:41], %8, %9, %5\n\t"
                         "v_mad_u64_u32 %6, s[40:41], %8, %9, %6\n\tv_mad_u64_u32 %7, s[40:41], %8, %9, %7"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "s40", "s41");
        } else {
            asm volatile("v_mad_u64_u32 %0, s[40:41], %8, %9, %0\n\tv_mad_u64_u32 %1, s[42:43], %8, %9, %1\n\tv_mad_u64_u32 %2, s[44:45], %8, %9, %2\n\t"
                         "v_mad_u64_u32 %3, s[46:47], %8, %9, %3\n\tv_mad_u64_u32 %4, s[48:49], %8, %9, %4\n\tv_mad_u64_u32 %5, s[50:51], %8, %9, %5\n\t"
                         "v_mad_u64_u32 %6, s[52:53], %8, %9, %6\n\tv_mad_u64_u32 %7, s[54:55], %8, %9, %7"
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(a), "v"(b)
                         : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
        }
    }
    const uint64_t t1 = clock64();
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[(size_t)gridDim.x * 256] = t1 - t0;
}

template <int MODE>
static int run(uint64_t* d_out, int w, const char* name) {
    const int blocks = 256 * w, iters = 8192;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d_out, 64);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cyc = 0;
    CK(hipMemcpy(&cyc, d_out + (size_t)blocks * 256, 8, hipMemcpyDeviceToHost));
    const double instr = (double)iters * 8;
    printf("MADCARRY %-28s waves/SIMD=%d  wave0 %.2f shader cycles per multiply-add  whole chip %.2f T mad/s\n", name, w, (double)cyc / instr,
           65536.0 * instr * w / (ms * 1e-3) / 1e12);
    return 0;
}

int main() {
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * (256 * 8 * 256 + 16)));
    for (int w : {1, 2, 4, 8}) {
        run<0>(d_out, w, "carry-out -> vcc");
        run<1>(d_out, w, "carry-out -> one sgpr pair");
        run<2>(d_out, w, "carry-out -> 8 sgpr pairs");
    }
    return 0;
}
