// How does gfx950 issue a DEPENDENT v_mad_u64_u32 chain, and what does occupancy buy?  (round 6: the product-scanning field products
// are one serial multiply-add chain per product.)
//   chains = 1, 2, 4, 8: that many independent accumulators, round-robin, each step depending on the same accumulator's last step
//   product kernels: Fp30<BLS12-381 Fq>::mul / sqr / mul_sub_fused in a dependent loop (x = x * y), as the library builds them
//     (this file compiled twice: product scanning as generated assembly, and -DG16_NO_FIPS = round 5's operand-scanning form)
// waves per SIMD = workgroups of 256 lanes per compute unit (register use permitting).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 [-DG16_NO_FIPS] tools/probe_chain.hip -o tools/bin/probe_chain[_nofips]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../groth16_amd/csrc/curve.hpp"
#include "../groth16_amd/csrc/fp30.hpp"
using namespace g16;
typedef Fp30<Bls12_381FqP> F;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int CH>
__global__ __launch_bounds__(256) void k_chain(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    uint64_t r[8];
    for (int i = 0; i < 8; ++i) r[i] = a + i;
    for (int i = 0; i < iters; ++i) {
        if (CH == 1)
            asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                         "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                         : "+v"(r[0]) : "v"(a), "v"(b) : "vcc");
        else if (CH == 2)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1\n\t"
                         "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1\n\tv_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_mad_u64_u32 %1, vcc, %2, %3, %1"
                         : "+v"(r[0]), "+v"(r[1]) : "v"(a), "v"(b) : "vcc");
        else if (CH == 4)
            asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3\n\t"
                         "v_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_mad_u64_u32 %1, vcc, %4, %5, %1\n\tv_mad_u64_u32 %2, vcc, %4, %5, %2\n\tv_mad_u64_u32 %3, vcc, %4, %5, %3"
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]) : "v"(a), "v"(b) : "vcc");
        else
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\t"
                         "v_mad_u64_u32 %3, vcc, %8, %9, %3\n\tv_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\t"
                         "v_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                         : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t s = 0;
    for (int i = 0; i < 8; ++i) s += r[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// KIND 0: x = x * y; 1: x = x^2 * ... (sqr); 2: x = x y - y x' (fused two-sweep form); 3: two INDEPENDENT products per iteration
template <int KIND>
__global__ __launch_bounds__(256) void k_prod(uint32_t* out, int iters) {
    F x, y, z;
    for (int i = 0; i < F::NL; ++i) {
        x.l[i] = (threadIdx.x * 2654435761u + i * 40503u) & F::MASK;
        y.l[i] = (blockIdx.x * 2246822519u + i * 3266489917u + 7u) & F::MASK;
        z.l[i] = (threadIdx.x * 374761393u + i * 668265263u + 11u) & F::MASK;
    }
    x.l[F::NL - 1] &= 0xfffffu; y.l[F::NL - 1] &= 0xfffffu; z.l[F::NL - 1] &= 0xfffffu;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) x = x.mul(y);
        else if (KIND == 1) x = x.sqr();
        else if (KIND == 2) x = F::mul_sub_fused(x, y, z, F::zero().template sub<2>(F::zero()));
        else { x = x.mul(y); z = z.mul(y); }
    }
    uint32_t s = 0;
    for (int i = 0; i < F::NL; ++i) s ^= x.l[i] ^ z.l[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K>
static int time_kernel(K launch, double* ms_out) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(4);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    launch(0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    *ms_out = ms;
    return 0;
}

int main() {
    void* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * (256 * 8 * 256 + 16)));
#ifdef G16_NO_FIPS
    const char* form = "operand scanning (round 5)";
#else
    const char* form = "product scanning (generated assembly)";
#endif
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int blocks = 256 * w, iters = 4096;
        double ms;
#ifndef G16_NO_FIPS
#define CHAIN(CHN)                                                                                                                       \
        if (time_kernel([&](int it) { hipLaunchKernelGGL((k_chain<CHN>), dim3(blocks), dim3(256), 0, 0, (uint64_t*)d_out, it ? it : iters); }, &ms)) return 1; \
        printf("CHAIN chains=%d waves/SIMD=%d  %.2f T mad/s\n", CHN, w, 65536.0 * iters * 8 * w / (ms * 1e-3) / 1e12);
        CHAIN(1) CHAIN(2) CHAIN(4) CHAIN(8)
#endif
#define PROD(KIND, NAME, PER)                                                                                                              \
        if (time_kernel([&](int it) { hipLaunchKernelGGL((k_prod<KIND>), dim3(blocks), dim3(256), 0, 0, (uint32_t*)d_out, it ? it : 512); }, &ms)) return 1; \
        printf("PROD  %-40s %-10s waves/SIMD=%d  %.1f G products/s\n", form, NAME, w, 65536.0 * 512 * PER * w / (ms * 1e-3) / 1e9);
        PROD(0, "mul", 1) PROD(1, "sqr", 1) PROD(2, "mul_sub", 1) PROD(3, "2 x mul", 2)
    }
    return 0;
}
