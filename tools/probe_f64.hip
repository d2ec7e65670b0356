// Round-4 probes, memory-free, whole chip:
//   1. MADCLOCK  v_mad_u64_u32 issue: cycles per wave-instruction per SIMD from the shader clock (s_memtime) AND the shader clock itself
//      (s_memtime ticks per s_memrealtime tick of 100 MHz) while the chip is busy -> lanes * f / cycles-per-issue, the independent
//      check of the "measured issue peak" bench.py divides by (g16_diag_valu times the same instruction with HIP events only).
//   2. MUL30     the product's own Fp30<Bls12_381 Fq>::mul, dependent chain per lane, 8 / 4 / 2 waves per SIMD -> products/s.
//   3. MULF64    the candidate: 8 limbs of 48 bits held as doubles; a limb product is split exactly by two FMAs
//                (hi = fma(a, b, C) - C, lo = fma(a, b, -hi)); 64 + 64 limb products per Montgomery-shaped product, column carries by
//                the add-magic / subtract-magic floor.  This is an OPERATION-MIX probe (the instruction counts and dependencies of
//                a correct implementation; the limb values are not checked): if this mix is not >= 1.25x MUL30 there is nothing to build.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Igroth16_amd/csrc -Iinclude tools/probe_f64.hip -o tools/bin/probe_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "internal.hpp"
#include "fp30.hpp"
using namespace g16;
namespace g16 { void set_last_error(const char* w, hipError_t e, const char* f, int l) { printf("HIP error %s: %s (%s:%d)\n", w, hipGetErrorString(e), f, l); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void mad_clock_kernel(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    uint64_t r[8];
    for (int k = 0; k < 8; ++k) r[k] = a * (k + 1);
    const uint64_t t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
    }
    const uint64_t t1 = clock64(), w1 = wall_clock64();
    uint64_t acc = 0;
    for (int k = 0; k < 8; ++k) acc += r[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[(size_t)gridDim.x * 256] = t1 - t0; out[(size_t)gridDim.x * 256 + 1] = w1 - w0; }
}

typedef Bls12_381::Fq::Params FqP;
__global__ void mul30_kernel(const Fp<FqP>* __restrict__ in, Fp<FqP>* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp30<FqP> a = Fp30<FqP>::unpack(in[t & 1023].v), b = Fp30<FqP>::unpack(in[(t + 1) & 1023].v);
    for (int i = 0; i < iters; ++i) { a = a.mul(b); b = b.mul(a); }
    a.add(b).canonical_lt2p().pack(out[t].v);
}

// 8 x 48-bit limbs as doubles.  C: 1.5 * 2^(52+48) makes fma(a, b, C) round a 96-bit product to its upper 48 bits (+ C)
struct F64x8 { double l[8]; };
__device__ __forceinline__ void limb_mul_acc(double a, double b, double& colh, double& coll) {
    const double C = 0x1.8p100;
    const double hi = __fma_rn(a, b, C) - C;      // multiple of 2^48, exact
    const double lo = __fma_rn(a, b, -hi);         // exact remainder, |lo| <= 2^47
    colh += hi;                                    // < 2^53 * 2^48 after 32 terms: exact
    coll += lo;
}
__device__ __forceinline__ F64x8 mulf64(const F64x8& a, const F64x8& b, const double* __restrict__ pl, double pinv) {
    double H[16], L[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) H[c] = L[c] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) limb_mul_acc(a.l[i], b.l[j], H[i + j], L[i + j]);
    const double M48 = 0x1p48, I48 = 0x1p-48, MAGIC = 0x1.8p100;   // (x + MAGIC) - MAGIC = x rounded to a multiple of 2^48
    double carry = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {     // Montgomery rows: m = low 48 bits of (column * pinv), then += m * p
        const double col = L[i] + carry;
        const double q = __fma_rn(col, pinv, MAGIC) - MAGIC;          // upper part of col * pinv
        const double m = __fma_rn(col, pinv, -q);                      // low 48 bits (signed)
#pragma unroll
        for (int j = 0; j < 8; ++j) limb_mul_acc(m, pl[j], H[i + j], L[i + j]);
        const double t = L[i] + carry;                                 // now a multiple of 2^48
        carry = t * I48 + H[i] * I48;                                  // into the next column
    }
    F64x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {     // normalise the upper columns to 48-bit limbs
        const double v = L[8 + j] + carry;
        const double up = (v + MAGIC) - MAGIC;
        r.l[j] = v - up;
        carry = up * I48 + H[8 + j] * I48;
    }
    (void)M48;
    return r;
}
__global__ void mulf64_kernel(const double* __restrict__ in, double* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F64x8 a, b;
    double pl[8];
    for (int k = 0; k < 8; ++k) { a.l[k] = in[(t + k) & 1023]; b.l[k] = in[(t + 8 + k) & 1023]; pl[k] = in[(16 + k) & 1023]; }
    const double pinv = in[40];
    for (int i = 0; i < iters; ++i) { a = mulf64(a, b, pl, pinv); b = mulf64(b, a, pl, pinv); }
    double s = 0;
    for (int k = 0; k < 8; ++k) s += a.l[k] + b.l[k];
    out[t] = s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d nominal clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    {
        uint64_t* d_out;
        CK(hipMalloc(&d_out, sizeof(uint64_t) * (256 * 8 * 256 + 16)));
        for (int w : {8, 2}) {
            const int blocks = 256 * w, iters = 8192;
            hipLaunchKernelGGL(mad_clock_kernel, dim3(blocks), dim3(256), 0, 0, d_out, 64);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mad_clock_kernel, dim3(blocks), dim3(256), 0, 0, d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t cw[2];
            CK(hipMemcpy(cw, d_out + (size_t)blocks * 256, 16, hipMemcpyDeviceToHost));
            const double instr = (double)iters * 8;                 // wave-instructions issued by one wave
            const double mhz = (double)cw[0] / ((double)cw[1] / 100.0);   // s_memtime ticks per microsecond of s_memrealtime (100 MHz)
            const double cyc_per_issue = (double)cw[0] / (instr * w);     // shader cycles per wave-instruction per SIMD (w waves share it)
            const double lanes = 256.0 * 4 * 64;
            printf("MADCLOCK waves/SIMD=%d  kernel %.3f ms  shader clock %.0f MHz (s_memtime / s_memrealtime)  %.2f cycles per wave-instruction per SIMD  "
                   "=> %.2f T mad/s from clock x issue rate; %.2f T mad/s from the event time\n", w, ms, mhz, cyc_per_issue,
                   lanes * mhz * 1e6 / cyc_per_issue / 1e12, lanes / 64.0 * instr * w * 64.0 / (ms * 1e-3) / 1e12 / 1.0);
        }
        (void)hipFree(d_out);
    }
    {
        Fp<FqP>* d_in; Fp<FqP>* d_o; double* d_din; double* d_do;
        const uint32_t T = 256 * 4 * 64 * 8;
        CK(hipMalloc(&d_in, sizeof(Fp<FqP>) * 1024)); CK(hipMalloc(&d_o, sizeof(Fp<FqP>) * T));
        CK(hipMalloc(&d_din, 8 * 1024)); CK(hipMalloc(&d_do, 8 * T));
        std::vector<uint32_t> h(1024 * 12);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)(i * 2654435761u) >> ((i % 12) == 11 ? 4 : 0);
        CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        std::vector<double> hd(1024);
        for (size_t i = 0; i < hd.size(); ++i) hd[i] = (double)((i * 2654435761ull) & 0xffffffffffffull);
        hd[40] = 123456789012345.0;
        CK(hipMemcpy(d_din, hd.data(), 8 * 1024, hipMemcpyHostToDevice));
        for (int w : {8, 4, 2}) {
            const uint32_t threads = 256 * 4 * 64 * w;
            const int iters = 128;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mul30_kernel, dim3(threads / 256), dim3(256), 0, 0, d_in, d_o, iters);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double muls = (double)threads * iters * 2;
            printf("MUL30   waves/SIMD<=%d  %.3f ms  %.2f G products/s (30-bit limbs x13, v_mad_u64_u32)\n", w, ms, muls / ms / 1e6);
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(mulf64_kernel, dim3(threads / 256), dim3(256), 0, 0, d_din, d_do, iters);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
            }
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("MULF64  waves/SIMD<=%d  %.3f ms  %.2f G products/s (48-bit limbs x8 in doubles, v_fma_f64 split; operation-mix probe)\n", w, ms, muls / ms / 1e6);
        }
    }
    return 0;
}
