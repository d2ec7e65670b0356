// Instruction-throughput and accumulate-variant micro-benchmarks for gfx950 (run on the GPU box).
// Build: see tools/build_ubench.sh.  Prints one line per measurement.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "internal.hpp"
#include "fp30.hpp"
using namespace g16;
namespace g16 { void set_last_error(const char* w, hipError_t e, const char* f, int l) { printf("HIP error %s: %s (%s:%d)\n", w, hipGetErrorString(e), f, l); } }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

enum { OP_MAD64 = 0, OP_MULLO, OP_MULHI, OP_ADDC, OP_LSHLADD64, OP_MOV, OP_MAD24, OP_FMA64, OP_ADD32, OP_MADLO, OP_COUNT };
static const char* OPN[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co+v_addc(pair)", "v_lshl_add_u64", "v_mov_b32",
                            "v_mad_u32_u24", "v_fma_f64", "v_add_u32", "v_mad_u64_u32(dep-chain x8 indep)"};

template <int OP>
__global__ __launch_bounds__(256) void ub_kernel(uint64_t* out, int iters) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    uint64_t r[8];
    double d[8];
    for (int k = 0; k < 8; ++k) { r[k] = a * (k + 1); d[k] = (double)(a + k); }
    uint32_t lo[8], hi[8];
    for (int k = 0; k < 8; ++k) { lo[k] = a + k; hi[k] = b + k; }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (OP == OP_MAD64 || OP == OP_MADLO) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
            else if (OP == OP_MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(b));
            else if (OP == OP_MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(b));
            else if (OP == OP_ADDC) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(lo[k]), "+v"(hi[k]) : "v"(a), "v"(b) : "vcc");
            else if (OP == OP_LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(r[k]) : "v"(r[(k + 1) & 7]));
            else if (OP == OP_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(lo[k]) : "v"(hi[k]));
            else if (OP == OP_MAD24) asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(lo[k]) : "v"(a), "v"(b));
            else if (OP == OP_FMA64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[k]) : "v"(d[(k + 1) & 7]));
            else if (OP == OP_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(lo[k]) : "v"(b));
        }
    }
    long long t1 = clock64();
    uint64_t acc = 0;
    for (int k = 0; k < 8; ++k) acc += r[k] + lo[k] + hi[k] + (uint64_t)d[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc + (uint64_t)(t1 - t0);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[(size_t)gridDim.x * 256] = (uint64_t)(t1 - t0);
}

template <int OP>
static int run_op(uint64_t* d_out, int waves_per_simd) {
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (4 waves per block = 1 per SIMD) x waves_per_simd
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ub_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_out, 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((ub_kernel<OP>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    uint64_t cyc = 0;
    CK(hipMemcpy(&cyc, d_out + (size_t)blocks * 256, 8, hipMemcpyDeviceToHost));
    const double ninstr = (double)iters * 8 * (OP == OP_ADDC ? 2 : 1);
    // per SIMD: waves_per_simd waves each issuing ninstr wave-instructions
    const double wave_instr_per_simd = ninstr * waves_per_simd;
    printf("UBENCH op=%-36s waves/SIMD=%d  time=%.3f ms  wave0_cycles/instr=%.2f  ns_per_waveinstr_per_SIMD=%.3f (=> cycles@2.4GHz %.2f)\n",
           OPN[OP], waves_per_simd, ms, (double)cyc / ninstr, ms * 1e6 / wave_instr_per_simd, ms * 1e6 / wave_instr_per_simd * 2.4);
    return 0;
}

// ---- accumulate variants -------------------------------------------------------------------
template <class F>
__global__ __launch_bounds__(128) void acc_probe(const Affine<F>* __restrict__ bases, uint32_t nb, uint32_t len, XYZZ<F>* __restrict__ out) {
    const uint32_t t = blockIdx.x * 128 + threadIdx.x;
    XYZZ<F> acc = XYZZ<F>::identity();
    uint32_t x = t * 2654435761u + 1u;
    for (uint32_t e = 0; e < len; ++e) {
        x = x * 1664525u + 1013904223u;
        Affine<F> p = bases[(x >> 4) % nb];
        if (x & 1) p.y = p.y.neg();
        acc.add_affine(p);
    }
    out[t] = acc;
}

template <class F>
__global__ void mul_probe(const F* __restrict__ in, F* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    F a = in[t], b = in[t + 1];
    for (int i = 0; i < iters; ++i) { a = a * b; b = b * a; }
    out[t] = a + b;
}

template <class C, class F>
static int run_acc(const char* name, int g2) {
    const uint32_t nb = 1 << 16, len = 64;
    const uint32_t threads = 256 * 4 * 64 * 4;  // 4 waves per SIMD
    Affine<F>* d_b; XYZZ<F>* d_o;
    CK(hipMalloc(&d_b, sizeof(Affine<F>) * nb));
    CK(hipMalloc(&d_o, sizeof(XYZZ<F>) * threads));
    if (synth_bases_device<C>(g2, 1, 0, nb, d_b, 0) != 0) { printf("synth failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((acc_probe<F>), dim3(threads / 128), dim3(128), 0, 0, d_b, nb, len, d_o);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double adds = (double)threads * len;
    printf("ACCPROBE %-28s variant=%s  %.3f ms for %.0f mixed adds => %.2f Gadd/s  (%.1f ns per add per SIMD-lane-slot)\n", name, G16_VARIANT,
           ms, adds, adds / ms / 1e6, ms * 1e6 / (adds / (256.0 * 4 * 64)));
    // field-mul throughput
    {
        typedef typename C::Fq Fq;
        Fq *d_in, *d_out;
        const uint32_t T = 256 * 4 * 64 * 4;
        CK(hipMalloc(&d_in, sizeof(Fq) * (T + 1)));
        CK(hipMalloc(&d_out, sizeof(Fq) * T));
        for (uint32_t off = 0; off < T + 1; off += 65536) {
            const uint32_t cnt = (T + 1 - off) < 65536 ? (T + 1 - off) : 65536;
            CK(hipMemcpy(d_in + off, d_b, sizeof(Fq) * cnt, hipMemcpyDeviceToDevice));
        }
        const int iters = 256;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((mul_probe<Fq>), dim3(T / 256), dim3(256), 0, 0, d_in, d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double muls = (double)T * iters * 2;
        printf("MULPROBE %-28s variant=%s  %.3f ms for %.0f Fq muls => %.2f Gmul/s  (%.0f SIMD-cycles@2.4GHz per wave-mul)\n", name, G16_VARIANT, ms,
               muls, muls / ms / 1e6, ms * 1e-3 * 2.4e9 / (muls / 64 / 1024));
        (void)hipFree(d_in); (void)hipFree(d_out);
    }
    (void)hipFree(d_b); (void)hipFree(d_o);
    return 0;
}

template <class F30>
__global__ __launch_bounds__(128) void acc30_probe(const Affine<typename F30::Std>* __restrict__ bases, uint32_t nb, uint32_t len,
                                                   XYZZ<typename F30::Std>* __restrict__ out) {
    const uint32_t t = blockIdx.x * 128 + threadIdx.x;
    Acc30<F30> acc = Acc30<F30>::identity();
    uint32_t x = t * 2654435761u + 1u;
    for (uint32_t e = 0; e < len; ++e) {
        x = x * 1664525u + 1013904223u;
        const Affine<typename F30::Std> p = bases[(x >> 4) % nb];
        const F30 px = F30::from_packed(p.x);
        F30 py = F30::from_packed(p.y);
        if (x & 1) py = py.neg2();
        acc.add_affine(px, py);
    }
    out[t] = acc.to_std();
}
template <class P>
__global__ void mul30_probe(const Fp<P>* __restrict__ in, Fp<P>* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fp30<P> a = Fp30<P>::unpack(in[t].v), b = Fp30<P>::unpack(in[t + 1].v);
    for (int i = 0; i < iters; ++i) { a = a.mul(b); b = b.mul(a); }
    a.add(b).canonical_lt2p().pack(out[t].v);
}

template <class C>
static int run_acc30_g2(const char* name, int waves_per_simd) {
    typedef typename C::Fq2 F;
    typedef Fp2x30<typename C::Fq::Params> F30;
    const uint32_t nb = 1 << 15, len = 48;
    const uint32_t threads = 256 * 4 * 64 * waves_per_simd;
    Affine<F>* d_b; XYZZ<F>* d_o;
    CK(hipMalloc(&d_b, sizeof(Affine<F>) * nb));
    CK(hipMalloc(&d_o, sizeof(XYZZ<F>) * threads));
    if (synth_bases_device<C>(1, 1, 0, nb, d_b, 0) != 0) { printf("synth failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((acc30_probe<F30>), dim3(threads / 128), dim3(128), 0, 0, d_b, nb, len, d_o);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double adds = (double)threads * len;
    printf("ACC30PROBE %-20s variant=%s waves/SIMD=%d  %.3f ms for %.0f mixed adds => %.2f Gadd/s\n", name, G16_VARIANT, waves_per_simd, ms, adds,
           adds / ms / 1e6);
    (void)hipFree(d_b); (void)hipFree(d_o);
    return 0;
}

template <class C>
static int run_acc30(const char* name, int waves_per_simd) {
    typedef typename C::Fq Fq;
    typedef typename Fq::Params P;
    const uint32_t nb = 1 << 16, len = 64;
    const uint32_t threads = 256 * 4 * 64 * waves_per_simd;
    Affine<Fq>* d_b; XYZZ<Fq>* d_o;
    CK(hipMalloc(&d_b, sizeof(Affine<Fq>) * nb));
    CK(hipMalloc(&d_o, sizeof(XYZZ<Fq>) * threads));
    if (synth_bases_device<C>(0, 1, 0, nb, d_b, 0) != 0) { printf("synth failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((acc30_probe<Fp30<P>>), dim3(threads / 128), dim3(128), 0, 0, d_b, nb, len, d_o);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double adds = (double)threads * len;
    printf("ACC30PROBE %-20s variant=%s waves/SIMD=%d  %.3f ms for %.0f mixed adds => %.2f Gadd/s\n", name, G16_VARIANT, waves_per_simd, ms, adds,
           adds / ms / 1e6);
    {
        Fq *d_in, *d_out;
        const uint32_t T = threads;
        CK(hipMalloc(&d_in, sizeof(Fq) * (T + 1)));
        CK(hipMalloc(&d_out, sizeof(Fq) * T));
        for (uint32_t off = 0; off < T + 1; off += 65536) {
            const uint32_t cnt = (T + 1 - off) < 65536 ? (T + 1 - off) : 65536;
            CK(hipMemcpy(d_in + off, d_b, sizeof(Fq) * cnt, hipMemcpyDeviceToDevice));
        }
        const int iters = 256;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL((mul30_probe<P>), dim3(T / 256), dim3(256), 0, 0, d_in, d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
        }
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double muls = (double)T * iters * 2;
        printf("MUL30PROBE %-20s variant=%s waves/SIMD=%d  %.3f ms for %.0f muls => %.2f Gmul/s  (%.0f SIMD-cycles@2.4GHz per wave-mul)\n", name,
               G16_VARIANT, waves_per_simd, ms, muls, muls / ms / 1e6, ms * 1e-3 * 2.4e9 / (muls / 64 / 1024));
        (void)hipFree(d_in); (void)hipFree(d_out);
    }
    (void)hipFree(d_b); (void)hipFree(d_o);
    return 0;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs=%d clock=%d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
#ifdef G16_UBENCH_OPS
    uint64_t* d_out;
    CK(hipMalloc(&d_out, sizeof(uint64_t) * (256 * 8 * 256 + 16)));
    for (int w : {1, 2, 4, 8}) {
        run_op<OP_MAD64>(d_out, w); run_op<OP_MULLO>(d_out, w); run_op<OP_MULHI>(d_out, w); run_op<OP_ADDC>(d_out, w);
        run_op<OP_LSHLADD64>(d_out, w); run_op<OP_MOV>(d_out, w); run_op<OP_MAD24>(d_out, w); run_op<OP_FMA64>(d_out, w);
        run_op<OP_ADD32>(d_out, w);
    }
#endif
    for (int w : {1, 2}) { run_acc30<Bls12_381>("bls12_381 G1", w); run_acc30<Bn254>("bn254 G1", w); }
    for (int w : {1, 2}) { run_acc30_g2<Bls12_381>("bls12_381 G2", w); run_acc30_g2<Bn254>("bn254 G2", w); }
#ifdef G16_UBENCH_OPS
    run_acc<Bls12_381, Bls12_381::Fq>("bls12_381 G1", 0);
    run_acc<Bn254, Bn254::Fq>("bn254 G1", 0);
    run_acc<Bls12_381, Bls12_381::Fq2>("bls12_381 G2", 1);
    run_acc<Bn254, Bn254::Fq2>("bn254 G2", 1);
#endif
    return 0;
}
