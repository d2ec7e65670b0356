// Synthetic workload generators for bench.py and the parity tests.
//  * synthetic proving-key bases: n distinct non-identity points P_i = (s0 + first + i) * G,
//    generated on the GPU straight into HBM (a valid CRS would need the CPU setup of
//    /root/reference/src/generator.rs:47-208, minutes at 2^22);
//  * SYN(k, seed): the Fibonacci-product-chain R1CS mirroring benches/bench.rs:23-64
//    (2 instance variables, n_c = 2^k - 2 so the FFT domain is exactly 2^k).
#include "internal.hpp"

namespace g16 {

static constexpr int SYNTH_RUN = 16;  // consecutive points per lane (amortises the initial scalar mul)

template <class F>
__global__ __launch_bounds__(64) void synth_bases_kernel(Affine<F> gen, uint64_t s_first, uint64_t n, Affine<F>* __restrict__ out) {
    const uint64_t t = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    const uint64_t lo = t * SYNTH_RUN;
    if (lo >= n) return;
    const uint64_t hi = min(n, lo + (uint64_t)SYNTH_RUN);
    const uint64_t k = s_first + lo;
    uint32_t kw[2] = {(uint32_t)k, (uint32_t)(k >> 32)};
    XYZZ<F> p = XYZZ<F>::from_affine(gen).mul_bits(kw, 64);
    for (uint64_t i = lo; i < hi; ++i) {
        out[i] = p.to_affine();
        p.add_affine(gen);
    }
}

// v_mad_u64_u32 issue-rate probe (g16_diag_valu): 8 independent accumulator chains per lane, so the multiplier pipeline is never
// waiting on a dependent result; 8 waves per SIMD on every CU
__global__ __launch_bounds__(256) void mad_rate_kernel(uint64_t* __restrict__ out, int iters) {
    const uint32_t a = threadIdx.x * 2654435761u + 12345u, b = blockIdx.x * 40503u + 7u;
    uint64_t r[8];
    for (int k = 0; k < 8; ++k) r[k] = (uint64_t)a * (uint32_t)(k + 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[k]) : "v"(a), "v"(b) : "vcc");
    }
    uint64_t acc = 0;
    for (int k = 0; k < 8; ++k) acc += r[k];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

int mad_rate_device(hipStream_t st, double* mad_per_s) {
    int dev = 0, cus = 0;
    G16_HIP_TRY(hipGetDevice(&dev));
    G16_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int blocks = cus * 8;   // 4 waves per block = one per SIMD; x8 = 8 waves per SIMD
    const int iters = 8192;
    uint64_t* d_out = nullptr;
    G16_HIP_TRY(hipMalloc((void**)&d_out, (size_t)blocks * 256 * sizeof(uint64_t)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = G16_OK;
    float ms = 0.f;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = G16_ERR_HIP;
    for (int rep = 0; rc == G16_OK && rep < 2; ++rep) {   // first launch warms the clocks
        if (hipEventRecord(e0, st) != hipSuccess) { rc = G16_ERR_HIP; break; }
        hipLaunchKernelGGL(mad_rate_kernel, dim3(blocks), dim3(256), 0, st, d_out, iters);
        if (hipEventRecord(e1, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = G16_ERR_HIP;
    }
    if (rc == G16_OK && hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = G16_ERR_HIP;
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_out);
    if (rc == G16_OK) *mad_per_s = (double)blocks * 256.0 * iters * 8.0 / ((double)ms * 1e-3);
    return rc;
}

static uint64_t splitmix_next(uint64_t& s) {
    s += 0x9E3779B97F4A7C15ULL;
    uint64_t z = s;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

template <class C>
int synth_bases_device(int g2, uint64_t seed, uint64_t first, uint64_t n, void* out_dev, hipStream_t st) {
    if (n == 0) return G16_OK;
    uint64_t s = seed ^ 0xBA5E5ULL;
    const uint64_t s0 = (splitmix_next(s) >> 8) | 1ULL;
    const unsigned blocks = (unsigned)(((n + SYNTH_RUN - 1) / SYNTH_RUN + 63) / 64);
    if (!g2) {
        typedef typename C::Fq F;
        hipLaunchKernelGGL((synth_bases_kernel<F>), dim3(blocks), dim3(64), 0, st, C::g1_generator(), s0 + first, n,
                           reinterpret_cast<Affine<F>*>(out_dev));
    } else {
        typedef typename C::Fq2 F;
        hipLaunchKernelGGL((synth_bases_kernel<F>), dim3(blocks), dim3(64), 0, st, C::g2_generator(), s0 + first, n,
                           reinterpret_cast<Affine<F>*>(out_dev));
    }
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template int synth_bases_device<Bls12_381>(int, uint64_t, uint64_t, uint64_t, void*, hipStream_t);
template int synth_bases_device<Bn254>(int, uint64_t, uint64_t, uint64_t, void*, hipStream_t);

// 512 pseudo-random bits reduced mod r (Horner in Fr over eight 64-bit limbs, MSB limb first)
template <class Fr>
static Fr rand_fr(uint64_t& state) {
    const Fr two64 = Fr::from_u64(1ULL << 32).sqr();
    Fr acc = Fr::zero();
    for (int k = 0; k < 8; ++k) acc = acc * two64 + Fr::from_u64(splitmix_next(state));
    return acc;
}

template <class C>
static int synth_circuit_host(int k, uint64_t seed, uint64_t* z_out, uint64_t* row_ptr, uint32_t* colA, uint32_t* colB, uint32_t* colC,
                              uint64_t* val) {
    typedef typename C::Fr Fr;
    if (k < 2 || k > 30) return G16_ERR_BAD_ARG;
    const uint64_t nc = ((uint64_t)1 << k) - 2;
    uint64_t st = seed;
    Fr* z = reinterpret_cast<Fr*>(z_out);  // Fr is 32 B of limbs; callers pass 32-byte records
    std::vector<Fr> u(nc + 2);
    u[0] = rand_fr<Fr>(st);
    u[1] = rand_fr<Fr>(st);
    for (uint64_t i = 0; i < nc; ++i) u[i + 2] = u[i] * u[i + 1];
    const Fr one = Fr::one();
    memcpy(z_out, &one, sizeof(Fr));
    memcpy(z_out + 4, &u[nc + 1], sizeof(Fr));
    memcpy(z_out + 8, u.data(), sizeof(Fr) * (nc + 1));
    (void)z;
    for (uint64_t i = 0; i < nc; ++i) {
        row_ptr[i] = i;
        colA[i] = (uint32_t)(2 + i);
        colB[i] = (uint32_t)(2 + i + 1);
        colC[i] = (i + 2 == nc + 1) ? 1u : (uint32_t)(2 + i + 2);
        memcpy(val + 4 * i, &one, sizeof(Fr));
    }
    row_ptr[nc] = nc;
    return G16_OK;
}

}  // namespace g16

extern "C" int g16_synth_circuit(int curve, int k, uint64_t seed, uint64_t* z_out, uint64_t* row_ptr, uint32_t* colA, uint32_t* colB,
                                 uint32_t* colC, uint64_t* val) {
    if (curve == G16_BLS12_381) return g16::synth_circuit_host<g16::Bls12_381>(k, seed, z_out, row_ptr, colA, colB, colC, val);
    if (curve == G16_BN254) return g16::synth_circuit_host<g16::Bn254>(k, seed, z_out, row_ptr, colA, colB, colC, val);
    return G16_ERR_BAD_ARG;
}
