// Round 6 probe (round-5 verdict, item 5): could the CONSTANT-operand half of the Montgomery reduction -- m * p, a Toeplitz-matrix
// x batch-of-vectors contraction -- run on the matrix unit (v_mfma_i32_16x16x64_i8) instead of 169 v_mad_u64_u32 per product?
//
// The MFMA consumes 7-bit digits (signed i8 operands, non-negative) and returns 111 digit-columns of the convolution as int32.
// Whatever the matrix unit's speed, the VECTOR pipe still has to
//   (a) split m's 13 30-bit limbs into 56 base-128 digits, four per dword            (split)
//   (b) move them into the MFMA's B layout and the results back (ds_bpermute_b32: LDS pipe, not counted against the vector pipe here)
//   (c) fold the 111 int32 columns (weight 2^(7c)) back into the 30-bit column accumulators: one multiply-add per column   (recombine)
// This probe times (a) + (c) -- bit-exactly validated against a per-lane digit convolution -- beside the 169-multiply-add sweep they
// would replace, and the 28 MFMAs + 128 ds_bpermute per 64 products on their own pipes.  If (a) + (c) alone cost the vector pipe as
// much as the sweep, no matrix-unit speed can win (the bucket passes are bound by vector-instruction issue: profiles/r06_pmc_*).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=200000 tools/probe_mfma.hip -o tools/bin/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../groth16_amd/csrc/curve.hpp"
#include "../groth16_amd/csrc/fp30.hpp"
using namespace g16;
typedef Bls12_381FqP P;
typedef Fp30<P> F;
static constexpr int NL = F::NL, ND = 56, NC = 2 * ND - 1;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t digit(const uint32_t* l, int d) {   // 7-bit digit d of the 390-bit value in 30-bit limbs
    const int bit = 7 * d, q = bit / 30, off = bit % 30;
    uint32_t v = l[q] >> off;
    if (off + 7 > 30 && q + 1 < NL) v |= l[q + 1] << (30 - off);
    return v & 127u;
}
__device__ __forceinline__ void split(const uint32_t* l, uint32_t* w /* ND / 4 dwords */) {
#pragma unroll
    for (int k = 0; k < ND / 4; ++k) w[k] = digit(l, 4 * k) | digit(l, 4 * k + 1) << 8 | digit(l, 4 * k + 2) << 16 | digit(l, 4 * k + 3) << 24;
}
// T[q] (64-bit, weight 2^(30 q)) += col_c * 2^(7 c - 30 q): ONE v_mad_u64_u32 per digit column
__device__ __forceinline__ void recombine(const uint32_t* col, uint64_t* T) {
    uint32_t pw[30];   // 2^s as opaque scalars: keeps  T[q] + col * 2^s  ONE v_mad_u64_u32 (a 64-bit shift and a 64-bit add otherwise)
#pragma unroll
    for (int k = 0; k < 30; ++k) { pw[k] = 1u << k; asm("" : "+s"(pw[k])); }
#pragma unroll
    for (int q = 0; q < 2 * NL; ++q) T[q] = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int bit = 7 * c, q = bit / 30, s = bit % 30;
        T[q] += (uint64_t)col[c] * pw[s];
    }
}
__device__ __forceinline__ uint32_t pdigit(int d) {
    uint32_t pl[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) pl[i] = P::p30(i);
    return d < ND ? digit(pl, d) : 0u;
}
// validation: split -> per-lane digit convolution with p's digits (what the MFMA would return) -> recombine == the VALU sweep m * p
__global__ __launch_bounds__(64) void k_check(const uint32_t* in, uint32_t* bad) {
    uint32_t m[NL];
    for (int i = 0; i < NL; ++i) m[i] = in[threadIdx.x * NL + i] & F::MASK;
    uint32_t w[ND / 4], col[NC];
    split(m, w);
    for (int c = 0; c < NC; ++c) {
        uint32_t s = 0;
        for (int k = 0; k < ND; ++k) {
            const int j = c - k;
            if (j >= 0 && j < ND) s += ((w[k / 4] >> (8 * (k % 4))) & 255u) * pdigit(j);
        }
        col[c] = s;
    }
    uint64_t T[2 * NL], W[2 * NL];
    recombine(col, T);
    for (int q = 0; q < 2 * NL; ++q) W[q] = 0;
    for (int i = 0; i < NL; ++i)
        for (int j = 0; j < NL; ++j) W[i + j] += (uint64_t)m[i] * P::p30(j);
    // both are redundant column forms of the same integer: compare after a carry pass
    uint64_t c1 = 0, c2 = 0;
    uint32_t diff = 0;
    for (int q = 0; q < 2 * NL; ++q) {
        const uint64_t a = T[q] + c1, b = W[q] + c2;
        diff |= (uint32_t)(a & F::MASK) ^ (uint32_t)(b & F::MASK);
        c1 = a >> 30; c2 = b >> 30;
    }
    if (diff || c1 != c2) atomicAdd(bad, 1u);
}
// timed: KIND 0 = the sweep (169 multiply-adds, m * p into 26 columns); 1 = split + recombine (columns stand-ins: the digits' dwords)
template <int KIND>
__global__ __launch_bounds__(256) void k_time(uint32_t* out, int iters) {
    uint32_t m[NL];
    for (int i = 0; i < NL; ++i) m[i] = (threadIdx.x * 2654435761u + blockIdx.x * 40503u + i * 977u) & F::MASK;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint64_t T[2 * NL];
        if (KIND == 0) {
            uint32_t pl[NL];   // opaque: the limbs of p as the kernels see them (scalar registers), not folded into shifts and adds
#pragma unroll
            for (int j = 0; j < NL; ++j) { pl[j] = P::p30(j); asm("" : "+s"(pl[j])); }
#pragma unroll
            for (int q = 0; q < 2 * NL; ++q) T[q] = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i)
#pragma unroll
                for (int j = 0; j < NL; ++j) T[i + j] += (uint64_t)m[i] * pl[j];
        } else {
            uint32_t w[ND / 4], col[NC];
            split(m, w);
#pragma unroll
            for (int c = 0; c < NC; ++c) col[c] = w[c % (ND / 4)] >> (c / (ND / 4));   // stand-in for the MFMA's result registers
            recombine(col, T);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) m[i] = ((uint32_t)T[i] ^ (uint32_t)(T[i] >> 32) ^ (uint32_t)(T[i + NL] >> 7) ^ (uint32_t)(T[i + NL] >> 39) ^ m[i]) & F::MASK;   // feed back: nothing is dead
        acc ^= m[0];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}
// the matrix unit and the LDS crossbar on their own: 28 v_mfma_i32_16x16x64_i8 (7 row tiles x 4 batch tiles) and 128 ds_bpermute_b32
// (16 in, 112 out) per 64 products
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_mfma(int* out, int iters) {
    v4i a = {(int)threadIdx.x, 3, 5, 7}, b = {(int)blockIdx.x, 11, 13, 17};
    v4i c[7];
    for (int r = 0; r < 7; ++r) c[r] = (v4i){0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 7; ++r) c[r] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, c[r], 0, 0, 0);
    }
    int s = 0;
    for (int r = 0; r < 7; ++r) s += c[r][0] + c[r][1] + c[r][2] + c[r][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_bperm(int* out, int iters) {
    int v = threadIdx.x * 7 + 1, idx = ((threadIdx.x * 5 + 3) & 63) << 2;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 128; ++k) v = __builtin_amdgcn_ds_bpermute(idx, v + k);
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    uint32_t *d_in, *d_out, *d_bad;
    CK(hipMalloc(&d_in, 64 * NL * 4)); CK(hipMalloc(&d_out, sizeof(uint32_t) * 256 * 8 * 256)); CK(hipMalloc(&d_bad, 4));
    std::vector<uint32_t> h(64 * NL);
    uint64_t s = 88172645463325252ull;
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)s; }
    for (int i = 0; i < NL; ++i) { h[i] = 0x3fffffffu; h[NL + i] = 0; }   // all-ones limbs, zero
    CK(hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_bad, 0, 4));
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, d_in, d_bad);
    uint32_t bad = 1;
    CK(hipMemcpy(&bad, d_bad, 4, hipMemcpyDeviceToHost));
    printf("MFMA-GLUE check: split -> digit convolution -> recombine == m * p for 64 values (all-ones and zero among them): %s\n", bad ? "MISMATCH" : "ok");
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w : {2, 3, 4, 8}) {
        const int blocks = 256 * w, iters = 2048;
        float ms[4];
        for (int kind = 0; kind < 4; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                if (kind == 0) hipLaunchKernelGGL((k_time<0>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
                else if (kind == 1) hipLaunchKernelGGL((k_time<1>), dim3(blocks), dim3(256), 0, 0, d_out, iters);
                else if (kind == 2) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, (int*)d_out, iters / 64);   // 28 MFMAs serve 64 products
                else hipLaunchKernelGGL(k_bperm, dim3(blocks), dim3(256), 0, 0, (int*)d_out, iters / 64);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms[kind], e0, e1));
            }
        }
        const double prods = 65536.0 * w * iters;
        printf("MFMA-GLUE waves/SIMD=%d  per product: sweep (169 v_mad_u64_u32) %.2f ps of chip time | split + recombine on the vector pipe %.2f | 28 MFMA / 64 products %.2f | 128 ds_bpermute / 64 products %.2f   "
               "=> vector-pipe time of the MFMA variant / sweep = %.2f\n", w, ms[0] * 1e9 / prods, ms[1] * 1e9 / prods, ms[2] * 1e9 / prods, ms[3] * 1e9 / prods, ms[1] / ms[0]);
    }
    return 0;
}
