// Radix-2 NTT over the scalar field for gfx950.
//
// Replaces ark-poly's Radix2EvaluationDomain::{fft,ifft}_in_place and the coset variants as
// called from /root/reference/src/r1cs_to_qap.rs:201-207,220-221,232.
//
// Shape: an n-point transform is split into passes; each pass loads a tile of 2^11 elements
// (64 KiB) into LDS, runs up to 11 butterfly stages there and writes the tile back, so one
// NTT is ceil(log2 n / ~8) sweeps over HBM instead of log2 n.  The last/first pass works on
// contiguous tiles; the others gather 2^T-element runs (>= 1 KiB contiguous for T >= 5) so
// that every wave-level access is a full-line access.  Decimation-in-frequency (natural ->
// bit-reversed) is paired with decimation-in-time (bit-reversed -> natural) so the witness map
// never needs a separate bit-reversal between its inverse and forward transforms; the coset
// shift g^k and the 1/n factor ride along as a pre-scale on the DIT load.
// Bound: HBM bandwidth (2*32*n bytes per sweep) with the Fr Montgomery product
// (8 limbs, 136 v_mad_u64_u32) as the competing VALU term; no MFMA (integer modular work).
#include "internal.hpp"

namespace g16 {

static constexpr int NTT_TILE_LOG = 11;   // 2^11 Fr = 64 KiB of LDS per workgroup
static constexpr int NTT_THREADS = 256;

template <class Fr>
struct PowTable { Fr p[32]; };  // base^(2^j)

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

// out[i] = scale * base^e(i),  e(i) = bitrev(i) if rev else i
template <class Fr>
__global__ void gen_powers_kernel(Fr* __restrict__ out, size_t n, PowTable<Fr> tab, Fr scale, int rev_bits) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e = rev_bits ? bitrev32((uint32_t)i, rev_bits) : (uint32_t)i;
    Fr acc = scale;
    for (int j = 0; j < 32; ++j) {
        if ((e >> j) & 1) acc = acc * tab.p[j];
    }
    out[i] = acc;
}

// One LDS-tiled pass over stages [s_lo, s_hi).  T = log2 of the contiguous run (ignored when s_lo == 0).
template <class Fr, bool DIT>
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(Fr* __restrict__ data, const Fr* __restrict__ tw,
                                                                const Fr* __restrict__ prescale, int log_n, int s_lo,
                                                                int s_hi, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Fr* lds = reinterpret_cast<Fr*>(smem);
    const int K = s_hi - s_lo;
    const int TT = (s_lo == 0) ? 0 : T;
    const uint32_t E = 1u << (K + TT);
    const uint64_t blk = blockIdx.x;
    uint64_t base;
    if (s_lo == 0) {
        base = blk << K;
    } else {
        const int mid_bits = s_lo - T;                          // index bits [T, s_lo)
        const uint64_t lowpart = blk & ((1ull << mid_bits) - 1);
        const uint64_t hipart = blk >> mid_bits;                // index bits [s_hi, log_n)
        base = (lowpart << T) | (hipart << s_hi);
    }
    auto gidx = [&](uint32_t e) -> uint64_t {
        if (s_lo == 0) return base + e;
        const uint32_t low = e & ((1u << T) - 1), mid = e >> T;
        return base + low + ((uint64_t)mid << s_lo);
    };
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
        const uint64_t g = gidx(e);
        Fr x = data[g];
        if (prescale) x = x * prescale[g];
        lds[e] = x;
    }
    __syncthreads();
    for (int k = 0; k < K; ++k) {
        const int s = DIT ? (s_lo + k) : (s_hi - 1 - k);
        const int bitpos = (s - s_lo) + TT;
        for (uint32_t b = threadIdx.x; b < E / 2; b += NTT_THREADS) {
            const uint32_t lower = b & ((1u << bitpos) - 1), upper = b >> bitpos;
            const uint32_t e0 = (upper << (bitpos + 1)) | lower, e1 = e0 | (1u << bitpos);
            const uint64_t j = gidx(e0) & ((1ull << s) - 1);
            const Fr w = tw[j << (log_n - 1 - s)];
            Fr u = lds[e0], v = lds[e1];
            if (DIT) {
                v = v * w;
                lds[e0] = u + v;
                lds[e1] = u - v;
            } else {
                lds[e0] = u + v;
                lds[e1] = (u - v) * w;
            }
        }
        __syncthreads();
    }
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) data[gidx(e)] = lds[e];
}

template <class Fr>
__global__ void bitrev_scale_kernel(Fr* __restrict__ out, const Fr* __restrict__ in, const Fr* __restrict__ table, Fr cst,
                                    int has_cst, int log_n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ((size_t)1 << log_n)) return;
    Fr x = in[bitrev32((uint32_t)k, log_n)];
    if (table) x = x * table[k];
    if (has_cst) x = x * cst;
    out[k] = x;
}

template <class Fr>
__global__ void scale_table_kernel(Fr* __restrict__ data, const Fr* __restrict__ table, size_t n) {
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    data[k] = data[k] * table[k];
}

// ---------------------------------------------------------------------------------------------
struct PassPlan { int s_lo, s_hi, T; };

static std::vector<PassPlan> plan_passes(int log_n) {
    // ascending stage order (the DIT order); DIF walks the list backwards
    std::vector<PassPlan> p;
    if (log_n <= NTT_TILE_LOG) {
        p.push_back({0, log_n, 0});
        return p;
    }
    p.push_back({0, NTT_TILE_LOG, 0});
    int rest = log_n - NTT_TILE_LOG;
    const int npass = (rest + 7) / 8;
    int s = NTT_TILE_LOG;
    for (int i = 0; i < npass; ++i) {
        const int left = npass - i;
        const int k = (rest + left - 1) / left;  // split the remaining stages evenly
        p.push_back({s, s + k, NTT_TILE_LOG - k});
        s += k;
        rest -= k;
    }
    return p;
}

template <class Fr, bool DIT>
static int launch_pass(Fr* data, const Fr* tw, const Fr* prescale, int log_n, const PassPlan& pp, hipStream_t st) {
    const int K = pp.s_hi - pp.s_lo;
    const int TT = pp.s_lo == 0 ? 0 : pp.T;
    const size_t E = (size_t)1 << (K + TT);
    const size_t blocks = ((size_t)1 << log_n) / E;
    const size_t lds_bytes = E * sizeof(Fr);
    static bool attr_set = false;
    if (!attr_set) {
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt_pass_kernel<Fr, DIT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(Fr) << NTT_TILE_LOG)));
        attr_set = true;
    }
    hipLaunchKernelGGL((ntt_pass_kernel<Fr, DIT>), dim3((unsigned)blocks), dim3(NTT_THREADS), lds_bytes, st, data, tw, prescale,
                       log_n, pp.s_lo, pp.s_hi, pp.T);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
int ntt_dif(const Domain<C>* d, typename C::Fr* data, bool inverse, hipStream_t st) {
    typedef typename C::Fr Fr;
    if (d->log_n == 0) return G16_OK;
    auto passes = plan_passes(d->log_n);
    const Fr* tw = inverse ? d->tw_inv : d->tw_fwd;
    for (size_t i = passes.size(); i-- > 0;) G16_TRY((launch_pass<Fr, false>(data, tw, nullptr, d->log_n, passes[i], st)));
    return G16_OK;
}

template <class C>
int ntt_dit(const Domain<C>* d, typename C::Fr* data, bool inverse, const typename C::Fr* prescale, hipStream_t st) {
    typedef typename C::Fr Fr;
    if (d->log_n == 0) {
        if (prescale) G16_TRY((scale_by_table<C>(data, prescale, 1, st)));
        return G16_OK;
    }
    auto passes = plan_passes(d->log_n);
    const Fr* tw = inverse ? d->tw_inv : d->tw_fwd;
    for (size_t i = 0; i < passes.size(); ++i)
        G16_TRY((launch_pass<Fr, true>(data, tw, i == 0 ? prescale : nullptr, d->log_n, passes[i], st)));
    return G16_OK;
}

template <class C>
int bitrev_scale(const Domain<C>* d, typename C::Fr* out, const typename C::Fr* in, const typename C::Fr* table,
                 const typename C::Fr* cst, hipStream_t st) {
    typedef typename C::Fr Fr;
    const size_t n = d->n;
    Fr c = cst ? *cst : Fr::one();
    hipLaunchKernelGGL((bitrev_scale_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, in, table, c,
                       cst ? 1 : 0, d->log_n);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
int scale_by_table(typename C::Fr* data, const typename C::Fr* table, size_t n, hipStream_t st) {
    typedef typename C::Fr Fr;
    hipLaunchKernelGGL((scale_table_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, data, table, n);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class Fr>
static int gen_powers(Fr* out, size_t n, const Fr& base, const Fr& scale, int rev_bits, hipStream_t st) {
    if (n == 0) return G16_OK;
    PowTable<Fr> tab;
    Fr p = base;
    for (int j = 0; j < 32; ++j) { tab.p[j] = p; p = p.sqr(); }
    hipLaunchKernelGGL((gen_powers_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, tab, scale, rev_bits);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
int domain_create(int log_n, hipStream_t st, Domain<C>** out) {
    typedef typename C::Fr Fr;
    if (log_n > C::TWO_ADICITY || log_n > 31) return G16_ERR_DEGREE_TOO_LARGE;
    Domain<C>* d = new Domain<C>();
    d->log_n = log_n;
    d->n = (size_t)1 << log_n;
    const size_t n = d->n, half = n / 2 ? n / 2 : 1;
    Fr omega = C::two_adic_root();
    for (int i = log_n; i < C::TWO_ADICITY; ++i) omega = omega.sqr();
    Fr omega_inv = omega.inverse();
    d->n_inv = Fr::from_u64((uint64_t)n).inverse();
    const Fr g = C::fr_generator(), g_inv = C::fr_generator_inv();
    d->zinv = (g.pow_u64((uint64_t)n) - Fr::one()).inverse();  // 1 / Z(g), r1cs_to_qap.rs:223-226
    int rc = G16_OK;
    auto fail = [&](int code) { domain_destroy<C>(d); return code; };
    if (hipMalloc((void**)&d->tw_fwd, half * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->tw_inv, half * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->s1_br, n * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->s2, n * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if ((rc = gen_powers<Fr>(d->tw_fwd, half, omega, Fr::one(), 0, st)) != G16_OK) return fail(rc);
    if ((rc = gen_powers<Fr>(d->tw_inv, half, omega_inv, Fr::one(), 0, st)) != G16_OK) return fail(rc);
    if ((rc = gen_powers<Fr>(d->s1_br, n, g, d->n_inv, log_n, st)) != G16_OK) return fail(rc);
    if ((rc = gen_powers<Fr>(d->s2, n, g_inv, d->n_inv, 0, st)) != G16_OK) return fail(rc);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(G16_ERR_HIP);
    *out = d;
    return G16_OK;
}

template <class C>
int domain_ensure_gpow(Domain<C>* d, hipStream_t st) {
    typedef typename C::Fr Fr;
    if (d->g_pow) return G16_OK;
    G16_HIP_TRY(hipMalloc((void**)&d->g_pow, d->n * sizeof(Fr)));
    return gen_powers<Fr>(d->g_pow, d->n, C::fr_generator(), Fr::one(), 0, st);
}

template <class C>
void domain_destroy(Domain<C>* d) {
    if (!d) return;
    (void)hipFree(d->tw_fwd); (void)hipFree(d->tw_inv); (void)hipFree(d->s1_br); (void)hipFree(d->s2); (void)hipFree(d->g_pow);
    delete d;
}

#define G16_INSTANTIATE_NTT(C)                                                                                     \
    template int domain_create<C>(int, hipStream_t, Domain<C>**);                                                  \
    template void domain_destroy<C>(Domain<C>*);                                                                   \
    template int domain_ensure_gpow<C>(Domain<C>*, hipStream_t);                                                   \
    template int ntt_dif<C>(const Domain<C>*, typename C::Fr*, bool, hipStream_t);                                 \
    template int ntt_dit<C>(const Domain<C>*, typename C::Fr*, bool, const typename C::Fr*, hipStream_t);          \
    template int bitrev_scale<C>(const Domain<C>*, typename C::Fr*, const typename C::Fr*, const typename C::Fr*,  \
                                 const typename C::Fr*, hipStream_t);                                              \
    template int scale_by_table<C>(typename C::Fr*, const typename C::Fr*, size_t, hipStream_t);

G16_INSTANTIATE_NTT(Bls12_381)
G16_INSTANTIATE_NTT(Bn254)

}  // namespace g16
