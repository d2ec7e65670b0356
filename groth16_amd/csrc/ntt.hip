// Radix-2 NTT over the scalar field for gfx950.
//
// Replaces ark-poly's Radix2EvaluationDomain::{fft,ifft}_in_place and the coset variants as
// called from /root/reference/src/r1cs_to_qap.rs:201-207,220-221,232.
//
// Shape: an n-point transform is split into passes; each pass stages a tile of 2^10 elements in LDS,
// runs up to 10 butterfly stages there and writes the tile back, so one NTT is ceil(log2 n / ~8) sweeps
// over HBM instead of log2 n.  The last/first pass works on contiguous tiles; the others gather
// 2^T-element runs (>= 128 B contiguous) so that every wave-level access is a full-line access.
// Inside a pass the stages are taken two at a time: a lane pulls 4 elements out of LDS, does 4
// butterflies in registers (radix-4 block) and puts them back -- 5 LDS round trips for 10 stages, <= 122 registers,
// four workgroups per CU (round 3; round 2 used 2^11-element tiles and radix-8 blocks at two waves per SIMD).
// The tile is kept as structure-of-arrays of 30-bit limbs ([9][1024+pad] words, 38 KB): consecutive lanes
// touch consecutive words, and one pad word per 32 elements breaks the strided pattern of the first round.
//
// Arithmetic: Fr in the 30-bit lazy representation of fp30.hpp (9 limbs; a product is 162 carry-free
// v_mad_u64_u32).  Data stays in the arkworks Montgomery form (x*R); the twiddle / scale tables are stored as
// w*R' (R' = 2^270), so mont30(x*R, w*R') = x*w*R and no conversion of the data is ever needed.  Butterfly
// outputs are left loosely reduced for the whole pass (DIT: sums grow by <= 2p per stage; DIF: sums double per
// stage and the subtraction adds the matching redundant 2^(k+1) p) and canonicalised once, by a product with
// R' mod p, when the tile is written back.
// Decimation-in-frequency (natural -> bit-reversed) is paired with decimation-in-time (bit-reversed ->
// natural) so the witness map never needs a separate bit-reversal between its inverse and forward transforms;
// the coset shift g^k and the 1/n factor ride along as a pre-scale on the DIT load.
// Bound: nominally HBM (2*32*n bytes per sweep); in practice the 46 M Fr products of a 2^22-point transform
// (VALU) dominate -- see DESIGN.md 4.2.  No MFMA (integer modular work).
#include "internal.hpp"
#include "fp30.hpp"

namespace g16 {

// Tile shape (round 3, profiles/r03_ab_ntt_tile_radix_occupancy.txt, one box, 2^22): 2^10-element tiles (38 KB of LDS) with radix-4
// register blocks keep the pass kernels at <= 122 registers, so FOUR workgroups of 256 lanes share a CU (four waves per SIMD) and a
// wave waiting at a barrier or on LDS is covered by three others: 5.48 ms for the seven transforms against 6.03 ms for the round-2
// shape (2^11-element tiles, radix-8 blocks, 187 registers, two waves per SIMD).  Same number of sweeps (10 + 6 + 6 stages at 2^22).
// Also measured: 2^11 tiles / 512 lanes / radix-4 / 4 waves 5.78 ms; 2^12 tiles / 1024 lanes / radix-4 (two sweeps) 6.36 ms;
// 2^12 tiles / 512 lanes / radix-8 / 2 waves 6.06 ms.  -DG16_NTT_TILE_LOG / _THREADS / _MAX_R / _MIN_WAVES rebuild any of them.
#ifndef G16_NTT_TILE_LOG
#define G16_NTT_TILE_LOG 10
#endif
static constexpr int NTT_TILE_LOG = G16_NTT_TILE_LOG;
static constexpr int NTT_TILE = 1 << NTT_TILE_LOG;
#ifdef G16_NTT_THREADS
static constexpr int NTT_THREADS = G16_NTT_THREADS;
#else
static constexpr int NTT_THREADS = NTT_TILE_LOG >= 12 ? 512 : 256;
#endif
// stages taken together in registers: 2 = radix-4 blocks (4 elements), 3 = radix-8 blocks (8 elements, ~187 registers)
#ifndef G16_NTT_MAX_R
#define G16_NTT_MAX_R 2
#endif
static constexpr int NTT_MAX_R = G16_NTT_MAX_R;
#ifndef G16_NTT_MIN_WAVES
#define G16_NTT_MIN_WAVES 4
#endif
static constexpr int NTT_MAX_STRIDED = NTT_TILE_LOG >= 12 ? 10 : 8;   // stages per strided pass (runs of 2^(TILE_LOG - k) elements)
static constexpr int NTT_ROW = NTT_TILE + NTT_TILE / 32;  // padded row length (words) of one limb plane

template <class Fr>
struct PowTable { Fr p[32]; };  // base^(2^j)

// One twiddle as the butterflies consume it.  G16_NTT_TW_UNPACKED = 1: the NL 30-bit limbs themselves (9 words for the 255-bit scalar
// fields instead of the 8 packed ones) -- no shift / mask / align work per butterfly; 0: the packed w*R' words of rounds 1-3.
#ifndef G16_NTT_TW_UNPACKED
#define G16_NTT_TW_UNPACKED 1
#endif
template <class P>
struct Tw { uint32_t w[G16_NTT_TW_UNPACKED ? Fp30<P>::NL : P::N]; };
template <class P>
__device__ __forceinline__ Fp30<P> load_tw(const Tw<P>& e) {
#if G16_NTT_TW_UNPACKED
    Fp30<P> r;
    G16_UNROLL for (int i = 0; i < Fp30<P>::NL; ++i) r.l[i] = e.w[i];
    return r;
#else
    return Fp30<P>::unpack(e.w);
#endif
}
template <class P>
__global__ void tw_convert_kernel(const Fp<P>* __restrict__ in, Tw<P>* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Tw<P> e;
#if G16_NTT_TW_UNPACKED
    const Fp30<P> x = Fp30<P>::unpack(in[i].v);
    G16_UNROLL for (int k = 0; k < Fp30<P>::NL; ++k) e.w[k] = x.l[k];
#else
    G16_UNROLL for (int k = 0; k < P::N; ++k) e.w[k] = in[i].v[k];
#endif
    out[i] = e;
}

__device__ __forceinline__ uint32_t bitrev32(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }
__device__ __forceinline__ uint32_t lds_col(uint32_t e) { return e + (e >> 5); }

// out[i] = scale * base^e(i),  e(i) = bitrev(i) if rev else i     (standard Montgomery arithmetic; `scale` may
// carry the R'/R factor that turns the table into the w*R' form the 30-bit butterflies consume)
// layered_log > 0: the per-stage twiddle layout of the NTT passes -- entry 2^s - 1 + k (k < 2^s, s < layered_log) holds
// base^(k << (layered_log - 1 - s)), i.e. the 2^s twiddles of butterfly stage s CONTIGUOUSLY (n = 2^layered_log - 1 entries)
template <class Fr>
__global__ void gen_powers_kernel(Fr* __restrict__ out, size_t n, PowTable<Fr> tab, Fr scale, int rev_bits, int layered_log) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t e = rev_bits ? bitrev32((uint32_t)i, rev_bits) : (uint32_t)i;
    if (layered_log > 0) {
        const int s = 31 - __clz((uint32_t)i + 1u);
        e = (((uint32_t)i + 1u) - (1u << s)) << (layered_log - 1 - s);
    }
    Fr acc = scale;
    for (int j = 0; j < 32; ++j) {
        if ((e >> j) & 1) acc = acc * tab.p[j];
    }
    out[i] = acc;
}

// R stages (a radix-2^R block) on every group of 2^R tile elements whose indices differ in the R bits [p0, p0+R):
// load from the LDS planes, R x 2^(R-1) butterflies in registers, store back.
// unit0: at global stage 0 every twiddle is w^0 = 1 and the product is skipped -- DIT: the caller guarantees that the stage's
// inputs are reduced (< 2p: canonical data or a pre-scale product); DIF: the difference stays lazy (< 2^(kdone + 2) p), which
// the canonicalising / pre-scale product that always follows a DIF stage 0 absorbs.
template <class P, bool DIT, int R, class GIdx>
__device__ __forceinline__ void ntt30_round(uint32_t* lds, const Tw<P>* __restrict__ tw, int log_n, int s_lo, int q0, int TT, uint32_t E,
                                            int done, const GIdx& gidx, bool unit0) {
    typedef Fp30<P> F;
    constexpr int NL = F::NL;
    constexpr int NE = 1 << R;
    const int p0 = q0 + TT;   // bit position of the round's lowest stage inside the tile index
    for (uint32_t g = threadIdx.x; g < (E >> R); g += NTT_THREADS) {
        const uint32_t low = g & ((1u << p0) - 1), high = g >> p0;
        const uint32_t ebase = (high << (p0 + R)) | low;
        F x[NE];
        G16_UNROLL for (int j = 0; j < NE; ++j) {
            const uint32_t c = lds_col(ebase | ((uint32_t)j << p0));
            G16_UNROLL for (int l = 0; l < NL; ++l) x[j].l[l] = lds[l * NTT_ROW + c];
        }
        const uint64_t gbase = gidx(ebase);  // global index of element 0 of the group (the R varying bits are zero there)
        G16_UNROLL for (int qq = 0; qq < R; ++qq) {
            const int q = DIT ? qq : (R - 1 - qq);   // stage inside the round
            const int s = s_lo + q0 + q;             // global stage: pairs differ in index bit s
            const int kdone = done + qq;             // DIF: operands are < 2^kdone * 1.1 p
            G16_UNROLL for (int j = 0; j < NE; ++j) {
                if (!((j >> q) & 1)) {
                    const int j2 = j | (1 << q);
                    // twiddle exponent = (global index of the lower element) mod 2^s, scaled to the n/2-entry table
                    // (stage s's 2^s twiddles are contiguous in the layered table: lanes that walk consecutive low index bits read
                    // consecutive 32-byte entries, where the flat n/2-entry table made every lane of a wave touch its own cache line)
                    const uint32_t gi = (uint32_t)gbase + ((uint32_t)j << (s_lo + q0));
                    const uint32_t widx = ((1u << s) - 1u) + (gi & ((1u << s) - 1u));
                    if (unit0 && s == 0) {   // uniform across the workgroup
                        const F u = x[j], v = x[j2];
                        x[j] = u.add(v);
                        x[j2] = DIT ? u.template sub<2>(v) : u.sub_pow2(v, kdone);
                        continue;
                    }
                    const F w = load_tw<P>(tw[widx]);
                    if (DIT) {
                        const F v = x[j2].mul_impl(w);           // < 1.01 p
                        const F u = x[j];
                        x[j] = u.add(v);
                        x[j2] = u.template sub<2>(v);
                    } else {
                        const F u = x[j], v = x[j2];
                        x[j] = u.add(v);
                        x[j2] = u.sub_pow2(v, kdone).mul_impl(w);
                    }
                }
            }
        }
        G16_UNROLL for (int j = 0; j < NE; ++j) {
            const uint32_t c = lds_col(ebase | ((uint32_t)j << p0));
            G16_UNROLL for (int l = 0; l < NL; ++l) lds[l * NTT_ROW + c] = x[j].l[l];
        }
    }
}

// One LDS-tiled pass over stages [s_lo, s_hi).  T = log2 of the contiguous run (ignored when s_lo == 0).
// Up to three independent transforms over the same domain in ONE launch (blockIdx.y): the a, b, c chains of the witness map
// (r1cs_to_qap.rs:201-207,220-221).  A sweep of one 2^22-point array is 2048 workgroups for 512 slots (and a rank's 2^19-point share
// of the distributed map only 256); three chains per launch fill the chip and cut the launches of the six transforms to a third.
template <class P>
struct NttBatch { Fp<P>* p[3]; };
// Optional fused input of a pass (b != nullptr): the element is (data * b - c) * zinv instead of data -- the pointwise quotient
// (a b - c) / Z(g) of r1cs_to_qap.rs:223-230 computed where the first sweep of the last transform reads it, instead of a kernel of
// its own that writes q for this sweep to read back (standard Montgomery products, as that kernel used).
template <class P>
struct NttQuot { const Fp<P>* b; const Fp<P>* c; Fp<P> zinv; };

template <class P, bool DIT>
__global__ __launch_bounds__(NTT_THREADS, G16_NTT_MIN_WAVES) void ntt30_pass_kernel(NttBatch<P> batch, const Tw<P>* __restrict__ tw,
                                                                 const Fp<P>* __restrict__ prescale, int log_n, int s_lo, int s_hi,
                                                                 int T, NttQuot<P> quot) {
    typedef Fp30<P> F;
    constexpr int NL = F::NL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);  // [NL][NTT_ROW]
    Fp<P>* __restrict__ data = batch.p[blockIdx.y];
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
    auto lds_load = [&](uint32_t e) -> F {
        F x;
        const uint32_t c = lds_col(e);
        G16_UNROLL for (int l = 0; l < NL; ++l) x.l[l] = lds[l * NTT_ROW + c];
        return x;
    };
    auto lds_store = [&](uint32_t e, const F& x) {
        const uint32_t c = lds_col(e);
        G16_UNROLL for (int l = 0; l < NL; ++l) lds[l * NTT_ROW + c] = x.l[l];
    };
    // ---- load the tile (coalesced 32-byte elements), optional pre-scale
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
        const uint64_t g = gidx(e);
        F x;
        if (quot.b) {
            const Fp<P> t = (data[g] * quot.b[g] - quot.c[g]) * quot.zinv;
            x = F::unpack(t.v);
        } else {
            x = F::unpack(data[g].v);
        }
        if (prescale) x = x.mul_impl(F::unpack(prescale[g].v));
        lds_store(e, x);
    }
    __syncthreads();
    // ---- stages, three at a time in registers
    int done = 0;  // stages of this pass already applied (= how often sums may have doubled, DIF)
    while (done < K) {
        const int R = (K - done) >= NTT_MAX_R ? NTT_MAX_R : (K - done);
        const int q0 = DIT ? done : (K - done - R);    // first (lowest) stage of this round, relative to s_lo
        if (R == 3) ntt30_round<P, DIT, 3>(lds, tw, log_n, s_lo, q0, TT, E, done, gidx, true);
        else if (R == 2) ntt30_round<P, DIT, 2>(lds, tw, log_n, s_lo, q0, TT, E, done, gidx, true);
        else ntt30_round<P, DIT, 1>(lds, tw, log_n, s_lo, q0, TT, E, done, gidx, true);
        __syncthreads();
        done += R;
    }
    // ---- canonicalise (Fp30::canonical_quick: quotient estimate from the top limb, one row of multiply-adds, three conditional
    // subtractions -- a third of the issue slots of the product with R' mod p used until round 3) and write back
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
        const F x = lds_load(e).canonical_quick();
        Fp<P> o;
        x.pack(o.v);
        data[gidx(e)] = o;
    }
}

// The last pass of a DIF transform and the first pass of the DIT transform that follows it work on the same contiguous
// tiles (stages [0, K)): fused, the tile stays in LDS in between -- one HBM round trip, one canonicalisation and one launch
// less per DIF/DIT pair (three pairs per witness map).  `prescale` is applied between the two (x lazy, < 2^K p: fine for
// the product).
template <class P>
__global__ __launch_bounds__(NTT_THREADS, G16_NTT_MIN_WAVES) void ntt30_dif_dit_kernel(NttBatch<P> batch, const Tw<P>* __restrict__ tw_dif,
                                                                    const Tw<P>* __restrict__ tw_dit, const Fp<P>* __restrict__ prescale,
                                                                    int log_n, int K) {
    typedef Fp30<P> F;
    constexpr int NL = F::NL;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* lds = reinterpret_cast<uint32_t*>(smem);  // [NL][NTT_ROW]
    Fp<P>* __restrict__ data = batch.p[blockIdx.y];
    const uint32_t E = 1u << K;
    const uint64_t base = (uint64_t)blockIdx.x << K;
    auto gidx = [&](uint32_t e) -> uint64_t { return base + e; };
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
        const F x = F::unpack(data[base + e].v);
        const uint32_t c = lds_col(e);
        G16_UNROLL for (int l = 0; l < NL; ++l) lds[l * NTT_ROW + c] = x.l[l];
    }
    __syncthreads();
    for (int done = 0; done < K;) {
        const int R = (K - done) >= NTT_MAX_R ? NTT_MAX_R : (K - done);
        const int q0 = K - done - R;
        // (without a pre-scale the DIT half below multiplies only the odd element of a stage-0 pair; DIF's lazy difference is
        // such an element, its lazy sum is bounded like every other DIF sum: unit0 is safe either way)
        if (R == 3) ntt30_round<P, false, 3>(lds, tw_dif, log_n, 0, q0, 0, E, done, gidx, true);
        else if (R == 2) ntt30_round<P, false, 2>(lds, tw_dif, log_n, 0, q0, 0, E, done, gidx, true);
        else ntt30_round<P, false, 1>(lds, tw_dif, log_n, 0, q0, 0, E, done, gidx, true);
        __syncthreads();
        done += R;
    }
    if (prescale) {   // each lane rewrites the columns it reads: no barrier inside the loop
        for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
            const uint32_t c = lds_col(e);
            F x;
            G16_UNROLL for (int l = 0; l < NL; ++l) x.l[l] = lds[l * NTT_ROW + c];
            x = x.mul_impl(F::unpack(prescale[base + e].v));
            G16_UNROLL for (int l = 0; l < NL; ++l) lds[l * NTT_ROW + c] = x.l[l];
        }
        __syncthreads();
    }
    for (int done = 0; done < K;) {
        const int R = (K - done) >= NTT_MAX_R ? NTT_MAX_R : (K - done);
        const bool unit0 = prescale != nullptr;   // stage-0 inputs are pre-scale products (< 1.01 p); otherwise lazy DIF outputs
        if (R == 3) ntt30_round<P, true, 3>(lds, tw_dit, log_n, 0, done, 0, E, done, gidx, unit0);
        else if (R == 2) ntt30_round<P, true, 2>(lds, tw_dit, log_n, 0, done, 0, E, done, gidx, unit0);
        else ntt30_round<P, true, 1>(lds, tw_dit, log_n, 0, done, 0, E, done, gidx, unit0);
        __syncthreads();
        done += R;
    }
    for (uint32_t e = threadIdx.x; e < E; e += NTT_THREADS) {
        const uint32_t c = lds_col(e);
        F x;
        G16_UNROLL for (int l = 0; l < NL; ++l) x.l[l] = lds[l * NTT_ROW + c];
        Fp<P> o;
        x.canonical_quick().pack(o.v);
        data[base + e] = o;
    }
}

// out[k] = in[bitrev(k)] * table[k] * cst      (table / cst in the w*R' form; either may be absent)
template <class P>
__global__ void bitrev_scale_kernel(Fp<P>* __restrict__ out, const Fp<P>* __restrict__ in, const Fp<P>* __restrict__ table, Fp<P> cst,
                                    int has_cst, int log_n) {
    typedef Fp30<P> F;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ((size_t)1 << log_n)) return;
    Fp<P> o = in[bitrev32((uint32_t)k, log_n)];
    if (table || has_cst) {
        F x = F::unpack(o.v);
        if (table) x = x.mul_impl(F::unpack(table[k].v));
        if (has_cst) x = x.mul_impl(F::unpack(cst.v));
        x.canonical_lt2p().pack(o.v);
    }
    out[k] = o;
}

template <class P>
__global__ void scale_table_kernel(Fp<P>* __restrict__ data, const Fp<P>* __restrict__ table, size_t n) {
    typedef Fp30<P> F;
    size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Fp<P> o;
    F::unpack(data[k].v).mul_impl(F::unpack(table[k].v)).canonical_lt2p().pack(o.v);
    data[k] = o;
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
    const int npass = (rest + NTT_MAX_STRIDED - 1) / NTT_MAX_STRIDED;
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

template <class P, bool DIT>
static int launch_pass(const NttBatch<P>& batch, int nbatch, const Tw<P>* tw, const Fp<P>* prescale, int log_n, const PassPlan& pp, hipStream_t st,
                       const NttQuot<P>* quot = nullptr) {
    const int K = pp.s_hi - pp.s_lo;
    const int TT = pp.s_lo == 0 ? 0 : pp.T;
    const size_t E = (size_t)1 << (K + TT);
    const size_t blocks = ((size_t)1 << log_n) / E;
    const size_t lds_bytes = (size_t)Fp30<P>::NL * NTT_ROW * sizeof(uint32_t);
    static PerDeviceOnce attr_once;
    std::atomic<bool>& attr_set = attr_once.flag();
    if (!attr_set) {
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt30_pass_kernel<P, DIT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        attr_set = true;
    }
    NttQuot<P> q;
    if (quot) q = *quot;
    else { q.b = nullptr; q.c = nullptr; q.zinv = Fp<P>::zero(); }
    hipLaunchKernelGGL((ntt30_pass_kernel<P, DIT>), dim3((unsigned)blocks, (unsigned)nbatch), dim3(NTT_THREADS), lds_bytes, st, batch, tw, prescale,
                       log_n, pp.s_lo, pp.s_hi, pp.T, q);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
static NttBatch<typename C::Fr::Params> make_batch(typename C::Fr* const* data, int nbatch) {
    NttBatch<typename C::Fr::Params> b;
    for (int i = 0; i < 3; ++i) b.p[i] = data[i < nbatch ? i : 0];
    return b;
}

template <class C>
int ntt_dif_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool inverse, hipStream_t st) {
    typedef typename C::Fr::Params P;
    if (nbatch < 1 || nbatch > 3) return G16_ERR_INTERNAL;
    if (d->log_n == 0) return G16_OK;
    auto passes = plan_passes(d->log_n);
    const Tw<P>* tw = reinterpret_cast<const Tw<P>*>(inverse ? d->tw_inv : d->tw_fwd);
    const NttBatch<P> b = make_batch<C>(data, nbatch);
    for (size_t i = passes.size(); i-- > 0;) G16_TRY((launch_pass<P, false>(b, nbatch, tw, nullptr, d->log_n, passes[i], st)));
    return G16_OK;
}
template <class C>
int ntt_dif(const Domain<C>* d, typename C::Fr* data, bool inverse, hipStream_t st) { return ntt_dif_batch<C>(d, &data, 1, inverse, st); }

// ntt_dif of q = (a .* b - c) * zinv, in place in `a`, with the pointwise quotient fused into the load of the first sweep
template <class C>
int ntt_dif_quotient(const Domain<C>* d, typename C::Fr* a, const typename C::Fr* b, const typename C::Fr* c, const typename C::Fr& zinv,
                     bool inverse, hipStream_t st) {
    typedef typename C::Fr::Params P;
    auto passes = plan_passes(d->log_n);
    const Tw<P>* tw = reinterpret_cast<const Tw<P>*>(inverse ? d->tw_inv : d->tw_fwd);
    typename C::Fr* data[1] = {a};
    const NttBatch<P> batch = make_batch<C>(data, 1);
    NttQuot<P> q;
    q.b = b; q.c = c; q.zinv = zinv;
    if (d->log_n == 0) return G16_ERR_INTERNAL;   // (a one-point domain has no pass to fuse into; the caller keeps the plain kernel)
    for (size_t i = passes.size(); i-- > 0;)
        G16_TRY((launch_pass<P, false>(batch, 1, tw, nullptr, d->log_n, passes[i], st, i + 1 == passes.size() ? &q : nullptr)));
    return G16_OK;
}

template <class C>
int ntt_dit_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool inverse, const typename C::Fr* prescale, hipStream_t st) {
    typedef typename C::Fr::Params P;
    if (nbatch < 1 || nbatch > 3) return G16_ERR_INTERNAL;
    if (d->log_n == 0) {
        if (prescale) for (int i = 0; i < nbatch; ++i) G16_TRY((scale_by_table<C>(data[i], prescale, 1, st)));
        return G16_OK;
    }
    auto passes = plan_passes(d->log_n);
    const Tw<P>* tw = reinterpret_cast<const Tw<P>*>(inverse ? d->tw_inv : d->tw_fwd);
    const NttBatch<P> b = make_batch<C>(data, nbatch);
    for (size_t i = 0; i < passes.size(); ++i)
        G16_TRY((launch_pass<P, true>(b, nbatch, tw, i == 0 ? prescale : nullptr, d->log_n, passes[i], st)));
    return G16_OK;
}
template <class C>
int ntt_dit(const Domain<C>* d, typename C::Fr* data, bool inverse, const typename C::Fr* prescale, hipStream_t st) {
    return ntt_dit_batch<C>(d, &data, 1, inverse, prescale, st);
}

// ntt_dif(inverse = dif_inverse) followed by ntt_dit(inverse = !dif_inverse, prescale), with the two innermost passes fused
template <class C>
int ntt_dif_dit_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool dif_inverse, const typename C::Fr* prescale, hipStream_t st) {
    typedef typename C::Fr::Params P;
    if (nbatch < 1 || nbatch > 3) return G16_ERR_INTERNAL;
    if (d->log_n == 0) {
        if (prescale) for (int i = 0; i < nbatch; ++i) G16_TRY((scale_by_table<C>(data[i], prescale, 1, st)));
        return G16_OK;
    }
    auto passes = plan_passes(d->log_n);
    const Tw<P>* tw1 = reinterpret_cast<const Tw<P>*>(dif_inverse ? d->tw_inv : d->tw_fwd);
    const Tw<P>* tw2 = reinterpret_cast<const Tw<P>*>(dif_inverse ? d->tw_fwd : d->tw_inv);
    const NttBatch<P> b = make_batch<C>(data, nbatch);
    for (size_t i = passes.size(); i-- > 1;) G16_TRY((launch_pass<P, false>(b, nbatch, tw1, nullptr, d->log_n, passes[i], st)));
    {
        const int K = passes[0].s_hi;
        const size_t blocks = ((size_t)1 << d->log_n) >> K;
        const size_t lds_bytes = (size_t)Fp30<P>::NL * NTT_ROW * sizeof(uint32_t);
        static PerDeviceOnce attr_once;
        std::atomic<bool>& attr_set = attr_once.flag();
        if (!attr_set) {
            G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ntt30_dif_dit_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)lds_bytes));
            attr_set = true;
        }
        hipLaunchKernelGGL((ntt30_dif_dit_kernel<P>), dim3((unsigned)blocks, (unsigned)nbatch), dim3(NTT_THREADS), lds_bytes, st, b, tw1, tw2, prescale,
                           d->log_n, K);
        G16_LAUNCH_CHECK();
    }
    for (size_t i = 1; i < passes.size(); ++i) G16_TRY((launch_pass<P, true>(b, nbatch, tw2, nullptr, d->log_n, passes[i], st)));
    return G16_OK;
}
template <class C>
int ntt_dif_dit(const Domain<C>* d, typename C::Fr* data, bool dif_inverse, const typename C::Fr* prescale, hipStream_t st) {
    return ntt_dif_dit_batch<C>(d, &data, 1, dif_inverse, prescale, st);
}

template <class C>
int bitrev_scale(const Domain<C>* d, typename C::Fr* out, const typename C::Fr* in, const typename C::Fr* table,
                 const typename C::Fr* cst, hipStream_t st) {
    typedef typename C::Fr Fr;
    const size_t n = d->n;
    Fr c = cst ? *cst : Fr::zero();
    hipLaunchKernelGGL((bitrev_scale_kernel<typename Fr::Params>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, in, table, c,
                       cst ? 1 : 0, d->log_n);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
int scale_by_table(typename C::Fr* data, const typename C::Fr* table, size_t n, hipStream_t st) {
    typedef typename C::Fr Fr;
    hipLaunchKernelGGL((scale_table_kernel<typename Fr::Params>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, data, table, n);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

// table[i] = scale * base^e(i) * R'   (the form the 30-bit kernels multiply by)
template <class Fr>
static int gen_powers30(Fr* out, size_t n, const Fr& base, const Fr& scale, int rev_bits, hipStream_t st, int layered_log = 0) {
    if (n == 0) return G16_OK;
    PowTable<Fr> tab;
    Fr p = base;
    for (int j = 0; j < 32; ++j) { tab.p[j] = p; p = p.sqr(); }
    const Fr scale30 = Fp30<typename Fr::Params>::std_to_r30(scale);
    hipLaunchKernelGGL((gen_powers_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, tab, scale30, rev_bits, layered_log);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

// out[i] = scale * base^i, i < n: in the w*R' form the 30-bit kernels multiply by (r30_form) or in the standard Montgomery form
template <class C>
int gen_power_table(typename C::Fr* out, size_t n, const typename C::Fr& base, const typename C::Fr& scale, bool r30_form, hipStream_t st) {
    typedef typename C::Fr Fr;
    if (r30_form) return gen_powers30<Fr>(out, n, base, scale, 0, st);
    if (n == 0) return G16_OK;
    PowTable<Fr> tab;
    Fr p = base;
    for (int j = 0; j < 32; ++j) { tab.p[j] = p; p = p.sqr(); }
    hipLaunchKernelGGL((gen_powers_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out, n, tab, scale, 0, 0);
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
    const size_t n = d->n;
    Fr omega = C::two_adic_root();
    for (int i = log_n; i < C::TWO_ADICITY; ++i) omega = omega.sqr();
    Fr omega_inv = omega.inverse();
    const Fr n_inv = Fr::from_u64((uint64_t)n).inverse();
    d->n_inv = Fp30<typename Fr::Params>::std_to_r30(n_inv);  // w*R' form, like every table below
    const Fr g = C::fr_generator(), g_inv = C::fr_generator_inv();
    d->zinv = (g.pow_u64((uint64_t)n) - Fr::one()).inverse();  // 1 / Z(g), r1cs_to_qap.rs:223-226 (standard form)
    int rc = G16_OK;
    const size_t ntw = n > 1 ? n - 1 : 1;   // layered: 2^s entries for every stage s < log_n
    typedef Tw<typename Fr::Params> TwE;
    Fr* tw_tmp = nullptr;   // the generator writes packed words; the tables hold the butterflies' own form (Tw)
    // every error exit gives back the staging buffer too (the OOM exits are exactly where its 32 n bytes matter); launches already
    // queued may still be reading it, so the stream is drained first
    auto fail = [&](int code) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(tw_tmp);
        domain_destroy<C>(d);
        return code;
    };
    if (hipMalloc((void**)&d->tw_fwd, ntw * sizeof(TwE)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->tw_inv, ntw * sizeof(TwE)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&tw_tmp, ntw * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->s1_br, n * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    if (hipMalloc((void**)&d->s2, n * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
    for (int dir = 0; dir < 2; ++dir) {
        if ((rc = gen_powers30<Fr>(tw_tmp, ntw, dir ? omega_inv : omega, Fr::one(), 0, st, log_n)) != G16_OK) return fail(rc);
        hipLaunchKernelGGL((tw_convert_kernel<typename Fr::Params>), dim3((unsigned)((ntw + 255) / 256)), dim3(256), 0, st, tw_tmp,
                           reinterpret_cast<TwE*>(dir ? d->tw_inv : d->tw_fwd), ntw);
        if (hipGetLastError() != hipSuccess) return fail(G16_ERR_HIP);
    }
    if ((rc = gen_powers30<Fr>(d->s1_br, n, g, n_inv, log_n, st)) != G16_OK) return fail(rc);
    if ((rc = gen_powers30<Fr>(d->s2, n, g_inv, n_inv, 0, st)) != G16_OK) return fail(rc);
    if (hipStreamSynchronize(st) != hipSuccess) return fail(G16_ERR_HIP);
    (void)hipFree(tw_tmp);
    *out = d;
    return G16_OK;
}

template <class C>
int domain_ensure_gpow(Domain<C>* d, hipStream_t st) {
    typedef typename C::Fr Fr;
    if (d->g_pow) return G16_OK;
    G16_HIP_TRY(hipMalloc((void**)&d->g_pow, d->n * sizeof(Fr)));
    return gen_powers30<Fr>(d->g_pow, d->n, C::fr_generator(), Fr::one(), 0, st);
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
    template int ntt_dif_quotient<C>(const Domain<C>*, typename C::Fr*, const typename C::Fr*, const typename C::Fr*, const typename C::Fr&, bool, hipStream_t); \
    template int ntt_dit<C>(const Domain<C>*, typename C::Fr*, bool, const typename C::Fr*, hipStream_t);          \
    template int ntt_dif_dit<C>(const Domain<C>*, typename C::Fr*, bool, const typename C::Fr*, hipStream_t);      \
    template int ntt_dif_batch<C>(const Domain<C>*, typename C::Fr* const*, int, bool, hipStream_t);              \
    template int ntt_dit_batch<C>(const Domain<C>*, typename C::Fr* const*, int, bool, const typename C::Fr*, hipStream_t);   \
    template int ntt_dif_dit_batch<C>(const Domain<C>*, typename C::Fr* const*, int, bool, const typename C::Fr*, hipStream_t); \
    template int bitrev_scale<C>(const Domain<C>*, typename C::Fr*, const typename C::Fr*, const typename C::Fr*,  \
                                 const typename C::Fr*, hipStream_t);                                              \
    template int scale_by_table<C>(typename C::Fr*, const typename C::Fr*, size_t, hipStream_t);                   \
    template int gen_power_table<C>(typename C::Fr*, size_t, const typename C::Fr&, const typename C::Fr&, bool, hipStream_t);

G16_INSTANTIATE_NTT(Bls12_381)
G16_INSTANTIATE_NTT(Bn254)

}  // namespace g16
