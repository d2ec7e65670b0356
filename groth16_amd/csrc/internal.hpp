// Internal plumbing shared by the translation units of libg16_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/g16_mi355x.h"
#include "curve.hpp"

namespace g16 {

void set_last_error(const char* what, hipError_t e, const char* file, int line);

#define G16_HIP_TRY(expr)                                                     \
    do {                                                                      \
        hipError_t e__ = (expr);                                              \
        if (e__ != hipSuccess) {                                              \
            ::g16::set_last_error(#expr, e__, __FILE__, __LINE__);            \
            return e__ == hipErrorOutOfMemory ? G16_ERR_OOM : G16_ERR_HIP;    \
        }                                                                     \
    } while (0)

#define G16_TRY(expr)                  \
    do {                               \
        int rc__ = (expr);             \
        if (rc__ != G16_OK) return rc__; \
    } while (0)

#define G16_LAUNCH_CHECK() G16_HIP_TRY(hipGetLastError())

// hipFuncSetAttribute is per device: one "already raised the dynamic-LDS limit" flag per (call site, device).  Atomic: the
// per-device host threads of a multi-device context reach the same call site concurrently (a lost race only repeats the call).
struct PerDeviceOnce {
    std::atomic<bool> done[64] = {};
    std::atomic<bool>& flag() {
        int dev = 0;
        (void)hipGetDevice(&dev);
        return done[dev & 63];
    }
};

// Device scratch arena: chunks are retained across calls (first call pays hipMalloc), offsets
// reset at the start of each top-level call.  Sized for 288 GB of HBM: nothing is ever freed
// mid-proof, so there is no hipFree-induced device sync on the hot path.
struct Arena {
    struct Chunk { char* p; size_t cap; size_t used; };
    std::vector<Chunk> chunks;
    size_t min_chunk = (size_t)256 << 20;
    void reset() { for (auto& c : chunks) c.used = 0; }
    int alloc(size_t bytes, void** out);
    template <class T> int alloc_n(size_t n, T** out) { return alloc(n * sizeof(T), (void**)out); }
    void release();
    size_t total() const { size_t t = 0; for (auto& c : chunks) t += c.cap; return t; }
};

struct EventTimer {
    hipEvent_t a = nullptr, b = nullptr;
    bool used = false;
    int start(hipStream_t s) {
        if (!a) { G16_HIP_TRY(hipEventCreate(&a)); G16_HIP_TRY(hipEventCreate(&b)); }
        used = true;
        G16_HIP_TRY(hipEventRecord(a, s));
        return G16_OK;
    }
    int stop(hipStream_t s) { G16_HIP_TRY(hipEventRecord(b, s)); return G16_OK; }
    double ms() {
        if (!used) return 0.0;
        float t = 0.f;
        if (hipEventElapsedTime(&t, a, b) != hipSuccess) return 0.0;
        return (double)t;
    }
    void destroy() { if (a) { (void)hipEventDestroy(a); (void)hipEventDestroy(b); a = b = nullptr; } }
};

// ---- NTT domain (ntt.hip) ---------------------------------------------------------------
template <class C>
struct Domain {
    typedef typename C::Fr Fr;
    int log_n = 0;
    size_t n = 0;
    // twiddles, layered per butterfly stage: entry 2^s - 1 + k = w^(k * n / 2^(s+1)), k < 2^s, s < log_n  (n - 1 entries;
    // stage s's twiddles are contiguous, so the lanes of a wave read consecutive entries instead of one cache line each)
    Fr* tw_fwd = nullptr;    // (entries in the butterflies' own form: ntt.hip `Tw` -- the 30-bit limbs, not packed words)
    Fr* tw_inv = nullptr;    // the same over w^-1
    Fr* s1_br = nullptr;     // n^-1 * g^bitrev(i)   (between inverse-DIF and coset-DIT)
    Fr* s2 = nullptr;        // n^-1 * g^-k          (after the final inverse-DIF, natural index)
    Fr* g_pow = nullptr;     // g^k natural          (unit-level coset fft only; lazily built)
    Fr n_inv, zinv;          // 1/n ; 1/(g^n - 1)
};
template <class C> int domain_create(int log_n, hipStream_t st, Domain<C>** out);
template <class C> void domain_destroy(Domain<C>* d);
template <class C> int domain_ensure_gpow(Domain<C>* d, hipStream_t st);

// in-place passes over n = 2^log_n elements on device
// natural -> bit-reversed (Gentleman-Sande), roots = d->tw_inv if inverse else d->tw_fwd, no scaling
template <class C> int ntt_dif(const Domain<C>* d, typename C::Fr* data, bool inverse, hipStream_t st);
// the same transform of q = (a .* b - c) * zinv (in place in a), the pointwise quotient fused into the first sweep's load
template <class C> int ntt_dif_quotient(const Domain<C>* d, typename C::Fr* a, const typename C::Fr* b, const typename C::Fr* c,
                                        const typename C::Fr& zinv, bool inverse, hipStream_t st);
// bit-reversed -> natural (Cooley-Tukey); each input element is first multiplied by prescale[i] if non-null
template <class C> int ntt_dit(const Domain<C>* d, typename C::Fr* data, bool inverse, const typename C::Fr* prescale, hipStream_t st);
// ntt_dif(inverse = dif_inverse) then ntt_dit(inverse = !dif_inverse, prescale); the two innermost passes share one kernel
template <class C> int ntt_dif_dit(const Domain<C>* d, typename C::Fr* data, bool dif_inverse, const typename C::Fr* prescale, hipStream_t st);
// the same three over nbatch <= 3 arrays of one domain in ONE launch per sweep (the a, b, c chains of the witness map)
template <class C> int ntt_dif_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool inverse, hipStream_t st);
template <class C> int ntt_dit_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool inverse, const typename C::Fr* prescale,
                                     hipStream_t st);
template <class C> int ntt_dif_dit_batch(const Domain<C>* d, typename C::Fr* const* data, int nbatch, bool dif_inverse,
                                         const typename C::Fr* prescale, hipStream_t st);
// out[k] = in[bitrev(k)] * table[k] * cst  (table may be null; has_cst selects the constant factor)
template <class C> int bitrev_scale(const Domain<C>* d, typename C::Fr* out, const typename C::Fr* in, const typename C::Fr* table,
                                    const typename C::Fr* cst, hipStream_t st);
template <class C> int scale_by_table(typename C::Fr* data, const typename C::Fr* table, size_t n, hipStream_t st);
// out[i] = scale * base^i for i < n, in the w*R' form of the 30-bit kernels (r30_form) or in the standard Montgomery form
template <class C> int gen_power_table(typename C::Fr* out, size_t n, const typename C::Fr& base, const typename C::Fr& scale, bool r30_form,
                                       hipStream_t st);

// ---- R1CS on device + witness map (witness_map.hip) -------------------------------------------
template <class C>
struct DeviceCircuit {
    typedef typename C::Fr Fr;
    uint64_t num_inputs = 0, num_constraints = 0, num_variables = 0;
    uint64_t* row_ptr[3] = {nullptr, nullptr, nullptr};
    uint32_t* col[3] = {nullptr, nullptr, nullptr};
    Fr* val[3] = {nullptr, nullptr, nullptr};
    uint64_t nnz[3] = {0, 0, 0};
    Domain<C>* dom = nullptr;
    // need_col[k]: the largest column any of the rows [0, (k + 1) n / Z_CHUNKS) of A, B, C reads (the instance copies of
    // r1cs_to_qap.rs:195-199 included) -- computed once at g16_circuit_load.  A host assignment is uploaded in Z_CHUNKS pieces and the
    // sparse mat-vec of row block k starts as soon as the piece holding need_col[k] has landed (ZUpload below).
    static constexpr int Z_CHUNKS = 8;
    uint64_t need_col[Z_CHUNKS] = {};
};
// a host assignment on its way to the device: witness_map_device issues the copies itself (in DeviceCircuit::Z_CHUNKS pieces on
// `copy_stream`, one event each) and interleaves them with the row blocks of the mat-vec that each piece unlocks
struct ZUpload {
    const void* host = nullptr;      // n_assign field elements
    hipStream_t copy_stream = nullptr;
    hipEvent_t landed[8] = {};       // landed[k]: piece k is in HBM; the LAST used one (index pieces - 1) covers the whole assignment
    int pieces = 0;                  // out: how many pieces were used
};
// d_z: full assignment on device; d_h: n Fr out (natural order).  Scratch comes from the arena.
// after the CSR upload: flag the unit coefficients in the DEVICE column indices (bit 31), see witness_map.hip
template <class C> int mark_unit_coefficients(DeviceCircuit<C>* ck, hipStream_t st);
// up != nullptr: d_z is an empty device buffer and the assignment still lies in host memory (ZUpload)
template <class C> int witness_map_device(const DeviceCircuit<C>* ck, const typename C::Fr* d_z, typename C::Fr* d_h, Arena& arena,
                                          hipStream_t st, EventTimer* ntt_timers = nullptr, ZUpload* up = nullptr);

// ---- distributed witness map (witness_map.hip): the same h over N ranks, one all-to-all per transform --------------------
// n = N * M, blk = M / N (needs N^2 | n, N a power of two <= 16).  Two distributions of an n-vector over the ranks:
//   residue:  rank r holds x[r + N * i2],  i2 < M          block:  rank r holds x[(r * blk + j) + M * k1],  j < blk, k1 < N
// An n-point transform is a local M-point transform, a twiddle, ONE exchange (chunk p of M / N elements goes to rank p) and a
// local N-point transform ("4-step"); residue -> block and block -> residue alternate, so ifft, coset fft, pointwise, coset ifft
// need no redistribution in between and every rank ends with the h coefficients of its block indices -- the cut h_query then
// has to be sharded by.  Modelled and index-checked in oracle/pymodel.py::distributed_witness_map.
template <class C>
struct DistWm {
    typedef typename C::Fr Fr;
    int rank = 0, world = 1, log_world = 0;
    size_t M = 0, blk = 0;
    Domain<C>* dom_m = nullptr;   // the M-point domain (root w_n^N)
    Fr* tw1 = nullptr;            // [M]       w_n^(-r k2)                        (w*R' form)
    Fr* sc_mid = nullptr;         // [N][blk]  n^-1 g^((r blk + j) + M k1)        (standard form, like the three below)
    Fr* tw2 = nullptr;            // [N][blk]  w_n^((r blk + j) ka)
    Fr* sc_out = nullptr;         // [N][blk]  n^-1 g^-((r blk + j) + M k1)
    Fr wn_fwd[8], wn_inv[8];      // w_N^k, w_N^-k for k < N / 2 (w_N = w_n^M)
    Fr zinv;
};
template <class C> int dwm_create(const DeviceCircuit<C>* ck, int rank, int world, hipStream_t st, DistWm<C>** out);
template <class C> void dwm_destroy(DistWm<C>* d);
// stage 0: z -> work[0..2] (a, b, c; M Fr each);  exchange work -> recv;   stage 1: recv[0..2] -> work[0..2];  exchange;
// stage 2: recv[0..2] -> work[0] (the quotient);   exchange work[0] -> recv[0];   stage 3: recv[0] -> h_local (M Fr)
template <class C> int dwm_stage(const DeviceCircuit<C>* ck, const DistWm<C>* dw, int stage, const typename C::Fr* d_z,
                                 typename C::Fr* const work[3], typename C::Fr* const recv[3], typename C::Fr* h_local, hipStream_t st);

// ---- MSM (msm.hip) ----------------------------------------------------------------------------
struct MsmPlan {
    int c = 0;          // window bits
    int W = 0;          // windows
    // Bucket groups.  Per-window mode (ad-hoc bases): one group per window, B = 2^(c-1) buckets each, c <= 16.
    // Merged mode (proving-key queries with precomputed window tables T[j][i] = 2^(cj) P_i): every window's digits fall
    // into ONE set of 2^(c-1) buckets (c <= 20), cut into `groups` = 2^(c-1) / B classes of B <= 2^15 buckets so that
    // the LDS-resident counting sort and the reductions keep their shape.
    int groups = 0;
    bool merged = false;
    uint32_t B = 0;     // buckets per group
    uint32_t Lmax = 0;  // segment length: entries one bucket-pass lane walks (any integer >= 8)
    // Batched-affine plan (batch_affine.hpp): R > 0 pads every bucket region of the sorted list to a multiple of 2^R entries
    // (holes = the identity); R levels of pairwise affine additions with one shared inversion per lane batch then shrink the
    // list 2^R-fold before the XYZZ bucket pass walks what is left (<= ceil(m / 2^R) points per bucket of m entries).
    int affine_levels = 0;
    uint32_t chunk = 0; // points per histogram/scatter block
    uint32_t K[10];     // signed-digit bias  sum_w 2^(c-1) 2^(cw)
    // BUCKET-SPACE SHARD of a merged plan (round 5; one rank of shard_n, every rank holding the WHOLE window table): the rank owns
    // the buckets b with b mod shard_n == shard_r and indexes them locally by k = b / shard_n, so its sort keeps only the entries of
    // that residue class, its bucket pass folds ~1/shard_n of them, and -- unlike a shard of the BASES, which leaves every rank all
    // 2^(c-1) buckets -- its reductions shrink shard_n-fold too.  groups * B >= ceil(2^(c-1) / shard_n) local buckets; fold_windows
    // turns the local sums into the rank's share  sum_{b owned} (b+1) S_b = shard_n sum_k (k+1) S_k + (shard_r + 1 - shard_n) sum_k S_k.
    int shard_n = 1, shard_r = 0;
    uint32_t buckets() const { return B * (uint32_t)groups; }
    // Reduction of one group of B buckets: chunks of min(G, B) buckets per lane give (sum_b (b - b_lo + 1) S_b, sum_b S_b);
    // the chunk offsets b_lo = G * ch are applied WITHOUT a scalar multiplication in the dependent chain: the window level
    // sums the chunk sums once per bit of ch (masked tree sums, all in parallel) and the host recombines
    //   sum_b (b + 1) S_b = P_0 + G * sum_k 2^k P_(2+k),   sum_b S_b = P_1.
    uint32_t G = 8;     // buckets per lane of the bucket reduction (power of two; 8 -- other sizes measured no faster, see make_msm_plan)
    uint32_t chunk_buckets() const { return B >= G ? G : B; }
    uint32_t chunks() const { return B / chunk_buckets(); }
    int chunk_bits() const { int k = 0; while ((1u << k) < chunks()) ++k; return k; }
    int planes() const { return 2 + chunk_bits(); }                // sums per group leaving the device
    int outputs() const { return groups * planes(); }             // XYZZ sums leaving the device per MSM (<= 224)
};
static constexpr int MSM_MAX_OUTPUTS = 256;
static constexpr int MSM_MERGED_MIN_C = 9, MSM_MERGED_MAX_C = 20;   // merged entries pack window j < 32 and point i < 2^26
static constexpr uint64_t MSM_MERGED_MAX_N = (uint64_t)1 << 26;
// merged_c = 0: per-window plan, window size from the cost model (or G16_MSM_WINDOW); else a merged plan with that c
// shard_n > 1 (merged plans only): the plan of rank shard_r's bucket-space shard (MsmPlan::shard_n)
int make_msm_plan(uint64_t n, int scalar_bits, const uint32_t* modulus_words, int mod_nwords, int merged_c, MsmPlan* plan, int shard_n = 1,
                  int shard_r = 0);
int msm_window_override();  // env G16_MSM_WINDOW (0 = auto)
// window size for the precomputed tables of a query with n bases (0 = no tables: disabled by G16_MSM_PRECOMP=0 or n too large)
int merged_window_bits(uint64_t n, int scalar_bits, const uint32_t* modulus_words, int mod_nwords);
int msm_plan_windows(int c, int scalar_bits, const uint32_t* modulus_words, int mod_nwords);

// digit extraction + bucket sort of one scalar array, shared by every MSM over those scalars
struct ScalarSort {
    MsmPlan plan;
    uint64_t n = 0;
    uint32_t* sorted = nullptr;    // [<= n*W] point index | sign<<31 (merged: | window<<26), grouped by bucket
    uint32_t* offsets = nullptr;   // [buckets + 1] exclusive prefix of bucket sizes
    uint32_t* task_off = nullptr;  // [buckets + 1] exclusive prefix of per-bucket partial-sum slots (one per segment a bucket touches)
    uint32_t* heavy = nullptr;     // [0] = number of buckets with more than HEAVY_PARTS partials, then their ids
    uint32_t max_tasks = 0;        // host-side upper bound on task_off[buckets] (partial slots)
    uint32_t max_segments = 0;     // host-side upper bound on ceil(entries the bucket pass walks / Lmax)
    uint64_t max_sorted = 0;       // host-side upper bound on offsets[buckets]: entries + padding of the batched-affine plan
};
template <class C> int sort_scalars(const typename C::Fr* d_scalars, uint64_t n, int merged_c, Arena& arena, hipStream_t st, ScalarSort* out,
                                    int shard_n = 1, int shard_r = 0);

// Pippenger over one base array using a ScalarSort, in two stream-separable halves:
//   msm_bucket_pass  the throughput-bound bucket accumulation.  Sorted index p addresses bases[p + shift] when
//                    0 <= p + shift < base_count (other entries are skipped: lets l_query reuse the sort made for a/b);
//                    with a merged plan `d_bases` is the window table and entry (j, p) addresses [j * base_count + p + shift]
//   msm_reduce       heavy-bucket combine, bucket reduction, window reduction (latency-bound, few waves): meant to
//                    run on another stream underneath the next MSM's bucket pass.  Leaves plan.outputs() sums (standard
//                    Montgomery form, XYZZ) in buf.window_sums: plan.planes() sums per group (see MsmPlan).
template <class F>
struct MsmBuffers {
    void* partials = nullptr;    // AccRaw records (lazy limbs), one per (bucket, segment) pair
    void* chunk_out = nullptr;   // AccRaw records, one per reduction chunk
    XYZZ<F>* window_sums = nullptr;
};
template <class F> int msm_bucket_pass(const Affine<F>* d_bases, int64_t shift, uint64_t base_count, const ScalarSort& ss, Arena& arena,
                                       hipStream_t st, MsmBuffers<F>* out, EventTimer* bucket_timer);
// the bucket passes of n <= 4 MSMs over one bucket layout (B, groups, segment length) as ONE launch: one tail instead of n
template <class F>
struct PassJob {
    const Affine<F>* bases;
    int64_t shift;
    uint64_t base_count;
    const ScalarSort* ss;
    MsmBuffers<F>* out;
};
template <class F> int msm_bucket_pass_batch(const PassJob<F>* jobs, int n, Arena& arena, hipStream_t st, EventTimer* bucket_timer);
template <class F> int msm_reduce(const MsmBuffers<F>& buf, const ScalarSort& ss, hipStream_t st);
// the reductions of n <= 4 MSMs whose plans have the same bucket layout as ONE launch per stage (they fill the chip together
// instead of each occupying a corner of it underneath the next MSM's bucket pass); G16_ERR_INTERNAL if the plans differ
// heavy_done: the heavy-bucket combines already ran (msm_heavy_reduce, one MSM at a time underneath the following pass)
template <class F> int msm_reduce_batch(const MsmBuffers<F>* const* bufs, const ScalarSort* const* sorts, int n, hipStream_t st, bool heavy_done);
template <class F> int msm_heavy_reduce(const MsmBuffers<F>& buf, const ScalarSort& ss, hipStream_t st);
template <class F> int msm_heavy_reduce_batch(const MsmBuffers<F>* const* bufs, const ScalarSort* const* sorts, int n, hipStream_t st);
// MSM bases are kept on the device in the bucket kernel's own Montgomery radix (x*R' with R' = 2^(30 NL), canonical,
// packed in the usual words): converted in place, once, after upload (F = Fq for G1, Fq2 for G2).
template <class F> int convert_bases(Affine<F>* d_bases, uint64_t n, hipStream_t st);
// host: per group T = P_0 + G sum_k 2^k P_(2+k), S = P_1; then sum_w 2^(c w) T_w (per-window plan) or sum_q T_q + B sum_q q S_q (merged)
template <class F> XYZZ<F> fold_windows(const XYZZ<F>* window_sums, const MsmPlan& plan);
// window tables for a merged plan: table[j * n + i] = 2^(c j) * src[i] for j < W, affine, in the bucket kernel's radix
// (what convert_bases leaves); src holds standard-form affine points and is not modified
// park: window_table_park_bytes<F>(n, W) of device scratch (rows waiting for the backward sweep of the shared inversion) that
// stays valid until `st` has run the launches; nullptr = allocated inside, and the call returns with the table finished
template <class F> int build_window_tables(const Affine<F>* d_src, uint64_t n, int c, int W, Affine<F>* d_table, hipStream_t st, void* park = nullptr);
template <class F> size_t window_table_park_bytes(uint64_t n, int W);

// ---- CRS generation (setup.hip) -------------------------------------------------------------------
template <class C>
int generate_parameters_device(hipStream_t st, Arena& arena, const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv,
                               const g16_toxic_waste* tw, const uint64_t* g1_gen, const uint64_t* g2_gen, const g16_params_view* out);
template <class C>
int qap_evaluations_host(const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv, const uint64_t* t, uint64_t* a_out, uint64_t* b_out,
                         uint64_t* c_out, uint64_t* zt_out);

// ---- point (de)serialisation (serialize.hip; host only) -----------------------------------------------
int serialize_points(int curve, int g2, int compressed, const uint64_t* points, uint64_t n, uint8_t* out);
int deserialize_points(int curve, int g2, int compressed, const uint8_t* in, uint64_t n, int validate, uint64_t* points_out);
uint64_t serialized_point_size(int curve, int g2, int compressed);

// ---- synthetic generators (synth.hip) -----------------------------------------------------------
int mad_rate_device(hipStream_t st, double* mad_per_s);   // measured v_mad_u64_u32 lane-operations per second (diagnostic)
template <class C> int synth_bases_device(int g2, uint64_t seed, uint64_t first, uint64_t n, void* out_dev, hipStream_t st);

}  // namespace g16
