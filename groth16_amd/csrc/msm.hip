// Variable-base multi-scalar multiplication (Pippenger bucket method) for gfx950.
//
// Replaces ark-ec's VariableBaseMSM::msm_bigint at its five call sites in the prover
// (/root/reference/src/prover.rs:66, 74 and 262 via calculate_coeff :92,105,113).
// The reference runs one serial bucket scan per window, windows in parallel (<= 17-way); the
// result of an MSM is a canonical group element, so the GPU is free to organise the work
// differently:
//
//   1. digits_kernel        scalars (Montgomery Fr) -> canonical (prover.rs:63-65 into_bigint)
//                           -> biased signed digits, written as W coalesced u16 "digit planes"
//   2. bucket_count_kernel  per (point-chunk, window) workgroup: LDS-resident bucket histogram
//                           (2^(c-1) counters, <= 128 KiB of the 160 KiB LDS) flushed with one
//                           global atomic per non-empty bucket
//   3. scan_*_kernel, bucket_slots_kernel   prefix sums: bucket offsets, then partial-sum slot offsets
//   4. bucket_scatter_kernel  same LDS histogram, then one global atomic per bucket reserves a
//                           slice and LDS atomics hand out the slots: a counting sort of
//                           (point index | sign) by (window, bucket) -- order inside a bucket is
//                           irrelevant because group addition commutes
//      (1, 2, 4 as written serve ad-hoc bases -- g16_msm_g1 / _g2; proving-key MSMs use the merged-window variants below:
//       class_count / class_partition, bucket_count_merged / bucket_wg_scan / bucket_scatter_merged -- no global atomics, 32 KB
//       histograms that fit on a compute unit beside the bucket passes)
//   5. bucket_accumulate30_kernel  THE hot kernel: one lane (G1) / lane pair (G2) per 64-entry SEGMENT of the
//                           sorted list gathers affine bases (96 B / 192 B each) and folds them with XYZZ
//                           mixed additions (8M+2S, no inversion; Y3 under one reduction) in 30-bit lazy arithmetic
//                           (fp30.hpp), the running sum's four coordinates parked in LDS (AccParked, round 4),
//                           flushing one partial sum per (bucket, segment) it touches.  ONE launch walks up to four MSMs
//                           (PassBatch, round 5: the G1 MSMs of a proof that are ready together -- one tail instead of four)
//   5b. heavy_reduce_kernel cooperative combine of the FEW buckets with MANY partial sums (> 16)
//   5c. bucket_combine_kernel (round 5)  every other bucket's partial sums added up by one lane per BUCKET, so that ...
//   6. bucket_reduce_kernel / window_reduce_kernel   ... sum_b (b+1) S_b by chunked running sums reads ONE sum per bucket: a chain
//                           of 2 G dependent additions per lane; chunk offsets through bit-plane sums of the chunk totals
//                           (MsmPlan in internal.hpp), not a scalar multiplication
//   7. host                 recombine the planes per group, then sum_w 2^(cw) R_w  (<= 256 doublings); a bucket-space shard
//                           (MsmPlan::shard_n, round 5: the rank owns the buckets b mod N == rank and indexes them by b / N)
//                           turns its local sums into its share  N sum_k (k+1) S_k - (N - 1 - rank) sum_k S_k
//
// Steps 1-4 depend only on the scalars and are shared by every MSM over the same scalar
// vector (a_query, b_g1_query, b_g2_query and l_query all use the witness: one sort, four
// accumulations).
//
// MERGED WINDOWS (proving-key queries).  The bases of a proving key are fixed, so g16_pk_load precomputes the window
// tables T[j][i] = 2^(cj) P_i (build_window_tables_kernel).  Window j's digit of scalar i then selects T[j][i] and ALL
// windows share one set of 2^(c-1) buckets: sum_i s_i P_i = sum_b (b+1) sum_{(i,j): |d_ij|-1 = b} +-T[j][i].  The bucket
// count no longer grows with the number of windows, so c rises from 16 to 20 and the bucket pass folds n*13 instead
// of n*16 points (BLS12-381 and BN254 scalars); the 2^19 buckets are cut into 64 sort classes of 2^13 for the LDS histogram
// (*_merged_kernel below) and into 16 groups of 2^15 for the reductions, which treat a group like a window.  Costs 13x the key's
// memory (31 GB at 2^22 constraints on BLS12-381 -- HBM capacity is what this GPU has to spare) and a one-off table build at load time.
// Roofline: step 5 reads ~N*W*(sizeof(Affine)+4) bytes (6.9 GB measured per MSM at 2^22, G1, merged windows) but spends ~10
// field products (3 169 v_mad_u64_u32) per 100 bytes, i.e. it is integer-VALU bound, not HBM bound; the
// bytes/s it sustains is reported against the 8 TB/s roofline by bench.py regardless.
#include "internal.hpp"
#include "msm_common.hpp"
#include "fp30.hpp"
#include "batch_affine.hpp"
#include "window_tables.hpp"
#include <algorithm>
#include <cstdlib>

namespace g16 {

static constexpr int SORT_THREADS = 1024;
static constexpr int ACC_THREADS = 64;   // one wave per workgroup: finer re-dispatch granularity.  Same box, full proof at 2^22 (round 3): 64 lanes
                                         // 79.56 ms (G2 pass 25.87), 128 lanes 80.38 / 80.17 (26.53 / 26.44), 256 lanes 80.81 (26.76); G1 passes equal
static constexpr int RED_THREADS = 64;
static constexpr int WIN_THREADS = 256;  // lanes of the per-window reduction
#ifndef G16_REDUCE_G
#define G16_REDUCE_G 8
#endif
static constexpr uint32_t REDUCE_G = G16_REDUCE_G;   // buckets per lane in the bucket reduction

struct PlanDev {
    int c, W;
    uint32_t B;
    uint32_t K[10];
    // bucket-space shard (MsmPlan::shard_n): keep the keys with key mod shard_n == shard_r, re-index them as key / shard_n.
    // shard_magic = ceil(2^32 / shard_n): umulhi(key, magic) == key / shard_n exactly for key < 2^20, shard_n <= 64
    // (key * (magic * shard_n - 2^32) < 2^20 * 64 < 2^32).  shard_n == 1: no filter.
    uint32_t shard_n, shard_r, shard_magic;
};

// false: the key belongs to another rank's residue class; true: *key is now the rank's local bucket index
__device__ __forceinline__ bool shard_local_key(const PlanDev& plan, uint32_t* key) {
    if (plan.shard_n <= 1) return true;
    const uint32_t q = __umulhi(*key, plan.shard_magic);
    if (*key - q * plan.shard_n != plan.shard_r) return false;
    *key = q;
    return true;
}

// ---------------------------------------------------------------------------------------------
// 1. digit planes
// ---------------------------------------------------------------------------------------------
template <class Fr>
__global__ __launch_bounds__(256) void digits_kernel(const Fr* __restrict__ scalars, uint64_t n, PlanDev plan,
                                                     uint16_t* __restrict__ planes) {
    __shared__ uint32_t sw[MSM_SWORDS][256];
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t s[Fr::N];
    scalars[i].to_canonical(s);
    uint64_t carry = 0;
    G16_UNROLL for (int k = 0; k < 10; ++k) {
        carry += (uint64_t)(k < Fr::N ? s[k] : 0u) + plan.K[k];
        sw[k][threadIdx.x] = (uint32_t)carry;
        carry >>= 32;
    }
    sw[10][threadIdx.x] = 0;
    // each thread reads back only its own column: no barrier needed
    for (int w = 0; w < plan.W; ++w) {
        const int bit = w * plan.c, word = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)sw[word][threadIdx.x] | ((uint64_t)sw[word + 1][threadIdx.x] << 32);
        planes[(uint64_t)w * n + i] = (uint16_t)((uint32_t)(two >> sh) & ((1u << plan.c) - 1u));
    }
}

// ---------------------------------------------------------------------------------------------
// 2. histogram   grid = (chunks, W)
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(SORT_THREADS) void bucket_count_kernel(const uint16_t* __restrict__ planes, uint64_t n, uint32_t chunk,
                                                                    int c, uint32_t B, uint32_t* __restrict__ counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const int w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) hist[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    const uint16_t* plane = planes + (uint64_t)w * n;
    for (uint64_t p = lo + threadIdx.x; p < hi; p += SORT_THREADS) {
        uint32_t bucket, neg;
        if (digit_to_bucket(plane[p], c, &bucket, &neg)) atomicAdd(&hist[bucket], 1u);
    }
    __syncthreads();
    uint32_t* out = counts + (uint64_t)w * B;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) {
        const uint32_t v = hist[b];
        if (v) atomicAdd(&out[b], v);
    }
}

// ---------------------------------------------------------------------------------------------
// 1m/2m/4m. merged-window variants: every (point, window) digit is an entry of ONE bucket set of 2^(c-1) buckets, cut
// into Q sort classes of 2^blog <= 2^13 buckets (what the LDS histogram holds: SORT_CLASS_LOG below).  Two levels:
//   class_count_kernel / class_partition_kernel   entries -> Q contiguous class regions (block-local LDS cursors on top
//                           of scanned per-(class, block) counts); an entry is a u16 key (bucket in class | sign << 15)
//                           and a u32 tag (point | window << 26)
//   bucket_count_merged_kernel / bucket_wg_scan_kernel / bucket_scatter_merged_kernel   per class: an LDS counting sort over the
//                           class region, its histograms exchanged through a [class][workgroup][bucket] matrix instead of global
//                           atomics.   Sorted entry = point | window << 26 | sign << 31.
// ---------------------------------------------------------------------------------------------
static constexpr int CLASS_THREADS = 256;
static constexpr int MAX_CLASSES = 64;
// The merged counting sort works on SORT CLASSES of 2^13 buckets (round 5; 2^15 before) with 256-lane workgroups: a 32 KB histogram and
// four waves (<= 32 registers each) fit on a compute unit BESIDE eight bucket-pass workgroups (8 x 13 KB of LDS, 2 x 190 / 2 x 239 of a
// SIMD's 512 registers), so h's sort -- which cannot start before the witness map ends, i.e. when the passes begin -- runs underneath the
// passes instead of waiting for a kernel boundary: with 128 KB histograms its count and scatter kernels each sat out a whole 23 ms
// pass and the h pass then waited 2.3 ms for the scatter (kernel trace of round 5).  The sorted list does not depend on the class
// size (classes are contiguous bucket ranges); the reductions keep their own grouping (MsmPlan::B, groups).
static constexpr uint32_t SORT_CLASS_LOG = 13;
static constexpr int SORT_M_THREADS = 256;

// the thread's scalar, biased, as little-endian words in its column of sw
template <class Fr>
__device__ __forceinline__ void biased_scalar(const Fr& sc, const PlanDev& plan, uint32_t (*sw)[CLASS_THREADS]) {
    uint32_t s[Fr::N];
    sc.to_canonical(s);
    uint64_t carry = 0;
    G16_UNROLL for (int k = 0; k < 10; ++k) {
        carry += (uint64_t)(k < Fr::N ? s[k] : 0u) + plan.K[k];
        sw[k][threadIdx.x] = (uint32_t)carry;
        carry >>= 32;
    }
    sw[10][threadIdx.x] = 0;
}
__device__ __forceinline__ uint32_t window_of(const uint32_t (*sw)[CLASS_THREADS], int w, int c) {
    const int bit = w * c, word = bit >> 5, sh = bit & 31;
    const uint64_t two = (uint64_t)sw[word][threadIdx.x] | ((uint64_t)sw[word + 1][threadIdx.x] << 32);
    return (uint32_t)(two >> sh) & ((1u << c) - 1u);
}

// block_counts[q * gridDim.x + block] = entries of class q among this block's 256 points
template <class Fr>
__global__ __launch_bounds__(CLASS_THREADS) void class_count_kernel(const Fr* __restrict__ scalars, uint64_t n, PlanDev plan, uint32_t blog,
                                                                    uint32_t Q, uint32_t* __restrict__ block_counts) {
    __shared__ uint32_t sw[MSM_SWORDS][CLASS_THREADS];
    __shared__ uint32_t cnt[MAX_CLASSES];
    if (threadIdx.x < MAX_CLASSES) cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * CLASS_THREADS + threadIdx.x;
    if (i < n) {
        biased_scalar(scalars[i], plan, sw);
        for (int w = 0; w < plan.W; ++w) {
            uint32_t key, neg;
            if (digit_to_bucket(window_of(sw, w, plan.c), plan.c, &key, &neg) && shard_local_key(plan, &key)) atomicAdd(&cnt[key >> blog], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < Q) block_counts[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = cnt[threadIdx.x];
}

template <class Fr>
__global__ __launch_bounds__(CLASS_THREADS) void class_partition_kernel(const Fr* __restrict__ scalars, uint64_t n, PlanDev plan, uint32_t blog,
                                                                        uint32_t Q, const uint32_t* __restrict__ block_off,
                                                                        uint16_t* __restrict__ ent_key, uint32_t* __restrict__ ent_tag) {
    __shared__ uint32_t sw[MSM_SWORDS][CLASS_THREADS];
    __shared__ uint32_t cur[MAX_CLASSES];
    if (threadIdx.x < Q) cur[threadIdx.x] = block_off[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x];
    __syncthreads();
    const uint64_t i = (uint64_t)blockIdx.x * CLASS_THREADS + threadIdx.x;
    if (i >= n) return;
    biased_scalar(scalars[i], plan, sw);
    for (int w = 0; w < plan.W; ++w) {
        uint32_t key, neg;
        if (digit_to_bucket(window_of(sw, w, plan.c), plan.c, &key, &neg) && shard_local_key(plan, &key)) {
            const uint32_t pos = atomicAdd(&cur[key >> blog], 1u);
            ent_key[pos] = (uint16_t)((key & ((1u << blog) - 1u)) | (neg << 15));
            ent_tag[pos] = (uint32_t)i | ((uint32_t)w << 26);
        }
    }
}

// grid = (G, Q): workgroup (x, q) owns the tiles x, x + G, x + 2G, ... (SORT_M_THREADS entries each) of class region q.
// NO global atomics (round 5).  The first version flushed every workgroup's histogram with one global atomic per non-empty bucket and
// reserved scatter slices with one RETURNING global atomic per non-empty bucket: 2048 workgroups x ~18 000 buckets = 37 M device-scope
// atomics per kernel at 2^22 -- on this GPU those are performed at the memory side (eight XCDs, eight L2s) -- 0.9 ms for the count and
// 2.5 ms for the scatter of a 54.5 M-entry sort whose compulsory traffic is ~1.3 GB.  Now each workgroup STORES its histogram as one
// row of a [class][workgroup][bucket] matrix (coalesced), a small kernel turns the columns into exclusive prefixes over the
// workgroups (one thread per bucket, coalesced across threads) and totals, and the scatter reads its row back: its slice of bucket b
// starts at offsets[b] + prefix[workgroup][b].  The scatter no longer re-counts its tiles either.
static __global__ __launch_bounds__(SORT_M_THREADS) void bucket_count_merged_kernel(const uint16_t* __restrict__ ent_key,
                                                                           const uint32_t* __restrict__ class_off, uint32_t nb, uint32_t B,
                                                                           uint32_t* __restrict__ wg_counts) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const uint32_t q = blockIdx.y;
    const uint32_t lo = class_off[(uint64_t)q * nb], hi = class_off[(uint64_t)(q + 1) * nb];
    for (uint32_t b = threadIdx.x; b < B; b += SORT_M_THREADS) hist[b] = 0;
    __syncthreads();
    for (uint64_t p = (uint64_t)lo + blockIdx.x * SORT_M_THREADS + threadIdx.x; p < hi; p += (uint64_t)gridDim.x * SORT_M_THREADS)
        atomicAdd(&hist[ent_key[p] & 0x7fffu], 1u);
    __syncthreads();
    uint32_t* out = wg_counts + ((uint64_t)q * gridDim.x + blockIdx.x) * B;   // a workgroup without tiles stores its row of zeros
    for (uint32_t b = threadIdx.x; b < B; b += SORT_M_THREADS) out[b] = hist[b];
}

// one thread per (class, bucket): wg_counts[q][x][b] <- sum of the rows x' < x (in place), counts[q * B + b] <- the column's total
static __global__ __launch_bounds__(256) void bucket_wg_scan_kernel(uint32_t* __restrict__ wg_counts, uint32_t G, uint32_t B, uint32_t M,
                                                                    uint32_t* __restrict__ counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    const uint32_t q = i / B, b = i - q * B;
    uint32_t* col = wg_counts + (uint64_t)q * G * B + b;
    uint32_t run = 0;
    for (uint32_t x = 0; x < G; ++x) {
        const uint32_t v = col[(uint64_t)x * B];
        col[(uint64_t)x * B] = run;
        run += v;
    }
    counts[i] = run;
}

static __global__ __launch_bounds__(SORT_M_THREADS) void bucket_scatter_merged_kernel(const uint16_t* __restrict__ ent_key,
                                                                             const uint32_t* __restrict__ ent_tag,
                                                                             const uint32_t* __restrict__ class_off, uint32_t nb, uint32_t B,
                                                                             const uint32_t* __restrict__ offsets,
                                                                             const uint32_t* __restrict__ wg_counts, uint32_t* __restrict__ sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const uint32_t q = blockIdx.y;
    const uint32_t lo = class_off[(uint64_t)q * nb], hi = class_off[(uint64_t)(q + 1) * nb];
    if ((uint64_t)lo + (uint64_t)blockIdx.x * SORT_M_THREADS >= hi) return;
    // hist[b] = this workgroup's write cursor in bucket b: the bucket's offset + what the workgroups before this one put there
    const uint64_t qb = (uint64_t)q * B;
    const uint32_t* mine = wg_counts + ((uint64_t)q * gridDim.x + blockIdx.x) * B;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_M_THREADS) hist[b] = offsets[qb + b] + mine[b];
    __syncthreads();
    const uint64_t first = (uint64_t)lo + blockIdx.x * SORT_M_THREADS + threadIdx.x, step = (uint64_t)gridDim.x * SORT_M_THREADS;
    for (uint64_t p = first; p < hi; p += step) {
        const uint32_t d = ent_key[p];
        const uint32_t pos = atomicAdd(&hist[d & 0x7fffu], 1u);
        sorted[pos] = ent_tag[p] | ((d >> 15) << 31);
    }
}

// ---------------------------------------------------------------------------------------------
// 3. scans over the M = W*B buckets: an exclusive prefix sum of a u32 array in three small coalesced launches
//    (block sums -> scan of block sums -> write), 2048 values per workgroup.  Used twice: bucket counts -> bucket
//    offsets, and per-bucket partial-slot counts -> slot offsets.
// ---------------------------------------------------------------------------------------------
static constexpr int SCAN_THREADS = 256;
static constexpr int SCAN_PER_THREAD = 8;
static constexpr int SCAN_TILE = SCAN_THREADS * SCAN_PER_THREAD;
// A bucket with more partial sums than this is combined cooperatively by a workgroup (heavy_reduce_kernel: FEW buckets with MANY partial
// sums -- the top window's short digit range, repeated scalars); up to this many are added up by one lane (bucket_combine_kernel).
// 8 until round 5: a bucket-space shard's top-window buckets hold ~2x the mean (224 entries = 8 - 9 segments of 32), thousands of them
// just over the threshold, and the cooperative kernel -- a 128-lane tree per bucket -- took 0.5 - 0.9 ms per MSM in the tail of the proof.
static constexpr uint32_t HEAVY_PARTS = 16;

// exclusive scan of one value per thread across the workgroup; returns the workgroup total through *total
__device__ __forceinline__ uint32_t block_exclusive(uint32_t v, uint32_t* sh, uint32_t* total) {
    const uint32_t tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_THREADS; d <<= 1) {
        const uint32_t add = tid >= d ? sh[tid - d] : 0u;
        __syncthreads();
        sh[tid] += add;
        __syncthreads();
    }
    const uint32_t incl = sh[tid];
    *total = sh[SCAN_THREADS - 1];
    __syncthreads();
    return incl - v;
}

// `pad` (a power of two minus one, or 0): every value is rounded up to a multiple of pad + 1 before it is summed -- bucket
// regions of the batched-affine plan start and end on multiples of 2^R entries
static __global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums_kernel(const uint32_t* __restrict__ vals, uint32_t M, uint32_t pad,
                                                                       uint32_t* __restrict__ block_sums) {
    __shared__ uint32_t sa[SCAN_THREADS];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t sum = 0, tot;
    G16_UNROLL for (int j = 0; j < SCAN_PER_THREAD; ++j) sum += (base + j < M) ? ((vals[base + j] + pad) & ~pad) : 0u;
    (void)block_exclusive(sum, sa, &tot);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

static __global__ __launch_bounds__(SCAN_THREADS) void scan_block_offsets_kernel(uint32_t* __restrict__ block_sums, uint32_t nblocks, uint32_t M,
                                                                          uint32_t* __restrict__ prefix) {
    __shared__ uint32_t sa[SCAN_THREADS];
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nblocks; base += SCAN_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t a = i < nblocks ? block_sums[i] : 0u;
        uint32_t tot;
        const uint32_t ea = block_exclusive(a, sa, &tot);
        if (i < nblocks) block_sums[i] = carry + ea;
        carry += tot;
    }
    if (threadIdx.x == 0) prefix[M] = carry;
}

static __global__ __launch_bounds__(SCAN_THREADS) void scan_write_kernel(const uint32_t* __restrict__ vals, uint32_t M, uint32_t pad,
                                                                  const uint32_t* __restrict__ block_base, uint32_t* __restrict__ prefix) {
    __shared__ uint32_t sa[SCAN_THREADS];
    const uint32_t base = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_PER_THREAD;
    uint32_t cv[SCAN_PER_THREAD], sum = 0, tot;
    G16_UNROLL for (int j = 0; j < SCAN_PER_THREAD; ++j) { cv[j] = (base + j < M) ? ((vals[base + j] + pad) & ~pad) : 0u; sum += cv[j]; }
    uint32_t a = block_base[blockIdx.x] + block_exclusive(sum, sa, &tot);
    G16_UNROLL for (int j = 0; j < SCAN_PER_THREAD; ++j) {
        if (base + j < M) { prefix[base + j] = a; a += cv[j]; }
    }
}

// partial-sum slots of bucket b = number of length-L segments of the sorted list its entries touch
// (batched-affine plan: the bucket pass walks the level-R list, whose bucket regions are offsets >> R)
static __global__ void bucket_slots_kernel(const uint32_t* __restrict__ offsets, uint32_t M, uint32_t off_shift, uint32_t lseg,
                                    uint32_t* __restrict__ nparts, uint32_t* __restrict__ heavy /* [0] = count, then bucket ids */) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= M) return;
    const uint32_t lo = offsets[b] >> off_shift, hi = offsets[b + 1] >> off_shift;
    const uint32_t np = hi > lo ? ((hi - 1) / lseg - lo / lseg + 1u) : 0u;
    nparts[b] = np;
    if (np > HEAVY_PARTS) heavy[1 + atomicAdd(&heavy[0], 1u)] = b;
}

// ---------------------------------------------------------------------------------------------
// 4. scatter   grid = (chunks, W)
// ---------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(SORT_THREADS) void bucket_scatter_kernel(const uint16_t* __restrict__ planes, uint64_t n, uint32_t chunk,
                                                                      int c, uint32_t B, const uint32_t* __restrict__ offsets,
                                                                      uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
    const int w = blockIdx.y;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) hist[b] = 0;
    __syncthreads();
    const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = min(n, lo + chunk);
    const uint16_t* plane = planes + (uint64_t)w * n;
    for (uint64_t p = lo + threadIdx.x; p < hi; p += SORT_THREADS) {
        uint32_t bucket, neg;
        if (digit_to_bucket(plane[p], c, &bucket, &neg)) atomicAdd(&hist[bucket], 1u);
    }
    __syncthreads();
    // reserve a slice of every non-empty bucket; hist[b] becomes this block's write cursor
    const uint64_t wb = (uint64_t)w * B;
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) {
        const uint32_t v = hist[b];
        if (v) hist[b] = offsets[wb + b] + atomicAdd(&cursor[wb + b], v);
    }
    __syncthreads();
    for (uint64_t p = lo + threadIdx.x; p < hi; p += SORT_THREADS) {
        uint32_t bucket, neg;
        if (digit_to_bucket(plane[p], c, &bucket, &neg)) {
            const uint32_t pos = atomicAdd(&hist[bucket], 1u);
            sorted[pos] = (uint32_t)p | (neg << 31);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 5. bucket accumulation (THE hot kernel), 30-bit lazy arithmetic (fp30.hpp); F30 = Fp30<P> (G1) or Fp2p30<P> (G2, lane pair).
// Work unit = a SEGMENT of Lseg consecutive entries of the bucket-sorted list, not a bucket: every lane walks exactly
// Lseg entries, so lanes of a wave finish together whatever the bucket-size distribution (one lane per bucket left ~18 %
// of lane time idle at a mean bucket load of 128, ~35 % at 16).  When the walk crosses a bucket boundary the running sum
// is flushed -- raw lazy limbs, no conversion -- into that bucket's slot for this segment (slot offsets from the scan),
// so a bucket ends up with one partial sum per segment it touches; the reduction kernels add them up.
// `bases` hold canonical x*R', y*R' packed in 32-bit words (convert_bases30_kernel).
// DIRECT (batched-affine plan): the walk is over the level-R list itself -- slot e holds an affine point (or the identity),
// bucket b owns slots [offsets[b] >> off_shift, offsets[b + 1] >> off_shift) -- instead of over sorted entry words.
// The accumulator's coordinates in LDS (AccParked, fp30.hpp): value v, limb i of lane t.  Limbs are grouped in fours so that a
// coordinate moves as ds_read_b128 / ds_write_b128 (each lane its own 16 bytes, consecutive lanes consecutive: conflict-free) plus
// single words for the NL mod 4 tail rows.  4 * NL * 64 words per 64-lane workgroup: 13 KB (NL = 13), i.e. 104 of the CU's 160 KB at
// two waves per SIMD (eight workgroups per CU).
template <class F30>
struct LdsAccStore {
    static constexpr int NL = F30::PREFIX_LIMBS;
    static constexpr int QUADS = NL / 4, TAIL = NL % 4;
    static constexpr int WORDS_PER_VALUE = NL * ACC_THREADS;
    uint32_t* quad;   // lds + 4 * lane
    uint32_t* tail;   // lds + 4 * QUADS * ACC_THREADS + lane
    __device__ __forceinline__ F30 ld(int v) const {
        asm volatile("" ::: "memory");   // a FRESH read every time: the point of parking is that the value is not kept live
        uint32_t w[NL];
        const uint32_t* q = quad + v * WORDS_PER_VALUE;
        G16_UNROLL for (int g = 0; g < QUADS; ++g) {
            const uint4 t = *reinterpret_cast<const uint4*>(q + g * 4 * ACC_THREADS);
            w[4 * g] = t.x; w[4 * g + 1] = t.y; w[4 * g + 2] = t.z; w[4 * g + 3] = t.w;
        }
        const uint32_t* r = tail + v * WORDS_PER_VALUE;
        G16_UNROLL for (int i = 0; i < TAIL; ++i) w[4 * QUADS + i] = r[i * ACC_THREADS];
        return F30::from_limbs(w);
    }
    __device__ __forceinline__ void st(int v, const F30& a) const {
        uint32_t w[NL];
        a.get_limbs(w);
        uint32_t* q = quad + v * WORDS_PER_VALUE;
        G16_UNROLL for (int g = 0; g < QUADS; ++g)
            *reinterpret_cast<uint4*>(q + g * 4 * ACC_THREADS) = make_uint4(w[4 * g], w[4 * g + 1], w[4 * g + 2], w[4 * g + 3]);
        uint32_t* r = tail + v * WORDS_PER_VALUE;
        G16_UNROLL for (int i = 0; i < TAIL; ++i) r[i * ACC_THREADS] = w[4 * QUADS + i];
        asm volatile("" ::: "memory");
    }
};

// The reduction kernels add with acc_add_streamed (fp30.hpp) for the fields whose limbs can be moved word by word (the same
// fields whose bucket pass parks its accumulator); -DG16_REDUCE_STREAMED=0 rebuilds the register-resident form (A/B).
#ifndef G16_REDUCE_STREAMED
#define G16_REDUCE_STREAMED 1
#endif
template <class F30>
static constexpr bool REDUCE_STREAMED = F30::ACC_PARKED && (G16_REDUCE_STREAMED != 0);

// Stores for the reductions' streamed additions (acc_add_streamed, fp30.hpp).
// An AccRaw record in LDS or global memory as the lane(s) of one task see it: coordinate k of a one-lane field is the k-th F30 of the
// record; of the lane pair, component (lane parity) of the k-th Fq2.  Identity <=> zz is all-zero limbs (AccRaw).
template <class F30>
struct RawAccStore {
    AccRaw<typename F30::Raw>* p;
    __device__ __forceinline__ F30 ld(int k) const {
        asm volatile("" ::: "memory");   // a fresh read where it is needed, not a value kept live from the top of the formula
        if constexpr (F30::LANES_PER_TASK == 1) {
            static_assert(sizeof(F30) == sizeof(typename F30::Raw), "one-lane fields share the raw layout");
            return reinterpret_cast<const F30*>(p)[k];
        } else {
            typedef typename F30::B B30;
            return F30{reinterpret_cast<const B30*>(p)[2 * k + (F30::lane_hi() ? 1 : 0)]};
        }
    }
    __device__ __forceinline__ void st(int k, const F30& a) const {
        if constexpr (F30::LANES_PER_TASK == 1) {
            reinterpret_cast<F30*>(p)[k] = a;
        } else {
            typedef typename F30::B B30;
            reinterpret_cast<B30*>(p)[2 * k + (F30::lane_hi() ? 1 : 0)] = a.c;
        }
        asm volatile("" ::: "memory");
    }
    __device__ __forceinline__ bool inf() const {
        const F30 zz = ld(2);
        if constexpr (F30::LANES_PER_TASK == 1) return zz.raw_zero();
        else return F30::both(zz.c.raw_zero());
    }
    __device__ __forceinline__ void set_inf() const { st(2, F30::zero()); }
};
// four coordinates in registers (one-lane fields: the running sum of the bucket reduction)
template <class F30>
struct RegAccStore {
    F30 v[4];
    __device__ __forceinline__ F30 ld(int k) const { return v[k]; }
    __device__ __forceinline__ void st(int k, const F30& a) { v[k] = a; }
};
template <class F30, class St>
__device__ __forceinline__ Acc30<F30> gather_acc(const St& st, bool inf) {
    Acc30<F30> a;
    a.inf = inf;
    if (inf) { a.x = a.y = a.zz = a.zzz = F30::zero(); return a; }
    a.x = st.ld(0); a.y = st.ld(1); a.zz = st.ld(2); a.zzz = st.ld(3);
    return a;
}

// ONE launch walks up to PASS_BATCH MSMs (blockIdx.y = MSM of the batch: its own bases, sorted list, offsets and partial-sum slots; one
// plan).  A pass ends with a tail -- the last segments run on a chip that is emptying: about half a segment's time when the lanes are
// re-filled dynamically -- and a launch's tail cannot be filled by the next launch of the same stream.  The G1 MSMs of a proof that are
// ready together (l, a, b_g1 share the witness sort; h joins when its sort is done in time) therefore go as ONE launch: one tail
// instead of three or four (a rank's share of an 8-way sharded 2^22 proof: 1.3-1.4 ms per pass of 1.1 ms of work, round 5).
static constexpr int PASS_BATCH = 4;
template <class F30>
struct PassBatch {
    const Affine<typename F30::Std>* bases[PASS_BATCH];
    int64_t shift[PASS_BATCH];
    uint64_t base_count[PASS_BATCH];
    const uint32_t* sorted[PASS_BATCH];
    const uint32_t* offsets[PASS_BATCH];
    const uint32_t* slot_off[PASS_BATCH];
    AccRaw<typename F30::Raw>* partials[PASS_BATCH];
};

template <class F30, bool DIRECT>
__global__ __launch_bounds__(ACC_THREADS, F30::ACC_MIN_WAVES) void bucket_accumulate30_kernel(
    PassBatch<F30> batch, uint32_t M, uint32_t off_shift, uint32_t lseg, uint32_t merged) {
    // (uniform index: the seven values come out of the kernel arguments with scalar loads)
    const Affine<typename F30::Std>* __restrict__ bases = batch.bases[blockIdx.y];
    const int64_t shift = batch.shift[blockIdx.y];
    const uint64_t base_count = batch.base_count[blockIdx.y];
    const uint32_t* __restrict__ sorted = batch.sorted[blockIdx.y];
    const uint32_t* __restrict__ offsets = batch.offsets[blockIdx.y];
    const uint32_t* __restrict__ slot_off = batch.slot_off[blockIdx.y];
    AccRaw<typename F30::Raw>* __restrict__ partials = batch.partials[blockIdx.y];
    const uint32_t t = (blockIdx.x * ACC_THREADS + threadIdx.x) / F30::LANES_PER_TASK;   // lanes of one task are adjacent
    const uint32_t S = offsets[M] >> off_shift;                                            // entries in total
    const uint64_t start64 = (uint64_t)t * lseg;
    if (start64 >= S) return;
    const uint32_t start = (uint32_t)start64;
    const uint32_t end = (uint32_t)min((uint64_t)S, start64 + lseg);
    uint32_t lo = 0, hi = M;  // bucket containing `start`: offsets[lo] <= start < offsets[lo + 1]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if ((offsets[mid] >> off_shift) <= start) lo = mid; else hi = mid;
    }
    uint32_t b = lo, b_first = offsets[b] >> off_shift, b_end = offsets[b + 1] >> off_shift;
    // F30::ACC_PARKED: the running sum's coordinates live in LDS (AccParked); otherwise in registers (Acc30)
    __shared__ __attribute__((aligned(16))) uint32_t acc_lds[F30::ACC_PARKED ? 4 * F30::PREFIX_LIMBS * ACC_THREADS : 4];
    typedef typename std::conditional<F30::ACC_PARKED, AccParked<F30, LdsAccStore<F30>>, Acc30<F30>>::type Acc;
    Acc acc;
    if constexpr (F30::ACC_PARKED) {
        acc.s.quad = acc_lds + 4 * threadIdx.x;
        acc.s.tail = acc_lds + 4 * LdsAccStore<F30>::QUADS * ACC_THREADS + threadIdx.x;
        acc.inf = true;
    } else {
        acc = Acc30<F30>::identity();
    }
    auto flush = [&](AccRaw<typename F30::Raw>* dst) {
        if constexpr (F30::ACC_PARKED) { acc.gather().store_raw(dst); acc.inf = true; }
        else { acc.store_raw(dst); acc = Acc30<F30>::identity(); }
    };
    // software pipeline: the base point of entry e+1 is gathered (and the sorted word of entry e+2 loaded) before the
    // ~20k-instruction addition of entry e, so the HBM latencies hide under arithmetic.  (Touching the cache lines of
    // entry e+2 with direct-to-LDS loads was tried for the window tables' wider gather: 25 % slower.)
    uint32_t v_next = 0;
    // a parked accumulator takes y PACKED (12 words as gathered): the signed digit's negation costs one subtract-with-borrow per word
    // there, and the unpacking happens once, when the point is added (AccParked::add_affine_packed)
    typedef typename std::conditional<F30::ACC_PARKED, typename F30::PackedC, F30>::type YNext;
    F30 px_next = F30::zero();
    YNext py_next = YNext::zero();
    bool ok_next = false;
    auto decode = [&](uint32_t v, int64_t* at) -> bool {
        // per-window plan: entry = point | sign<<31 ; merged plan: point | window<<26 | sign<<31, base = table[window][point]
        const uint32_t pt = merged ? (v & 0x3ffffffu) : (v & 0x7fffffffu);
        const uint64_t row = merged ? (uint64_t)((v >> 26) & 31u) * base_count : 0;
        const int64_t idx = (int64_t)pt + shift;
        *at = (int64_t)row + idx;
        return idx >= 0 && (uint64_t)idx < base_count;
    };
    auto fetch = [&](uint32_t v) {   // DIRECT: v is the slot index itself
        v_next = DIRECT ? 0u : v;
        int64_t at = (int64_t)v;
        ok_next = DIRECT ? true : decode(v, &at);
        if (ok_next) {
            if constexpr (F30::ACC_PARKED) ok_next = F30::load_point_py(bases, at, px_next, py_next);
            else ok_next = F30::load_point(bases, at, px_next, py_next);
        }
    };
    fetch(DIRECT ? start : sorted[start]);
    uint32_t v_fetch = DIRECT ? start + 1 : (start + 1 < end ? sorted[start + 1] : 0u);   // entry e+1's word, loaded one iteration early
    for (uint32_t e = start; e < end; ++e) {
        if (e == b_end) {  // crossed into the next non-empty bucket: flush this segment's share of bucket b
            flush(&partials[slot_off[b] + (t - b_first / lseg)]);
            do { ++b; b_end = offsets[b + 1] >> off_shift; } while (b_end <= e);
            b_first = offsets[b] >> off_shift;
        }
        const uint32_t v = v_next;
        const F30 px = px_next;
        YNext py = py_next;
        const bool ok = ok_next;
        auto advance = [&]() {
            if (e + 1 < end) fetch(v_fetch);
            if constexpr (DIRECT) v_fetch = e + 2;
            else v_fetch = e + 2 < end ? sorted[e + 2] : 0u;
        };
        if constexpr (F30::ACC_PREFETCH) advance();
        if (ok) {
            if constexpr (F30::ACC_PARKED) {
                acc.add_affine_packed(px, py, (v >> 31) != 0);   // the digit's sign and the parked sum's sign are one flip (AccParked)
            } else {
                if (v >> 31) py = py.neg2();
                acc.add_affine(px, py);
            }
        }
        if constexpr (!F30::ACC_PREFETCH) advance();
    }
    flush(&partials[slot_off[b] + (t - b_first / lseg)]);
}

// ---------------------------------------------------------------------------------------------
// 5a. batched-affine tree levels (batch_affine.hpp): one launch per level, AFF_THREADS / 64 waves per workgroup
// ---------------------------------------------------------------------------------------------
static constexpr int AFF_THREADS = 128;
template <class F30, bool LEVEL0>
__global__ __launch_bounds__(AFF_THREADS, 2) void affine_level_kernel(AffineLevelArgs<F30> args) {
    const uint32_t wave = (blockIdx.x * AFF_THREADS + threadIdx.x) >> 6;
    AffineLevel<F30, LEVEL0>::run(args, wave, threadIdx.x & 63u);
}

template <class P> G16_HD Fp<P> to_r30(const Fp<P>& x) { return Fp30<P>::std_to_r30(x); }
template <class P> G16_HD Fp2<P> to_r30(const Fp2<P>& x) { return {Fp30<P>::std_to_r30(x.c0), Fp30<P>::std_to_r30(x.c1)}; }

template <class F>
__global__ void convert_bases30_kernel(Affine<F>* __restrict__ pts, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Affine<F> a = pts[i];
    a.x = to_r30(a.x);  // 0 stays 0: the identity encoding (0, 0) is preserved
    a.y = to_r30(a.y);
    pts[i] = a;
}

template <class F> struct Lazy30;   // field used by the reduction kernels (and by the G1 bucket pass)
template <class P> struct Lazy30<Fp<P>> { typedef Fp30<P> type; typedef Fp30<P> acc_type; };
template <class P> struct Lazy30<Fp2<P>> {
#if defined(G16_G2_REDUCE_FP2K30)
    typedef Fp2k30<P> type;       // register-passed Karatsuba (same raw limb layout): less scratch, but one wave per SIMD --
                                  // measured slower for the reductions (82.1 vs 80.8 ms per proof at 2^22, same box)
#elif defined(G16_G2_REDUCE_FP2X30)
    typedef Fp2x30<P> type;       // one lane per task, 4-product lazy Fq2 with out-of-line products (2.8 KB of scratch per lane)
#else
    typedef Fp2p30<P> type;       // reductions, too, on lane pairs: no scratch, each dependent group operation ~1.7x shorter --
                                  // what bounds a rank's share of a sharded proof is the G2 reduction chain
#endif
#if defined(G16_G2_ACC_FP2X30)
    typedef Fp2x30<P> acc_type;
#elif defined(G16_G2_ACC_FP2K30)
    typedef Fp2k30<P> acc_type;   // register-passed Karatsuba, one lane per bucket
#else
    typedef Fp2p30<P> acc_type;   // bucket pass: lane-pair Fq2 (two lanes per bucket, one component each)
#endif
};

// ---------------------------------------------------------------------------------------------
// 5b. heavy buckets: a bucket with many partial sums (short top window, repeated scalars such as the all-equal
//     witness of benches/bench.rs:52-54, a boolean witness, ...) has them combined cooperatively by one workgroup; the
//     result replaces the bucket's first partial (the bucket reduction reads only that one for such buckets).
// ---------------------------------------------------------------------------------------------
static constexpr int HEAVY_THREADS = 128;
// Waves per SIMD the one-lane (G1) first-stage reduction kernels are compiled for.  -DG16_FIRST_STAGE_WAVES=4 holds them to 128 registers
// (92 - 112 B of scratch per lane on BLS12-381) so that one of their waves fits BESIDE two G1-pass waves and the stage runs underneath
// the next pass instead of waiting for its workgroups to retire -- measured and lost (round 5, same box, profiles/
// r05_ab_first_stage_128_registers.txt): 65.88 / 65.89 vs 65.57 / 65.52 ms per proof, 8-way share 10.85 / 10.88 vs 10.62 / 10.60 ms.
// The stage is off the critical path either way; co-resident it takes issue slots from the pass and pays for its spills.
#ifndef G16_FIRST_STAGE_WAVES
#define G16_FIRST_STAGE_WAVES 2
#endif
static constexpr int HEAVY_BLOCKS = 512;

// The reductions of up to REDUCE_BATCH MSMs run as ONE launch each (blockIdx.y / .z = MSM of the batch): a reduction is a chain of
// dependent additions in a few hundred waves, so one MSM's reduction alone leaves most of the chip idle, and run underneath the next
// MSM's bucket pass it takes register slots from it for milliseconds (8-way shard at 2^22: 1.8-2.1 ms per pass instead of 1.2).
// The prover therefore runs its four G1 bucket passes back to back and reduces them together.  All MSMs of a batch share the plan
// (bucket count, chunking); slot offsets / heavy lists are per MSM (h has its own sort).
static constexpr int REDUCE_BATCH = 4;
template <class Raw, class X>
struct ReduceBatch {
    Raw* partials[REDUCE_BATCH];
    const uint32_t* slot_off[REDUCE_BATCH];
    const uint32_t* heavy[REDUCE_BATCH];
    Raw* chunk_out[REDUCE_BATCH];       // weighted chunk sums, then (at + chunks) the plain chunk sums
    X* window_sums[REDUCE_BATCH];
};

// (task = one lane, or one lane pair for the lane-pair Fq2: both lanes of a pair run the same control flow)
template <class F30>
__global__ __launch_bounds__(HEAVY_THREADS, (REDUCE_STREAMED<F30> && F30::LANES_PER_TASK == 1) ? G16_FIRST_STAGE_WAVES : REDUCE_STREAMED<F30> ? 2 : 1) void heavy_reduce_kernel(ReduceBatch<AccRaw<typename F30::Raw>, XYZZ<typename F30::Std>> batch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    AccRaw<typename F30::Raw>* sh = reinterpret_cast<AccRaw<typename F30::Raw>*>(smem);
    constexpr uint32_t LPT = F30::LANES_PER_TASK, TASKS = HEAVY_THREADS / LPT;
    AccRaw<typename F30::Raw>* __restrict__ partials = batch.partials[blockIdx.y];
    const uint32_t* __restrict__ slot_off = batch.slot_off[blockIdx.y];
    const uint32_t* __restrict__ heavy = batch.heavy[blockIdx.y];
    const uint32_t nheavy = heavy[0], task = threadIdx.x / LPT;
    for (uint32_t i = blockIdx.x; i < nheavy; i += gridDim.x) {
        const uint32_t b = heavy[1 + i];
        const uint32_t t0 = slot_off[b], t1 = slot_off[b + 1];
        if constexpr (REDUCE_STREAMED<F30>) {
            // streamed additions (acc_add_streamed): the task's sum lives in its LDS record from the start, every operand
            // coordinate is fetched where it is consumed -- no 8-coordinate operand pair in registers
            __syncthreads();  // previous iteration's readers are done with sh / partials[t0]
            const RawAccStore<F30> mine{&sh[task]};
            bool inf = true;
            for (uint32_t q = t0 + task; q < t1; q += TASKS) {
                const RawAccStore<F30> src{&partials[q]};
                acc_add_streamed<F30>(mine, inf, src, src.inf());
            }
            if (inf) mine.set_inf();
            __syncthreads();
            for (uint32_t d = TASKS / 2; d > 0; d >>= 1) {
                if (task < d) {
                    const RawAccStore<F30> other{&sh[task + d]};
                    bool mi = mine.inf();
                    const bool was = mi;
                    acc_add_streamed<F30>(mine, mi, other, other.inf());
                    if (mi && !was) mine.set_inf();
                }
                __syncthreads();
            }
            if (task == 0) Acc30<F30>::load_raw(sh[0]).store_raw(&partials[t0]);
        } else {
        Acc30<F30> acc = Acc30<F30>::identity();
        for (uint32_t q = t0 + task; q < t1; q += TASKS) acc.add(Acc30<F30>::load_raw(partials[q]));
        __syncthreads();  // previous iteration's readers are done with sh / partials[t0]
        acc.store_raw(&sh[task]);
        __syncthreads();
        for (uint32_t d = TASKS / 2; d > 0; d >>= 1) {
            if (task < d) {
                Acc30<F30> x = Acc30<F30>::load_raw(sh[task]);
                x.add(Acc30<F30>::load_raw(sh[task + d]));
                x.store_raw(&sh[task]);
            }
            __syncthreads();
        }
        if (task == 0) Acc30<F30>::load_raw(sh[0]).store_raw(&partials[t0]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 5c. every other bucket with more than one partial sum (2 .. HEAVY_PARTS of them): ONE task per BUCKET adds them up into the bucket's
//     first slot, so that the bucket reduction below -- one lane per G buckets, a chain of dependent additions whose length IS the time
//     it takes (a lone wave issues a dependent instruction every ~7.5 cycles: ~27 us per G1 addition) -- reads one sum per bucket:
//     G (np + 1) additions per lane become (np - 1) here, spread over G times the lanes, and 2 G there.  Round 5: the reductions are
//     what a rank's share of a sharded proof ends with (8-way at 2^22: 1.0 ms -> 0.45 ms of bucket level per G1 batch), and the G2
//     reduction runs its chain underneath the G1 passes.
// ---------------------------------------------------------------------------------------------
template <class F30>
__global__ __launch_bounds__(RED_THREADS, F30::LANES_PER_TASK == 1 ? G16_FIRST_STAGE_WAVES : 2) void bucket_combine_kernel(ReduceBatch<AccRaw<typename F30::Raw>, XYZZ<typename F30::Std>> batch, uint32_t M) {
    AccRaw<typename F30::Raw>* __restrict__ partials = batch.partials[blockIdx.y];
    const uint32_t* __restrict__ slot_off = batch.slot_off[blockIdx.y];
    const uint32_t b = (blockIdx.x * RED_THREADS + threadIdx.x) / F30::LANES_PER_TASK;   // lanes of one task are adjacent
    if (b >= M) return;
    const uint32_t t0 = slot_off[b], np = slot_off[b + 1] - t0;
    if (np < 2 || np > HEAVY_PARTS) return;   // (heavy buckets were combined into [t0] by heavy_reduce_kernel)
    if constexpr (REDUCE_STREAMED<F30>) {
        const RawAccStore<F30> mine{&partials[t0]};
        bool inf = mine.inf();
        const bool was = inf;
        for (uint32_t q = 1; q < np; ++q) {
            const RawAccStore<F30> src{&partials[t0 + q]};
            acc_add_streamed<F30>(mine, inf, src, src.inf());
        }
        if (inf && !was) mine.set_inf();
    } else {
        Acc30<F30> acc = Acc30<F30>::load_raw(partials[t0]);
        for (uint32_t q = 1; q < np; ++q) acc.add(Acc30<F30>::load_raw(partials[t0 + q]));
        acc.store_raw(&partials[t0]);
    }
}

// ---------------------------------------------------------------------------------------------
// 6. bucket reduction: chunk of G buckets per lane, then one workgroup per window
// ---------------------------------------------------------------------------------------------
template <class F30>
__global__ __launch_bounds__(RED_THREADS, (F30::LANES_PER_TASK == 1 || REDUCE_STREAMED<F30>) ? 2 : 1) void bucket_reduce_kernel(ReduceBatch<AccRaw<typename F30::Raw>, XYZZ<typename F30::Std>> batch, uint32_t B, int W,
                                                                    uint32_t G) {
    const AccRaw<typename F30::Raw>* __restrict__ partials = batch.partials[blockIdx.y];
    const uint32_t* __restrict__ slot_off = batch.slot_off[blockIdx.y];
    const uint32_t cpw = B / G;
    AccRaw<typename F30::Raw>* __restrict__ chunk_out = batch.chunk_out[blockIdx.y];
    AccRaw<typename F30::Raw>* __restrict__ chunk_sum = chunk_out + (size_t)cpw * W;
    const uint32_t t = (blockIdx.x * RED_THREADS + threadIdx.x) / F30::LANES_PER_TASK;   // lanes of one task are adjacent
    if (t >= cpw * (uint32_t)W) return;
    const uint32_t w = t / cpw, ch = t % cpw, b_lo = ch * G;
    if constexpr (REDUCE_STREAMED<F30>) {
        // two running sums, `run` (the buckets so far) and `tot` (the runs), added to with acc_add_streamed: the partial sums come
        // straight from their records in memory, `tot` lives in LDS, and `run` in LDS too for the lane pair (for the one-lane field it
        // stays in registers: 13 KB of LDS per workgroup keeps eight workgroups per CU for the batched G1 reduction of the tail)
        constexpr bool PAIR = F30::LANES_PER_TASK == 2;
        constexpr int SLOTS = PAIR ? 8 : 4;
        __shared__ __attribute__((aligned(16))) uint32_t red_lds[SLOTS * F30::PREFIX_LIMBS * RED_THREADS];
        static_assert(RED_THREADS == ACC_THREADS, "LdsAccStore is laid out for ACC_THREADS lanes");
        LdsAccStore<F30> tot_s;
        tot_s.quad = red_lds + 4 * threadIdx.x;
        tot_s.tail = red_lds + 4 * LdsAccStore<F30>::QUADS * RED_THREADS + threadIdx.x;
        typename std::conditional<PAIR, LdsAccStore<F30>, RegAccStore<F30>>::type run_s;
        if constexpr (PAIR) {
            run_s.quad = tot_s.quad + 4 * LdsAccStore<F30>::WORDS_PER_VALUE;
            run_s.tail = tot_s.tail + 4 * LdsAccStore<F30>::WORDS_PER_VALUE;
        }
        bool run_inf = true, tot_inf = true;
        for (uint32_t bb = G; bb-- > 0;) {
            const uint32_t gb = w * B + b_lo + bb;
            const uint32_t t0 = slot_off[gb];
            if (slot_off[gb + 1] != t0) {   // the bucket's partial sums were combined into [t0] (heavy_reduce_kernel / bucket_combine_kernel)
                const RawAccStore<F30> src{const_cast<AccRaw<typename F30::Raw>*>(&partials[t0])};
                acc_add_streamed<F30>(run_s, run_inf, src, src.inf());
            }
            acc_add_streamed<F30>(tot_s, tot_inf, run_s, run_inf);
        }
        gather_acc<F30>(run_s, run_inf).store_raw(&chunk_sum[t]);
        gather_acc<F30>(tot_s, tot_inf).store_raw(&chunk_out[t]);
    } else {
    Acc30<F30> run = Acc30<F30>::identity(), tot = Acc30<F30>::identity();
    for (uint32_t bb = G; bb-- > 0;) {
        const uint32_t gb = w * B + b_lo + bb;
        const uint32_t t0 = slot_off[gb];
        if (slot_off[gb + 1] != t0) run.add(Acc30<F30>::load_raw(partials[t0]));   // combined into [t0] by the two kernels above
        tot.add(run);
    }
    // sum_b (b+1) S_b over the chunk = tot + b_lo * run: the b_lo * run part is assembled from bit-plane sums of `run` over
    // the chunks by the window level and the host (MsmPlan) -- no scalar multiplication in this chain of dependent additions
    run.store_raw(&chunk_sum[t]);
    tot.store_raw(&chunk_out[t]);
    }
}

// grid = (groups, planes): plane 0 sums the chunks' weighted sums, plane 1 their plain sums, plane 2 + k the plain sums of the
// chunks whose index has bit k set
template <class F30>
__global__ __launch_bounds__(WIN_THREADS, REDUCE_STREAMED<F30> ? 2 : 1) void window_reduce_kernel(ReduceBatch<AccRaw<typename F30::Raw>, XYZZ<typename F30::Std>> batch, uint32_t cpw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    AccRaw<typename F30::Raw>* sh = reinterpret_cast<AccRaw<typename F30::Raw>*>(smem);
    constexpr uint32_t LPT = F30::LANES_PER_TASK, TASKS = WIN_THREADS / LPT;
    const uint32_t w = blockIdx.x, p = blockIdx.y, task = threadIdx.x / LPT;
    const AccRaw<typename F30::Raw>* __restrict__ chunk_out = batch.chunk_out[blockIdx.z];
    const AccRaw<typename F30::Raw>* __restrict__ chunk_sum = chunk_out + (size_t)cpw * gridDim.x;
    XYZZ<typename F30::Std>* __restrict__ window_sums = batch.window_sums[blockIdx.z];
    const AccRaw<typename F30::Raw>* src = (p == 0 ? chunk_out : chunk_sum) + (uint64_t)w * cpw;
    const uint32_t mask = p >= 2 ? 1u << (p - 2) : 0u;
    if constexpr (REDUCE_STREAMED<F30>) {   // streamed additions, as in heavy_reduce_kernel
        const RawAccStore<F30> mine{&sh[task]};
        bool inf = true;
        for (uint32_t j = task; j < cpw; j += TASKS)
            if (!mask || (j & mask)) {
                const RawAccStore<F30> s_j{const_cast<AccRaw<typename F30::Raw>*>(&src[j])};
                acc_add_streamed<F30>(mine, inf, s_j, s_j.inf());
            }
        if (inf) mine.set_inf();
        __syncthreads();
        for (uint32_t d = TASKS / 2; d > 0; d >>= 1) {
            if (task < d) {
                const RawAccStore<F30> other{&sh[task + d]};
                bool mi = mine.inf();
                const bool was = mi;
                acc_add_streamed<F30>(mine, mi, other, other.inf());
                if (mi && !was) mine.set_inf();
            }
            __syncthreads();
        }
    } else {
    Acc30<F30> acc = Acc30<F30>::identity();
    for (uint32_t j = task; j < cpw; j += TASKS)
        if (!mask || (j & mask)) acc.add(Acc30<F30>::load_raw(src[j]));
    acc.store_raw(&sh[task]);
    __syncthreads();
    for (uint32_t d = TASKS / 2; d > 0; d >>= 1) {
        if (task < d) {
            Acc30<F30> x = Acc30<F30>::load_raw(sh[task]);
            x.add(Acc30<F30>::load_raw(sh[task + d]));
            x.store_raw(&sh[task]);
        }
        __syncthreads();
    }
    }
    // the plane sums are what leaves the device: standard arkworks Montgomery radix
    if (task == 0) Acc30<F30>::load_raw(sh[0]).store_std(&window_sums[(uint64_t)w * gridDim.y + p]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// This file is compiled twice (G16_MSM_PART = 0: BLS12-381 + the curve-independent host helpers, 1: BN254) so that the
// two sets of template instantiations build in parallel.
#ifndef G16_MSM_PART
#define G16_MSM_PART 0
#endif
#if G16_MSM_PART == 0
int msm_window_override() {
    const char* e = getenv("G16_MSM_WINDOW");
    return e ? atoi(e) : 0;
}

// W for a given c: smallest W with (modulus - 1) + K < 2^(cW)
static int plan_windows(int c, int scalar_bits, const uint32_t* modulus_words, int mod_nwords, uint32_t* Kout) {
    int W = (scalar_bits + 1 + c - 1) / c;
    for (;; ++W) {
        if (c * W > 320) return -1;
        uint32_t K[MSM_SWORDS] = {0};
        for (int w = 0; w < W; ++w) {
            const int bit = w * c + c - 1;
            K[bit >> 5] |= 1u << (bit & 31);
        }
        uint32_t sum[MSM_SWORDS];
        uint64_t carry = 0;
        for (int k = 0; k < MSM_SWORDS; ++k) {
            carry += (uint64_t)K[k] + (k < mod_nwords ? modulus_words[k] : 0u);
            sum[k] = (uint32_t)carry;
            carry >>= 32;
        }
        bool ok = true;  // sum = modulus + K > (modulus - 1) + K ; require sum < 2^(cW)  (conservative by one)
        for (int bit = MSM_SWORDS * 32 - 1; bit >= c * W; --bit)
            if ((sum[bit >> 5] >> (bit & 31)) & 1) { ok = false; break; }
        if (ok) {
            if (Kout) for (int k = 0; k < 10; ++k) Kout[k] = K[k];
            return W;
        }
    }
}

int msm_plan_windows(int c, int scalar_bits, const uint32_t* modulus_words, int mod_nwords) {
    return plan_windows(c, scalar_bits, modulus_words, mod_nwords, nullptr);
}

// seconds, MI355X measurements of round 1: the bucket pass folds entries at ~4.5 G mixed adds/s; the bucket reduction is
// buckets/G lanes of (2G + ~24) dependent full additions, latency-bound (~12 us each) until the lanes exceed two waves
// per SIMD; per-group bookkeeping ~4 us
static double plan_cost(uint64_t n, int W, double buckets, int groups) {
    const double acc = (double)n * W / 4.5e9;
    const double lanes = buckets / REDUCE_G;
    const double red = (2.0 * REDUCE_G + 24.0) * 12e-6 * (lanes > 131072.0 ? lanes / 131072.0 : 1.0);
    return acc + red + 4e-6 * groups;
}

int merged_window_bits(uint64_t n, int scalar_bits, const uint32_t* modulus_words, int mod_nwords) {
    const char* e = getenv("G16_MSM_PRECOMP");
    if (e && atoi(e) == 0) return 0;
    if (n == 0 || n > MSM_MERGED_MAX_N) return 0;
    const char* f = getenv("G16_MSM_PRECOMP_WINDOW");
    int c = f ? atoi(f) : 0;
    if (c <= 0) {
        double best = 1e300;
        for (int cc = MSM_MERGED_MIN_C; cc <= MSM_MERGED_MAX_C; ++cc) {
            const int W = plan_windows(cc, scalar_bits, modulus_words, mod_nwords, nullptr);
            if (W < 0 || W > 32) continue;
            const double buckets = (double)(1u << (cc - 1));
            const int groups = cc > 16 ? 1 << (cc - 16) : 1;
            // the class filter of the merged counting sort re-reads the digit planes once per class
            const double t = plan_cost(n, W, buckets, groups) + (double)n * W * groups * 3.0 / 1.5e12;
            if (t < best) { best = t; c = cc; }
        }
    }
    if (c < MSM_MERGED_MIN_C) c = MSM_MERGED_MIN_C;
    if (c > MSM_MERGED_MAX_C) c = MSM_MERGED_MAX_C;
    const int W = plan_windows(c, scalar_bits, modulus_words, mod_nwords, nullptr);
    return (W < 0 || W > 32) ? 0 : c;
}

int make_msm_plan(uint64_t n, int scalar_bits, const uint32_t* modulus_words, int mod_nwords, int merged_c, MsmPlan* plan, int shard_n,
                  int shard_r) {
    if (shard_n < 1 || shard_n > 64 || shard_r < 0 || shard_r >= shard_n || (shard_n > 1 && merged_c <= 0)) return G16_ERR_INTERNAL;
    int c = merged_c;
    if (c <= 0) {
        c = msm_window_override();
        if (c <= 0) {
            double best = 1e300;
            for (int cc = 4; cc <= 16; ++cc) {
                const int W = plan_windows(cc, scalar_bits, modulus_words, mod_nwords, nullptr);
                if (W < 0) continue;
                const double t = plan_cost(n, W, (double)W * (1u << (cc - 1)), W);
                if (t < best) { best = t; c = cc; }
            }
        }
        if (c < 3) c = 3;
        if (c > 16) c = 16;
    } else if (c < MSM_MERGED_MIN_C || c > MSM_MERGED_MAX_C || n > MSM_MERGED_MAX_N) {
        return G16_ERR_INTERNAL;
    }
    const int W = plan_windows(c, scalar_bits, modulus_words, mod_nwords, plan->K);
    if (W < 0 || (merged_c > 0 && W > 32)) return G16_ERR_INTERNAL;
    plan->c = c;
    plan->W = W;
    plan->merged = merged_c > 0;
    plan->shard_n = shard_n;
    plan->shard_r = shard_r;
    if (plan->merged) {
        // classes of B <= 2^15 buckets (what the LDS histogram holds) over the rank's buckets: all 2^(c-1) of them, or -- bucket-space
        // shard -- the ceil(2^(c-1) / shard_n) local ones (bucket shard_n k + shard_r has local index k)
        const uint32_t all = 1u << (c - 1), local = (all + (uint32_t)shard_n - 1) / (uint32_t)shard_n;
        uint32_t B = 1;
        while (B < local && B < (1u << 15)) B <<= 1;
        plan->B = B;
        plan->groups = (int)((local + B - 1) / B);
    } else {
        plan->B = 1u << (c - 1);
        plan->groups = W;
    }
    // buckets per lane of the bucket reduction: 8.  16 halves the window level's tree sums but doubles the bucket level's chain;
    // measured at 2^22 on one box (round 3, tools/ab_reduce_g.sh): 8: 77.50 / 78.11 ms, 16: 78.27 / 77.81 ms, 32: 80.56 ms -- no
    // gain, so 8 stays.  G16_MSM_REDUCE_G forces 4 / 8 / 16 / 32 (tests, A/B).
    plan->G = REDUCE_G;
    if (const char* e = getenv("G16_MSM_REDUCE_G")) {
        const int v = atoi(e);
        if (v == 4 || v == 8 || v == 16 || v == 32) plan->G = (uint32_t)v;
    }
    const uint64_t all_entries = n * (uint64_t)W / (uint64_t)shard_n;   // EXPECTED entries of this rank (uniform digits); buffers
                                                                          // are sized for the worst case in sort_scalars
    // batched-affine levels: they pay when buckets are long (each level halves a bucket's entries at ~0.6 of the XYZZ cost
    // per addition, but a lane needs >= ~16 pairs to amortise its inversion and the chip >= ~2 k waves to stay busy)
    plan->affine_levels = 0;
    {
        const uint64_t mean_load = all_entries / plan->buckets();
        // Measured (profiles/r02_affine_v1_kernel_stats.csv, 2^22, BLS12-381): the levels LOSE on this GPU -- level 0 gathers
        // every operand twice and spills a prefix product per pair (~750 B of mostly random HBM traffic per addition against
        // ~137 B for the XYZZ walk), 8.0 ms for the 27 M additions that cost the XYZZ pass 6.3 ms -- so the default is 0 and the
        // path stays as a measured, tested experiment.
        (void)mean_load;
        int R = 0;
        if (const char* e = getenv("G16_MSM_AFFINE_LEVELS")) {   // 0 disables, 1..4 forces (tests run it at tiny sizes)
            const int v = atoi(e);
            if (v >= 0 && v <= 4) R = v;
        }
        if (all_entries + (uint64_t)plan->buckets() * ((1u << R) - 1u) >= ((uint64_t)1 << 32)) R = 0;
        if (shard_n > 1) R = 0;   // (an experiment that is off by default: not carried into the bucket-space shard)
        plan->affine_levels = R;
    }
    const uint64_t entries = (all_entries >> plan->affine_levels) + (plan->affine_levels ? plan->buckets() : 0);   // what the bucket pass walks
    // Segment length of the bucket pass (entries one lane walks; the kernels take any integer >= 8): 64 entries per lane,
    // shorter for small inputs so that the pass still has >= ~4 segments per lane slot of the chip (256 CUs x 4 SIMDs x 2 waves x 64
    // lanes).  Measured at 2^22 (profiles/r03_ab_segment_length.txt, one box): 60 and 64 tie, 70 / 84 +0.8 %, 104 +1.2 %, 139 +3 %;
    // for an 8-way shard 16 and 19 tie, 28 +2 %, 56 (ONE full round of the chip) +6 %.  Lanes do not take equal time (a lane that
    // crosses more bucket boundaries flushes more partial sums), workgroups are re-dispatched one by one as others retire, so many
    // short segments balance better than few long ones, and "fill whole rounds of the chip" -- tried in round 3 -- does not pay.
    uint32_t l = 64;
    while (l > 16 && entries / l < 4ull * 131072ull) l >>= 1;
    // ... but never much shorter than the mean bucket load: a bucket of ~mean entries then touches at most ~5 segments and
    // stays below the heavy-bucket threshold (otherwise EVERY bucket would go through the cooperative combine)
    const uint64_t mean = entries / plan->buckets() + 1;
    while (l < 128 && (uint64_t)l * 4 < mean) l <<= 1;
    if (const char* e = getenv("G16_MSM_SEGMENT")) {   // experiments: force the segment length (8 .. 4096)
        const int v = atoi(e);
        if (v >= 8 && v <= 4096) l = (uint32_t)v;
    }
    plan->Lmax = l;
    // histogram / scatter chunking: ~2048 workgroups in total, chunk a multiple of 1024 points
    uint64_t nchunks = 2048 / (uint64_t)plan->groups;
    if (nchunks < 1) nchunks = 1;
    uint64_t chunk = (n + nchunks - 1) / nchunks;
    chunk = (chunk + 1023) / 1024 * 1024;
    if (chunk < 1024) chunk = 1024;
    plan->chunk = (uint32_t)chunk;
    return G16_OK;
}

#endif  // G16_MSM_PART == 0

static int ilog2(uint32_t v) { int l = 0; while ((1u << l) < v) ++l; return l; }

template <class C>
int sort_scalars(const typename C::Fr* d_scalars, uint64_t n, int merged_c, Arena& arena, hipStream_t st, ScalarSort* out, int shard_n,
                 int shard_r) {
    typedef typename C::Fr Fr;
    if (n >= ((uint64_t)1 << 31)) return G16_ERR_BAD_LENGTH;
    uint32_t modw[Fr::N];
    for (int i = 0; i < Fr::N; ++i) modw[i] = Fr::Params::mod(i);
    MsmPlan plan;
    G16_TRY(make_msm_plan(n, Fr::Params::BITS, modw, Fr::N, merged_c, &plan, shard_n, shard_r));
    out->plan = plan;
    out->n = n;
    const uint32_t M = plan.buckets();
    const uint64_t nw = n * (uint64_t)plan.W;
    if (nw >= ((uint64_t)1 << 32)) return G16_ERR_BAD_LENGTH;
    uint16_t* planes = nullptr;     // per-window plan: digit planes; merged plan: entry keys, class-partitioned
    uint32_t* ent_tag = nullptr;    // merged plan: entry tags (point | window << 26), class-partitioned
    uint32_t *class_cnt = nullptr, *class_off = nullptr;
    uint32_t *counts = nullptr, *cursor = nullptr, *nparts = nullptr, *block_sums = nullptr;
    // merged plan: sort classes of Bs = 2^13 buckets (or the whole bucket set when it is smaller)
    uint32_t sort_class_log = SORT_CLASS_LOG;
    if (const char* e = getenv("G16_SORT_CLASS_LOG")) {   // A/B: 15 = histograms of 128 KB (no room beside the bucket passes)
        const int v = atoi(e);
        if (v >= 8 && v <= 15) sort_class_log = (uint32_t)v;
    }
    const uint32_t slog = plan.merged ? std::min<uint32_t>(sort_class_log, (uint32_t)ilog2(plan.B)) : 0u;
    const uint32_t Bs = plan.merged ? 1u << slog : plan.B;
    const uint32_t nb = (uint32_t)((n + CLASS_THREADS - 1) / CLASS_THREADS), Q = plan.merged ? M >> slog : (uint32_t)plan.groups;
    G16_TRY(arena.alloc_n(nw ? nw : 1, &planes));
    if (plan.merged) {
        if (Q > MAX_CLASSES) return G16_ERR_INTERNAL;
        G16_TRY(arena.alloc_n(nw ? nw : 1, &ent_tag));
        G16_TRY(arena.alloc_n((size_t)Q * nb + 1, &class_cnt));
        G16_TRY(arena.alloc_n((size_t)Q * nb + 1, &class_off));
    }
    G16_TRY(arena.alloc_n((size_t)3 * M + 1, &counts));  // counts | scatter cursors | heavy count + list
    cursor = counts + M;
    out->heavy = counts + 2 * (size_t)M;
    G16_TRY(arena.alloc_n((size_t)M + 1, &out->offsets));
    G16_TRY(arena.alloc_n((size_t)M + 1, &out->task_off));
    G16_TRY(arena.alloc_n((size_t)M, &nparts));
    // batched-affine plan: every bucket region is padded to a multiple of 2^R entries; the padding slots keep the hole
    // word the list is pre-filled with, and the bucket pass walks the level-R list (1 / 2^R of the slots)
    const int R = plan.affine_levels;
    const uint32_t pad = (1u << R) - 1u;
    const uint64_t max_sorted = nw + (uint64_t)M * pad;
    if (max_sorted >= ((uint64_t)1 << 32)) return G16_ERR_BAD_LENGTH;
    out->max_sorted = max_sorted;
    G16_TRY(arena.alloc_n(max_sorted ? max_sorted : 1, &out->sorted));
    if (R) G16_HIP_TRY(hipMemsetAsync(out->sorted, 0xff, max_sorted * sizeof(uint32_t), st));   // SORT_HOLE
    const uint64_t walked = R ? (max_sorted >> R) : nw;
    out->max_segments = (uint32_t)((walked + plan.Lmax - 1) / plan.Lmax);
    out->max_tasks = out->max_segments + M;   // a bucket gets one slot per segment it touches: <= segments + buckets in total
    G16_HIP_TRY(hipMemsetAsync(counts, 0, ((size_t)2 * M + 1) * sizeof(uint32_t), st));
    PlanDev pd;
    pd.c = plan.c; pd.W = plan.W; pd.B = plan.B;
    for (int k = 0; k < 10; ++k) pd.K[k] = plan.K[k];
    pd.shard_n = (uint32_t)plan.shard_n;
    pd.shard_r = (uint32_t)plan.shard_r;
    pd.shard_magic = plan.shard_n > 1 ? (uint32_t)((((uint64_t)1 << 32) + (uint64_t)plan.shard_n - 1) / (uint64_t)plan.shard_n) : 0u;
    const size_t lds = (size_t)Bs * sizeof(uint32_t);
    static PerDeviceOnce attr_once;
    std::atomic<bool>& attr_set = attr_once.flag();
    if (!attr_set) {
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bucket_count_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        128 * 1024));
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bucket_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        128 * 1024));
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bucket_count_merged_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&bucket_scatter_merged_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        attr_set = true;
    }
    const unsigned nchunks = (unsigned)((n + plan.chunk - 1) / plan.chunk);
    const uint32_t blog = plan.merged ? slog : (uint32_t)ilog2(plan.B);
    const uint32_t max_scan = std::max(M, plan.merged ? Q * nb : 0u);
    G16_TRY(arena.alloc_n((size_t)(max_scan + SCAN_TILE - 1) / SCAN_TILE, &block_sums));
    auto prefix_scan = [&](const uint32_t* vals, uint32_t* prefix, uint32_t count, uint32_t padmask = 0) -> int {
        const uint32_t nblocks = (count + SCAN_TILE - 1) / SCAN_TILE;
        hipLaunchKernelGGL(scan_block_sums_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, st, vals, count, padmask, block_sums);
        G16_LAUNCH_CHECK();
        hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(SCAN_THREADS), 0, st, block_sums, nblocks, count, prefix);
        G16_LAUNCH_CHECK();
        hipLaunchKernelGGL(scan_write_kernel, dim3(nblocks), dim3(SCAN_THREADS), 0, st, vals, count, padmask, block_sums, prefix);
        G16_LAUNCH_CHECK();
        return G16_OK;
    };
    // merged plan: ~4096 workgroups of 256 lanes over the class regions (a class region may hold anything between nothing and all
    // entries), at most 256 per class: every workgroup zeroes, stores and re-reads a histogram row whatever it counts
    unsigned gx = std::max(1u, std::min(std::min(4096u / Q, 256u), (unsigned)((nw + SORT_M_THREADS - 1) / SORT_M_THREADS)));
    if (const char* e = getenv("G16_SORT_GX")) {   // experiments: workgroups per class of the merged counting sort
        const int v = atoi(e);
        if (v >= 1 && v <= 4096) gx = (unsigned)v;
    }
    uint32_t* wg_counts = nullptr;   // merged plan: [class][workgroup][bucket] histograms, then their prefixes over the workgroups
    if (plan.merged) G16_TRY(arena.alloc_n((size_t)Q * gx * Bs, &wg_counts));
    if (n) {
        if (plan.merged) {
            hipLaunchKernelGGL((class_count_kernel<Fr>), dim3(nb), dim3(CLASS_THREADS), 0, st, d_scalars, n, pd, blog, Q, class_cnt);
            G16_LAUNCH_CHECK();
            G16_TRY(prefix_scan(class_cnt, class_off, Q * nb));
            hipLaunchKernelGGL((class_partition_kernel<Fr>), dim3(nb), dim3(CLASS_THREADS), 0, st, d_scalars, n, pd, blog, Q, class_off, planes,
                               ent_tag);
            G16_LAUNCH_CHECK();
            hipLaunchKernelGGL(bucket_count_merged_kernel, dim3(gx, Q), dim3(SORT_M_THREADS), lds, st, planes, class_off, nb, Bs, wg_counts);
            G16_LAUNCH_CHECK();
            hipLaunchKernelGGL(bucket_wg_scan_kernel, dim3((M + 255) / 256), dim3(256), 0, st, wg_counts, gx, Bs, M, counts);
        } else {
            hipLaunchKernelGGL((digits_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_scalars, n, pd, planes);
            G16_LAUNCH_CHECK();
            hipLaunchKernelGGL(bucket_count_kernel, dim3(nchunks, plan.W), dim3(SORT_THREADS), lds, st, planes, n, plan.chunk, plan.c, plan.B,
                               counts);
        }
        G16_LAUNCH_CHECK();
    }
    G16_TRY(prefix_scan(counts, out->offsets, M, pad));
    hipLaunchKernelGGL(bucket_slots_kernel, dim3((M + 255) / 256), dim3(256), 0, st, out->offsets, M, (uint32_t)R, plan.Lmax, nparts,
                       out->heavy);
    G16_LAUNCH_CHECK();
    G16_TRY(prefix_scan(nparts, out->task_off, M));
    if (n) {
        if (plan.merged)
            hipLaunchKernelGGL(bucket_scatter_merged_kernel, dim3(gx, Q), dim3(SORT_M_THREADS), lds, st, planes, ent_tag, class_off, nb, Bs,
                               out->offsets, wg_counts, out->sorted);
        else
            hipLaunchKernelGGL(bucket_scatter_kernel, dim3(nchunks, plan.W), dim3(SORT_THREADS), lds, st, planes, n, plan.chunk, plan.c, plan.B,
                               out->offsets, cursor, out->sorted);
        G16_LAUNCH_CHECK();
    }
    return G16_OK;
}

// buffers of one MSM's pass and reductions
template <class F>
static int alloc_msm_buffers(const ScalarSort& ss, Arena& arena, MsmBuffers<F>* out) {
    typedef AccRaw<typename Lazy30<F>::type::Raw> Raw;
    const MsmPlan& plan = ss.plan;
    const uint32_t cpw = plan.B / plan.chunk_buckets();
    Raw *partials = nullptr, *chunk_out = nullptr;
    G16_TRY(arena.alloc_n((size_t)ss.max_tasks ? ss.max_tasks : 1, &partials));
    G16_TRY(arena.alloc_n((size_t)cpw * plan.groups * 2, &chunk_out));   // weighted chunk sums, then plain chunk sums
    G16_TRY(arena.alloc_n((size_t)plan.outputs(), &out->window_sums));
    out->partials = partials;
    out->chunk_out = chunk_out;
    return G16_OK;
}

template <class F>
int msm_bucket_pass(const Affine<F>* d_bases, int64_t shift, uint64_t base_count, const ScalarSort& ss, Arena& arena, hipStream_t st,
                    MsmBuffers<F>* out, EventTimer* bucket_timer) {
    typedef typename Lazy30<F>::acc_type F30;
    typedef AccRaw<typename Lazy30<F>::type::Raw> Raw;
    const MsmPlan& plan = ss.plan;
    const uint32_t M = plan.buckets();
    const int R = plan.affine_levels;
    if (!R || !ss.max_sorted) {
        const PassJob<F> job{d_bases, shift, base_count, &ss, out};
        return msm_bucket_pass_batch<F>(&job, 1, arena, st, bucket_timer);
    }
    G16_TRY(alloc_msm_buffers<F>(ss, arena, out));
    Raw* partials = static_cast<Raw*>(out->partials);
    // batched-affine levels: two ping-pong lists (level 1 holds max_sorted / 2 points, level 2 a quarter, level 3 reuses the
    // first, ...) and the prefix-product scratch of the widest level
    Affine<F>* lists[2] = {nullptr, nullptr};
    Word4* prefix = nullptr;
    constexpr uint32_t T = 64 / F30::LANES_PER_TASK;
    constexpr int QUADS = AffineLevel<F30, true>::QUADS;
    auto level_K = [&](int r) -> uint32_t {   // steps per task: as many as keep >= ~2 k waves in flight, 8 .. 64
        const uint64_t pairs = ss.max_sorted >> (r + 1);
        uint32_t K = 64;
        while (K > 8 && pairs / ((uint64_t)K * T) < 2048) K >>= 1;
        return K;
    };
    auto level_waves = [&](int r) -> uint64_t { return ((ss.max_sorted >> (r + 1)) + (uint64_t)level_K(r) * T - 1) / ((uint64_t)level_K(r) * T); };
    {
        G16_TRY(arena.alloc_n((size_t)(ss.max_sorted >> 1) + 1, &lists[0]));
        if (R > 1) G16_TRY(arena.alloc_n((size_t)(ss.max_sorted >> 2) + 1, &lists[1]));
        size_t recs = 0;
        for (int r = 0; r < R; ++r) recs = std::max(recs, (size_t)(level_waves(r) * level_K(r)) * QUADS * 64);
        G16_TRY(arena.alloc_n(recs ? recs : 1, &prefix));
    }
    if (bucket_timer) G16_TRY(bucket_timer->start(st));
    {
        AffineLevelArgs<F30> a;
        a.sorted = ss.sorted;
        a.total = ss.offsets + M;
        a.prefix = prefix;
        a.shift = shift;
        a.base_count = base_count;
        a.merged = plan.merged ? 1u : 0u;
        for (int r = 0; r < R; ++r) {
            a.in = r == 0 ? d_bases : lists[(r - 1) & 1];
            a.out = lists[r & 1];
            a.level = (uint32_t)r;
            a.K = level_K(r);
            const uint64_t waves = level_waves(r);
            const unsigned blocks = (unsigned)((waves * 64 + AFF_THREADS - 1) / AFF_THREADS);
            if (r == 0) hipLaunchKernelGGL((affine_level_kernel<F30, true>), dim3(blocks), dim3(AFF_THREADS), 0, st, a);
            else hipLaunchKernelGGL((affine_level_kernel<F30, false>), dim3(blocks), dim3(AFF_THREADS), 0, st, a);
            G16_LAUNCH_CHECK();
        }
    }
    if (ss.max_segments) {
        const uint64_t lanes = (uint64_t)ss.max_segments * F30::LANES_PER_TASK;
        const dim3 grid((unsigned)((lanes + ACC_THREADS - 1) / ACC_THREADS));
        PassBatch<F30> b;
        for (int i = 0; i < PASS_BATCH; ++i) {
            b.bases[i] = lists[(R - 1) & 1]; b.shift[i] = 0; b.base_count[i] = 0; b.sorted[i] = nullptr;
            b.offsets[i] = ss.offsets; b.slot_off[i] = ss.task_off; b.partials[i] = partials;
        }
        hipLaunchKernelGGL((bucket_accumulate30_kernel<F30, true>), grid, dim3(ACC_THREADS), 0, st, b, M, (uint32_t)R, plan.Lmax, 0u);
        G16_LAUNCH_CHECK();
    }
    if (bucket_timer) G16_TRY(bucket_timer->stop(st));
    return G16_OK;
}

// Workgroups of the bucket pass per compute unit.  Since round 6 (product-scanning field products: no 2 NL-column array) the G1 kernel
// needs ~140 registers and THREE waves per SIMD fit; what is co-resident is then decided by LDS: 13 312 B of parked accumulator per
// 64-lane workgroup + this pad (unused dynamic LDS).  G16_PASS_WG_PER_CU=n pads so that exactly n workgroups fit (A/B knob).
template <class F30>
static unsigned pass_lds_pad() {
    static const int want = [] { const char* e = getenv("G16_PASS_WG_PER_CU"); return e ? atoi(e) : 0; }();
    if (want <= 0) return 0;
    const unsigned own = F30::ACC_PARKED ? 4u * F30::PREFIX_LIMBS * ACC_THREADS * 4u : 16u;
    const unsigned per = (160u * 1024u / (unsigned)want) & ~255u;
    return per > own ? per - own : 0;
}

// the bucket passes of n <= PASS_BATCH MSMs with one bucket layout as ONE launch (PassBatch above)
template <class F>
int msm_bucket_pass_batch(const PassJob<F>* jobs, int n, Arena& arena, hipStream_t st, EventTimer* bucket_timer) {
    typedef typename Lazy30<F>::acc_type F30;
    typedef AccRaw<typename Lazy30<F>::type::Raw> Raw;
    if (n < 1 || n > PASS_BATCH) return G16_ERR_INTERNAL;
    const MsmPlan& plan = jobs[0].ss->plan;
    for (int i = 0; i < n; ++i) {
        const MsmPlan& q = jobs[i].ss->plan;
        if (q.B != plan.B || q.groups != plan.groups || q.Lmax != plan.Lmax || q.merged != plan.merged) return G16_ERR_INTERNAL;
        if (q.affine_levels && jobs[i].ss->max_sorted) return G16_ERR_INTERNAL;   // that plan walks level lists: msm_bucket_pass
    }
    PassBatch<F30> b;
    uint32_t max_segments = 0;
    for (int i = 0; i < n; ++i) G16_TRY(alloc_msm_buffers<F>(*jobs[i].ss, arena, jobs[i].out));
    for (int i = 0; i < PASS_BATCH; ++i) {
        const PassJob<F>& j = jobs[i < n ? i : 0];
        b.bases[i] = j.bases; b.shift[i] = j.shift; b.base_count[i] = j.base_count; b.sorted[i] = j.ss->sorted;
        b.offsets[i] = j.ss->offsets; b.slot_off[i] = j.ss->task_off; b.partials[i] = static_cast<Raw*>(j.out->partials);
        if (i < n) max_segments = std::max(max_segments, j.ss->max_segments);
    }
    if (bucket_timer) G16_TRY(bucket_timer->start(st));
    if (max_segments) {
        const uint64_t lanes = (uint64_t)max_segments * F30::LANES_PER_TASK;
        const dim3 grid((unsigned)((lanes + ACC_THREADS - 1) / ACC_THREADS), (unsigned)n);
        hipLaunchKernelGGL((bucket_accumulate30_kernel<F30, false>), grid, dim3(ACC_THREADS), pass_lds_pad<F30>(), st, b, plan.buckets(), 0u,
                           plan.Lmax, plan.merged ? 1u : 0u);
        G16_LAUNCH_CHECK();
    }
    if (bucket_timer) G16_TRY(bucket_timer->stop(st));
    return G16_OK;
}

template <class F>
static int reduce_setup(size_t* lds_heavy, size_t* lds_win) {
    typedef typename Lazy30<F>::type F30;
    typedef AccRaw<typename F30::Raw> Raw;
    static PerDeviceOnce attr_once;
    std::atomic<bool>& attr_set = attr_once.flag();
    *lds_heavy = sizeof(Raw) * HEAVY_THREADS / F30::LANES_PER_TASK;
    *lds_win = sizeof(Raw) * WIN_THREADS / F30::LANES_PER_TASK;
    if (!attr_set) {
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&heavy_reduce_kernel<F30>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)*lds_heavy));
        G16_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&window_reduce_kernel<F30>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)*lds_win));
        attr_set = true;
    }
    return G16_OK;
}

template <class F>
static ReduceBatch<AccRaw<typename Lazy30<F>::type::Raw>, XYZZ<F>> make_reduce_batch(const MsmBuffers<F>* const* bufs, const ScalarSort* const* sorts, int n) {
    typedef AccRaw<typename Lazy30<F>::type::Raw> Raw;
    ReduceBatch<Raw, XYZZ<F>> batch;
    for (int i = 0; i < REDUCE_BATCH; ++i) {
        const int k = i < n ? i : 0;
        batch.partials[i] = static_cast<Raw*>(bufs[k]->partials);
        batch.slot_off[i] = sorts[k]->task_off;
        batch.heavy[i] = sorts[k]->heavy;
        batch.chunk_out[i] = static_cast<Raw*>(bufs[k]->chunk_out);
        batch.window_sums[i] = bufs[k]->window_sums;
    }
    return batch;
}

// the cooperative combine of one MSM's heavy buckets alone (a few hundred workgroups at most: cheap enough to run underneath
// the next bucket pass, which takes it off the batched reduction's chain)
// (n MSMs with one bucket layout: one launch per kernel)
template <class F>
int msm_heavy_reduce_batch(const MsmBuffers<F>* const* bufs, const ScalarSort* const* sorts, int n, hipStream_t st) {
    typedef typename Lazy30<F>::type F30;
    if (n < 1 || n > REDUCE_BATCH) return G16_ERR_INTERNAL;
    const MsmPlan& plan = sorts[0]->plan;
    for (int i = 1; i < n; ++i)
        if (sorts[i]->plan.B != plan.B || sorts[i]->plan.groups != plan.groups) return G16_ERR_INTERNAL;
    size_t lds_heavy, lds_win;
    G16_TRY(reduce_setup<F>(&lds_heavy, &lds_win));
    const auto batch = make_reduce_batch<F>(bufs, sorts, n);
    hipLaunchKernelGGL((heavy_reduce_kernel<F30>), dim3(HEAVY_BLOCKS, n), dim3(HEAVY_THREADS), lds_heavy, st, batch);
    G16_LAUNCH_CHECK();
    const uint32_t M = plan.buckets();
    hipLaunchKernelGGL((bucket_combine_kernel<F30>), dim3((M * F30::LANES_PER_TASK + RED_THREADS - 1) / RED_THREADS, n), dim3(RED_THREADS), 0, st, batch, M);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class F>
int msm_heavy_reduce(const MsmBuffers<F>& buf, const ScalarSort& ss, hipStream_t st) {
    const MsmBuffers<F>* b = &buf;
    const ScalarSort* s = &ss;
    return msm_heavy_reduce_batch<F>(&b, &s, 1, st);
}

template <class F>
int msm_reduce_batch(const MsmBuffers<F>* const* bufs, const ScalarSort* const* sorts, int n, hipStream_t st, bool heavy_done) {
    typedef typename Lazy30<F>::type F30;
    if (n < 1 || n > REDUCE_BATCH) return G16_ERR_INTERNAL;
    const MsmPlan& plan = sorts[0]->plan;
    for (int i = 1; i < n; ++i)
        if (sorts[i]->plan.B != plan.B || sorts[i]->plan.groups != plan.groups) return G16_ERR_INTERNAL;   // one launch = one plan
    const uint32_t G = plan.chunk_buckets();
    const uint32_t cpw = plan.B / G;
    size_t lds_heavy, lds_win;
    G16_TRY(reduce_setup<F>(&lds_heavy, &lds_win));
    const auto batch = make_reduce_batch<F>(bufs, sorts, n);
    if (!heavy_done) G16_TRY((msm_heavy_reduce_batch<F>(bufs, sorts, n, st)));
    hipLaunchKernelGGL((bucket_reduce_kernel<F30>), dim3((cpw * plan.groups * F30::LANES_PER_TASK + RED_THREADS - 1) / RED_THREADS, n), dim3(RED_THREADS), 0,
                       st, batch, plan.B, plan.groups, G);
    G16_LAUNCH_CHECK();
    hipLaunchKernelGGL((window_reduce_kernel<F30>), dim3(plan.groups, plan.planes(), n), dim3(WIN_THREADS), lds_win, st, batch, cpw);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class F>
int msm_reduce(const MsmBuffers<F>& buf, const ScalarSort& ss, hipStream_t st) {
    const MsmBuffers<F>* b = &buf;
    const ScalarSort* s = &ss;
    return msm_reduce_batch<F>(&b, &s, 1, st, false);
}

template <class F>
int convert_bases(Affine<F>* d_bases, uint64_t n, hipStream_t st) {
    if (n == 0) return G16_OK;
    hipLaunchKernelGGL((convert_bases30_kernel<F>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_bases, n);
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class F>
XYZZ<F> fold_windows(const XYZZ<F>* ws, const MsmPlan& plan) {
    const int NP = plan.planes(), bits = plan.chunk_bits();
    // per group: T = sum_b (b+1) S_b = P_0 + G * sum_k 2^k P_(2+k)
    auto group_T = [&](int g) -> XYZZ<F> {
        const XYZZ<F>* P = ws + (size_t)g * NP;
        XYZZ<F> hi = XYZZ<F>::identity();
        for (int k = bits - 1; k >= 0; --k) { hi = hi.dbl(); hi.add(P[2 + k]); }
        for (uint32_t b = plan.chunk_buckets(); b > 1; b >>= 1) hi = hi.dbl();
        hi.add(P[0]);
        return hi;
    };
    XYZZ<F> total = XYZZ<F>::identity();
    if (plan.merged) {
        // bucket value of (group q, bucket b) is q*B + b + 1:  sum = sum_q T_q + B * sum_q q S_q
        const int Q = plan.groups;
        if (Q > 1) {
            XYZZ<F> run = XYZZ<F>::identity();
            for (int q = Q - 1; q >= 1; --q) { run.add(ws[(size_t)q * NP + 1]); total.add(run); }
            for (uint32_t b = plan.B; b > 1; b >>= 1) total = total.dbl();
        }
        for (int q = 0; q < Q; ++q) total.add(group_T(q));
        if (plan.shard_n > 1) {
            // bucket-space shard: `total` is sum_k (k+1) S_k over the LOCAL indices; the rank's buckets are b = shard_n k + shard_r, so its
            // share of the MSM is  shard_n * total - (shard_n - 1 - shard_r) * sum_k S_k   (two multiplications by numbers below 64)
            XYZZ<F> plain = XYZZ<F>::identity();
            for (int q = 0; q < Q; ++q) plain.add(ws[(size_t)q * NP + 1]);
            const uint32_t kn = (uint32_t)plan.shard_n, km = (uint32_t)(plan.shard_n - 1 - plan.shard_r);
            total = total.mul_bits(&kn, 7);
            if (km) total.add(plain.mul_bits(&km, 7).neg());
        }
        return total;
    }
    for (int w = plan.W - 1; w >= 0; --w) {
        for (int k = 0; k < plan.c; ++k) total = total.dbl();
        total.add(group_T(w));
    }
    return total;
}

// table[j * n + i] = 2^(c j) P_i (window_tables.hpp): one lane (G1) or one lane pair (G2) per point in the 30-bit lazy arithmetic,
// Jacobian doublings, ONE field inversion per point; the rows' (X, Y, Z, prefix product) wait for the backward sweep in `park`,
// a limb-planar HBM buffer of 4 (W - 1) NL words per lane.  Runs once per key, in chunks of TABLE_CHUNK points.
static constexpr int TABLE_MAX_W = 32;
static constexpr int TABLE_THREADS = 128;
static constexpr uint64_t TABLE_CHUNK = (uint64_t)1 << 18;   // points per launch: 2^18 (G1) / 2^19 (G2) lanes, park <= 1.3 GB
template <class F30>
__global__ __launch_bounds__(TABLE_THREADS, 2) void build_window_tables_kernel(const Affine<typename F30::Std>* __restrict__ src, uint64_t first,
                                                                              uint64_t count, uint64_t n, int c, int W,
                                                                              Affine<typename F30::Std>* __restrict__ table, uint32_t* __restrict__ park) {
    typedef TableDeviceIO<F30> IO;
    typedef typename IO::W32 W32;
    const uint64_t lane = (uint64_t)blockIdx.x * TABLE_THREADS + threadIdx.x;
    const uint64_t t = lane / F30::LANES_PER_TASK;   // lanes of one task are adjacent
    if (t >= count) return;
    const uint64_t i = first + t;
    IO io{reinterpret_cast<const W32*>(src + i), reinterpret_cast<W32*>(table + i), n * 2 * IO::TL::PARTS, park + lane, count * F30::LANES_PER_TASK};
    window_table_task<F30>(io, c, W);
}

template <class F>
size_t window_table_park_bytes(uint64_t n, int W) {
    typedef typename Lazy30<F>::acc_type F30;
    const uint64_t chunk = std::min(n ? n : 1, TABLE_CHUNK);
    return (size_t)4 * (W > 1 ? W - 1 : 1) * TableLane<F30>::B::NL * chunk * F30::LANES_PER_TASK * sizeof(uint32_t);
}

// park: window_table_park_bytes<F>(n, W) of device scratch the caller keeps alive until the stream has run the launches (then
// nothing here waits for the GPU: g16_pk_load queues the builds of all five queries behind one another and allocates the next
// table while the previous one is being built); nullptr: allocated here, and the call returns with the table finished.
template <class F>
int build_window_tables(const Affine<F>* d_src, uint64_t n, int c, int W, Affine<F>* d_table, hipStream_t st, void* park_buf) {
    typedef typename Lazy30<F>::acc_type F30;
    static_assert(sizeof(Affine<F>) == 2 * TableLane<F30>::PARTS * sizeof(typename TableLane<F30>::B::Std), "affine point = x parts | y parts");
    if (n == 0) return G16_OK;
    if (W > TABLE_MAX_W) return G16_ERR_INTERNAL;
    const uint64_t chunk = std::min(n, TABLE_CHUNK);
    uint32_t* park = static_cast<uint32_t*>(park_buf);
    if (!park && hipMalloc((void**)&park, window_table_park_bytes<F>(n, W)) != hipSuccess) { (void)hipGetLastError(); return G16_ERR_OOM; }
    int rc = G16_OK;
    for (uint64_t first = 0; first < n && rc == G16_OK; first += chunk) {
        const uint64_t count = std::min(chunk, n - first);
        const uint64_t lanes = count * F30::LANES_PER_TASK;
        hipLaunchKernelGGL((build_window_tables_kernel<F30>), dim3((unsigned)((lanes + TABLE_THREADS - 1) / TABLE_THREADS)), dim3(TABLE_THREADS), 0, st,
                           d_src, first, count, n, c, W, d_table, park);
        if (hipGetLastError() != hipSuccess) rc = G16_ERR_HIP;
    }
    if (!park_buf) {
        if (hipStreamSynchronize(st) != hipSuccess && rc == G16_OK) rc = G16_ERR_HIP;
        (void)hipFree(park);
    }
    return rc;
}

#define G16_INSTANTIATE_MSM(C)                                                                                               \
    template int convert_bases<typename C::Fq>(Affine<typename C::Fq>*, uint64_t, hipStream_t);                             \
    template int convert_bases<typename C::Fq2>(Affine<typename C::Fq2>*, uint64_t, hipStream_t);                           \
    template int sort_scalars<C>(const typename C::Fr*, uint64_t, int, Arena&, hipStream_t, ScalarSort*, int, int);                   \
    template int build_window_tables<typename C::Fq>(const Affine<typename C::Fq>*, uint64_t, int, int, Affine<typename C::Fq>*, hipStream_t, void*);    \
    template int build_window_tables<typename C::Fq2>(const Affine<typename C::Fq2>*, uint64_t, int, int, Affine<typename C::Fq2>*, hipStream_t, void*); \
    template size_t window_table_park_bytes<typename C::Fq>(uint64_t, int);                                                 \
    template size_t window_table_park_bytes<typename C::Fq2>(uint64_t, int);                                                \
    template int msm_bucket_pass<typename C::Fq>(const Affine<typename C::Fq>*, int64_t, uint64_t, const ScalarSort&, Arena&, \
                                                 hipStream_t, MsmBuffers<typename C::Fq>*, EventTimer*);                      \
    template int msm_bucket_pass<typename C::Fq2>(const Affine<typename C::Fq2>*, int64_t, uint64_t, const ScalarSort&,     \
                                                  Arena&, hipStream_t, MsmBuffers<typename C::Fq2>*, EventTimer*);            \
    template int msm_bucket_pass_batch<typename C::Fq>(const PassJob<typename C::Fq>*, int, Arena&, hipStream_t, EventTimer*);   \
    template int msm_bucket_pass_batch<typename C::Fq2>(const PassJob<typename C::Fq2>*, int, Arena&, hipStream_t, EventTimer*); \
    template int msm_reduce<typename C::Fq>(const MsmBuffers<typename C::Fq>&, const ScalarSort&, hipStream_t);             \
    template int msm_reduce_batch<typename C::Fq>(const MsmBuffers<typename C::Fq>* const*, const ScalarSort* const*, int, hipStream_t, bool); \
    template int msm_heavy_reduce<typename C::Fq>(const MsmBuffers<typename C::Fq>&, const ScalarSort&, hipStream_t);     \
    template int msm_heavy_reduce_batch<typename C::Fq>(const MsmBuffers<typename C::Fq>* const*, const ScalarSort* const*, int, hipStream_t); \
    template int msm_reduce<typename C::Fq2>(const MsmBuffers<typename C::Fq2>&, const ScalarSort&, hipStream_t);           \
    template XYZZ<typename C::Fq> fold_windows<typename C::Fq>(const XYZZ<typename C::Fq>*, const MsmPlan&);               \
    template XYZZ<typename C::Fq2> fold_windows<typename C::Fq2>(const XYZZ<typename C::Fq2>*, const MsmPlan&);

#if G16_MSM_PART == 0
G16_INSTANTIATE_MSM(Bls12_381)
#else
G16_INSTANTIATE_MSM(Bn254)
#endif

}  // namespace g16
