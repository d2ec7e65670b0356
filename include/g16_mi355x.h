/* g16_mi355x.h -- C ABI of the MI355X-native Groth16 prover hot path.
 *
 * Drop-in boundary for ark-groth16 0.5.0 (paths relative to /root/reference):
 *   g16_prove            <->  Groth16::create_proof_with_reduction_and_matrices   src/prover.rs:26-51
 *                             (= witness_map_from_matrices  src/r1cs_to_qap.rs:172-235
 *                              + create_proof_with_assignment src/prover.rs:54-132)
 *   g16_witness_map      <->  LibsnarkReduction::witness_map_from_matrices        src/r1cs_to_qap.rs:172-235
 *   g16_msm_g1 / _g2     <->  VariableBaseMSM::msm_bigint call sites              src/prover.rs:66,74,262
 *   g16_ntt              <->  EvaluationDomain::{fft,ifft}_in_place (+coset)      src/r1cs_to_qap.rs:201-232
 *   g16_pk_load          <->  &ProvingKey<E>                                      src/data_structures.rs:125-143
 *   g16_circuit_load     <->  &ConstraintMatrices<F>, num_inputs, num_constraints src/prover.rs:30-32
 *   g16_prove_partial / g16_prove_finalize: the same proof with the MSM base set sharded
 *                             over several GPUs (one process per GPU; the host exchanges the
 *                             fixed-size g16_partial records, e.g. one RCCL all-gather).
 * The reference has no FFI of its own (#![forbid(unsafe_code)], src/lib.rs:13); INTEGRATION.md
 * shows the Rust binding a maintainer would add.
 *
 * Data conventions (zero conversion on the Rust side):
 *   field element  : arkworks in-memory form -- little-endian u64 limbs of a*R mod p (Montgomery),
 *                    Fr: 4 limbs (both curves); Fq: 6 limbs (BLS12-381) / 4 limbs (BN254)
 *   G1 affine      : x | y              (2*FQ limbs)
 *   G2 affine      : x.c0 x.c1 y.c0 y.c1 (4*FQ limbs)
 *   identity       : all limbs zero (arkworks Affine::identity() has x = y = 0, infinity = true;
 *                    (0,0) is not on either curve, so the flag is redundant)
 *   CSR matrix     : row_ptr u64[num_constraints+1], col u32[nnz], val Fr[nnz]; built from
 *                    ConstraintMatrices' Vec<Vec<(F, usize)>> rows
 * All functions return 0 on success or a g16_status; they never throw or unwind.
 * A g16_ctx is bound to one HIP device (g16_ctx_create) or to several (g16_ctx_create_multi) and is thread-compatible
 * (one call in flight per ctx).  A g16_pk / g16_circuit may be used by every single-device context on the GPU it was loaded on
 * (its device data is read-only during a proof; everything a proof writes belongs to the calling context), from different
 * threads at once: two contexts on one GPU proving side by side over one key is the THROUGHPUT mode -- the head (witness map) and
 * tail (reductions, host glue) of one proof run under the bucket passes of the other (bench.py reports it as `pipelined`).
 * Free a key / circuit only when no call on any context is using it.
 */
#ifndef G16_MI355X_H
#define G16_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    G16_OK = 0,
    G16_ERR_DEGREE_TOO_LARGE = 1, /* SynthesisError::PolynomialDegreeTooLarge (r1cs_to_qap.rs:178-179) */
    G16_ERR_BAD_LENGTH = 2,       /* the reference would panic on a slice bound (prover.rs:44-45)      */
    G16_ERR_BAD_ARG = 3,
    G16_ERR_HIP = 4,
    G16_ERR_OOM = 5,
    G16_ERR_NO_DEVICE = 6,
    G16_ERR_INTERNAL = 7,
    G16_ERR_UNEXPECTED_IDENTITY = 8, /* SynthesisError::UnexpectedIdentity: gamma or delta is zero (generator.rs:110-111) */
    G16_ERR_INVALID_DATA = 9,        /* SerializationError::InvalidData: bytes that are not a point of the group        */
    G16_ERR_NO_PEER_ACCESS = 10      /* g16_ctx_create_multi with G16_MULTI_REQUIRE_PEER=1: a device pair without peer access */
} g16_status;

typedef enum { G16_BLS12_381 = 0, G16_BN254 = 1 } g16_curve;

typedef struct g16_ctx g16_ctx;
typedef struct g16_pk g16_pk;
typedef struct g16_circuit g16_circuit;

/* one query vector (or a contiguous shard of it) */
typedef struct {
    const uint64_t* points; /* affine points, host memory (or device memory if G16_PK_DEVICE_PTRS) */
    uint64_t count;         /* points in this shard                                               */
    uint64_t start;         /* index of points[0] within the logical MSM base array               */
} g16_query;

#define G16_PK_DEVICE_PTRS 1u /* query `points` are device pointers on the ctx's GPU */

/* g16_pk_load never modifies or retains the caller's arrays.  What it keeps on the GPU per query is either the
 * bases (in the bucket kernel's radix) or -- the default -- their WINDOW TABLE: W rows 2^(c j) * query, c = 20 and
 * W = 13 for keys of 2^20 points and more, i.e. 13x the memory of the key (31 GB for 2^22 BLS12-381 constraints),
 * which cuts the prover's dominant kernel by a fifth.  Environment: G16_MSM_PRECOMP=0 keeps plain
 * bases; G16_PK_TABLE_BUDGET_MB caps one query's table; a failed allocation falls back to plain bases. */

/* ProvingKey<E> as the prover reads it (src/data_structures.rs:125-143; vk fields prover.rs:92,105,113).
 * MSM base arrays, in the index space the prover uses:
 *   a, b_g1, b_g2 : query[1..]   (index i <-> full_assignment[1+i]),  logical length m
 *   l             : l_query      (index j <-> full_assignment[num_inputs+j]), logical length w
 *   h             : h_query      (index k <-> h[k]), logical length n-1
 * query[0] of a / b_g1 / b_g2 is passed separately (calculate_coeff, prover.rs:261). */
typedef struct {
    const uint64_t* alpha_g1; /* vk.alpha_g1 */
    const uint64_t* beta_g1;
    const uint64_t* delta_g1;
    const uint64_t* beta_g2;  /* vk.beta_g2  */
    const uint64_t* delta_g2; /* vk.delta_g2 */
    const uint64_t* a_query0;
    const uint64_t* b_g1_query0;
    const uint64_t* b_g2_query0;
    g16_query a, b_g1, b_g2, h, l;
    uint32_t flags;
} g16_pk_view;

typedef struct {
    const uint64_t* row_ptr;
    const uint32_t* col;
    const uint64_t* val;
} g16_csr_view;

/* Proof<E> (src/data_structures.rs:8-16), affine Montgomery limbs; only the first
 * 2*FQ / 4*FQ / 2*FQ limbs of a / b / c are meaningful. */
typedef struct {
    uint64_t a[12];
    uint64_t b[24];
    uint64_t c[12];
} g16_proof;

/* Partial MSM sums of one shard, extended-Jacobian (X, Y, ZZ, ZZZ) Montgomery limbs.
 * Fixed size so that it can be all-gathered as raw bytes. */
typedef struct {
    uint64_t h[24], l[24], a[24], b_g1[24]; /* 4*FQ limbs used */
    uint64_t b_g2[48];                      /* 8*FQ limbs used */
} g16_partial;

/* phase timings of the last g16_prove on this ctx, milliseconds (GPU events + host clock);
 * names follow the reference's timers (prover.rs:36,62,89,99,111,119) */
typedef struct {
    double witness_map_ms; /* for a HOST assignment (assignment_on_device == 0) the upload's pieces land UNDER the map's first kernels, whose
                              row blocks wait for them: the figure then includes the ~2.4 ms (2^22) the 128 MiB take to arrive; with
                              the assignment resident, or G16_UPLOAD_CHUNKED=0 (one copy ahead of the timer), it is the map alone */
    double msm_h_ms, msm_l_ms, msm_a_ms, msm_b_g1_ms, msm_b_g2_ms;
    double scalar_prep_ms; /* into_bigint + digit extraction + bucket sort (shared by the MSMs) */
    double finish_ms;      /* host glue: scalar muls, final adds, into_affine */
    double total_ms;
    double bucket_pass_ms; /* sum over the 5 MSMs of the bucket-accumulation kernel */
    double bucket_ms[5];   /* that kernel per MSM: h, l, a, b_g1 (G1 kernel), b_g2 (G2 kernel); HIP events on the ctx stream (see g1_pass_launches) */
    double window_bits;    /* Pippenger window size c of the witness MSMs in the last call ... */
    double windows;        /* ... and their window count W: a bucket pass folds (bases in the shard) * W points */
    double ntt_ms;         /* the seven transforms of r1cs_to_qap.rs:201-232 alone (inside witness_map_ms, which also covers
                              the three sparse mat-vecs and the pointwise pass) */
    double g1_pass_launches; /* launches of the G1 bucket kernel in the last call: MSMs that are ready together share ONE launch
                              (l, a, b_g1; h too in a sharded proof), bucket_ms[] then holds each one's share of it by points */
} g16_timings;

int g16_ctx_create(int curve, int device_id, g16_ctx** out);
/* One context over n_dev GPUs of the node, so that the reference's single call -- Groth16::prove, src/lib.rs:76-82
 * -> create_proof_with_reduction, src/prover.rs:173-217 -- stays a single g16_prove: g16_pk_load (given the WHOLE key: every
 * query.start == 0, host pointers) cuts the five MSM base arrays into n_dev contiguous shards, one per device;
 * g16_circuit_load replicates the matrices; g16_prove (host full_assignment) runs one host thread per device and folds the
 * n_dev partial records on the host.  When n_dev is a power of two <= 16 with n_dev^2 | domain_size the witness map is
 * distributed as well (the g16_dwm_* stages below on every device, the all-to-all as peer copies between the devices' buffers
 * over xGMI) and g16_pk_load gathers every h_query shard in the block order that leaves h in (the key must then be the
 * circuit's: h_query holds domain_size - 1 bases, generator.rs:168; anything else is G16_ERR_BAD_LENGTH at g16_prove).  g16_prove_partial is the per-device form and is refused on such a context;
 * g16_prove_finalize and the unit-level entry points run on the first device.  device_ids may repeat.  n_dev == 1 is
 * g16_ctx_create.
 * How g16_pk_load cuts the key over the devices (round 5; G16_MULTI_SHARD_MODE=auto|base|bucket, default auto): by BASE RANGES as above, or in
 * BUCKET SPACE (g16_pk_load_bucket_shard on every device: the whole key's window tables per device, device i owns the buckets
 * b mod n_dev == i, and with the distributed witness map every device pulls ALL blocks of h -- the all-gather as peer copies); auto picks
 * bucket space while the whole key's tables stay below 80 GiB and fit in every device's free memory, and falls back to base ranges if
 * such a load then runs out of memory (a FORCED bucket-space load does not fall back: G16_ERR_OOM).  g16_pk_get_info says which.
 * EXPERIMENTAL for n_dev > 1 over distinct devices: every test so far ran with one physical GPU listed several times (the build
 * pool has one-GPU boxes), so peer access, hipMemcpyPeerAsync between devices and the cross-device event waits of g16_prove have
 * not met real hardware; tests/test_gpu_parity.py::test_multi_device_context_distinct_gpus runs wherever >= 2 GPUs are visible.
 * The process-per-GPU form (g16_prove_partial + one all-gather, bench.py --gpus N) is the supported multi-GPU path. */
int g16_ctx_create_multi(int curve, const int* device_ids, int n_dev, g16_ctx** out);
int g16_ctx_num_devices(const g16_ctx* ctx);
/* 1: device i of the context reaches device j's memory directly (hipDeviceCanAccessPeer said yes and the access was enabled, or
 * i and j are the same physical device); 0: copies between the two are staged through host memory by the runtime (correct, slow --
 * G16_MULTI_REQUIRE_PEER=1 makes g16_ctx_create_multi fail with G16_ERR_NO_PEER_ACCESS instead); -1: bad index. */
int g16_ctx_peer_access(const g16_ctx* ctx, int i, int j);
void g16_ctx_destroy(g16_ctx* ctx);
/* HIP stream the ctx launches on (hipStream_t); lets callers bracket work with their own events */
void* g16_ctx_stream(g16_ctx* ctx);

int g16_pk_load(g16_ctx* ctx, const g16_pk_view* view, g16_pk** out);
void g16_pk_free(g16_pk* pk);

/* BUCKET-SPACE SHARD of the five MSMs (src/prover.rs:66,74,262) over `world` ranks, the second way to cut them (the first: base
 * ranges, g16_query.start / count above).  Every rank loads the WHOLE key (`view` as for one GPU: every query.start == 0) as
 * window tables -- 31 GB at 2^22 BLS12-381 constraints, 126 GB at 2^24: what 288 GB of HBM per GPU are for -- and owns the buckets
 * b with  b mod world == rank  of the merged-window bucket set (interleaved, so that the short top window's entries spread
 * evenly).  A rank's sort keeps only the (point, window) entries of its residue class, its bucket passes fold ~1/world of them
 * at the full window size (c = 20, W = 13, ~100 entries per bucket), and -- the point of the mode -- its bucket REDUCTIONS shrink
 * world-fold too, which a base-range shard's do not (every rank keeps all 2^(c-1) buckets of all five MSMs there).  With local
 * index k = b / world:  sum_{b owned} (b+1) S_b = world * sum_k (k+1) S_k + (rank + 1 - world) * sum_k S_k.
 * g16_prove_partial / g16_prove_partial_h over such a key yield the rank's partial sums; the exchange (one all-gather of the
 * g16_partial records) and g16_prove_finalize are unchanged.  The scalars are needed WHOLE on every rank: the assignment as for
 * one GPU, and h either from the replicated witness map (g16_prove_partial) or -- distributed map -- all-gathered (n Fr; the
 * ranks' blocks back to back, h_query loaded in that same order) and passed to g16_prove_partial_h.
 * No silent fall-back: G16_ERR_OOM if the whole key's tables do not fit, G16_ERR_BAD_ARG for a key that cannot have tables
 * (G16_MSM_PRECOMP=0) or a view that is itself a base-range shard.  world == 1 is g16_pk_load. */
int g16_pk_load_bucket_shard(g16_ctx* ctx, const g16_pk_view* view, int rank, int world, g16_pk** out);
/* The resident tables do not depend on the rank: re-label a whole key held as window tables (g16_pk_load, or a bucket-space shard)
 * as rank `rank` of `world` (world == 1: the whole bucket set again).  Not while a call is using the key.  Lets one GPU walk through
 * every rank's share (parity tests at 2^24, bench.py --sim-shards); a base-range shard or a plain-bases key is G16_ERR_BAD_ARG. */
int g16_pk_rebind_bucket_shard(g16_pk* pk, int rank, int world);

/* How a loaded key is held (no reference counterpart).  table_fallback says WHY a key is held as plain bases -- a slower prover:
 * per-window buckets, c <= 16, more windows -- instead of window tables: 0 window tables (the default);
 * 1 plain bases by request (G16_MSM_PRECOMP=0); 2 a query too long for merged entries; 3 the tables did not fit next to what
 * already lives on the GPU (allocation failed, or G16_PK_TABLE_BUDGET_MB). */
typedef struct {
    int window_bits_z;      /* window size c of the witness MSMs' tables (0: plain bases)  */
    int window_bits_h;      /* ... of h_query's                                          */
    int table_fallback;     /* reason code above                                         */
    int bucket_shard_rank;  /* g16_pk_load_bucket_shard: rank, world (0, 1 otherwise)    */
    int bucket_shard_world;
    int n_devices;          /* 1, or the devices of a multi-device key (fields above: device 0's shard) */
    uint64_t device_bytes;  /* HBM held by the five query arrays (per device)            */
} g16_pk_info;
int g16_pk_get_info(const g16_pk* pk, g16_pk_info* out);

/* num_variables = num_instance_variables + num_witness_variables (len of full_assignment) */
int g16_circuit_load(g16_ctx* ctx, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints,
                     uint64_t num_variables, g16_circuit** out);
void g16_circuit_free(g16_circuit* c);
uint64_t g16_circuit_domain_size(const g16_circuit* c);

/* full_assignment: n_assign Fr (host memory unless assignment_on_device != 0); r, s: one Fr each */
int g16_prove(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment,
              uint64_t n_assign, int assignment_on_device, const uint64_t r[4], const uint64_t s[4], g16_proof* out);

/* sharded proof: every rank runs _partial on its pk shard, the host gathers the records, any rank
 * (all ranks, typically) runs _finalize over all of them */
int g16_prove_partial(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment,
                      uint64_t n_assign, int assignment_on_device, int skip_b_g1, g16_partial* out);
int g16_prove_finalize(g16_ctx* ctx, const g16_pk* pk, const g16_partial* parts, int n_parts, const uint64_t r[4],
                       const uint64_t s[4], g16_proof* out);

/* Optional, before (or while) the ranks' g16_prove_partial calls run: start on a host thread the half of the glue of
 * src/prover.rs:76-131 that depends only on r, s and the key's fixed points -- r delta, s delta, r s delta and, by linearity of
 * :94 and :114, s (r delta_g1 + a_query[0] + alpha_g1) and r (s delta_g1 + b_g1_query[0] + beta_g1).  The next g16_prove_finalize on
 * this context over the same (pk, r, s) then only multiplies the two MSM sums by s and r, adds, and converts to affine; any other
 * (pk, r, s) ignores the prepared half.  g16_prove does this by itself. */
int g16_prove_finalize_prepare(g16_ctx* ctx, const g16_pk* pk, const uint64_t r[4], const uint64_t s[4]);

/* ---- distributed witness map (the Amdahl term of the sharded proof) ----
 * h = witness_map_from_matrices (src/r1cs_to_qap.rs:172-235) over `world` ranks (a power of two <= 16 with world^2 | domain_size),
 * every n-point transform as a local (n / world)-point transform, a twiddle, ONE all-to-all and a local world-point transform.
 * Rank r ends with the h coefficients of its BLOCK indices  (r * blk + j) + M * k1,  j < blk = M / world, k1 < world, M = n / world,
 * in the order [k1][j] -- which is how that rank's h_query shard has to be gathered (g16_pk_view.h = those bases, start 0).
 * The caller owns the buffers (device memory, M = g16_dwm_local_size() Fr each) and the exchange: after stages 0, 1 (three
 * arrays: a, b, c) and 2 (one array), chunk p (M / world elements) of each work array goes to rank p, which stores the chunk
 * from rank q at position q of the matching recv array -- an all-to-all (RCCL over xGMI: torch.distributed.all_to_all_single).
 *   stage 0: full_assignment -> work[0..2]     stage 1: recv[0..2] -> work[0..2]     stage 2: recv[0..2] -> work[0]
 *   stage 3: recv[0] -> h_local.               Each call returns with its device work finished. */
typedef struct g16_dwm g16_dwm;
int g16_dwm_create(g16_ctx* ctx, const g16_circuit* circuit, int rank, int world, g16_dwm** out);
void g16_dwm_free(g16_dwm* d);
uint64_t g16_dwm_local_size(const g16_dwm* d);
int g16_dwm_stage(g16_ctx* ctx, g16_dwm* d, int stage, const uint64_t* full_assignment, uint64_t n_assign, int assignment_on_device,
                  uint64_t* const work[3], uint64_t* const recv[3], uint64_t* h_local);
/* The same stages WITHOUT host synchronisation: the stage is enqueued on the context's witness-map stream (g16_ctx_wm_stream) and the
 * call returns; the caller enqueues its exchange on that SAME stream (e.g. torch.cuda.ExternalStream), so stages and exchanges are
 * ordered by the stream alone.  full_assignment must be device memory.  A following g16_prove_partial_h orders its h sort and its
 * bucket passes after everything enqueued on that stream; its witness digit/sort pass (shared by the MSMs of prover.rs:74,92,105,
 * 113) does not wait and runs beside the map's stages and exchanges. */
void* g16_ctx_wm_stream(g16_ctx* ctx);
int g16_dwm_stage_async(g16_ctx* ctx, g16_dwm* d, int stage, const uint64_t* full_assignment_dev, uint64_t n_assign,
                        uint64_t* const work[3], uint64_t* const recv[3], uint64_t* h_local);
/* Optional, before the stages above: enqueue NOW the witness digit/sort pass (shared by the MSMs of prover.rs:74,92,105,113) that the
 * next g16_prove_partial / g16_prove_partial_h over the same (pk shard, device assignment) would otherwise enqueue after the map's
 * forty launches -- so that sort and map run side by side from the start.  The next such call consumes it; any other call on the
 * context drops it (the sort's buffers live in the per-call arena). */
int g16_prove_partial_prepare(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment_dev,
                              uint64_t n_assign);
/* g16_prove_partial with h supplied by the caller (device memory, h_len Fr; the key's h shard indexes it from h.start) */
int g16_prove_partial_h(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign,
                        int assignment_on_device, const uint64_t* h_dev, uint64_t h_len, int skip_b_g1, g16_partial* out);

/* the same fold + glue without a GPU context: only the eight fixed points of `fixed` are read.  Lets a host
 * process that merely aggregates shard records (or the CPU multi-rank tests) finish a proof. */
int g16_finalize_host(int curve, const g16_pk_view* fixed, const g16_partial* parts, int n_parts, const uint64_t r[4],
                      const uint64_t s[4], g16_proof* out);

int g16_get_timings(g16_ctx* ctx, g16_timings* out);

/* diagnostics behind bench.py's `roofline.valu_bound` (no reference counterpart): the issue rate of v_mad_u64_u32 -- the
 * instruction the field products are made of -- measured on this GPU now (all CUs, 8 waves per SIMD, independent chains), and
 * the multiply-adds one mixed addition of the G1 / G2 bucket kernels executes, counted from the same constexpr tables the
 * kernels are generated from (limb count, relaxed columns) */
typedef struct {
    double mad_per_s;          /* measured: v_mad_u64_u32 lane-operations per second, whole GPU                  */
    double mads_per_add_g1;    /* static: multiply-adds per G1 mixed addition (8 products + 2 squarings)           */
    double mads_per_add_g2;    /* static: per G2 mixed addition, both lanes of the pair together                   */
    double mads_per_product;   /* static: one base-field product (limb products + Montgomery reduction + relaxation) */
    int limbs;                 /* 30-bit limbs of the base field                                                   */
} g16_diag;
int g16_diag_valu(g16_ctx* ctx, g16_diag* out);

/* ---- unit-level entry points (parity tests, micro-benchmarks) ---- */

/* h_out: domain_size Fr, natural order */
int g16_witness_map(g16_ctx* ctx, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign,
                    int on_device, uint64_t* h_out);
/* bases affine, scalars Fr in Montgomery form (into_bigint is applied on the GPU, prover.rs:63-65);
 * out: affine result */
int g16_msm_g1(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine);
int g16_msm_g2(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine);
/* rank's bucket-space share of the same MSM (g16_pk_load_bucket_shard's cut, window tables built on the fly): the `world`
 * results add up to g16_msm_g1 / _g2 of the same inputs.  Parity tests. */
int g16_msm_bucket_shard(g16_ctx* ctx, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int rank, int world,
                         uint64_t* out_affine);
/* in place, natural order in and out, n = 2^log_n Fr in host memory */
int g16_ntt(g16_ctx* ctx, uint64_t* data, int log_n, int inverse, int coset);

/* ---- synthetic workload generators (bench.py) ---- */

/* n distinct non-identity points P_i = (s0 + first + i) * G written to DEVICE memory `out_dev`
 * (g2 = 0: G1 affine, 1: G2 affine) */
int g16_synth_bases(g16_ctx* ctx, int g2, uint64_t seed, uint64_t first, uint64_t n, uint64_t* out_dev);
/* SYN(k, seed) Fibonacci-product-chain R1CS: n_c = 2^k - 2, 2 instance variables, 2^k + 1 variables.
 * Host buffers: z_out (2^k+1) Fr, row_ptr n_c+1, colA/B/C n_c each, val n_c Fr (all one, shared). */
int g16_synth_circuit(int curve, int k, uint64_t seed, uint64_t* z_out, uint64_t* row_ptr, uint32_t* colA,
                      uint32_t* colB, uint32_t* colC, uint64_t* val);

/* ---- CRS generation: Groth16::generate_parameters_with_qap, src/generator.rs:47-208 ----
 * The matrices-level form, like g16_prove: the caller synthesises the circuit (generator.rs:62-76) and passes
 * cs.to_matrices().  The reference draws t from the rng (generator.rs:90); here it comes with the rest of the toxic
 * waste so that a run can be replayed.  Scalar work (Lagrange coefficients at t, a/b/c(t), l, gamma_abc, h scalars:
 * r1cs_to_qap.rs:120-170, 236-246) runs on the host, the ~5n fixed-base multiplications on the GPU. */
typedef struct {
    uint64_t alpha[4], beta[4], gamma[4], delta[4], t[4]; /* Fr, arkworks Montgomery limbs */
} g16_toxic_waste;

#define G16_PARAMS_DEVICE_PTRS 1u /* the five query arrays below are device memory of the ctx's GPU */

typedef struct {
    uint64_t *alpha_g1, *beta_g1, *delta_g1; /* one G1 affine each, host memory                                   */
    uint64_t *beta_g2, *delta_g2, *gamma_g2; /* one G2 affine each, host memory                                   */
    uint64_t* gamma_abc_g1;                  /* num_inputs G1, host memory (VerifyingKey, data_structures.rs:39)  */
    uint64_t *a_query, *b_g1_query;          /* num_variables G1 each; entry 0 belongs to the constant-one variable */
    uint64_t* b_g2_query;                    /* num_variables G2                                                  */
    uint64_t* h_query;                       /* domain_size - 1 G1                                                */
    uint64_t* l_query;                       /* num_variables - num_inputs G1                                     */
    uint32_t flags;
} g16_params_view;

/* status: G16_ERR_DEGREE_TOO_LARGE as in the prover; G16_ERR_UNEXPECTED_IDENTITY for gamma == 0 or delta == 0;
 * G16_ERR_BAD_ARG if t lies in the evaluation domain (sample_element_outside_domain never returns such a t) */
int g16_generate_parameters(g16_ctx* ctx, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints,
                            uint64_t num_variables, const g16_toxic_waste* toxic_waste, const uint64_t* g1_generator,
                            const uint64_t* g2_generator, const g16_params_view* out);
/* CPU only: LibsnarkReduction::instance_map_with_evaluation (r1cs_to_qap.rs:120-170) -- a, b, c: num_variables Fr each */
int g16_host_qap_evaluations(int curve, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints,
                             uint64_t num_variables, const uint64_t t[4], uint64_t* a_out, uint64_t* b_out,
                             uint64_t* c_out, uint64_t zt_out[4]);

/* ---- canonical (de)serialisation of points (CPU) ----
 * The element format behind `#[derive(CanonicalSerialize, CanonicalDeserialize)]` on Proof / VerifyingKey / ProvingKey
 * (src/data_structures.rs:8,31,125): BLS12-381 in the zcash / IETF form that ark-bls12-381 uses, BN254 in ark-ec's
 * default short-Weierstrass form (see serialize.hip; restated from the published formats, not checkable against the
 * reference here).  Containers (Vec<T> = u64 little-endian length + elements, struct = fields in order) are assembled by the
 * caller (groth16_amd/serialize.py).  `points`: affine Montgomery limbs as everywhere in this ABI. */
uint64_t g16_serialized_point_size(int curve, int g2, int compressed);
int g16_serialize_points(int curve, int g2, int compressed, const uint64_t* points, uint64_t n, uint8_t* out);
/* validate: 0 = Validate::No, 1 = on-curve check, 2 = on-curve + prime-order subgroup (Validate::Yes).
 * G16_ERR_INVALID_DATA for a non-canonical coordinate, inconsistent flags, an x with no point, or a failed check */
int g16_deserialize_points(int curve, int g2, int compressed, const uint8_t* in, uint64_t n, int validate,
                           uint64_t* points_out);

/* ---- host-side arithmetic self-test hooks (CPU; used by the `not gpu` tests) ----
 * The same field / group code the kernels use, compiled for the host.  They ship IN the product library on purpose: the CPU tier --
 * the only tier that runs where the library is built, a container without a GPU -- checks the 30-bit lazy arithmetic, its overflow
 * and bound proofs, the window-table tasks and the bucket-method model (g16_host_selftest, g16_host_msm_model[_shard]) on exactly the
 * code objects the kernels are generated from.  None of them is on a proof's path; none touches a GPU.
 * which: 0 Fr, 1 Fq.  op: 0 add, 1 sub, 2 mul, 3 inverse(a), 4 to_canonical(a), 5 from_canonical(a) */
int g16_host_field_op(int curve, int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out);
/* g2: 0/1.  op: 0 p+q (affine in, affine out), 1 k*p (k canonical 4 limbs), 2 p+q via XYZZ+XYZZ add */
int g16_host_group_op(int curve, int g2, int op, const uint64_t* p, const uint64_t* q_or_k, uint64_t* out);
/* CPU model of the MSM bucket method exactly as the kernels run it (signed digits, window c):
 * checks digit extraction + bucket reduction + window fold logic without a GPU */
int g16_host_msm_model(int curve, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int c,
                       uint64_t* out_affine);
/* the same model of rank's bucket-space share (c < 0: merged plan with window size -c) */
int g16_host_msm_model_shard(int curve, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int c, int rank,
                             int world, uint64_t* out_affine);

/* randomized CPU self-test of the reduced-radix (30-bit limb) arithmetic used by the bucket kernel against the
 * standard field / group code; 0 = all checks passed, otherwise the number of the first failing check */
int g16_host_selftest(int curve, uint64_t seed, int iters);

const char* g16_strerror(int status);
/* text of the last HIP error seen on this thread ("" if none) */
const char* g16_last_error(void);
const char* g16_version(void);
/* ABI revision of this header: bumped whenever a struct written by the library GROWS (fields are only ever appended).  2 = round 5's
 * g16_timings.g1_pass_launches and g16_pk_info.  A caller compiled against an older header must not let the library write the
 * longer struct into its shorter one: it checks g16_abi_version() == G16_ABI_VERSION at start-up, or asks for the library's struct
 * sizes (g16_struct_size), or uses the *_sized readers below, which copy min(size, library's size) bytes and never more. */
#define G16_ABI_VERSION 2
int g16_abi_version(void);
enum { G16_STRUCT_TIMINGS = 0, G16_STRUCT_PK_INFO = 1, G16_STRUCT_DIAG = 2, G16_STRUCT_PROOF = 3, G16_STRUCT_PARTIAL = 4, G16_STRUCT_PK_VIEW = 5 };
/* sizeof the library's own idea of a struct of this header; 0 for an unknown `which` */
uint64_t g16_struct_size(int which);
/* g16_get_timings / g16_pk_get_info into a caller struct of `size` bytes (its sizeof at ITS compile time) */
int g16_get_timings_sized(g16_ctx* ctx, void* out, uint64_t size);
int g16_pk_get_info_sized(const g16_pk* pk, void* out, uint64_t size);

#ifdef __cplusplus
}
#endif
#endif /* G16_MI355X_H */
