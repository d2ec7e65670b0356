// C ABI of libg16_mi355x.so (declared in include/g16_mi355x.h): contexts, device-resident proving
// keys and circuits, the prover orchestration and the unit-level entry points.
//
// Orchestration restates Groth16::create_proof_with_reduction_and_matrices
// (/root/reference/src/prover.rs:26-51) and create_proof_with_assignment (:54-132): the witness
// map and the five MSMs run on the GPU; the O(1) glue of :76-131 (six scalar multiples, a
// handful of additions, three into_affine) runs on the host with the same field code.
//
// Layout of this translation unit: api_types.hpp (handles), key_load.hpp (g16_pk_load / g16_circuit_load), prover.hpp (the GPU
// schedule of one proof), proof_glue.hpp (the host glue of prover.rs:76-131), unit_api.hpp (unit-level entry points),
// multi_device.hpp (host threads of a multi-device context); below: curve dispatch and the extern "C" functions themselves.
#include "api_types.hpp"
#include "key_load.hpp"
#include "prover.hpp"
#include "proof_glue.hpp"
#include "unit_api.hpp"

namespace {
template <class C>
struct Impl : KeyLoader<C>, Prover<C>, ProofGlue<C>, UnitApi<C> {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
};
}  // namespace

#define G16_DISPATCH(curve, EXPR)                                             \
    do {                                                                      \
        try {                                                                 \
            if ((curve) == G16_BLS12_381) { typedef Impl<Bls12_381> I; return EXPR; } \
            if ((curve) == G16_BN254) { typedef Impl<Bn254> I; return EXPR; }  \
            return G16_ERR_BAD_ARG;                                           \
        } catch (const std::bad_alloc&) {                                     \
            return G16_ERR_OOM;                                               \
        } catch (...) {                                                       \
            return G16_ERR_INTERNAL;                                          \
        }                                                                     \
    } while (0)


#include "multi_device.hpp"

static void diag_counts(g16_diag* out) {
    constexpr int NL = B30::NL;
    out->limbs = NL;
#ifdef G16_FIPS
    // product scanning (round 6): NL^2 multiply-adds per sweep and per reduction, 2 per extra accumulator of the column plan -- the same
    // constexpr plan the assembly blocks are generated from (fp30.hpp fips_plan) -- and NL - 1 per fused subtraction (K p_j - s_j joins
    // its column with one multiply-add by 1)
    constexpr auto p1 = B30::template fips_plan<1>();
    constexpr auto p2 = B30::template fips_plan<2>();
    constexpr auto p4 = B30::template fips_plan<4>();
    const double fsub = NL - 1;
    const double mul = 2.0 * NL * NL + 2.0 * p1.extra, sqr = NL * (NL + 1) / 2.0 + NL * NL + 2.0 * p1.extra;
    const double two_sweeps = 3.0 * NL * NL + 2.0 * p2.extra;    // two product sweeps + ONE reduction: a lane's Fq2 product; Fp30::mul2_cols
    const double four_sweeps = 5.0 * NL * NL + 2.0 * p4.extra;   // the lane pair's fused Y
    out->mads_per_product = mul;
    // the mixed addition as the bucket pass runs it (AccParked): U2 - x and S2 - y with the subtraction fused, PP, PPP, Q, ZZ*PP, ZZZ*PPP,
    // R^2 with the X3 subtraction fused, Y as two products under one reduction
#ifdef G16_NO_FUSED_SUB
    const double fs = 0.0;
#else
    const double fs = fsub;
#endif
    out->mads_per_add_g1 = 6 * mul + 2 * sqr + two_sweeps + 3 * fs;
    // per lane of the pair: six pair products (two sweeps each), two pair squarings (one product each), Y as four sweeps + one reduction
    out->mads_per_add_g2 = 2 * (6 * two_sweeps + 2 * mul + four_sweeps + 3 * fs);
#else
    int relax[5] = {0, 0, 0, 0, 0};   // relax[k]: columns relaxed when k sweeps are in and one more (a sweep or the reduction) follows
    for (int c = 0; c + 1 < 2 * NL; ++c)
        for (int k = 1; k <= 4; ++k)
            if ((k + 1) * B30::col_count(c) > G16_RELAX_LIMIT) ++relax[k];
    const double mul = 2.0 * NL * NL + relax[1], sqr = NL * (NL + 1) / 2.0 + NL * NL + relax[1];
    const double two_sweeps = 3.0 * NL * NL + relax[1] + relax[2];   // two product sweeps + ONE reduction: a lane's Fq2 product; Fp30::mul_sub_cols
    const double four_sweeps = 5.0 * NL * NL + relax[1] + relax[2] + relax[3] + relax[4];   // Fp2p30::pair_mul_sub
    out->mads_per_product = mul;
    out->mads_per_add_g1 = 6 * mul + 2 * sqr + two_sweeps;
    out->mads_per_add_g2 = 2 * (6 * two_sweeps + 2 * mul + four_sweeps);
#endif
}

// g16_pk_load on a multi-device context: the retry with base ranges after an automatic bucket-space load ran out of memory
static thread_local bool g_multi_force_base = false;

extern "C" {

int g16_ctx_create(int curve, int device_id, g16_ctx** out) {
    if (!out || (curve != G16_BLS12_381 && curve != G16_BN254)) return G16_ERR_BAD_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
        g_last_error = "no HIP device visible (the prover has no CPU fallback)";
        return G16_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= count) return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(device_id));
    g16_ctx* c = new (std::nothrow) g16_ctx();
    if (!c) return G16_ERR_OOM;
    c->curve = curve;
    c->device = device_id;
    memset(&c->tm, 0, sizeof(c->tm));
    // streams 2, 3 and the reduction streams carry the short, latency-bound work (digit/sort passes, reductions): give them
    // priority over the long bucket passes they run underneath (measured effect: small -- a kernel whose waves hold every
    // register slot for milliseconds is not displaced by priority; see prove_partial)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);   // numerically lower = higher priority
    bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_lo) == hipSuccess &&
              hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->stream_wm, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->stream_h2d, hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_dwm, hipEventDisableTiming) == hipSuccess &&
              hipEventCreate(&c->ev_edge[0]) == hipSuccess && hipEventCreate(&c->ev_edge[1]) == hipSuccess &&
              hipEventCreate(&c->ev_edge[2]) == hipSuccess && hipEventCreate(&c->ev_edge[3]) == hipSuccess &&
              hipEventCreate(&c->ev_edge[4]) == hipSuccess && hipEventCreate(&c->ev_edge[5]) == hipSuccess &&
              hipEventCreate(&c->ev_edge[6]) == hipSuccess && hipEventCreate(&c->ev_edge[7]) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_heavy[0], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_heavy[1], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_heavy[2], hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_heavy[3], hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithPriority(&c->red[0], hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->red[1], hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->red[2], hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->red[3], hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipStreamCreateWithPriority(&c->red[4], hipStreamNonBlocking, prio_hi) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_wm, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_z, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_h, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; ok && i < 5; ++i)
        ok = hipEventCreate(&c->ev_done[i]) == hipSuccess;
    for (int i = 0; ok && i < 8; ++i)
        ok = hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        g16::set_last_error("stream/event creation", hipGetLastError(), __FILE__, __LINE__);
        delete c;
        return G16_ERR_HIP;
    }
    *out = c;
    return G16_OK;
}

// one context over several GPUs of the node, so that the reference-shaped call stays ONE call
// (Groth16::prove, src/lib.rs:76-82): g16_pk_load cuts the key into n_dev contiguous shards (one per device),
// g16_circuit_load replicates the matrices, g16_prove runs one host thread per device (g16_prove_partial on its shard) and
// folds the n_dev partial records on the host.  The exchange needs no collective in this single-process form: a partial record
// is 1 152 bytes that each device's thread already leaves in host memory.  (The process-per-GPU form, bench.py --gpus N,
// exchanges the same records with one RCCL all-gather.)  Device ids may repeat (several shards on one GPU: tests).
int g16_ctx_create_multi(int curve, const int* device_ids, int n_dev, g16_ctx** out) {
    if (!out || !device_ids || n_dev < 1 || n_dev > 64 || (curve != G16_BLS12_381 && curve != G16_BN254)) return G16_ERR_BAD_ARG;
    if (n_dev == 1) return g16_ctx_create(curve, device_ids[0], out);
    g16_ctx* c = new (std::nothrow) g16_ctx();
    if (!c) return G16_ERR_OOM;
    c->curve = curve;
    c->device = device_ids[0];
    c->stream = c->stream2 = c->stream3 = nullptr;
    memset(&c->tm, 0, sizeof(c->tm));
    for (int i = 0; i < n_dev; ++i) {
        g16_ctx* sub = nullptr;
        const int rc = g16_ctx_create(curve, device_ids[i], &sub);
        if (rc) {
            for (g16_ctx* s : c->subs) g16_ctx_destroy(s);
            delete c;
            return rc;
        }
        c->subs.push_back(sub);
    }
    // Direct xGMI copies between the devices (the exchange of the distributed witness map).  Probed, not assumed: for every ordered
    // pair of DISTINCT devices hipDeviceCanAccessPeer is asked first and the answer kept (g16_ctx_peer_access).  A pair without peer
    // access still works -- hipMemcpyPeerAsync then stages through host memory -- but slowly, and a caller that sized its node for
    // xGMI wants to know: G16_MULTI_REQUIRE_PEER=1 turns a refusal into G16_ERR_NO_PEER_ACCESS at create time instead of a slow prover.
    // A pair that CAN be enabled and then fails to (anything but "already enabled") is an error in its own right.
    // G16_MULTI_FAKE_NO_PEER=1 (tests: one physical GPU listed several times has no distinct pair) treats every pair as refused.
    const bool require_peer = getenv("G16_MULTI_REQUIRE_PEER") && atoi(getenv("G16_MULTI_REQUIRE_PEER")) != 0;
    const bool fake_refusal = getenv("G16_MULTI_FAKE_NO_PEER") && atoi(getenv("G16_MULTI_FAKE_NO_PEER")) != 0;
    c->peer.assign((size_t)n_dev * (size_t)n_dev, 1);
    auto fail = [&](int rc) {
        for (g16_ctx* sub : c->subs) g16_ctx_destroy(sub);
        delete c;
        return rc;
    };
    for (int a = 0; a < n_dev; ++a)
        for (int b = 0; b < n_dev; ++b) {
            if (a == b) continue;
            int can = device_ids[a] == device_ids[b] ? 1 : 0;   // the same physical device: plain device-to-device copies
            if (!can && hipDeviceCanAccessPeer(&can, device_ids[a], device_ids[b]) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            if (fake_refusal) can = 0;
            if (can && device_ids[a] != device_ids[b]) {
                if (hipSetDevice(device_ids[a]) != hipSuccess) return fail(G16_ERR_HIP);
                const hipError_t e = hipDeviceEnablePeerAccess(device_ids[b], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    g16::set_last_error("hipDeviceEnablePeerAccess (the pair reports peer capability)", e, __FILE__, __LINE__);
                    return fail(G16_ERR_HIP);
                }
                (void)hipGetLastError();
            }
            c->peer[(size_t)a * n_dev + b] = (char)can;
            if (!can && require_peer) {
                char buf[160];
                snprintf(buf, sizeof(buf), "devices %d and %d of the context have no peer access (G16_MULTI_REQUIRE_PEER=1)", device_ids[a], device_ids[b]);
                g_last_error = buf;
                return fail(G16_ERR_NO_PEER_ACCESS);
            }
        }
    *out = c;
    return G16_OK;
}

int g16_ctx_peer_access(const g16_ctx* ctx, int i, int j) {
    if (!ctx) return -1;
    const int n = ctx->subs.empty() ? 1 : (int)ctx->subs.size();
    if (i < 0 || j < 0 || i >= n || j >= n) return -1;
    if (i == j || ctx->subs.empty()) return 1;
    return ctx->peer[(size_t)i * n + j] ? 1 : 0;
}

int g16_ctx_num_devices(const g16_ctx* ctx) { return ctx ? (ctx->subs.empty() ? 1 : (int)ctx->subs.size()) : 0; }

void g16_ctx_destroy(g16_ctx* ctx) {
    if (!ctx) return;
    if (!ctx->subs.empty()) {
        for (g16_ctx* sub : ctx->subs) g16_ctx_destroy(sub);
        delete ctx;
        return;
    }
    ctx->finprep.drop();
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->stream2);
    (void)hipStreamSynchronize(ctx->stream3);
    (void)hipStreamSynchronize(ctx->stream_wm);
    (void)hipStreamSynchronize(ctx->stream_h2d);
    for (int i = 0; i < 5; ++i) (void)hipStreamSynchronize(ctx->red[i]);
    ctx->arena.release();
    ctx->t_wm.destroy(); ctx->t_prep_h.destroy(); ctx->t_prep_z.destroy(); ctx->t_ntt[0].destroy(); ctx->t_ntt[1].destroy();
    for (int i = 0; i < 5; ++i) {
        ctx->t_bucket[i].destroy();
        (void)hipEventDestroy(ctx->ev_done[i]);
    }
    (void)hipEventDestroy(ctx->ev_z); (void)hipEventDestroy(ctx->ev_h); (void)hipEventDestroy(ctx->ev_wm); (void)hipEventDestroy(ctx->ev_dwm);
    for (int i = 0; i < 4; ++i) (void)hipEventDestroy(ctx->ev_heavy[i]);
    for (int i = 0; i < 8; ++i) (void)hipEventDestroy(ctx->ev_edge[i]);
    for (int i = 0; i < 8; ++i) (void)hipEventDestroy(ctx->ev_up[i]);
    (void)hipStreamDestroy(ctx->stream_h2d);
    (void)hipStreamDestroy(ctx->stream_wm);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->stream2);
    (void)hipStreamDestroy(ctx->stream3);
    for (int i = 0; i < 5; ++i) (void)hipStreamDestroy(ctx->red[i]);
    delete ctx;
}

void* g16_ctx_stream(g16_ctx* ctx) { return ctx ? (void*)(ctx->subs.empty() ? ctx->stream : ctx->subs[0]->stream) : nullptr; }
void* g16_ctx_wm_stream(g16_ctx* ctx) { return ctx ? (void*)(ctx->subs.empty() ? ctx->stream_wm : ctx->subs[0]->stream_wm) : nullptr; }

int g16_pk_load(g16_ctx* ctx, const g16_pk_view* view, g16_pk** out) {
    if (!ctx || !view || !out) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) {
        // the WHOLE key, cut here: a / b_g1 / b_g2 identically, l at the matching places (l index j <-> a index j + num_inputs - 1,
        // and num_inputs - 1 = m - w for a whole key) so that every shard keeps sharing one witness sort, h evenly
        const int n = (int)ctx->subs.size();
        if (view->a.start || view->b_g1.start || view->b_g2.start || view->h.start || view->l.start) return G16_ERR_BAD_ARG;
        if (view->b_g1.count != view->a.count || view->b_g2.count != view->a.count || view->l.count > view->a.count) return G16_ERR_BAD_LENGTH;
        if (view->flags & G16_PK_DEVICE_PTRS)
            for (g16_ctx* sub : ctx->subs) if (sub->device != ctx->subs[0]->device) return G16_ERR_BAD_ARG;   // host pointers across devices
        g16_pk* h = new (std::nothrow) g16_pk{ctx->curve, ctx, nullptr};
        if (!h) return G16_ERR_OOM;
        h->subs.assign((size_t)n, nullptr);
        const uint64_t m = view->a.count, w = view->l.count, hl = view->h.count, skip = m - w;
        const size_t g1b = (ctx->curve == G16_BLS12_381 ? 12 : 8), g2b = 2 * g1b;   // u64 limbs per G1 / G2 affine point
        // h: when the device count admits the distributed witness map over the key's domain (hl + 1 points, generator.rs:168),
        // shard i holds the bases of ITS block of h -- indices (i blk + j) + M k1 in the order [k1][j], the last one (n - 1) has
        // no base -- instead of a contiguous range; g16_prove then never replicates the transforms
        const bool dist_h = !(view->flags & G16_PK_DEVICE_PTRS) && view->h.points && dist_wm_admissible(n, hl + 1);
        if (dist_h) h->dist_n = hl + 1;
        // How the five MSMs are cut over the devices.  Base ranges: 1 / n of the window tables per device.  Bucket space
        // (round 5): the WHOLE tables on every device, device i owns the buckets b mod n == i, and the reductions shrink with n too --
        // chosen (G16_MULTI_SHARD_MODE=auto, the default) while the whole key's tables stay below 80 GiB and fit in the free memory of
        // every device with room for the per-proof arena; =base / =bucket force.  A forced bucket-space load that does not fit is an
        // error (no silent fall-back in that mode); auto falls back to base ranges.
        bool bucket = false, bucket_forced = false;
        {
            const char* me = g_multi_force_base ? "base" : getenv("G16_MULTI_SHARD_MODE");
            if (me && strcmp(me, "bucket") == 0) bucket = bucket_forced = true;
            else if (!me || strcmp(me, "auto") == 0) {
                const double key_bytes = (double)(g1b * 8) * (double)(m * 2 + w + hl) + (double)(g2b * 8) * (double)m;
                const double rows = m >= ((uint64_t)1 << 20) ? 13.0 : 32.0;
                const double need = rows * key_bytes * 1.08 + 3.0 * key_bytes + 8.0 * 1073741824.0;
                bucket = rows * key_bytes <= 80.0 * 1073741824.0 && !(view->flags & G16_PK_DEVICE_PTRS);
                // a key that may not have merged window tables cannot be cut in bucket space (KeyLoader::pk_load refuses it): plain
                // bases requested, or nothing to build tables from -- base ranges take any key
                const char* pc = getenv("G16_MSM_PRECOMP");
                if ((pc && atoi(pc) == 0) || m <= 1 || hl == 0) bucket = false;
                for (g16_ctx* sub : ctx->subs) {
                    size_t fr = 0, tot = 0;
                    if (!bucket) break;
                    if (hipSetDevice(sub->device) != hipSuccess || hipMemGetInfo(&fr, &tot) != hipSuccess) { (void)hipGetLastError(); bucket = false; }
                    // a device listed k times holds k copies (tests)
                    size_t copies = 0;
                    for (g16_ctx* o : ctx->subs) copies += o->device == sub->device ? 1 : 0;
                    if (need * (double)copies > (double)fr) bucket = false;
                }
            }
            if (bucket && (view->flags & G16_PK_DEVICE_PTRS)) { delete h; return G16_ERR_BAD_ARG; }
        }
        h->bucket_mode = bucket;
        std::vector<uint64_t> hgather;   // bucket space + distributed map: h_query in the order the all-gathered blocks of h arrive in
        if (bucket && dist_h) {
            const uint64_t M = (hl + 1) / (uint64_t)n, blk = M / (uint64_t)n;
            hgather.reserve((size_t)(hl * g1b));
            for (uint64_t q = 0; q < (uint64_t)n; ++q)
                for (uint64_t k1 = 0; k1 < (uint64_t)n; ++k1)
                    for (uint64_t j = 0; j < blk; ++j) {
                        const uint64_t idx = q * blk + j + M * k1;
                        if (idx < hl) hgather.insert(hgather.end(), view->h.points + idx * g1b, view->h.points + (idx + 1) * g1b);
                    }
        }
        int rc = for_each_device(n, [&](int i) -> int {
            g16_pk_view v = *view;
            if (bucket) {
                if (dist_h) { v.h.points = hgather.data(); v.h.count = hgather.size() / g1b; v.h.start = 0; }
                return g16_pk_load_bucket_shard(ctx->subs[(size_t)i], &v, i, n, &h->subs[(size_t)i]);
            }
            const uint64_t a_lo = m * (uint64_t)i / n, a_hi = m * (uint64_t)(i + 1) / n;
            const uint64_t l_lo = std::min(w, a_lo > skip ? a_lo - skip : 0), l_hi = std::min(w, a_hi > skip ? a_hi - skip : 0);
            const uint64_t h_lo = hl * (uint64_t)i / n, h_hi = hl * (uint64_t)(i + 1) / n;
            std::vector<uint64_t> hblock;
            if (dist_h) {
                const uint64_t M = (hl + 1) / (uint64_t)n, blk = M / (uint64_t)n;
                hblock.reserve((size_t)(M * g1b));
                for (uint64_t k1 = 0; k1 < (uint64_t)n; ++k1)
                    for (uint64_t j = 0; j < blk; ++j) {
                        const uint64_t idx = (uint64_t)i * blk + j + M * k1;
                        if (idx < hl) hblock.insert(hblock.end(), view->h.points + idx * g1b, view->h.points + (idx + 1) * g1b);
                    }
            }
            auto cut = [](const g16_query& q, uint64_t lo, uint64_t hi, size_t limbs) {
                g16_query r;
                r.points = q.points ? q.points + lo * limbs : nullptr;
                r.count = hi - lo;
                r.start = lo;
                return r;
            };
            v.a = cut(view->a, a_lo, a_hi, g1b);
            v.b_g1 = cut(view->b_g1, a_lo, a_hi, g1b);
            v.b_g2 = cut(view->b_g2, a_lo, a_hi, g2b);
            v.l = cut(view->l, l_lo, l_hi, g1b);
            v.h = cut(view->h, h_lo, h_hi, g1b);
            if (dist_h) { v.h.points = hblock.data(); v.h.count = hblock.size() / g1b; v.h.start = 0; }
            return g16_pk_load(ctx->subs[(size_t)i], &v, &h->subs[(size_t)i]);
        }, serial_loads(ctx));
        if ((rc == G16_ERR_OOM || rc == G16_ERR_BAD_ARG) && bucket && !bucket_forced) {
            // auto guessed wrong (other tenants of the HBM; a window size the environment forces that admits no merged tables --
            // the bucket-space loader answers BAD_ARG): give the partial loads back and cut by base ranges instead.  A view that
            // is bad for its own reasons gets the same code from the retry.
            for (g16_pk*& sub : h->subs) { g16_pk_free(sub); sub = nullptr; }
            delete h;
            g_multi_force_base = true;
            const int rc2 = g16_pk_load(ctx, view, out);
            g_multi_force_base = false;
            return rc2;
        }
        if (rc) { g16_pk_free(h); return rc; }
        *out = h;
        return G16_OK;
    }
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::pk_load(ctx, view, out));
}

// Bucket-space shard (round 5): the rank holds the WHOLE key as window tables and owns the buckets b mod world == rank of all five
// MSMs; its g16_prove_partial[_h] then yields the partial sums over those buckets, and the ranks' records add up to the MSM sums in
// g16_prove_finalize exactly like base-range shards' do.
int g16_pk_load_bucket_shard(g16_ctx* ctx, const g16_pk_view* view, int rank, int world, g16_pk** out) {
    if (!ctx || !view || !out || world < 1 || world > 64 || rank < 0 || rank >= world) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return G16_ERR_BAD_ARG;   // per-device form only (one process per GPU)
    if (world == 1) return g16_pk_load(ctx, view, out);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::pk_load(ctx, view, out, world, rank));
}

// the same resident tables seen as another rank's bucket-space shard (the tables do not depend on the rank): lets ONE GPU walk through
// every rank's share of a sharded proof -- the parity tests at sizes where loading the key once per rank would dominate, and
// bench.py --sim-shards
int g16_pk_rebind_bucket_shard(g16_pk* pk, int rank, int world) {
    if (!pk || world < 1 || world > 64 || rank < 0 || rank >= world || !pk->subs.empty() || !pk->dp) return G16_ERR_BAD_ARG;
    auto rebind = [&](auto* dp) -> int {
        if (dp->c_z == 0 || dp->a_start || dp->b_g1_start || dp->b_g2_start || dp->h_start || dp->l_start) return G16_ERR_BAD_ARG;
        dp->shard_n = world;
        dp->shard_r = rank;
        return G16_OK;
    };
    if (pk->ctx->finprep.key_id == pk->id) pk->ctx->finprep.drop();
    pk->ctx->drop_prepared_sort(pk->id);   // a prepared sort kept the OLD residue class
    if (pk->curve == G16_BLS12_381) return rebind(static_cast<DevicePk<Bls12_381>*>(pk->dp));
    return rebind(static_cast<DevicePk<Bn254>*>(pk->dp));
}

int g16_pk_get_info(const g16_pk* pk, g16_pk_info* out) {
    if (!pk || !out) return G16_ERR_BAD_ARG;
    memset(out, 0, sizeof(*out));
    const g16_pk* one = pk->subs.empty() ? pk : pk->subs[0];
    if (!one || !one->dp) return G16_ERR_BAD_ARG;
    auto fill = [&](const auto* dp) {
        out->window_bits_z = dp->c_z;
        out->window_bits_h = dp->c_h;
        out->table_fallback = dp->table_fallback;
        out->bucket_shard_rank = dp->shard_r;
        out->bucket_shard_world = dp->shard_n;
        out->device_bytes = dp->table_bytes;
    };
    if (one->curve == G16_BLS12_381) fill(static_cast<const DevicePk<Bls12_381>*>(one->dp));
    else fill(static_cast<const DevicePk<Bn254>*>(one->dp));
    out->n_devices = pk->subs.empty() ? 1 : (int)pk->subs.size();
    return G16_OK;
}

void g16_pk_free(g16_pk* pk) {
    if (!pk) return;
    if (!pk->ctx->subs.empty()) {
        for (g16_pk* sub : pk->subs) g16_pk_free(sub);
        delete pk;
        return;
    }
    // a prepared finalize half of this key on its own context is forgotten here; halves prepared by OTHER contexts that share the
    // key hold their own reference to its host side (KeyGlue) and are matched by key id, so they neither dangle nor match a later key
    if (pk->ctx->finprep.key_id == pk->id) pk->ctx->finprep.drop();
    (void)hipSetDevice(pk->ctx->device);
    pk->ctx->drop_prepared_sort(pk->id);   // its buffers index THIS key's bases (and may still be running on stream 2)
    if (pk->curve == G16_BLS12_381) Impl<Bls12_381>::pk_free(static_cast<DevicePk<Bls12_381>*>(pk->dp));
    else Impl<Bn254>::pk_free(static_cast<DevicePk<Bn254>*>(pk->dp));
    delete pk;
}

int g16_circuit_load(g16_ctx* ctx, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                     g16_circuit** out) {
    if (!ctx || !abc || !out) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) {
        const int n = (int)ctx->subs.size();
        g16_circuit* h = new (std::nothrow) g16_circuit{ctx->curve, ctx, nullptr, 0};
        if (!h) return G16_ERR_OOM;
        h->subs.assign((size_t)n, nullptr);
        const int rc = for_each_device(n, [&](int i) -> int {
            return g16_circuit_load(ctx->subs[(size_t)i], abc, num_inputs, num_constraints, num_variables, &h->subs[(size_t)i]);
        }, serial_loads(ctx));
        if (rc) { g16_circuit_free(h); return rc; }
        h->domain_size = h->subs[0]->domain_size;
        h->num_variables = num_variables;
        if (dist_wm_admissible(n, h->domain_size)) {
            h->dist.assign((size_t)n, DwmSlot());
            const uint64_t M = h->domain_size / (uint64_t)n;
            const int rc2 = for_each_device(n, [&](int i) -> int {
                DwmSlot& sl = h->dist[(size_t)i];
                G16_TRY(g16_dwm_create(ctx->subs[(size_t)i], h->subs[(size_t)i], i, n, &sl.dwm));
                for (int k = 0; k < 3; ++k) {
                    if (hipMalloc((void**)&sl.work[k], M * 32) != hipSuccess || hipMalloc((void**)&sl.recv[k], M * 32) != hipSuccess) return G16_ERR_OOM;
                }
                if (hipMalloc((void**)&sl.h_local, M * 32) != hipSuccess || hipMalloc((void**)&sl.z_dev, (num_variables ? num_variables : 1) * 32) != hipSuccess)
                    return G16_ERR_OOM;   // (h_full: on first use by a bucket-space key, g16_prove)
                return G16_OK;
            }, serial_loads(ctx));
            if (rc2) { g16_circuit_free(h); return rc2; }
        }
        *out = h;
        return G16_OK;
    }
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::circuit_load(ctx, abc, num_inputs, num_constraints, num_variables, out));
}

void g16_circuit_free(g16_circuit* c) {
    if (!c) return;
    if (!c->ctx->subs.empty()) {
        for (size_t i = 0; i < c->dist.size(); ++i) {
            DwmSlot& sl = c->dist[i];
            (void)hipSetDevice(c->ctx->subs[i]->device);
            g16_dwm_free(sl.dwm);
            for (int k = 0; k < 3; ++k) { (void)hipFree(sl.work[k]); (void)hipFree(sl.recv[k]); }
            (void)hipFree(sl.h_local); (void)hipFree(sl.z_dev); (void)hipFree(sl.h_full);
        }
        for (g16_circuit* sub : c->subs) g16_circuit_free(sub);
        delete c;
        return;
    }
    (void)hipSetDevice(c->ctx->device);
    if (c->curve == G16_BLS12_381) Impl<Bls12_381>::circuit_free(static_cast<DeviceCircuit<Bls12_381>*>(c->dc));
    else Impl<Bn254>::circuit_free(static_cast<DeviceCircuit<Bn254>*>(c->dc));
    delete c;
}

uint64_t g16_circuit_domain_size(const g16_circuit* c) { return c ? c->domain_size : 0; }

// A key / circuit may be used by ANY single-device context on the GPU it was loaded on: the device data is read-only during a
// proof, everything a proof writes (arena, streams, events, pinned buffer, timers) belongs to the calling context.  Two contexts
// on one GPU proving side by side over one key is the throughput mode (the tail of one proof under the passes of the other).
static bool usable_on(const g16_ctx* owner, const g16_ctx* ctx) {
    return owner == ctx || (owner && ctx && owner->subs.empty() && ctx->subs.empty() && owner->device == ctx->device && owner->curve == ctx->curve);
}

int g16_prove_partial(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign,
                      int assignment_on_device, int skip_b_g1, g16_partial* out) {
    if (!ctx || !pk || !circuit || !full_assignment || !out) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve || circuit->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty() || !usable_on(pk->ctx, ctx) || !usable_on(circuit->ctx, ctx)) return G16_ERR_BAD_ARG;   // multi-device contexts: g16_prove
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::prove_partial(ctx, pk, circuit, full_assignment, n_assign, assignment_on_device, skip_b_g1, out));
}

int g16_prove_finalize(g16_ctx* ctx, const g16_pk* pk, const g16_partial* parts, int n_parts, const uint64_t r[4], const uint64_t s[4],
                       g16_proof* out) {
    if (!ctx || !pk || !parts || !r || !s || !out) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) {   // every shard carries the same eight fixed points
        if (pk->subs.empty()) return G16_ERR_BAD_ARG;
        return g16_prove_finalize(ctx->subs[0], pk->subs[0], parts, n_parts, r, s, out);
    }
    if (!pk->subs.empty() || !pk->dp) return G16_ERR_BAD_ARG;   // a multi-device key on a single-device context
    G16_DISPATCH(ctx->curve, I::prove_finalize(ctx, pk, parts, n_parts, r, s, out));
}

int g16_prove_finalize_prepare(g16_ctx* ctx, const g16_pk* pk, const uint64_t r[4], const uint64_t s[4]) {
    if (!ctx || !pk || !r || !s) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) {
        if (pk->subs.empty()) return G16_ERR_BAD_ARG;
        return g16_prove_finalize_prepare(ctx->subs[0], pk->subs[0], r, s);
    }
    if (!pk->subs.empty() || !pk->dp) return G16_ERR_BAD_ARG;   // a multi-device key on a single-device context
    G16_DISPATCH(ctx->curve, I::prove_finalize_prepare(ctx, pk, r, s));
}

int g16_finalize_host(int curve, const g16_pk_view* fixed, const g16_partial* parts, int n_parts, const uint64_t r[4], const uint64_t s[4],
                      g16_proof* out) {
    if (!fixed || !parts || !r || !s || !out) return G16_ERR_BAD_ARG;
    G16_DISPATCH(curve, I::finalize_host(fixed, parts, n_parts, r, s, out));
}

int g16_prove(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign,
              int assignment_on_device, const uint64_t r[4], const uint64_t s[4], g16_proof* out) {
    if (!r || !s || !out) return G16_ERR_BAD_ARG;
    g16_partial part;
    const uint64_t zero[4] = {0, 0, 0, 0};
    const int skip_b_g1 = memcmp(r, zero, 32) == 0;  // r == 0 skips B in G1 (prover.rs:98)
    if (ctx && !ctx->subs.empty()) {
        if (!pk || !circuit || !full_assignment || pk->ctx != ctx || circuit->ctx != ctx) return G16_ERR_BAD_ARG;
        const int n = (int)ctx->subs.size();
        if (assignment_on_device)   // a device pointer is valid on one GPU only
            for (g16_ctx* sub : ctx->subs) if (sub->device != ctx->subs[0]->device) return G16_ERR_BAD_ARG;
        const double t0 = now_ms();
        std::vector<g16_partial> parts((size_t)n);
        if (!pk->subs.empty()) (void)g16_prove_finalize_prepare(ctx->subs[0], pk->subs[0], r, s);   // host glue under the GPU work
        if (pk->dist_n && (circuit->dist.empty() || pk->dist_n != circuit->domain_size)) {
            g_last_error = "the key's h_query does not belong to this circuit's domain (h_query must hold domain_size - 1 bases)";
            ctx->subs[0]->finprep.drop();
            return G16_ERR_BAD_LENGTH;
        }
        int rc;
        double dwm_ms = 0.0;
        if (pk->dist_n) {
            // distributed witness map: every device's thread ENQUEUES its four stages on its witness-map stream; after stages 0, 1
            // (a, b, c) and 2 (the quotient) device i PULLS chunk i of every device's work array into slot q of its recv array
            // (peer copies over xGMI).  Ordering is by events, not by waiting for the GPU: a pull waits on the sources' stage
            // events, the next stage (which overwrites work[]) on the pullers' pull events; the host barriers only make sure an
            // event has been recorded before somebody waits on it.  Then the MSMs: the witness sort and the four h-independent
            // passes start at once, the h MSM follows the map (g16_prove_partial_h orders itself after the witness-map stream).
            if (n_assign != circuit->num_variables) { ctx->subs[0]->finprep.drop(); return G16_ERR_BAD_LENGTH; }
            HostBarrier bar(n);
            std::atomic<int> failed{0};
            const uint64_t M = circuit->domain_size / (uint64_t)n, blk = M / (uint64_t)n;
            std::vector<hipEvent_t> ev_stage((size_t)n, nullptr), ev_pull((size_t)n, nullptr), ev_up((size_t)n, nullptr);
            rc = for_each_device(n, [&](int i) -> int {
                g16_ctx* sub = ctx->subs[(size_t)i];
                const DwmSlot& sl = circuit->dist[(size_t)i];
                hipStream_t sw = sub->stream_wm;
                int my = G16_OK;
                int step_no = 0;
                const bool dbg = getenv("G16_DEBUG") != nullptr;
                auto step = [&](auto fn) {   // every thread passes every barrier, whatever failed where
                    if (dbg) fprintf(stderr, "[g16 multi] device %d step %d begin\n", i, step_no);
                    if (!failed.load() && my == G16_OK) { my = fn(); if (my) failed.store(1); }
                    if (dbg) fprintf(stderr, "[g16 multi] device %d step %d rc=%d, at barrier\n", i, step_no, my);
                    bar.wait();
                    ++step_no;
                };
                const uint64_t* zp = full_assignment;
                step([&]() -> int {
                    G16_HIP_TRY(hipSetDevice(sub->device));
                    G16_HIP_TRY(hipEventCreateWithFlags(&ev_stage[(size_t)i], hipEventDisableTiming));
                    G16_HIP_TRY(hipEventCreateWithFlags(&ev_pull[(size_t)i], hipEventDisableTiming));
                    G16_HIP_TRY(hipEventCreateWithFlags(&ev_up[(size_t)i], hipEventDisableTiming));
                    if (pk->bucket_mode && !sl.h_full && hipMalloc((void**)&sl.h_full, M * (uint64_t)n * 32) != hipSuccess) {
                        (void)hipGetLastError();
                        sl.h_full = nullptr;
                        return G16_ERR_OOM;
                    }
                    if (!assignment_on_device) {
                        G16_HIP_TRY(hipMemcpyAsync(sl.z_dev, full_assignment, n_assign * 32, hipMemcpyHostToDevice, sw));
                        zp = sl.z_dev;
                    }
                    G16_HIP_TRY(hipEventRecord(ev_up[(size_t)i], sw));
                    // the witness sort of this device's MSMs goes into the queues ahead of the map's stages (it reads the uploaded
                    // assignment: stream 2 waits for the upload)
                    G16_HIP_TRY(hipStreamWaitEvent(sub->stream2, ev_up[(size_t)i], 0));
                    G16_TRY(g16_prove_partial_prepare(sub, pk->subs[(size_t)i], circuit->subs[(size_t)i], zp, n_assign));
                    return G16_OK;
                });
                const double tw = now_ms();
                for (int st = 0; st < 4; ++st) {
                    step([&]() -> int {
                        G16_TRY(g16_dwm_stage_async(sub, sl.dwm, st, zp, n_assign, sl.work, sl.recv, sl.h_local));
                        if (st < 3 || pk->bucket_mode) G16_HIP_TRY(hipEventRecord(ev_stage[(size_t)i], sw));
                        return G16_OK;
                    });
                    if (st == 3) {
                        // bucket-space key: every device folds ALL of h (and 1 / n of the buckets) -- it pulls every device's block
                        // behind that device's last stage, into the order the key's h_query was loaded in (the all-gather as peer copies)
                        if (pk->bucket_mode)
                            step([&]() -> int {
                                G16_HIP_TRY(hipSetDevice(sub->device));
                                for (int q = 0; q < n; ++q) {
                                    G16_HIP_TRY(hipStreamWaitEvent(sw, ev_stage[(size_t)q], 0));
                                    G16_HIP_TRY(hipMemcpyPeerAsync(sl.h_full + (uint64_t)q * M * 4, sub->device, circuit->dist[(size_t)q].h_local,
                                                                   ctx->subs[(size_t)q]->device, M * 32, sw));
                                }
                                return G16_OK;
                            });
                        break;
                    }
                    step([&]() -> int {   // every device's stage event is recorded: pull
                        G16_HIP_TRY(hipSetDevice(sub->device));
                        for (int q = 0; q < n; ++q) G16_HIP_TRY(hipStreamWaitEvent(sw, ev_stage[(size_t)q], 0));
                        for (int a = 0; a < (st < 2 ? 3 : 1); ++a)
                            for (int q = 0; q < n; ++q)
                                G16_HIP_TRY(hipMemcpyPeerAsync(sl.recv[a] + (uint64_t)q * blk * 4, sub->device,
                                                               circuit->dist[(size_t)q].work[a] + (uint64_t)i * blk * 4, ctx->subs[(size_t)q]->device,
                                                               blk * 32, sw));
                        G16_HIP_TRY(hipEventRecord(ev_pull[(size_t)i], sw));
                        return G16_OK;
                    });
                    step([&]() -> int {   // every pull is enqueued: my next stage may overwrite work[] only after all of them
                        G16_HIP_TRY(hipSetDevice(sub->device));
                        for (int q = 0; q < n; ++q) G16_HIP_TRY(hipStreamWaitEvent(sw, ev_pull[(size_t)q], 0));
                        return G16_OK;
                    });
                }
                if (i == 0) dwm_ms = now_ms() - tw;   // host time to enqueue the map (the GPU runs it beside the witness sort)
                int out_rc = my;
                if (!failed.load() && my == G16_OK) {
                    // the uploaded assignment is read by the MSM streams too: order them after the upload
                    if (hipStreamWaitEvent(sub->stream, ev_up[(size_t)i], 0) != hipSuccess) out_rc = G16_ERR_HIP;
                    else
                        out_rc = g16_prove_partial_h(sub, pk->subs[(size_t)i], circuit->subs[(size_t)i], zp, n_assign, 1,
                                                     pk->bucket_mode ? sl.h_full : sl.h_local, pk->bucket_mode ? M * (uint64_t)n : M, skip_b_g1,
                                                     &parts[(size_t)i]);
                } else if (my == G16_OK) {
                    out_rc = SIBLING_FAILED;
                }
                (void)hipSetDevice(sub->device);
                (void)hipStreamSynchronize(sw);
                bar.wait();   // nobody destroys an event another device's stream may still be waiting on
                if (ev_stage[(size_t)i]) (void)hipEventDestroy(ev_stage[(size_t)i]);
                if (ev_pull[(size_t)i]) (void)hipEventDestroy(ev_pull[(size_t)i]);
                if (ev_up[(size_t)i]) (void)hipEventDestroy(ev_up[(size_t)i]);
                return out_rc;
            });
        } else {
            rc = for_each_device(n, [&](int i) -> int {
                return g16_prove_partial(ctx->subs[(size_t)i], pk->subs[(size_t)i], circuit->subs[(size_t)i], full_assignment, n_assign,
                                         assignment_on_device, skip_b_g1, &parts[(size_t)i]);
            });
        }
        if (rc) { ctx->subs[0]->finprep.drop(); return rc; }
        rc = g16_prove_finalize(ctx->subs[0], pk->subs[0], parts.data(), n, r, s, out);
        // timings: the slowest device's phases (the proof waits for it), wall time of the whole call
        int slow = 0;
        for (int i = 1; i < n; ++i) if (ctx->subs[(size_t)i]->tm.total_ms > ctx->subs[(size_t)slow]->tm.total_ms) slow = i;
        ctx->tm = ctx->subs[(size_t)slow]->tm;
        if (pk->dist_n) ctx->tm.witness_map_ms = dwm_ms;   // device 0's four stages + three exchanges (host clock)
        ctx->tm.finish_ms = ctx->subs[0]->tm.finish_ms;
        ctx->tm.total_ms = now_ms() - t0;
        return rc;
    }
    // (the argument checks of g16_prove_partial, made BEFORE a host thread is started on the key)
    if (!ctx || !pk || !circuit || !full_assignment) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve || circuit->curve != ctx->curve || !pk->subs.empty() || !pk->dp || !circuit->subs.empty() || !circuit->dc)
        return G16_ERR_BAD_ARG;
    if (!usable_on(pk->ctx, ctx) || !usable_on(circuit->ctx, ctx)) return G16_ERR_BAD_ARG;
    int rc = g16_prove_finalize_prepare(ctx, pk, r, s);   // the (r, s)-only host glue runs while the GPU works
    if (rc) return rc;
    rc = g16_prove_partial(ctx, pk, circuit, full_assignment, n_assign, assignment_on_device, skip_b_g1, &part);
    if (rc) { ctx->finprep.drop(); return rc; }
    return g16_prove_finalize(ctx, pk, &part, 1, r, s, out);
}

int g16_prove_partial_prepare(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment_dev, uint64_t n_assign) {
    if (!ctx || !pk || !circuit || !full_assignment_dev) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve || circuit->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty() || !usable_on(pk->ctx, ctx) || !usable_on(circuit->ctx, ctx)) return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::prove_partial_prepare(ctx, pk, circuit, full_assignment_dev, n_assign));
}

int g16_prove_partial_h(g16_ctx* ctx, const g16_pk* pk, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign,
                        int assignment_on_device, const uint64_t* h_dev, uint64_t h_len, int skip_b_g1, g16_partial* out) {
    if (!ctx || !pk || !circuit || !full_assignment || !h_dev || !out) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve || circuit->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty() || !usable_on(pk->ctx, ctx) || !usable_on(circuit->ctx, ctx)) return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::prove_partial(ctx, pk, circuit, full_assignment, n_assign, assignment_on_device, skip_b_g1, out,
                                              reinterpret_cast<const typename I::Fr*>(h_dev), h_len));
}

int g16_dwm_create(g16_ctx* ctx, const g16_circuit* circuit, int rank, int world, g16_dwm** out) {
    if (!ctx || !circuit || !out || circuit->curve != ctx->curve || circuit->ctx != ctx || !ctx->subs.empty()) return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(ctx->device));
    void* dw = nullptr;
    int rc;
    try {
        if (ctx->curve == G16_BLS12_381) {
            DistWm<Bls12_381>* d = nullptr;
            rc = dwm_create<Bls12_381>(static_cast<const DeviceCircuit<Bls12_381>*>(circuit->dc), rank, world, ctx->stream, &d);
            dw = d;
        } else {
            DistWm<Bn254>* d = nullptr;
            rc = dwm_create<Bn254>(static_cast<const DeviceCircuit<Bn254>*>(circuit->dc), rank, world, ctx->stream, &d);
            dw = d;
        }
    } catch (...) {
        return G16_ERR_OOM;
    }
    if (rc) return rc;
    g16_dwm* h = new (std::nothrow) g16_dwm{ctx->curve, ctx, circuit, dw, rank, world, circuit->domain_size / (uint64_t)world};
    if (!h) { g16_dwm tmp{ctx->curve, ctx, circuit, dw, rank, world, 0}; g16_dwm_free_impl(&tmp); return G16_ERR_OOM; }
    *out = h;
    return G16_OK;
}

void g16_dwm_free(g16_dwm* d) {
    if (!d) return;
    g16_dwm_free_impl(d);
    delete d;
}

uint64_t g16_dwm_local_size(const g16_dwm* d) { return d ? d->local_size : 0; }

int g16_dwm_stage(g16_ctx* ctx, g16_dwm* d, int stage, const uint64_t* full_assignment, uint64_t n_assign, int assignment_on_device,
                  uint64_t* const work[3], uint64_t* const recv[3], uint64_t* h_local) {
    if (!ctx || !d || d->ctx != ctx || !work || !recv || stage < 0 || stage > 3) return G16_ERR_BAD_ARG;
    if ((stage == 0 && (!work[0] || !work[1] || !work[2])) || (stage == 1 && (!work[0] || !work[1] || !work[2] || !recv[0] || !recv[1] || !recv[2])) ||
        (stage == 2 && (!work[0] || !recv[0] || !recv[1] || !recv[2])) || (stage == 3 && (!recv[0] || !h_local)))
        return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::dwm_stage_api(ctx, d->circuit, d->dw, stage, full_assignment, n_assign, assignment_on_device, work, recv, h_local, false));
}

int g16_dwm_stage_async(g16_ctx* ctx, g16_dwm* d, int stage, const uint64_t* full_assignment_dev, uint64_t n_assign, uint64_t* const work[3],
                        uint64_t* const recv[3], uint64_t* h_local) {
    if (!ctx || !d || d->ctx != ctx || !work || !recv || stage < 0 || stage > 3) return G16_ERR_BAD_ARG;
    if ((stage == 0 && (!work[0] || !work[1] || !work[2])) || (stage == 1 && (!work[0] || !work[1] || !work[2] || !recv[0] || !recv[1] || !recv[2])) ||
        (stage == 2 && (!work[0] || !recv[0] || !recv[1] || !recv[2])) || (stage == 3 && (!recv[0] || !h_local)))
        return G16_ERR_BAD_ARG;
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::dwm_stage_api(ctx, d->circuit, d->dw, stage, full_assignment_dev, n_assign, 1, work, recv, h_local, true));
}

int g16_get_timings(g16_ctx* ctx, g16_timings* out) {
    if (!ctx || !out) return G16_ERR_BAD_ARG;
    *out = ctx->tm;
    return G16_OK;
}

int g16_diag_valu(g16_ctx* ctx, g16_diag* out) {
    if (!ctx || !out) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_diag_valu(ctx->subs[0], out);
    memset(out, 0, sizeof(*out));
    if (ctx->curve == G16_BLS12_381) diag_counts<Fp30<Bls12_381::Fq::Params>>(out);
    else diag_counts<Fp30<Bn254::Fq::Params>>(out);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    return mad_rate_device(ctx->stream, &out->mad_per_s);
}

int g16_witness_map(g16_ctx* ctx, const g16_circuit* circuit, const uint64_t* full_assignment, uint64_t n_assign, int on_device,
                    uint64_t* h_out) {
    if (!ctx || !circuit || !full_assignment || !h_out || circuit->curve != ctx->curve || circuit->ctx != ctx) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_witness_map(ctx->subs[0], circuit->subs[0], full_assignment, n_assign, on_device, h_out);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::witness_map_api(ctx, circuit, full_assignment, n_assign, on_device, h_out));
}

int g16_msm_g1(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine) {
    if (!ctx || !out_affine || (n && (!bases || !scalars))) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_msm_g1(ctx->subs[0], bases, scalars, n, out_affine);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::template msm_api<typename I::Fq>(ctx, bases, scalars, n, out_affine));
}

int g16_msm_g2(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine) {
    if (!ctx || !out_affine || (n && (!bases || !scalars))) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_msm_g2(ctx->subs[0], bases, scalars, n, out_affine);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::template msm_api<typename I::Fq2>(ctx, bases, scalars, n, out_affine));
}

int g16_msm_bucket_shard(g16_ctx* ctx, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int rank, int world,
                         uint64_t* out_affine) {
    if (!ctx || !out_affine || (n && (!bases || !scalars)) || world < 1 || world > 64 || rank < 0 || rank >= world) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_msm_bucket_shard(ctx->subs[0], g2, bases, scalars, n, rank, world, out_affine);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    if (!g2) G16_DISPATCH(ctx->curve, I::template msm_api<typename I::Fq>(ctx, bases, scalars, n, out_affine, world, rank));
    G16_DISPATCH(ctx->curve, I::template msm_api<typename I::Fq2>(ctx, bases, scalars, n, out_affine, world, rank));
}

int g16_ntt(g16_ctx* ctx, uint64_t* data, int log_n, int inverse, int coset) {
    if (!ctx || !data) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_ntt(ctx->subs[0], data, log_n, inverse, coset);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::ntt_api(ctx, data, log_n, inverse, coset));
}

int g16_synth_bases(g16_ctx* ctx, int g2, uint64_t seed, uint64_t first, uint64_t n, uint64_t* out_dev) {
    if (!ctx || (n && !out_dev)) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) return g16_synth_bases(ctx->subs[0], g2, seed, first, n, out_dev);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    int rc;
    if (ctx->curve == G16_BLS12_381) rc = synth_bases_device<Bls12_381>(g2, seed, first, n, out_dev, ctx->stream);
    else rc = synth_bases_device<Bn254>(g2, seed, first, n, out_dev, ctx->stream);
    if (rc) return rc;
    G16_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return G16_OK;
}

int g16_host_field_op(int curve, int which, int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
    if (!a || !out) return G16_ERR_BAD_ARG;
    if (which == 0) G16_DISPATCH(curve, I::template field_op<typename I::Fr>(op, a, b, out));
    G16_DISPATCH(curve, I::template field_op<typename I::Fq>(op, a, b, out));
}

int g16_host_group_op(int curve, int g2, int op, const uint64_t* p, const uint64_t* q_or_k, uint64_t* out) {
    if (!p || !q_or_k || !out) return G16_ERR_BAD_ARG;
    if (!g2) G16_DISPATCH(curve, I::template group_op<typename I::Fq>(op, p, q_or_k, out));
    G16_DISPATCH(curve, I::template group_op<typename I::Fq2>(op, p, q_or_k, out));
}

int g16_host_msm_model(int curve, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int c, uint64_t* out_affine) {
    if (!out_affine || (n && (!bases || !scalars))) return G16_ERR_BAD_ARG;
    if (!g2) G16_DISPATCH(curve, I::template msm_model<typename I::Fq>(bases, scalars, n, c, out_affine));
    G16_DISPATCH(curve, I::template msm_model<typename I::Fq2>(bases, scalars, n, c, out_affine));
}

int g16_host_msm_model_shard(int curve, int g2, const uint64_t* bases, const uint64_t* scalars, uint64_t n, int c, int rank, int world,
                             uint64_t* out_affine) {
    if (!out_affine || (n && (!bases || !scalars)) || world < 1 || world > 64 || rank < 0 || rank >= world) return G16_ERR_BAD_ARG;
    if (!g2) G16_DISPATCH(curve, I::template msm_model<typename I::Fq>(bases, scalars, n, c, out_affine, world, rank));
    G16_DISPATCH(curve, I::template msm_model<typename I::Fq2>(bases, scalars, n, c, out_affine, world, rank));
}

int g16_generate_parameters(g16_ctx* ctx, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                            const g16_toxic_waste* tw, const uint64_t* g1_generator, const uint64_t* g2_generator, const g16_params_view* out) {
    if (!ctx || !abc || !tw || !g1_generator || !g2_generator || !out) return G16_ERR_BAD_ARG;
    if (!out->alpha_g1 || !out->beta_g1 || !out->delta_g1 || !out->beta_g2 || !out->delta_g2 || !out->gamma_g2) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty())
        return g16_generate_parameters(ctx->subs[0], abc, num_inputs, num_constraints, num_variables, tw, g1_generator, g2_generator, out);
    G16_HIP_TRY(hipSetDevice(ctx->device));
    if (ctx->prep.valid) (void)hipStreamSynchronize(ctx->stream2);   // a dropped prepared sort still owns arena buffers
    ctx->prep.valid = false;   // the generator resets the arena
    try {
        if (ctx->curve == G16_BLS12_381)
            return generate_parameters_device<Bls12_381>(ctx->stream, ctx->arena, abc, num_inputs, num_constraints, num_variables, tw,
                                                         g1_generator, g2_generator, out);
        if (ctx->curve == G16_BN254)
            return generate_parameters_device<Bn254>(ctx->stream, ctx->arena, abc, num_inputs, num_constraints, num_variables, tw,
                                                     g1_generator, g2_generator, out);
    } catch (const std::bad_alloc&) {
        return G16_ERR_OOM;
    }
    return G16_ERR_BAD_ARG;
}

int g16_host_qap_evaluations(int curve, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                             const uint64_t t[4], uint64_t* a_out, uint64_t* b_out, uint64_t* c_out, uint64_t zt_out[4]) {
    if (!abc || !t || !a_out || !b_out || !c_out || !zt_out) return G16_ERR_BAD_ARG;
    try {
        if (curve == G16_BLS12_381)
            return qap_evaluations_host<Bls12_381>(abc, num_inputs, num_constraints, num_variables, t, a_out, b_out, c_out, zt_out);
        if (curve == G16_BN254) return qap_evaluations_host<Bn254>(abc, num_inputs, num_constraints, num_variables, t, a_out, b_out, c_out, zt_out);
    } catch (const std::bad_alloc&) {
        return G16_ERR_OOM;
    }
    return G16_ERR_BAD_ARG;
}

uint64_t g16_serialized_point_size(int curve, int g2, int compressed) { return g16::serialized_point_size(curve, g2, compressed); }

int g16_serialize_points(int curve, int g2, int compressed, const uint64_t* points, uint64_t n, uint8_t* out) {
    if ((!points || !out) && n) return G16_ERR_BAD_ARG;
    return g16::serialize_points(curve, g2, compressed, points, n, out);
}

int g16_deserialize_points(int curve, int g2, int compressed, const uint8_t* in, uint64_t n, int validate, uint64_t* points_out) {
    if ((!in || !points_out) && n) return G16_ERR_BAD_ARG;
    if (validate < 0 || validate > 2) return G16_ERR_BAD_ARG;
    return g16::deserialize_points(curve, g2, compressed, in, n, validate, points_out);
}

const char* g16_strerror(int status) {
    switch (status) {
        case G16_OK: return "ok";
        case G16_ERR_DEGREE_TOO_LARGE: return "polynomial degree too large for the scalar field's 2-adicity";
        case G16_ERR_BAD_LENGTH: return "assignment / proving key / matrix length mismatch";
        case G16_ERR_BAD_ARG: return "bad argument";
        case G16_ERR_HIP: return "HIP runtime error (see g16_last_error)";
        case G16_ERR_OOM: return "out of memory";
        case G16_ERR_NO_DEVICE: return "no HIP device";
        case G16_ERR_INTERNAL: return "internal error";
        case G16_ERR_UNEXPECTED_IDENTITY: return "unexpected identity: gamma or delta is zero";
        case G16_ERR_INVALID_DATA: return "invalid data: the bytes do not encode a point of the group";
        case G16_ERR_NO_PEER_ACCESS: return "two devices of a multi-device context have no peer access (G16_MULTI_REQUIRE_PEER)";
        default: return "unknown status";
    }
}

const char* g16_last_error(void) { return g_last_error.c_str(); }
const char* g16_version(void) { return "g16_mi355x 0.2 (gfx950)"; }
int g16_abi_version(void) { return G16_ABI_VERSION; }
uint64_t g16_struct_size(int which) {
    switch (which) {
        case G16_STRUCT_TIMINGS: return sizeof(g16_timings);
        case G16_STRUCT_PK_INFO: return sizeof(g16_pk_info);
        case G16_STRUCT_DIAG: return sizeof(g16_diag);
        case G16_STRUCT_PROOF: return sizeof(g16_proof);
        case G16_STRUCT_PARTIAL: return sizeof(g16_partial);
        case G16_STRUCT_PK_VIEW: return sizeof(g16_pk_view);
        default: return 0;
    }
}
int g16_get_timings_sized(g16_ctx* ctx, void* out, uint64_t size) {
    if (!ctx || !out) return G16_ERR_BAD_ARG;
    memcpy(out, &ctx->tm, (size_t)std::min<uint64_t>(size, sizeof(g16_timings)));
    return G16_OK;
}
int g16_pk_get_info_sized(const g16_pk* pk, void* out, uint64_t size) {
    if (!out) return G16_ERR_BAD_ARG;
    g16_pk_info full;
    G16_TRY(g16_pk_get_info(pk, &full));
    memcpy(out, &full, (size_t)std::min<uint64_t>(size, sizeof(full)));
    return G16_OK;
}

}  // extern "C"
