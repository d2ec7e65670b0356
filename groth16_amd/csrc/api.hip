// C ABI of libg16_mi355x.so (declared in include/g16_mi355x.h): contexts, device-resident proving
// keys and circuits, the prover orchestration and the unit-level entry points.
//
// Orchestration restates Groth16::create_proof_with_reduction_and_matrices
// (/root/reference/src/prover.rs:26-51) and create_proof_with_assignment (:54-132): the witness
// map and the five MSMs run on the GPU; the O(1) glue of :76-131 (six scalar multiples, a
// handful of additions, three into_affine) runs on the host with the same field code.
#include "internal.hpp"
#include "msm_common.hpp"
#include "fp30.hpp"
#include "fixed_base.hpp"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <future>
#include <mutex>
#include <thread>
#include <new>
#include <type_traits>

namespace g16 {

static thread_local std::string g_last_error;

void set_last_error(const char* what, hipError_t e, const char* file, int line) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s:%d: %s -> %s", file, line, what, hipGetErrorString(e));
    g_last_error = buf;
}

int Arena::alloc(size_t bytes, void** out) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (bytes == 0) bytes = 256;
    for (auto& c : chunks) {
        if (c.cap - c.used >= bytes) {
            *out = c.p + c.used;
            c.used += bytes;
            return G16_OK;
        }
    }
    Chunk c;
    c.cap = bytes > min_chunk ? bytes : min_chunk;
    c.used = 0;
    c.p = nullptr;
    G16_HIP_TRY(hipMalloc((void**)&c.p, c.cap));
    c.used = bytes;
    *out = c.p;
    chunks.push_back(c);
    return G16_OK;
}
void Arena::release() {
    for (auto& c : chunks) (void)hipFree(c.p);
    chunks.clear();
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace g16

using namespace g16;

struct g16_ctx {
    int curve;
    int device;
    hipStream_t stream;   // witness map, the five bucket passes back to back, the batched G1 reduction
    hipStream_t stream2;  // witness digit/sort pass beside the witness map; the G2 reduction of a whole-key proof
    hipStream_t stream3;  // h's digit/sort pass, underneath the first bucket pass
    hipStream_t stream_wm = nullptr;   // g16_dwm_stage_async: the distributed witness map's stages (and the caller's exchanges between them)
    hipEvent_t ev_dwm = nullptr;
    hipEvent_t ev_heavy[4] = {};   // G1 MSM k's heavy-bucket combine (side stream) done
    hipEvent_t ev_edge[8] = {};    // timestamps on stream 1 at the boundaries of the bucket passes (see prove_partial)
    // g16_prove_partial_prepare: the witness digit/sort pass of the NEXT g16_prove_partial[_h] over (pk, z), already enqueued on
    // stream 2 (its buffers live in the arena, which that call then must not reset)
    struct Prepared {
        bool valid = false;
        const g16_pk* pk = nullptr;
        const uint64_t* z = nullptr;
        uint64_t n_assign = 0;
        ScalarSort sort_z;
    } prep;
    // a prepared sort that is being DROPPED (the next call is not the prove_partial it was made for, or a failed call left it
    // behind) may still be running on stream 2 inside arena buffers: wait for it before the arena is handed out again
    void reset_arena() {
        if (prep.valid) (void)hipStreamSynchronize(stream2);
        arena.reset();
        prep.valid = false;
    }
    hipStream_t red[5];   // per MSM: the heavy-bucket combine of G1 MSM k right after its pass ([0..3]); the G2 reduction of a sharded
                          // proof ([4]: short passes, it must not queue behind the witness sort on stream 2)
    Arena arena;
    g16_timings tm;
    EventTimer t_wm, t_prep_h, t_prep_z, t_bucket[5], t_ntt[2];
    hipEvent_t ev_z = nullptr, ev_h = nullptr, ev_wm = nullptr, ev_done[5] = {};
    // g16_prove_finalize_prepare: the (r, s)-only half of the host glue, running on a host thread (api.hip: FinalizePrep)
    struct FinPrep {
        bool valid = false;
        std::future<void> fut;
        std::shared_ptr<void> data;
        const g16_pk* pk = nullptr;
        uint64_t r[4] = {}, s[4] = {};
        bool matches(const g16_pk* p, const uint64_t* r_, const uint64_t* s_) const {
            return valid && pk == p && memcmp(r, r_, 32) == 0 && memcmp(s, s_, 32) == 0;
        }
        void drop() {   // wait for a running thread (it reads the key) and forget its result
            if (valid && fut.valid()) fut.wait();
            valid = false;
            data.reset();
            pk = nullptr;
        }
    } finprep;
    void* pinned = nullptr;  // window sums land here (hipHostMalloc)
    size_t pinned_bytes = 0;
    // g16_ctx_create_multi: a multi-device context owns one full context per device and no device state of its own
    std::vector<g16_ctx*> subs;
};

// Error exits of the entry points that launch on several streams: kernels still in flight reference arena memory that the
// next call resets and reuses, so an early return first drains every stream of the ctx.
struct DrainOnError {
    g16_ctx* ctx;
    bool armed = true;
    explicit DrainOnError(g16_ctx* c) : ctx(c) {}
    void dismiss() { armed = false; }
    ~DrainOnError() {
        if (!armed) return;
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream3);
        (void)hipStreamSynchronize(ctx->stream_wm);
        for (int i = 0; i < 5; ++i) (void)hipStreamSynchronize(ctx->red[i]);
    }
};

struct g16_dwm;
// multi-device context, device i: its side of the distributed witness map (g16_prove runs the four stages on every device's
// host thread and moves the chunks between the devices with peer copies -- the single-process form of the all-to-all)
struct DwmSlot {
    g16_dwm* dwm = nullptr;
    uint64_t *work[3] = {nullptr, nullptr, nullptr}, *recv[3] = {nullptr, nullptr, nullptr}, *h_local = nullptr;   // M Fr each
    uint64_t* z_dev = nullptr;   // num_variables Fr: a host assignment is uploaded once per proof and device
};

struct g16_circuit {
    int curve;
    g16_ctx* ctx;
    void* dc;  // DeviceCircuit<C>*
    uint64_t domain_size;
    std::vector<g16_circuit*> subs;   // multi-device context: the circuit replicated on every device (dc == nullptr)
    std::vector<DwmSlot> dist;        // multi-device context whose device count admits the distributed witness map
    uint64_t num_variables = 0;
};

template <class C>
struct DevicePk {
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    G1A alpha_g1, beta_g1, delta_g1, a_query0, b_g1_query0;
    G2A beta_g2, delta_g2, b_g2_query0;
    G1A *a = nullptr, *b_g1 = nullptr, *h = nullptr, *l = nullptr;
    G2A* b_g2 = nullptr;
    uint64_t a_start = 0, a_count = 0, b_g1_start = 0, b_g1_count = 0, b_g2_start = 0, b_g2_count = 0;
    uint64_t h_start = 0, h_count = 0, l_start = 0, l_count = 0;
    // window size of the precomputed window tables (msm.hip, merged windows): every query array then holds W rows of
    // `count` points, row j = 2^(cj) * query.  0 = no tables (plain bases, per-window buckets).  a, b_g1, b_g2 and l share
    // the witness sort and therefore one window size; h has its own.
    int c_z = 0, c_h = 0;
    // host: multiples of delta_g1 / delta_g2 for the glue of every proof over this key (fixed_base.hpp).  Built by the SECOND
    // finalize over the key (~35 ms of host work once, ~0.5 ms saved per later proof): a key that proves once never pays.
    mutable FixedBaseTable<typename C::G1X> delta1_tab;
    mutable FixedBaseTable<typename C::G2X> delta2_tab;
    mutable std::mutex tab_mu;
    mutable int finalize_calls = 0;
};

struct g16_pk {
    int curve;
    g16_ctx* ctx;
    void* dp;  // DevicePk<C>*
    std::vector<g16_pk*> subs;        // multi-device context: shard i of the key on device i (dp == nullptr)
    uint64_t dist_n = 0;              // != 0: the h shards are gathered in the block order of the distributed witness map over a
                                      // domain of dist_n points (h_query holds dist_n - 1 bases, generator.rs:168)
};

struct g16_dwm {
    int curve;
    g16_ctx* ctx;
    const g16_circuit* circuit;
    void* dw;   // DistWm<C>*
    int rank, world;
    uint64_t local_size;
};

namespace {

template <class T>
T load_pod(const uint64_t* p) {
    T t;
    memcpy(&t, p, sizeof(T));
    return t;
}

template <class C>
struct Impl {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    typedef typename C::G1X G1X;
    typedef typename C::G2X G2X;
    static constexpr int L = Fq::N / 2;  // 64-bit limbs per Fq

    // ---------------------------------------------------------------------------------------
    // one query array of the key on the device, ready for the bucket kernel: with c == 0 the bases themselves (converted
    // to the kernel's radix), else the W-row window table built from them; the caller's buffer is never modified
    // load-time scratch shared by the five queries of one key: the window-table builders' parking buffer and the staged copies of
    // host-side bases.  Nothing in load_query waits for the GPU; pk_load synchronises once and releases these.
    struct LoadScratch {
        void* park = nullptr;
        size_t park_bytes = 0;
        struct Staged { void* p; hipEvent_t built; };   // a staged host query and the event behind the build that reads it
        std::vector<Staged> staged;
        // free the staged copies whose builds have finished (wait: all of them)
        void reap(bool wait) {
            size_t keep = 0;
            for (Staged& st : staged) {
                if (wait ? ((void)hipEventSynchronize(st.built), true) : hipEventQuery(st.built) == hipSuccess) {
                    (void)hipFree(st.p);
                    (void)hipEventDestroy(st.built);
                } else {
                    staged[keep++] = st;
                }
            }
            staged.resize(keep);
        }
        void release() {
            reap(true);
            (void)hipFree(park);
            park = nullptr;
            park_bytes = 0;
        }
        // hipMalloc; if it fails, the staged copies of the queries already queued are given back (after their builds) and the
        // allocation is tried once more -- near the capacity limit a load then still gets its window tables instead of silently
        // falling back to plain bases (a slower prover).  No frees on the normal path: hipFree synchronises the device, and the
        // point of queueing the builds is that the host allocates the next table meanwhile.
        int alloc(void** out, size_t bytes) {
            if (hipMalloc(out, bytes) == hipSuccess) return G16_OK;
            (void)hipGetLastError();
            if (staged.empty()) return G16_ERR_OOM;
            reap(true);
            return hipMalloc(out, bytes) == hipSuccess ? G16_OK : G16_ERR_OOM;
        }
    };

    template <class F>
    static int load_query(g16_ctx* ctx, const g16_query& q, bool dev_ptrs, int c, Affine<F>** out, LoadScratch& ls) {
        typedef Affine<F> P;
        *out = nullptr;
        if (q.count == 0) return G16_OK;
        if (!q.points) return G16_ERR_BAD_ARG;
        if (c == 0) {
            if (hipMalloc((void**)out, q.count * sizeof(P)) != hipSuccess) return G16_ERR_OOM;
            G16_HIP_TRY(hipMemcpyAsync(*out, q.points, q.count * sizeof(P), dev_ptrs ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                                       ctx->stream));
            return convert_bases<F>(*out, q.count, ctx->stream);
        }
        uint32_t modw[Fr::N];
        for (int i = 0; i < Fr::N; ++i) modw[i] = Fr::Params::mod(i);
        const int W = msm_plan_windows(c, Fr::Params::BITS, modw, Fr::N);
        if (W <= 0) return G16_ERR_INTERNAL;
        // G16_PK_TABLE_BUDGET_MB caps one query's table (default: whatever hipMalloc grants); past it the key is held plain
        const char* cap = getenv("G16_PK_TABLE_BUDGET_MB");
        if (cap && (double)q.count * sizeof(P) * W > atof(cap) * 1048576.0) return G16_ERR_OOM;
        if (ls.alloc((void**)out, q.count * sizeof(P) * (size_t)W) != G16_OK) return G16_ERR_OOM;
        const P* src = reinterpret_cast<const P*>(q.points);
        P* staged = nullptr;
        if (!dev_ptrs) {
            if (ls.alloc((void**)&staged, q.count * sizeof(P)) != G16_OK) return G16_ERR_OOM;
            G16_HIP_TRY(hipMemcpyAsync(staged, q.points, q.count * sizeof(P), hipMemcpyHostToDevice, ctx->stream));
            src = staged;
        }
        // the parking buffer is used by one build at a time (they are queued on one stream); it only ever grows
        const size_t need = window_table_park_bytes<F>(q.count, W);
        if (need > ls.park_bytes) {
            if (ls.park) {   // an earlier build may still be using the smaller one
                G16_HIP_TRY(hipStreamSynchronize(ctx->stream));
                (void)hipFree(ls.park);
                ls.park = nullptr;
                ls.park_bytes = 0;
            }
            if (ls.alloc(&ls.park, need) != G16_OK) { if (staged) (void)hipFree(staged); return G16_ERR_OOM; }
            ls.park_bytes = need;
        }
        const int rc = build_window_tables<F>(src, q.count, c, W, *out, ctx->stream, ls.park);
        if (staged) {   // freed as soon as the build behind this event is done (LoadScratch::reap)
            hipEvent_t ev = nullptr;
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipEventRecord(ev, ctx->stream) != hipSuccess) {
                (void)hipStreamSynchronize(ctx->stream);
                (void)hipFree(staged);
                if (ev) (void)hipEventDestroy(ev);
            } else {
                ls.staged.push_back({staged, ev});
            }
        }
        return rc;
    }

    static void pk_free(DevicePk<C>* p) {
        if (!p) return;
        (void)hipFree(p->a); (void)hipFree(p->b_g1); (void)hipFree(p->b_g2); (void)hipFree(p->h); (void)hipFree(p->l);
        delete p;
    }

    static int pk_load(g16_ctx* ctx, const g16_pk_view* v, g16_pk** out) {
        if (!v->alpha_g1 || !v->beta_g1 || !v->delta_g1 || !v->beta_g2 || !v->delta_g2 || !v->a_query0 || !v->b_g1_query0 ||
            !v->b_g2_query0)
            return G16_ERR_BAD_ARG;
        DevicePk<C>* p = new (std::nothrow) DevicePk<C>();
        if (!p) return G16_ERR_OOM;
        p->alpha_g1 = load_pod<G1A>(v->alpha_g1);
        p->beta_g1 = load_pod<G1A>(v->beta_g1);
        p->delta_g1 = load_pod<G1A>(v->delta_g1);
        p->beta_g2 = load_pod<G2A>(v->beta_g2);
        p->delta_g2 = load_pod<G2A>(v->delta_g2);
        p->a_query0 = load_pod<G1A>(v->a_query0);
        p->b_g1_query0 = load_pod<G1A>(v->b_g1_query0);
        p->b_g2_query0 = load_pod<G2A>(v->b_g2_query0);
        const bool dev = (v->flags & G16_PK_DEVICE_PTRS) != 0;
        int rc = G16_OK;
        // window tables (merged windows, msm.hip): a, b_g1, b_g2 and l share the witness sort, hence one window size
        uint32_t modw[Fr::N];
        for (int i = 0; i < Fr::N; ++i) modw[i] = Fr::Params::mod(i);
        const uint64_t nz = std::max(std::max(v->a.count, v->b_g1.count), std::max(v->b_g2.count, v->l.count));
        // the one size limit of the MSM path, enforced where the key arrives instead of at the first proof: entry lists are
        // indexed with 32 bits, a per-window plan has at most 17 windows (c = 16), so a shard of 2^27 points or more of any
        // query could not be sorted (sort_scalars: n * W < 2^32).  Shard the key further (or over more GPUs) instead.
        if (std::max(nz, (uint64_t)v->h.count) >= ((uint64_t)1 << 27)) {
            delete p;
            g_last_error = "a proving-key shard holds 2^27 or more points of one query: shard it over more ranks";
            return G16_ERR_BAD_LENGTH;
        }
        p->c_z = merged_window_bits(nz, Fr::Params::BITS, modw, Fr::N);
        p->c_h = merged_window_bits(v->h.count, Fr::Params::BITS, modw, Fr::N);
        if (v->h.count == 0) p->c_h = p->c_z;
        if (nz == 0) p->c_z = p->c_h;
        if (p->c_z == 0 || p->c_h == 0) p->c_z = p->c_h = 0;   // tables for all queries or for none
        LoadScratch ls;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int cz = p->c_z, ch = p->c_h;
            // the G2 table first: it needs the largest parking buffer
            if ((rc = load_query<Fq2>(ctx, v->b_g2, dev, cz, &p->b_g2, ls)) || (rc = load_query<Fq>(ctx, v->a, dev, cz, &p->a, ls)) ||
                (rc = load_query<Fq>(ctx, v->b_g1, dev, cz, &p->b_g1, ls)) || (rc = load_query<Fq>(ctx, v->l, dev, cz, &p->l, ls)) ||
                (rc = load_query<Fq>(ctx, v->h, dev, ch, &p->h, ls))) {
                (void)hipStreamSynchronize(ctx->stream);
                ls.release();
                (void)hipFree(p->a); (void)hipFree(p->b_g1); (void)hipFree(p->b_g2); (void)hipFree(p->h); (void)hipFree(p->l);
                p->a = p->b_g1 = p->h = p->l = nullptr;
                p->b_g2 = nullptr;
                if (rc == G16_ERR_OOM && cz != 0) {   // the tables do not fit next to what already lives on this GPU: plain bases
                    (void)hipGetLastError();
                    p->c_z = p->c_h = 0;
                    continue;
                }
                pk_free(p);
                return rc;
            }
            break;
        }
        p->a_start = v->a.start; p->a_count = v->a.count;
        p->b_g1_start = v->b_g1.start; p->b_g1_count = v->b_g1.count;
        p->b_g2_start = v->b_g2.start; p->b_g2_count = v->b_g2.count;
        p->h_start = v->h.start; p->h_count = v->h.count;
        p->l_start = v->l.start; p->l_count = v->l.count;
        const bool load_ok = hipStreamSynchronize(ctx->stream) == hipSuccess;   // every table is built: the load-time scratch can go
        ls.release();
        if (!load_ok) { pk_free(p); return G16_ERR_HIP; }
        g16_pk* h = new (std::nothrow) g16_pk{C::CURVE_ID, ctx, p};
        if (!h) { pk_free(p); return G16_ERR_OOM; }
        *out = h;
        return G16_OK;
    }

    // ---------------------------------------------------------------------------------------
    static void circuit_free(DeviceCircuit<C>* dc) {
        if (!dc) return;
        for (int m = 0; m < 3; ++m) { (void)hipFree(dc->row_ptr[m]); (void)hipFree(dc->col[m]); (void)hipFree(dc->val[m]); }
        domain_destroy<C>(dc->dom);
        delete dc;
    }

    static int circuit_load(g16_ctx* ctx, const g16_csr_view abc[3], uint64_t num_inputs, uint64_t num_constraints, uint64_t num_variables,
                            g16_circuit** out) {
        if (num_inputs == 0 || num_variables < num_inputs) return G16_ERR_BAD_LENGTH;
        // D::new(num_constraints + num_inputs), r1cs_to_qap.rs:178-179
        const uint64_t need = num_constraints + num_inputs;
        int log_n = 0;
        while (((uint64_t)1 << log_n) < need) {
            ++log_n;
            if (log_n > 40) return G16_ERR_DEGREE_TOO_LARGE;
        }
        if (log_n > C::TWO_ADICITY) return G16_ERR_DEGREE_TOO_LARGE;
        if (log_n > 30) return G16_ERR_DEGREE_TOO_LARGE;  // 32-bit indices inside the kernels
        DeviceCircuit<C>* dc = new (std::nothrow) DeviceCircuit<C>();
        if (!dc) return G16_ERR_OOM;
        dc->num_inputs = num_inputs;
        dc->num_constraints = num_constraints;
        dc->num_variables = num_variables;
        auto fail = [&](int code) { circuit_free(dc); return code; };
        for (int m = 0; m < 3; ++m) {
            if (!abc[m].row_ptr) return fail(G16_ERR_BAD_ARG);
            // a malformed CSR would send spmv3_kernel out of bounds on the device: row_ptr must start at 0 and never decrease
            if (abc[m].row_ptr[0] != 0) return fail(G16_ERR_BAD_LENGTH);
            for (uint64_t i = 0; i < num_constraints; ++i)
                if (abc[m].row_ptr[i] > abc[m].row_ptr[i + 1]) return fail(G16_ERR_BAD_LENGTH);
            const uint64_t nnz = abc[m].row_ptr[num_constraints];
            dc->nnz[m] = nnz;
            if (nnz && (!abc[m].col || !abc[m].val)) return fail(G16_ERR_BAD_ARG);
            for (uint64_t k = 0; k < nnz; ++k)
                if (abc[m].col[k] >= num_variables) return fail(G16_ERR_BAD_LENGTH);
            if (hipMalloc((void**)&dc->row_ptr[m], (num_constraints + 1) * sizeof(uint64_t)) != hipSuccess) return fail(G16_ERR_OOM);
            if (hipMalloc((void**)&dc->col[m], (nnz ? nnz : 1) * sizeof(uint32_t)) != hipSuccess) return fail(G16_ERR_OOM);
            if (hipMalloc((void**)&dc->val[m], (nnz ? nnz : 1) * sizeof(Fr)) != hipSuccess) return fail(G16_ERR_OOM);
            if (hipMemcpyAsync(dc->row_ptr[m], abc[m].row_ptr, (num_constraints + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream) !=
                hipSuccess)
                return fail(G16_ERR_HIP);
            if (nnz) {
                if (hipMemcpyAsync(dc->col[m], abc[m].col, nnz * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    return fail(G16_ERR_HIP);
                if (hipMemcpyAsync(dc->val[m], abc[m].val, nnz * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
                    return fail(G16_ERR_HIP);
            }
        }
        int rc = domain_create<C>(log_n, ctx->stream, &dc->dom);
        if (rc) return fail(rc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(G16_ERR_HIP);
        g16_circuit* h = new (std::nothrow) g16_circuit{C::CURVE_ID, ctx, dc, (uint64_t)1 << log_n};
        if (!h) return fail(G16_ERR_OOM);
        *out = h;
        return G16_OK;
    }

    // ---------------------------------------------------------------------------------------
    static int stage_assignment(g16_ctx* ctx, const uint64_t* z, uint64_t n_assign, int on_device, const Fr** d_z) {
        if (on_device) { *d_z = reinterpret_cast<const Fr*>(z); return G16_OK; }
        Fr* buf = nullptr;
        G16_TRY(ctx->arena.alloc_n(n_assign, &buf));
        G16_HIP_TRY(hipMemcpyAsync(buf, z, n_assign * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        *d_z = buf;
        return G16_OK;
    }

    template <class X>
    static void store_xyzz(uint64_t* dst, const X& p) { memcpy(dst, &p, sizeof(X)); }
    template <class X>
    static X load_xyzz(const uint64_t* src) { X p; memcpy(&p, src, sizeof(X)); return p; }

    // the witness digit/sort pass of the next prove_partial over (pk shard, device assignment), enqueued NOW on stream 2: a caller that
    // runs the distributed witness map first (g16_dwm_stage_async + exchanges) calls this before it, so that the sort's dozen launches
    // are in the queues ahead of the map's forty and the two run side by side from t = 0
    static int prove_partial_prepare(g16_ctx* ctx, const g16_pk* pkh, const g16_circuit* ckh, const uint64_t* z_dev, uint64_t n_assign) {
        const DevicePk<C>* pk = static_cast<const DevicePk<C>*>(pkh->dp);
        const DeviceCircuit<C>* ck = static_cast<const DeviceCircuit<C>*>(ckh->dc);
        if (n_assign != ck->num_variables) return G16_ERR_BAD_LENGTH;
        if (pk->a_start + pk->a_count > n_assign - 1) return G16_ERR_BAD_LENGTH;
        DrainOnError drain(ctx);
        ctx->reset_arena();
        const Fr* d_z = reinterpret_cast<const Fr*>(z_dev);
        ScalarSort ss;
        G16_TRY((sort_scalars<C>(d_z + 1 + pk->a_start, pk->a_count, pk->c_z, ctx->arena, ctx->stream2, &ss)));
        G16_HIP_TRY(hipEventRecord(ctx->ev_z, ctx->stream2));
        ctx->prep.valid = true;
        ctx->prep.pk = pkh;
        ctx->prep.z = z_dev;
        ctx->prep.n_assign = n_assign;
        ctx->prep.sort_z = ss;
        drain.dismiss();
        return G16_OK;
    }

    // h_ext != nullptr: the witness map was computed elsewhere (the distributed map: this rank's block of h, h_ext_len
    // coefficients in the order the key's h shard was gathered in) -- it is taken as is and the map below is skipped
    static int prove_partial(g16_ctx* ctx, const g16_pk* pkh, const g16_circuit* ckh, const uint64_t* z, uint64_t n_assign, int on_device,
                             int skip_b_g1, g16_partial* out, const Fr* h_ext = nullptr, uint64_t h_ext_len = 0) {
        const DevicePk<C>* pk = static_cast<const DevicePk<C>*>(pkh->dp);
        const DeviceCircuit<C>* ck = static_cast<const DeviceCircuit<C>*>(ckh->dc);
        hipStream_t s1 = ctx->stream, s2 = ctx->stream2, s3 = ctx->stream3;
        if (n_assign != ck->num_variables) return G16_ERR_BAD_LENGTH;
        const uint64_t n = ck->dom->n, nin = ck->num_inputs;
        const uint64_t m = n_assign - 1, w = n_assign - nin;
        // the reference slices full_assignment[1..], [num_inputs..] (prover.rs:44-45) and msm_bigint
        // truncates to the shorter side; a shard must lie inside the scalar vector it indexes
        if (pk->a_start + pk->a_count > m || pk->l_start + pk->l_count > w || pk->h_start + pk->h_count > (h_ext ? h_ext_len : n))
            return G16_ERR_BAD_LENGTH;
        if (pk->b_g1_start != pk->a_start || pk->b_g1_count != pk->a_count || pk->b_g2_start != pk->a_start ||
            pk->b_g2_count != pk->a_count)
            return G16_ERR_BAD_ARG;  // a / b_g1 / b_g2 must be sharded identically (they share one bucket sort)
        memset(out, 0, sizeof(*out));
        const double t_begin = now_ms();
        DrainOnError drain(ctx);
        // g16_prove_partial_prepare ran for exactly this (key shard, device assignment): its sort is on stream 2 already
        const bool prepared = ctx->prep.valid && on_device && ctx->prep.pk == pkh && ctx->prep.z == z && ctx->prep.n_assign == n_assign;
        const ScalarSort prepared_sort = ctx->prep.sort_z;
        if (prepared) ctx->prep.valid = false;   // consumed: the arena keeps its contents for this call
        else ctx->reset_arena();
        const Fr* d_z = nullptr;
        G16_TRY(stage_assignment(ctx, z, n_assign, on_device, &d_z));
        if (!prepared) G16_HIP_TRY(hipEventRecord(ctx->ev_z, s1));

        // ---- witness map, h = QAP::witness_map_from_matrices (prover.rs:37-42); only the h MSM needs it.  It goes FIRST,
        // alone, on stream 1 (~6 ms at 2^22): underneath the bucket passes their long-lived waves starve it (60+ ms
        // measured) and everything queued behind it piles up at the end of the proof.
        Fr* d_h = nullptr;
        ScalarSort sort_h, sort_z, sort_l;
        G16_TRY(ctx->t_wm.start(s1));
        if (h_ext) {
            // h comes from the distributed map: whatever g16_dwm_stage_async (and the caller's exchanges) enqueued on the
            // witness-map stream must have finished before h is read.  The witness sort (stream 2) does NOT wait for it and runs
            // beside the map's stages and exchanges.  The bucket passes DO wait: the map is a chain of ~25 short dependent
            // kernels and 7 exchanges, and a kernel launched while a bucket pass holds every wave slot starts only when that
            // pass's workgroups retire -- underneath back-to-back passes each link of the chain would cost a whole pass (the
            // starvation measured on the single-GPU schedule: 60 ms for a 6 ms map).  So: map and sort side by side, then passes.
            d_h = const_cast<Fr*>(h_ext);
            ctx->t_ntt[0].used = ctx->t_ntt[1].used = false;
            G16_HIP_TRY(hipEventRecord(ctx->ev_dwm, ctx->stream_wm));
            G16_HIP_TRY(hipStreamWaitEvent(s3, ctx->ev_dwm, 0));
            G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_dwm, 0));
        } else {
            G16_TRY(ctx->arena.alloc_n(n, &d_h));
            G16_TRY((witness_map_device<C>(ck, d_z, d_h, ctx->arena, s1, ctx->t_ntt)));
        }
        G16_TRY(ctx->t_wm.stop(s1));
        G16_HIP_TRY(hipEventRecord(ctx->ev_wm, s1));

        // ---- stream 2, beside the witness map: assignment = full_assignment[1..] (prover.rs:80-85), ONE digit/sort
        // pass for a, b_g1, b_g2 (and l)
        if (prepared) {
            sort_z = prepared_sort;                      // ev_z was recorded on stream 2 behind the sort by the prepare call
            ctx->t_prep_z.used = false;
        } else {
            G16_HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_z, 0));
            G16_TRY(ctx->t_prep_z.start(s2));
            G16_TRY((sort_scalars<C>(d_z + 1 + pk->a_start, pk->a_count, pk->c_z, ctx->arena, s2, &sort_z)));
            G16_TRY(ctx->t_prep_z.stop(s2));
            G16_HIP_TRY(hipEventRecord(ctx->ev_z, s2));   // (re-recorded: now also covers the witness sort)
        }
        G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_z, 0));
        // ---- stream 3: h's digit/sort pass, underneath the first bucket pass (stream 2 stays free for the reductions)
        G16_HIP_TRY(hipStreamWaitEvent(s3, ctx->ev_wm, 0));
        G16_TRY(ctx->t_prep_h.start(s3));
        G16_TRY((sort_scalars<C>(d_h + pk->h_start, pk->h_count, pk->c_h, ctx->arena, s3, &sort_h)));
        G16_TRY(ctx->t_prep_h.stop(s3));
        G16_HIP_TRY(hipEventRecord(ctx->ev_h, s3));

        MsmBuffers<Fq> buf_h, buf_l, buf_a, buf_b1;
        MsmBuffers<Fq2> buf_b2;
        // window sums of MSM k land in the pinned host buffer; slot layout by MSM order (0 h, 1 l, 2 a, 3 b_g1, 4 b_g2)
        const size_t SLOT = MSM_MAX_OUTPUTS * sizeof(G2X);  // plan.outputs() <= 224 for every admissible plan
        if (ctx->pinned_bytes < 5 * SLOT) {
            if (ctx->pinned) (void)hipHostFree(ctx->pinned);
            G16_HIP_TRY(hipHostMalloc(&ctx->pinned, 5 * SLOT, hipHostMallocDefault));
            ctx->pinned_bytes = 5 * SLOT;
        }
        char* pin = static_cast<char*>(ctx->pinned);
        if (sort_z.plan.outputs() > MSM_MAX_OUTPUTS || sort_h.plan.outputs() > MSM_MAX_OUTPUTS) return G16_ERR_INTERNAL;
        // Bucket passes back to back on stream 1.  The G2 MSM goes first and its reduction (the longest chain: ~3x a G1 one) runs on
        // its own stream underneath the G1 passes; the four G1 reductions are NOT started one by one underneath the following
        // pass -- a reduction is a few hundred waves of dependent additions that hold register slots for milliseconds and slowed
        // every pass they ran under (8-way shard at 2^22: 1.8-2.1 ms per pass instead of 1.2) -- but run TOGETHER, one launch per
        // stage for all of them (msm_reduce_batch), after the last pass: 4x the waves per launch, one chain of latency instead of four.
        const bool short_passes = (uint64_t)pk->a_count * (uint64_t)sort_z.plan.W < 20000000ull;
        // Timestamps: ONE event per boundary between back-to-back passes (the end of pass k is the begin of pass k + 1) instead of a
        // start / stop / done / span-start record around every pass -- each record is a barrier packet the command processor retires
        // before it starts the next kernel, and the four of them cost ~0.13 ms of idle GPU between two passes (kernel trace of round 3).
        hipEvent_t pass_begin[5] = {}, pass_end[5] = {}, last_end = nullptr;
        int n_edge = 0;
        auto mark = [&](hipEvent_t* ev) -> int {
            if (n_edge >= 8) return G16_ERR_INTERNAL;
            *ev = ctx->ev_edge[n_edge++];
            G16_HIP_TRY(hipEventRecord(*ev, s1));
            return G16_OK;
        };
        auto run_pass = [&](int k, auto* bases, int64_t shift, uint64_t count, const ScalarSort& ss, auto* buf) -> int {
            if (last_end) pass_begin[k] = last_end;
            else G16_TRY(mark(&pass_begin[k]));
            G16_TRY((msm_bucket_pass(bases, shift, count, ss, ctx->arena, s1, buf, nullptr)));
            G16_TRY(mark(&pass_end[k]));
            last_end = pass_end[k];
            return G16_OK;
        };
        // a G1 MSM's heavy-bucket combine (buckets with many partial sums: the short top window's few hundred) goes underneath the
        // next pass on the MSM's own stream -- at most a few hundred workgroups -- so that the batched reduction starts at the bucket level
        auto heavy_early = [&](int k, const MsmBuffers<Fq>& buf, const ScalarSort& ss) -> int {
            G16_HIP_TRY(hipStreamWaitEvent(ctx->red[k], pass_end[k], 0));
            G16_TRY((msm_heavy_reduce<Fq>(buf, ss, ctx->red[k])));
            G16_HIP_TRY(hipEventRecord(ctx->ev_heavy[k], ctx->red[k]));
            return G16_OK;
        };
        auto copy_out = [&](int k, const auto& buf, const ScalarSort& ss, hipStream_t sr) -> int {
            G16_HIP_TRY(hipMemcpyAsync(pin + k * SLOT, buf.window_sums, sizeof(*buf.window_sums) * ss.plan.outputs(), hipMemcpyDeviceToHost, sr));
            G16_HIP_TRY(hipEventRecord(ctx->ev_done[k], sr));
            return G16_OK;
        };

        // l_aux_acc = msm(l_query, aux) (prover.rs:70-74); aux[j] = assignment[j + nin - 1]
        const bool l_covered = pk->l_count == 0 || (pk->l_start + nin - 1 >= pk->a_start &&
                                                    pk->l_start + pk->l_count + nin - 1 <= pk->a_start + pk->a_count);
        {
            hipStream_t sr = short_passes ? ctx->red[4] : s2;
            G16_TRY(run_pass(4, pk->b_g2, 0, pk->b_g2_count, sort_z, &buf_b2));                               // prover.rs:113
            G16_HIP_TRY(hipStreamWaitEvent(sr, pass_end[4], 0));
            G16_TRY((msm_reduce(buf_b2, sort_z, sr)));
            G16_TRY(copy_out(4, buf_b2, sort_z, sr));
        }
        struct G1Job { int k; MsmBuffers<Fq>* buf; const ScalarSort* ss; };
        G1Job jobs[4];
        int njobs = 0;
        if (l_covered) {
            const int64_t shift = (int64_t)pk->a_start - (int64_t)(nin - 1) - (int64_t)pk->l_start;
            G16_TRY(run_pass(1, pk->l, shift, pk->l_count, sort_z, &buf_l));
            G16_TRY(heavy_early(1, buf_l, sort_z));
            jobs[njobs++] = {1, &buf_l, &sort_z};
        } else {
            G16_TRY((sort_scalars<C>(d_z + nin + pk->l_start, pk->l_count, pk->c_z, ctx->arena, s1, &sort_l)));
            last_end = nullptr;   // the sort sits between the passes: this one gets its own begin mark
            G16_TRY(run_pass(1, pk->l, 0, pk->l_count, sort_l, &buf_l));
            G16_TRY(heavy_early(1, buf_l, sort_l));
            jobs[njobs++] = {1, &buf_l, &sort_l};
        }
        G16_TRY(run_pass(2, pk->a, 0, pk->a_count, sort_z, &buf_a));                                         // prover.rs:92
        G16_TRY(heavy_early(2, buf_a, sort_z));
        jobs[njobs++] = {2, &buf_a, &sort_z};
        if (!skip_b_g1) {                                                                                    // prover.rs:98-108
            G16_TRY(run_pass(3, pk->b_g1, 0, pk->b_g1_count, sort_z, &buf_b1));
            G16_TRY(heavy_early(3, buf_b1, sort_z));
            jobs[njobs++] = {3, &buf_b1, &sort_z};
        }
        // ---- h_acc = msm(h_query, h) (prover.rs:63-66): needs the witness map
        G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_h, 0));
        last_end = nullptr;       // whatever stream 1 waits for here is not the h pass
        G16_TRY(run_pass(0, pk->h, 0, pk->h_count, sort_h, &buf_h));
        G16_TRY(heavy_early(0, buf_h, sort_h));
        jobs[njobs++] = {0, &buf_h, &sort_h};
        // the G1 reductions, batched by bucket layout (h's window size may differ from the witness MSMs')
        bool done[4] = {false, false, false, false};
        for (int i = 0; i < njobs; ++i) {
            if (done[i]) continue;
            const MsmBuffers<Fq>* bb[4];
            const ScalarSort* sp[4];
            int idx[4], nb = 0;
            for (int q = i; q < njobs; ++q)
                if (!done[q] && jobs[q].ss->plan.B == jobs[i].ss->plan.B && jobs[q].ss->plan.groups == jobs[i].ss->plan.groups) {
                    bb[nb] = jobs[q].buf; sp[nb] = jobs[q].ss; idx[nb] = q; ++nb;
                    done[q] = true;
                }
            for (int q = 0; q < nb; ++q) G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_heavy[jobs[idx[q]].k], 0));
            G16_TRY((msm_reduce_batch<Fq>(bb, sp, nb, s1, /*heavy_done=*/true)));
            for (int q = 0; q < nb; ++q) G16_TRY(copy_out(jobs[idx[q]].k, *jobs[idx[q]].buf, *jobs[idx[q]].ss, s1));
        }

        // ---- host: fold the group sums of each MSM (merged plan: ~30 group operations per class, 0.25 ms per G1 MSM at c = 20,
        // 0.7 ms for G2).  G2's sums arrive early and are folded while the GPU runs the G1 passes; the four G1 MSMs' sums arrive
        // together at the very end, so their folds run side by side on host threads instead of one after the other.
        double fold_ms = 0.0;
        {
            G16_HIP_TRY(hipEventSynchronize(ctx->ev_done[4]));
            const double t0 = now_ms();
            store_xyzz(out->b_g2, fold_windows<Fq2>(reinterpret_cast<const G2X*>(pin + 4 * SLOT), sort_z.plan));
            fold_ms += now_ms() - t0;
        }
        if (skip_b_g1) store_xyzz(out->b_g1, G1X::identity());
        {
            // every queued G1 job's copy-out (the batches are grouped by bucket layout, so h's is not necessarily the last)
            for (int q = 0; q < njobs; ++q) G16_HIP_TRY(hipEventSynchronize(ctx->ev_done[jobs[q].k]));
            const double t0 = now_ms();
            struct FoldJob { int k; const MsmPlan* plan; uint64_t* dst; };
            const FoldJob fj[4] = {{1, l_covered ? &sort_z.plan : &sort_l.plan, out->l}, {2, &sort_z.plan, out->a}, {3, &sort_z.plan, out->b_g1},
                                   {0, &sort_h.plan, out->h}};
            auto fold_one = [&](const FoldJob& f) { store_xyzz(f.dst, fold_windows<Fq>(reinterpret_cast<const G1X*>(pin + f.k * SLOT), *f.plan)); };
            std::future<void> fut[3];
            int nf = 0;
            for (int q = 0; q < 3; ++q) {
                if (fj[q].k == 3 && skip_b_g1) continue;
                const FoldJob f = fj[q];
                fut[nf++] = std::async(std::launch::async, [&fold_one, f]() { fold_one(f); });
            }
            fold_one(fj[3]);
            for (int q = 0; q < nf; ++q) fut[q].get();
            fold_ms += now_ms() - t0;
        }
        G16_HIP_TRY(hipStreamSynchronize(s1));
        G16_HIP_TRY(hipStreamSynchronize(s2));
        G16_HIP_TRY(hipStreamSynchronize(s3));
        for (int k = 0; k < 5; ++k) G16_HIP_TRY(hipStreamSynchronize(ctx->red[k]));
        drain.dismiss();
        const double t_end = now_ms();

        g16_timings& tm = ctx->tm;
        memset(&tm, 0, sizeof(tm));
        auto span = [&](int k) -> double {  // bucket pass start (stream 1) -> group sums on the host (reduction stream)
            float t = 0.f;
            return (pass_begin[k] && hipEventElapsedTime(&t, pass_begin[k], ctx->ev_done[k]) == hipSuccess) ? (double)t : 0.0;
        };
        tm.witness_map_ms = ctx->t_wm.ms();
        tm.ntt_ms = ctx->t_ntt[0].ms() + ctx->t_ntt[1].ms();
        tm.scalar_prep_ms = ctx->t_prep_h.ms() + ctx->t_prep_z.ms();
        tm.msm_h_ms = span(0);
        tm.msm_l_ms = span(1);
        tm.msm_a_ms = span(2);
        tm.msm_b_g1_ms = skip_b_g1 ? 0.0 : span(3);
        tm.msm_b_g2_ms = span(4);
        for (int i = 0; i < 5; ++i) {
            float t = 0.f;
            tm.bucket_ms[i] = (pass_begin[i] && pass_end[i] && hipEventElapsedTime(&t, pass_begin[i], pass_end[i]) == hipSuccess) ? (double)t : 0.0;
            tm.bucket_pass_ms += tm.bucket_ms[i];
        }
        tm.finish_ms = fold_ms;
        tm.total_ms = t_end - t_begin;
        tm.window_bits = sort_z.plan.c;
        tm.windows = sort_z.plan.W;
        return G16_OK;
    }

    // prover.rs:76-131 glue over the summed MSM results
    struct FixedPoints {
        G1A alpha_g1, beta_g1, delta_g1, a_query0, b_g1_query0;
        G2A beta_g2, delta_g2, b_g2_query0;
    };
    // prover.rs:76-131 split by what it depends on.  Everything that needs only r, s and the key's eight fixed points -- r delta,
    // s delta, r s delta, and by linearity of :94 and :114 also s (r delta + a_query[0] + alpha) and r (s delta + b_g1_query[0] + beta)
    // -- is the PREPARED half: it can run on a host thread while the GPU is still busy with the MSMs (g16_prove starts it at entry;
    // the sharded path through g16_prove_finalize_prepare).  What is left once the five sums exist: s * sum_a and r * sum_b1 (two
    // variable-base multiplications, side by side), five additions and the three into_affine inversions.
    struct FinalizePrep {
        G1X A0;    // r delta_g1 + a_query[0] + alpha_g1                       (:90-92, :252-270 without the MSM term)
        G2X B2;    // s delta_g2 + b_g2_query[0] + beta_g2                      (:112-113)
        G1X C0;    // s A0 + r (s delta_g1 + b_g1_query[0] + beta_g1) - r s delta_g1   (:94, :114, :76; the r term vanishes for r = 0, :98-108)
        uint32_t rk[Fr::N], sk[Fr::N];
        bool r_zero;
    };
    static FinalizePrep finalize_prepare_core(const FixedPoints& pk, const uint64_t* r_, const uint64_t* s_,
                                              const FixedBaseTable<G1X>* d1 = nullptr, const FixedBaseTable<G2X>* d2 = nullptr) {
        FinalizePrep fp;
        const Fr r = load_pod<Fr>(r_), s = load_pod<Fr>(s_);
        uint32_t rsk[Fr::N];
        r.to_canonical(fp.rk);
        s.to_canonical(fp.sk);
        (r * s).to_canonical(rsk);
        fp.r_zero = r.is_zero();
        const int nb = Fr::Params::BITS;
        const G1X delta1 = G1X::from_affine(pk.delta_g1);
        static_assert(Fr::N == 8, "scalars are 8 words: fixed_base.hpp walks 32 bytes / 64 nibbles");
        auto mul_d1 = [&](const uint32_t* k) { return (d1 && d1->ready()) ? d1->mul(k) : delta1.mul_bits(k, nb); };
        fp.B2 = (d2 && d2->ready()) ? d2->mul(fp.sk) : G2X::from_affine(pk.delta_g2).mul_bits(fp.sk, nb);
        fp.B2.add_affine(pk.b_g2_query0);
        fp.B2.add_affine(pk.beta_g2);
        fp.A0 = mul_d1(fp.rk);
        fp.A0.add_affine(pk.a_query0);
        fp.A0.add_affine(pk.alpha_g1);
        fp.C0 = mul_window4(fp.A0, fp.sk);
        if (!fp.r_zero) {
            G1X b0 = mul_d1(fp.sk);
            b0.add_affine(pk.b_g1_query0);
            b0.add_affine(pk.beta_g1);
            fp.C0.add(mul_window4(b0, fp.rk));
        }
        fp.C0.add(mul_d1(rsk).neg());
        return fp;
    }
    // A and C leave through ONE base-field inversion (Montgomery's trick over the two ZZZ), B through its own in Fq2
    static void two_to_affine(const G1X& p, const G1X& q, G1A* pa, G1A* qa) {
        if (p.is_identity() || q.is_identity()) { *pa = p.to_affine(); *qa = q.to_affine(); return; }
        const Fq inv = (p.zzz * q.zzz).inverse();
        const Fq ip = inv * q.zzz, iq = inv * p.zzz;   // 1/ZZZ_p, 1/ZZZ_q
        const Fq zp = ip * p.zz, zq = iq * q.zz;       // 1/Z
        *pa = {p.x * zp.sqr(), p.y * ip};
        *qa = {q.x * zq.sqr(), q.y * iq};
    }
    static int finalize_finish(const FinalizePrep& fp, const g16_partial* parts, int n_parts, g16_proof* out) {
        if (n_parts < 1) return G16_ERR_BAD_ARG;
        G1X h_acc = G1X::identity(), l_acc = G1X::identity(), a_msm = G1X::identity(), b1_msm = G1X::identity();
        G2X b2_msm = G2X::identity();
        for (int i = 0; i < n_parts; ++i) {  // the N-way EC fold of the all-gathered shard records
            h_acc.add(load_xyzz<G1X>(parts[i].h));
            l_acc.add(load_xyzz<G1X>(parts[i].l));
            a_msm.add(load_xyzz<G1X>(parts[i].a));
            b1_msm.add(load_xyzz<G1X>(parts[i].b_g1));
            b2_msm.add(load_xyzz<G2X>(parts[i].b_g2));
        }
        // B in G2 and r * sum_b1 on two host threads, s * sum_a here
        auto fut_b2 = std::async(std::launch::async, [&]() {
            G2X g2_b = fp.B2;
            g2_b.add(b2_msm);
            return g2_b.to_affine();                                            // :129
        });
        const bool need_rb1 = !fp.r_zero && !b1_msm.is_identity();              // r == 0: :98-108
        std::future<G1X> fut_rb1;
        if (need_rb1) fut_rb1 = std::async(std::launch::async, [&]() { return mul_window4(b1_msm, fp.rk); });
        G1X g_a = fp.A0;                                                        // :90-92
        g_a.add(a_msm);
        G1X g_c = fp.C0;                                                        // :119-124, regrouped
        g_c.add(mul_window4(a_msm, fp.sk));
        g_c.add(l_acc);
        g_c.add(h_acc);
        if (need_rb1) g_c.add(fut_rb1.get());
        G1A pa, pc;
        two_to_affine(g_a, g_c, &pa, &pc);                                      // :128, :130
        const G2A pb = fut_b2.get();
        memset(out, 0, sizeof(*out));
        memcpy(out->a, &pa, sizeof(pa));
        memcpy(out->b, &pb, sizeof(pb));
        memcpy(out->c, &pc, sizeof(pc));
        return G16_OK;
    }
    static int finalize_core(const FixedPoints& pk, const g16_partial* parts, int n_parts, const uint64_t* r_, const uint64_t* s_,
                             g16_proof* out, const FixedBaseTable<G1X>* d1 = nullptr, const FixedBaseTable<G2X>* d2 = nullptr) {
        if (n_parts < 1) return G16_ERR_BAD_ARG;
        return finalize_finish(finalize_prepare_core(pk, r_, s_, d1, d2), parts, n_parts, out);
    }
    static FixedPoints fixed_points(const DevicePk<C>* pk) {
        return {pk->alpha_g1, pk->beta_g1, pk->delta_g1, pk->a_query0, pk->b_g1_query0, pk->beta_g2, pk->delta_g2, pk->b_g2_query0};
    }
    static void ensure_delta_tables(const DevicePk<C>* pk) {   // built on the second proof over a key (32 * 256 additions each)
        std::lock_guard<std::mutex> lk(pk->tab_mu);
        if (!pk->delta2_tab.ready() && ++pk->finalize_calls >= 2) {
            auto f1 = std::async(std::launch::async, [&]() { pk->delta1_tab.build(G1X::from_affine(pk->delta_g1)); });
            pk->delta2_tab.build(G2X::from_affine(pk->delta_g2));
            f1.get();
        }
    }
    // start the prepared half on a host thread; g16_prove_finalize over the same (key, r, s) picks it up
    static int prove_finalize_prepare(g16_ctx* ctx, const g16_pk* pkh, const uint64_t* r_, const uint64_t* s_) {
        const DevicePk<C>* pk = static_cast<const DevicePk<C>*>(pkh->dp);
        ctx->finprep.drop();
        auto data = std::make_shared<FinalizePrep>();
        ctx->finprep.data = data;
        ctx->finprep.pk = pkh;
        memcpy(ctx->finprep.r, r_, 32);
        memcpy(ctx->finprep.s, s_, 32);
        const uint64_t* rr = ctx->finprep.r;
        const uint64_t* ss = ctx->finprep.s;
        ctx->finprep.fut = std::async(std::launch::async, [pk, data, rr, ss]() {
            ensure_delta_tables(pk);
            *data = finalize_prepare_core(fixed_points(pk), rr, ss, &pk->delta1_tab, &pk->delta2_tab);
        });
        ctx->finprep.valid = true;
        return G16_OK;
    }
    static int prove_finalize(g16_ctx* ctx, const g16_pk* pkh, const g16_partial* parts, int n_parts, const uint64_t* r_, const uint64_t* s_,
                              g16_proof* out) {
        const DevicePk<C>* pk = static_cast<const DevicePk<C>*>(pkh->dp);
        const double t0 = now_ms();
        if (n_parts < 1) return G16_ERR_BAD_ARG;
        if (ctx->finprep.matches(pkh, r_, s_)) {   // prepared while the GPU was busy
            ctx->finprep.fut.get();
            const std::shared_ptr<void> keep = ctx->finprep.data;
            ctx->finprep.valid = false;
            G16_TRY(finalize_finish(*static_cast<const FinalizePrep*>(keep.get()), parts, n_parts, out));
        } else {
            ensure_delta_tables(pk);
            G16_TRY(finalize_core(fixed_points(pk), parts, n_parts, r_, s_, out, &pk->delta1_tab, &pk->delta2_tab));
        }
        const double dt = now_ms() - t0;
        ctx->tm.finish_ms += dt;
        ctx->tm.total_ms += dt;
        return G16_OK;
    }
    static int finalize_host(const g16_pk_view* v, const g16_partial* parts, int n_parts, const uint64_t* r_, const uint64_t* s_,
                             g16_proof* out) {
        if (!v->alpha_g1 || !v->beta_g1 || !v->delta_g1 || !v->beta_g2 || !v->delta_g2 || !v->a_query0 || !v->b_g1_query0 ||
            !v->b_g2_query0)
            return G16_ERR_BAD_ARG;
        const FixedPoints fp = {load_pod<G1A>(v->alpha_g1), load_pod<G1A>(v->beta_g1), load_pod<G1A>(v->delta_g1), load_pod<G1A>(v->a_query0),
                                load_pod<G1A>(v->b_g1_query0), load_pod<G2A>(v->beta_g2), load_pod<G2A>(v->delta_g2),
                                load_pod<G2A>(v->b_g2_query0)};
        return finalize_core(fp, parts, n_parts, r_, s_, out);
    }

    // ---------------------------------------------------------------------------------------
    static int witness_map_api(g16_ctx* ctx, const g16_circuit* ckh, const uint64_t* z, uint64_t n_assign, int on_device, uint64_t* h_out) {
        const DeviceCircuit<C>* ck = static_cast<const DeviceCircuit<C>*>(ckh->dc);
        if (n_assign != ck->num_variables) return G16_ERR_BAD_LENGTH;
        DrainOnError drain(ctx);
        ctx->reset_arena();
        const Fr* d_z = nullptr;
        G16_TRY(stage_assignment(ctx, z, n_assign, on_device, &d_z));
        Fr* d_h = nullptr;
        G16_TRY(ctx->arena.alloc_n(ck->dom->n, &d_h));
        G16_TRY((witness_map_device<C>(ck, d_z, d_h, ctx->arena, ctx->stream)));
        G16_HIP_TRY(hipMemcpyAsync(h_out, d_h, ck->dom->n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        G16_HIP_TRY(hipStreamSynchronize(ctx->stream));
        drain.dismiss();
        return G16_OK;
    }

    // async: enqueue on the witness-map stream and return (the caller's exchange goes on that stream too: g16_ctx_wm_stream)
    static int dwm_stage_api(g16_ctx* ctx, const g16_circuit* ckh, const void* dwp, int stage, const uint64_t* z, uint64_t n_assign, int on_device,
                             uint64_t* const work[3], uint64_t* const recv[3], uint64_t* h_local, bool async) {
        const DeviceCircuit<C>* ck = static_cast<const DeviceCircuit<C>*>(ckh->dc);
        const DistWm<C>* dw = static_cast<const DistWm<C>*>(dwp);
        DrainOnError drain(ctx);
        const Fr* d_z = nullptr;
        if (stage == 0) {
            if (!z || n_assign != ck->num_variables) return G16_ERR_BAD_LENGTH;
            if (async) {
                if (!on_device) return G16_ERR_BAD_ARG;   // a staged host copy would live in the arena the next call resets
                d_z = reinterpret_cast<const Fr*>(z);
            } else {
                ctx->reset_arena();
                G16_TRY(stage_assignment(ctx, z, n_assign, on_device, &d_z));
            }
        }
        Fr* w[3] = {reinterpret_cast<Fr*>(work[0]), reinterpret_cast<Fr*>(work[1]), reinterpret_cast<Fr*>(work[2])};
        Fr* rv[3] = {reinterpret_cast<Fr*>(recv[0]), reinterpret_cast<Fr*>(recv[1]), reinterpret_cast<Fr*>(recv[2])};
        G16_TRY((dwm_stage<C>(ck, dw, stage, d_z, w, rv, reinterpret_cast<Fr*>(h_local), async ? ctx->stream_wm : ctx->stream)));
        if (!async) G16_HIP_TRY(hipStreamSynchronize(ctx->stream));   // the caller's exchange runs on its own stream
        drain.dismiss();
        return G16_OK;
    }

    template <class F>
    static int msm_api(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine) {
        typedef Affine<F> A;
        typedef XYZZ<F> X;
        hipStream_t st = ctx->stream;
        DrainOnError drain(ctx);
        ctx->reset_arena();
        A* d_b = nullptr;
        Fr* d_s = nullptr;
        G16_TRY(ctx->arena.alloc_n(n ? n : 1, &d_b));
        G16_TRY(ctx->arena.alloc_n(n ? n : 1, &d_s));
        if (n) {
            G16_HIP_TRY(hipMemcpyAsync(d_b, bases, n * sizeof(A), hipMemcpyHostToDevice, st));
            G16_HIP_TRY(hipMemcpyAsync(d_s, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, st));
        }
        // ad-hoc bases: per-window buckets.  G16_MSM_API_PRECOMP=1 routes this entry point through the proving-key path
        // instead (window tables built on the fly, merged windows) so that it can be tested on arbitrary inputs.
        int merged_c = 0;
        const char* e = getenv("G16_MSM_API_PRECOMP");
        uint32_t modw[Fr::N];
        for (int i = 0; i < Fr::N; ++i) modw[i] = Fr::Params::mod(i);
        if (e && atoi(e) != 0 && n) merged_c = merged_window_bits(n, Fr::Params::BITS, modw, Fr::N);
        if (merged_c) {
            const int W = msm_plan_windows(merged_c, Fr::Params::BITS, modw, Fr::N);
            A* d_t = nullptr;
            G16_TRY(ctx->arena.alloc_n((size_t)n * W, &d_t));
            G16_TRY((build_window_tables<F>(d_b, n, merged_c, W, d_t, st)));
            d_b = d_t;
        } else {
            G16_TRY((convert_bases<F>(d_b, n, st)));
        }
        ScalarSort ss;
        G16_TRY((sort_scalars<C>(d_s, n, merged_c, ctx->arena, st, &ss)));
        MsmBuffers<F> buf;
        G16_TRY((msm_bucket_pass<F>(d_b, 0, n, ss, ctx->arena, st, &buf, &ctx->t_bucket[0])));
        G16_TRY((msm_reduce<F>(buf, ss, st)));
        std::vector<X> hws(ss.plan.outputs());
        G16_HIP_TRY(hipMemcpyAsync(hws.data(), buf.window_sums, sizeof(X) * ss.plan.outputs(), hipMemcpyDeviceToHost, st));
        G16_HIP_TRY(hipStreamSynchronize(st));
        drain.dismiss();
        const A res = fold_windows<F>(hws.data(), ss.plan).to_affine();
        memcpy(out_affine, &res, sizeof(A));
        ctx->tm.bucket_pass_ms = ctx->t_bucket[0].ms();
        ctx->tm.bucket_ms[0] = ctx->tm.bucket_pass_ms;
        ctx->tm.window_bits = ss.plan.c;
        ctx->tm.windows = ss.plan.W;
        return G16_OK;
    }

    static int ntt_api(g16_ctx* ctx, uint64_t* data, int log_n, int inverse, int coset) {
        if (log_n < 0 || log_n > 30) return G16_ERR_DEGREE_TOO_LARGE;
        hipStream_t st = ctx->stream;
        Domain<C>* dom = nullptr;
        G16_TRY((domain_create<C>(log_n, st, &dom)));
        const size_t n = dom->n;
        ctx->reset_arena();
        Fr *d_a = nullptr, *d_o = nullptr;
        int rc = G16_OK;
        auto body = [&]() -> int {
            G16_TRY(ctx->arena.alloc_n(n, &d_a));
            G16_TRY(ctx->arena.alloc_n(n, &d_o));
            G16_HIP_TRY(hipMemcpyAsync(d_a, data, n * sizeof(Fr), hipMemcpyHostToDevice, st));
            if (!inverse) {
                if (coset) {
                    G16_TRY((domain_ensure_gpow<C>(dom, st)));
                    G16_TRY((scale_by_table<C>(d_a, dom->g_pow, n, st)));
                }
                G16_TRY((ntt_dif<C>(dom, d_a, false, st)));
                G16_TRY((bitrev_scale<C>(dom, d_o, d_a, nullptr, nullptr, st)));
            } else {
                G16_TRY((ntt_dif<C>(dom, d_a, true, st)));
                if (coset) G16_TRY((bitrev_scale<C>(dom, d_o, d_a, dom->s2, nullptr, st)));
                else G16_TRY((bitrev_scale<C>(dom, d_o, d_a, nullptr, &dom->n_inv, st)));
            }
            G16_HIP_TRY(hipMemcpyAsync(data, d_o, n * sizeof(Fr), hipMemcpyDeviceToHost, st));
            G16_HIP_TRY(hipStreamSynchronize(st));
            return G16_OK;
        };
        rc = body();
        domain_destroy<C>(dom);
        return rc;
    }

    // ---------------------------------------------------------------------------------------
    // host-side hooks (no GPU)
    template <class F>
    static int field_op(int op, const uint64_t* a, const uint64_t* b, uint64_t* out) {
        F x = load_pod<F>(a), y = b ? load_pod<F>(b) : F::zero(), r;
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x.inverse(); break;
            case 4: x.to_canonical(r.v); break;
            case 5: r = F::from_canonical(x.v); break;
            default: return G16_ERR_BAD_ARG;
        }
        memcpy(out, &r, sizeof(F));
        return G16_OK;
    }
    template <class F>
    static int group_op(int op, const uint64_t* p_, const uint64_t* q_, uint64_t* out) {
        typedef Affine<F> A;
        typedef XYZZ<F> X;
        const A p = load_pod<A>(p_);
        X acc = X::from_affine(p);
        if (op == 0) {
            acc.add_affine(load_pod<A>(q_));
        } else if (op == 1) {
            uint32_t k[8];
            memcpy(k, q_, 32);
            acc = acc.mul_bits(k, 256);
        } else if (op == 2) {
            // exercise the projective + projective path with non-trivial ZZ on both sides
            X q = X::from_affine(load_pod<A>(q_));
            X p2 = acc.dbl(), q2 = q.dbl();   // 2p, 2q
            p2.add(q2);                       // 2p + 2q
            X np = acc.neg();
            p2.add(np);                       // p + 2q
            X nq = q.neg();
            p2.add(nq);                       // p + q
            acc = p2;
        } else {
            return G16_ERR_BAD_ARG;
        }
        const A r = acc.to_affine();
        memcpy(out, &r, sizeof(A));
        return G16_OK;
    }
    // CPU model of kernels 1-7 of msm.hip: same plan, same digit/bucket/sign mapping, same chunked running-sum reduction
    // and final fold.  c_override > 0: per-window plan with that window size; < 0: merged plan with window size -c_override
    // (window tables 2^(cj) P_i built here by repeated doubling); 0: per-window plan from the cost model.
    template <class F>
    static int msm_model(const uint64_t* bases_, const uint64_t* scalars_, uint64_t n, int c_override, uint64_t* out) {
        typedef Affine<F> A;
        typedef XYZZ<F> X;
        uint32_t modw[Fr::N];
        for (int i = 0; i < Fr::N; ++i) modw[i] = Fr::Params::mod(i);
        MsmPlan plan;
        if (c_override > 0) {
            char buf[16];
            snprintf(buf, sizeof(buf), "%d", c_override);
            setenv("G16_MSM_WINDOW", buf, 1);
        }
        int rc = make_msm_plan(n, Fr::Params::BITS, modw, Fr::N, c_override < 0 ? -c_override : 0, &plan);
        if (c_override > 0) unsetenv("G16_MSM_WINDOW");
        if (rc) return rc;
        const A* bases = reinterpret_cast<const A*>(bases_);
        std::vector<X> buckets((size_t)plan.buckets(), X::identity());
        for (uint64_t i = 0; i < n; ++i) {
            Fr s;
            memcpy(&s, scalars_ + 4 * i, sizeof(Fr));
            uint32_t can[Fr::N], sp[MSM_SWORDS];
            s.to_canonical(can);
            uint64_t carry = 0;
            for (int k = 0; k < 10; ++k) {
                carry += (uint64_t)(k < Fr::N ? can[k] : 0u) + plan.K[k];
                sp[k] = (uint32_t)carry;
                carry >>= 32;
            }
            sp[10] = 0;
            A p;
            memcpy(&p, bases + i, sizeof(A));
            for (int w = 0; w < plan.W; ++w) {
                if (plan.merged && w) {   // table row w: 2^(c w) P_i
                    X d = X::from_affine(p);
                    for (int k = 0; k < plan.c; ++k) d = d.dbl();
                    p = d.to_affine();
                }
                uint32_t bucket, neg;
                if (!digit_to_bucket(window_raw(sp, w, plan.c), plan.c, &bucket, &neg)) continue;
                A q = p;
                if (neg) q.y = q.y.neg();
                // merged: `bucket` is the key over all 2^(c-1) buckets = group * B + bucket-in-group already
                buckets[plan.merged ? (size_t)bucket : (size_t)w * plan.B + bucket].add_affine(q);
            }
        }
        const uint32_t G = plan.chunk_buckets(), cpw = plan.chunks();
        const int NP = plan.planes();
        std::vector<X> wsum(plan.outputs(), X::identity());
        for (int w = 0; w < plan.groups; ++w) {
            for (uint32_t ch = 0; ch < cpw; ++ch) {
                const uint32_t b_lo = ch * G;
                X run = X::identity(), tot = X::identity();
                for (uint32_t bb = G; bb-- > 0;) {
                    run.add(buckets[(size_t)w * plan.B + b_lo + bb]);
                    tot.add(run);
                }
                wsum[(size_t)w * NP + 0].add(tot);                                   // plane 0: weighted chunk sums
                wsum[(size_t)w * NP + 1].add(run);                                   // plane 1: plain chunk sums
                for (int k = 0; k < plan.chunk_bits(); ++k)
                    if ((ch >> k) & 1) wsum[(size_t)w * NP + 2 + k].add(run);        // plane 2 + k: chunks with bit k set
            }
        }
        const A res = fold_windows<F>(wsum.data(), plan).to_affine();
        memcpy(out, &res, sizeof(A));
        return G16_OK;
    }
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

static void g16_dwm_free_impl(g16_dwm* d) {
    (void)hipSetDevice(d->ctx->device);
    if (d->curve == G16_BLS12_381) dwm_destroy<Bls12_381>(static_cast<DistWm<Bls12_381>*>(d->dw));
    else dwm_destroy<Bn254>(static_cast<DistWm<Bn254>*>(d->dw));
    d->dw = nullptr;
}

// run fn(i) for i < n on n host threads.  Status: the first REAL failure by device index -- a thread that only gave up because
// a sibling failed returns SIBLING_FAILED, which never masks the sibling's own code -- and that thread's error text becomes the
// caller's g16_last_error() (g_last_error is thread_local).  Every thread is created before any of them runs fn (a start gate):
// if thread creation fails part-way nobody has entered a barrier that expects n participants.
// serial = true runs them one after the other on the calling thread.
static constexpr int SIBLING_FAILED = -1;
template <class Fn>
static int for_each_device(int n, Fn fn, bool serial = false) {
    if (serial) {
        for (int i = 0; i < n; ++i) {
            const int rc = fn(i);
            if (rc) return rc;
        }
        return G16_OK;
    }
    std::vector<int> rc((size_t)n, G16_OK);
    std::vector<std::string> msg((size_t)n);
    std::mutex mu;
    std::condition_variable cv;
    int gate = 0;   // 0: wait, 1: go, -1: abort
    auto body = [&](int i) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return gate != 0; });
            if (gate < 0) return;
        }
        rc[(size_t)i] = fn(i);
        if (rc[(size_t)i]) msg[(size_t)i] = g_last_error;
    };
    std::vector<std::thread> th;
    try {
        for (int i = 1; i < n; ++i) th.emplace_back(body, i);
    } catch (...) {
        { std::lock_guard<std::mutex> lk(mu); gate = -1; }
        cv.notify_all();
        for (auto& t : th) t.join();
        g_last_error = "could not start one host thread per device";
        return G16_ERR_INTERNAL;
    }
    { std::lock_guard<std::mutex> lk(mu); gate = 1; }
    cv.notify_all();
    rc[0] = fn(0);
    if (rc[0]) msg[0] = g_last_error;
    for (auto& t : th) t.join();
    int pick = -1;
    for (int i = 0; i < n && pick < 0; ++i) if (rc[(size_t)i] != G16_OK && rc[(size_t)i] != SIBLING_FAILED) pick = i;
    if (pick >= 0) { g_last_error = msg[(size_t)pick]; return rc[(size_t)pick]; }
    for (int v : rc) if (v) return G16_ERR_INTERNAL;   // only markers: cannot happen (a marker needs a failed sibling)
    return G16_OK;
}

// reusable barrier for the per-device host threads of one call (C++17: no std::barrier)
struct HostBarrier {
    std::mutex mu;
    std::condition_variable cv;
    int n, waiting = 0;
    uint64_t phase = 0;
    explicit HostBarrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> lk(mu);
        const uint64_t my = phase;
        if (++waiting == n) { waiting = 0; ++phase; cv.notify_all(); return; }
        cv.wait(lk, [&] { return phase != my; });
    }
};

// the distributed witness map runs over `world` ranks when world is a power of two in [2, 16] and world^2 divides the domain
static bool dist_wm_admissible(int world, uint64_t domain) {
    if (world < 2 || world > 16 || (world & (world - 1)) != 0) return false;
    if (domain == 0 || (domain & (domain - 1)) != 0) return false;
    return domain % ((uint64_t)world * (uint64_t)world) == 0;
}

// A context may list one physical device several times (tests: N shards on the one GPU of the box).  The LOAD paths of such a
// context run one after the other: n concurrent window-table builds on sibling queues of ONE device aborted inside the HIP runtime
// in the full test suite of round 2 (never in isolation, never root-caused; the builders were since rewritten without their 9-17 KB
// of scratch per lane and a concurrent run of the suite passed in round 3, but an abort cannot be caught and retried, so the safe
// order is the default).  Distinct devices -- the case that matters -- always load concurrently.  G16_MULTI_CONCURRENT_LOAD=1
// loads a repeated device concurrently too.
static bool serial_loads(const g16_ctx* ctx) {
    const char* e = getenv("G16_MULTI_CONCURRENT_LOAD");
    if (e && atoi(e) != 0) return false;
    for (size_t a = 0; a < ctx->subs.size(); ++a)
        for (size_t b = a + 1; b < ctx->subs.size(); ++b)
            if (ctx->subs[a]->device == ctx->subs[b]->device) return true;
    return false;
}

// static multiply-add counts of the bucket kernels' arithmetic, from the tables the kernels themselves are generated from
template <class B30>
static void diag_counts(g16_diag* out) {
    constexpr int NL = B30::NL;
    int relax[5] = {0, 0, 0, 0, 0};   // relax[k]: columns relaxed when k sweeps are in and one more (a sweep or the reduction) follows
    for (int c = 0; c + 1 < 2 * NL; ++c)
        for (int k = 1; k <= 4; ++k)
            if ((k + 1) * B30::col_count(c) > G16_RELAX_LIMIT) ++relax[k];
    const double mul = 2.0 * NL * NL + relax[1], sqr = NL * (NL + 1) / 2.0 + NL * NL + relax[1];
    const double two_sweeps = 3.0 * NL * NL + relax[1] + relax[2];   // two product sweeps + ONE reduction: a lane's Fq2 product; Fp30::mul_sub_cols
    const double four_sweeps = 5.0 * NL * NL + relax[1] + relax[2] + relax[3] + relax[4];   // Fp2p30::pair_mul_sub
    out->limbs = NL;
    out->mads_per_product = mul;
    // madd-2008-s as the bucket pass runs it (AccParked, fp30.hpp): U2 S2 PPP Q ZZ*PP ZZZ*PPP + the squarings PP, R^2 + Y3 = R (Q - X3) - Y1 PPP
    // under one reduction (the register-resident Acc30 of the other kernels spends two products on Y3)
    const bool fused_g1 = Fp30<typename B30::Params_t>::ACC_PARKED, fused_g2 = Fp2p30<typename B30::Params_t>::ACC_PARKED;
#ifdef G16_NO_MUL_SUB_FUSED
    const bool fused = false;
#else
    const bool fused = true;
#endif
    out->mads_per_add_g1 = (fused && fused_g1) ? 6 * mul + 2 * sqr + two_sweeps : 8 * mul + 2 * sqr;
    // per lane of the pair: pair products (two sweeps each), two pair squarings (one product each), Y3 as four sweeps + one reduction
    out->mads_per_add_g2 = 2 * ((fused && fused_g2) ? 6 * two_sweeps + 2 * mul + four_sweeps : 8 * two_sweeps + 2 * mul);
}

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
    // direct xGMI copies between the devices (the exchange of the distributed witness map); a refusal only means that
    // hipMemcpyPeerAsync stages through the host
    for (int a = 0; a < n_dev; ++a)
        for (int b = 0; b < n_dev; ++b)
            if (device_ids[a] != device_ids[b] && hipSetDevice(device_ids[a]) == hipSuccess) {
                (void)hipDeviceEnablePeerAccess(device_ids[b], 0);
                (void)hipGetLastError();
            }
    *out = c;
    return G16_OK;
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
        const int rc = for_each_device(n, [&](int i) -> int {
            g16_pk_view v = *view;
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
        if (rc) { g16_pk_free(h); return rc; }
        *out = h;
        return G16_OK;
    }
    G16_HIP_TRY(hipSetDevice(ctx->device));
    G16_DISPATCH(ctx->curve, I::pk_load(ctx, view, out));
}

void g16_pk_free(g16_pk* pk) {
    if (!pk) return;
    if (!pk->ctx->subs.empty()) {
        for (g16_pk* sub : pk->subs) g16_pk_free(sub);
        delete pk;
        return;
    }
    if (pk->ctx->finprep.pk == pk) pk->ctx->finprep.drop();   // a prepared finalize half may still be reading the key's fixed points
    (void)hipSetDevice(pk->ctx->device);
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
                    return G16_ERR_OOM;
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
            (void)hipFree(sl.h_local); (void)hipFree(sl.z_dev);
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
    G16_DISPATCH(ctx->curve, I::prove_finalize(ctx, pk, parts, n_parts, r, s, out));
}

int g16_prove_finalize_prepare(g16_ctx* ctx, const g16_pk* pk, const uint64_t r[4], const uint64_t s[4]) {
    if (!ctx || !pk || !r || !s) return G16_ERR_BAD_ARG;
    if (pk->curve != ctx->curve) return G16_ERR_BAD_ARG;
    if (!ctx->subs.empty()) {
        if (pk->subs.empty()) return G16_ERR_BAD_ARG;
        return g16_prove_finalize_prepare(ctx->subs[0], pk->subs[0], r, s);
    }
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
            if (n_assign != circuit->num_variables) return G16_ERR_BAD_LENGTH;
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
                        if (st < 3) G16_HIP_TRY(hipEventRecord(ev_stage[(size_t)i], sw));
                        return G16_OK;
                    });
                    if (st == 3) break;
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
                        out_rc = g16_prove_partial_h(sub, pk->subs[(size_t)i], circuit->subs[(size_t)i], zp, n_assign, 1, sl.h_local, M, skip_b_g1,
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
        if (rc) return rc;
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
    if (!ctx || !pk) return G16_ERR_BAD_ARG;
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
        default: return "unknown status";
    }
}

const char* g16_last_error(void) { return g_last_error.c_str(); }
const char* g16_version(void) { return "g16_mi355x 0.1 (gfx950)"; }

}  // extern "C"
