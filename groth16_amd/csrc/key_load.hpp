// g16_pk_load / g16_circuit_load: the proving key's five query arrays onto the GPU as window tables (or plain bases), the three CSR
// matrices and the NTT domain tables (ProvingKey: /root/reference/src/data_structures.rs:125-143; ConstraintMatrices: prover.rs:30-32).
#pragma once
#include "api_types.hpp"

namespace {

template <class C>
struct KeyLoader {
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

    // shard_n > 1: bucket-space shard `shard_r` of `shard_n` -- the view must be the WHOLE key, and the key must get its window
    // tables (merged plans only; there is no silent fall-back to plain bases in this mode: G16_ERR_OOM instead)
    static int pk_load(g16_ctx* ctx, const g16_pk_view* v, g16_pk** out, int shard_n = 1, int shard_r = 0) {
        if (!v->alpha_g1 || !v->beta_g1 || !v->delta_g1 || !v->beta_g2 || !v->delta_g2 || !v->a_query0 || !v->b_g1_query0 ||
            !v->b_g2_query0)
            return G16_ERR_BAD_ARG;
        DevicePk<C>* p = new (std::nothrow) DevicePk<C>();
        if (!p) return G16_ERR_OOM;
        p->glue->alpha_g1 = load_pod<G1A>(v->alpha_g1);
        p->glue->beta_g1 = load_pod<G1A>(v->beta_g1);
        p->glue->delta_g1 = load_pod<G1A>(v->delta_g1);
        p->glue->beta_g2 = load_pod<G2A>(v->beta_g2);
        p->glue->delta_g2 = load_pod<G2A>(v->delta_g2);
        p->glue->a_query0 = load_pod<G1A>(v->a_query0);
        p->glue->b_g1_query0 = load_pod<G1A>(v->b_g1_query0);
        p->glue->b_g2_query0 = load_pod<G2A>(v->b_g2_query0);
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
        if (p->c_z == 0 || p->c_h == 0) {   // tables for all queries or for none
            const char* e = getenv("G16_MSM_PRECOMP");
            p->table_fallback = (nz || v->h.count) ? ((e && atoi(e) == 0) ? 1 : 2) : 0;
            p->c_z = p->c_h = 0;
        }
        p->shard_n = shard_n;
        p->shard_r = shard_r;
        if (shard_n > 1) {
            if (v->a.start || v->b_g1.start || v->b_g2.start || v->h.start || v->l.start) {
                delete p;
                g_last_error = "a bucket-space shard is loaded from the WHOLE key (every query.start == 0): the ranks divide the buckets, not the bases";
                return G16_ERR_BAD_ARG;
            }
            if (p->c_z == 0) {
                delete p;
                g_last_error = "a bucket-space shard needs the key's window tables (merged windows): G16_MSM_PRECOMP=0 or an over-long query rules them out";
                return G16_ERR_BAD_ARG;
            }
        }
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
                if (rc == G16_ERR_OOM && cz != 0 && shard_n <= 1) {   // the tables do not fit next to what already lives on this GPU: plain bases
                    (void)hipGetLastError();
                    p->c_z = p->c_h = 0;
                    p->table_fallback = 3;
                    continue;
                }
                if (rc == G16_ERR_OOM && shard_n > 1)
                    g_last_error = "the whole key's window tables do not fit on this GPU: a bucket-space shard cannot be held (use base-range shards)";
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
        {
            const uint64_t rows_z = p->c_z ? (uint64_t)msm_plan_windows(p->c_z, Fr::Params::BITS, modw, Fr::N) : 1;
            const uint64_t rows_h = p->c_h ? (uint64_t)msm_plan_windows(p->c_h, Fr::Params::BITS, modw, Fr::N) : 1;
            p->table_bytes = rows_z * ((v->a.count + v->b_g1.count + v->l.count) * sizeof(G1A) + v->b_g2.count * sizeof(G2A)) +
                             rows_h * v->h.count * sizeof(G1A);
        }
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
        {   // need_col (DeviceCircuit): prefix maximum of the columns read by the rows of each of the Z_CHUNKS row blocks of the domain
            constexpr int ZC = DeviceCircuit<C>::Z_CHUNKS;
            const uint64_t n_dom = (uint64_t)1 << log_n;
            uint64_t run = 0;
            for (int k = 0; k < ZC; ++k) {
                const uint64_t r_lo = n_dom * (uint64_t)k / ZC, r_hi = n_dom * (uint64_t)(k + 1) / ZC;
                for (int m = 0; m < 3; ++m) {
                    const uint64_t a = std::min(r_lo, num_constraints), b = std::min(r_hi, num_constraints);
                    for (uint64_t e = abc[m].row_ptr[a]; e < abc[m].row_ptr[b]; ++e) run = std::max<uint64_t>(run, abc[m].col[e]);
                }
                if (r_hi > num_constraints && num_inputs) run = std::max(run, std::min(r_hi - num_constraints, num_inputs) - 1);   // a[nc + j] = z[j]
                dc->need_col[k] = run;
            }
        }
        int rc = mark_unit_coefficients<C>(dc, ctx->stream);
        if (rc) return fail(rc);
        rc = domain_create<C>(log_n, ctx->stream, &dc->dom);
        if (rc) return fail(rc);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return fail(G16_ERR_HIP);
        g16_circuit* h = new (std::nothrow) g16_circuit{C::CURVE_ID, ctx, dc, (uint64_t)1 << log_n};
        if (!h) return fail(G16_ERR_OOM);
        *out = h;
        return G16_OK;
    }

    // ---------------------------------------------------------------------------------------
};

}  // namespace
