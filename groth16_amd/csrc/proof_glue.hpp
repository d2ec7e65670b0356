// The host glue of /root/reference/src/prover.rs:76-131 over the (summed) MSM results: r delta, s delta, r s delta, s A, r B1, the
// additions and the three into_affine -- split into an (r, s)-only half that runs while the GPU works and a short finish.
#pragma once
#include "api_types.hpp"

namespace {

template <class C>
struct ProofGlue {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    typedef typename C::G1X G1X;
    typedef typename C::G2X G2X;
    static constexpr int L = Fq::N / 2;  // 64-bit limbs per Fq
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
        RoctxRange rr("Finish C");                                              // prover.rs:119
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
    static FixedPoints fixed_points(const KeyGlue<C>* pk) {
        return {pk->alpha_g1, pk->beta_g1, pk->delta_g1, pk->a_query0, pk->b_g1_query0, pk->beta_g2, pk->delta_g2, pk->b_g2_query0};
    }
    static void ensure_delta_tables(KeyGlue<C>* pk) {   // built on the second proof over a key (32 * 256 additions each)
        std::lock_guard<std::mutex> lk(pk->tab_mu);
        if (!pk->tabs_ready.load(std::memory_order_relaxed) && ++pk->finalize_calls >= 2) {
            auto f1 = std::async(std::launch::async, [&]() { pk->delta1_tab.build(G1X::from_affine(pk->delta_g1)); });
            pk->delta2_tab.build(G2X::from_affine(pk->delta_g2));
            f1.get();
            pk->tabs_ready.store(true, std::memory_order_release);
        }
    }
    // the tables, or nullptr while they do not exist / are being built by another context's thread (the glue then multiplies delta
    // bit by bit: correct, 0.5 ms slower, first proofs over a key only)
    static const FixedBaseTable<G1X>* table1(const KeyGlue<C>* pk) { return pk->tabs_ready.load(std::memory_order_acquire) ? &pk->delta1_tab : nullptr; }
    static const FixedBaseTable<G2X>* table2(const KeyGlue<C>* pk) { return pk->tabs_ready.load(std::memory_order_acquire) ? &pk->delta2_tab : nullptr; }
    // start the prepared half on a host thread; g16_prove_finalize over the same (key, r, s) picks it up.  The thread owns a
    // reference to the key's host half and copies of r and s: nothing it touches can be freed or rewritten under it.
    static int prove_finalize_prepare(g16_ctx* ctx, const g16_pk* pkh, const uint64_t* r_, const uint64_t* s_) {
        const std::shared_ptr<KeyGlue<C>> glue = static_cast<const DevicePk<C>*>(pkh->dp)->glue;
        ctx->finprep.drop();
        auto data = std::make_shared<FinalizePrep>();
        ctx->finprep.data = data;
        ctx->finprep.key_id = pkh->id;
        memcpy(ctx->finprep.r, r_, 32);
        memcpy(ctx->finprep.s, s_, 32);
        struct RS { uint64_t r[4], s[4]; } rs;
        memcpy(rs.r, r_, 32);
        memcpy(rs.s, s_, 32);
        ctx->finprep.fut = std::async(std::launch::async, [glue, data, rs]() {
            ensure_delta_tables(glue.get());
            *data = finalize_prepare_core(fixed_points(glue.get()), rs.r, rs.s, table1(glue.get()), table2(glue.get()));
        });
        ctx->finprep.valid = true;
        return G16_OK;
    }
    static int prove_finalize(g16_ctx* ctx, const g16_pk* pkh, const g16_partial* parts, int n_parts, const uint64_t* r_, const uint64_t* s_,
                              g16_proof* out) {
        KeyGlue<C>* pk = static_cast<const DevicePk<C>*>(pkh->dp)->glue.get();
        const double t0 = now_ms();
        if (n_parts < 1) return G16_ERR_BAD_ARG;
        if (ctx->finprep.matches(pkh->id, r_, s_)) {   // prepared while the GPU was busy
            ctx->finprep.fut.get();
            const std::shared_ptr<void> keep = ctx->finprep.data;
            ctx->finprep.valid = false;
            G16_TRY(finalize_finish(*static_cast<const FinalizePrep*>(keep.get()), parts, n_parts, out));
        } else {
            ensure_delta_tables(pk);
            G16_TRY(finalize_core(fixed_points(pk), parts, n_parts, r_, s_, out, table1(pk), table2(pk)));
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

};

}  // namespace
