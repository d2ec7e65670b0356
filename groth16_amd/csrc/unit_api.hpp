// Unit-level entry points behind the C ABI: witness map, distributed-map stages, ad-hoc MSMs, a single NTT, and the CPU-side
// self-test hooks (field / group operations, the bucket-method model).
#pragma once
#include "api_types.hpp"
#include "prover.hpp"

namespace {

template <class C>
struct UnitApi {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    typedef typename C::G1X G1X;
    typedef typename C::G2X G2X;
    static constexpr int L = Fq::N / 2;  // 64-bit limbs per Fq
    // ---------------------------------------------------------------------------------------
    static int witness_map_api(g16_ctx* ctx, const g16_circuit* ckh, const uint64_t* z, uint64_t n_assign, int on_device, uint64_t* h_out) {
        const DeviceCircuit<C>* ck = static_cast<const DeviceCircuit<C>*>(ckh->dc);
        if (n_assign != ck->num_variables) return G16_ERR_BAD_LENGTH;
        DrainOnError drain(ctx);
        ctx->reset_arena();
        const Fr* d_z = nullptr;
        G16_TRY(Prover<C>::stage_assignment(ctx, z, n_assign, on_device, &d_z));
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
                G16_TRY(Prover<C>::stage_assignment(ctx, z, n_assign, on_device, &d_z));
            }
        }
        Fr* w[3] = {reinterpret_cast<Fr*>(work[0]), reinterpret_cast<Fr*>(work[1]), reinterpret_cast<Fr*>(work[2])};
        Fr* rv[3] = {reinterpret_cast<Fr*>(recv[0]), reinterpret_cast<Fr*>(recv[1]), reinterpret_cast<Fr*>(recv[2])};
        G16_TRY((dwm_stage<C>(ck, dw, stage, d_z, w, rv, reinterpret_cast<Fr*>(h_local), async ? ctx->stream_wm : ctx->stream)));
        if (!async) G16_HIP_TRY(hipStreamSynchronize(ctx->stream));   // the caller's exchange runs on its own stream
        drain.dismiss();
        return G16_OK;
    }

    // shard_n > 1: rank shard_r's bucket-space share of the MSM (window tables built on the fly, merged windows): the ranks' results
    // add up to the MSM
    template <class F>
    static int msm_api(g16_ctx* ctx, const uint64_t* bases, const uint64_t* scalars, uint64_t n, uint64_t* out_affine, int shard_n = 1,
                       int shard_r = 0) {
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
        if (((e && atoi(e) != 0) || shard_n > 1) && n) merged_c = merged_window_bits(n, Fr::Params::BITS, modw, Fr::N);
        if (shard_n > 1 && n && !merged_c) return G16_ERR_BAD_ARG;   // G16_MSM_PRECOMP=0 / too many points for merged entries
        if (shard_n > 1 && !n) { memset(out_affine, 0, sizeof(A)); drain.dismiss(); return G16_OK; }
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
        G16_TRY((sort_scalars<C>(d_s, n, merged_c, ctx->arena, st, &ss, shard_n, shard_r)));
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
    // shard_n > 1 (merged plans): rank shard_r's bucket-space share, exactly as sort_scalars filters and re-indexes the keys
    template <class F>
    static int msm_model(const uint64_t* bases_, const uint64_t* scalars_, uint64_t n, int c_override, uint64_t* out, int shard_n = 1,
                         int shard_r = 0) {
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
        if (shard_n > 1 && c_override >= 0) return G16_ERR_BAD_ARG;
        int rc = make_msm_plan(n, Fr::Params::BITS, modw, Fr::N, c_override < 0 ? -c_override : 0, &plan, shard_n, shard_r);
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
                if (plan.shard_n > 1) {
                    if (bucket % (uint32_t)plan.shard_n != (uint32_t)plan.shard_r) continue;   // another rank's residue class
                    bucket /= (uint32_t)plan.shard_n;                                            // local index k: b = shard_n k + shard_r
                }
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
