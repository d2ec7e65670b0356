// g16_prove_partial: one proof's GPU schedule -- witness map, scalar sorts, the five bucket passes, reductions, window folds
// (/root/reference/src/prover.rs:26-51 and :54-132 up to the MSM sums; the glue of :76-131 is proof_glue.hpp).
#pragma once
#include "api_types.hpp"

namespace {

template <class C>
struct Prover {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    typedef typename C::G1X G1X;
    typedef typename C::G2X G2X;
    static constexpr int L = Fq::N / 2;  // 64-bit limbs per Fq
    static int stage_assignment(g16_ctx* ctx, const uint64_t* z, uint64_t n_assign, int on_device, const Fr** d_z) {
        if (on_device) { *d_z = reinterpret_cast<const Fr*>(z); return G16_OK; }
        Fr* buf = nullptr;
        G16_TRY(ctx->arena.alloc_n(n_assign, &buf));
        G16_HIP_TRY(hipMemcpyAsync(buf, z, n_assign * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        *d_z = buf;
        return G16_OK;
    }


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
        G16_TRY((sort_scalars<C>(d_z + 1 + pk->a_start, pk->a_count, pk->c_z, ctx->arena, ctx->stream2, &ss, pk->shard_n, pk->shard_r)));
        G16_HIP_TRY(hipEventRecord(ctx->ev_z, ctx->stream2));
        ctx->prep.valid = true;
        ctx->prep.pk = pkh;
        ctx->prep.key_id = pkh->id;
        ctx->prep.a_start = pk->a_start;
        ctx->prep.a_count = pk->a_count;
        ctx->prep.c_z = pk->c_z;
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
        // (... and for the key AS IT IS NOW: g16_pk_rebind_bucket_shard may have re-labelled it since, and the sort depends on the residue class)
        const bool prepared = ctx->prep.valid && on_device && ctx->prep.pk == pkh && ctx->prep.key_id == pkh->id && ctx->prep.z == z &&
                              ctx->prep.n_assign == n_assign && ctx->prep.a_start == pk->a_start && ctx->prep.a_count == pk->a_count &&
                              ctx->prep.c_z == pk->c_z && ctx->prep.sort_z.plan.shard_n == pk->shard_n &&
                              ctx->prep.sort_z.plan.shard_r == pk->shard_r;
        const ScalarSort prepared_sort = ctx->prep.sort_z;
        if (prepared) ctx->prep.valid = false;   // consumed: the arena keeps its contents for this call
        else ctx->reset_arena();                 // (drains a prepared sort that is being dropped before its buffers are reused)
        const Fr* d_z = nullptr;
        // a host assignment with the witness map running here: uploaded in pieces BY the map, whose mat-vec follows the pieces
        // (witness_map_device / ZUpload); every other reader of z waits for the last piece.  G16_UPLOAD_CHUNKED=0: one copy (A/B).
        ZUpload up;
        const bool chunked = !on_device && !h_ext && !prepared && !(getenv("G16_UPLOAD_CHUNKED") && atoi(getenv("G16_UPLOAD_CHUNKED")) == 0);
        if (chunked) {
            Fr* buf = nullptr;
            G16_TRY(ctx->arena.alloc_n(n_assign, &buf));
            d_z = buf;
            up.host = z;
            up.copy_stream = ctx->stream_h2d;
            for (int i = 0; i < 8; ++i) up.landed[i] = ctx->ev_up[i];
        } else {
            G16_TRY(stage_assignment(ctx, z, n_assign, on_device, &d_z));
            if (!prepared) G16_HIP_TRY(hipEventRecord(ctx->ev_z, s1));
        }

        // ---- witness map, h = QAP::witness_map_from_matrices (prover.rs:37-42); only the h MSM needs it.  It goes FIRST,
        // alone, on stream 1 (~6 ms at 2^22): underneath the bucket passes their long-lived waves starve it (60+ ms
        // measured) and everything queued behind it piles up at the end of the proof.
        // G16_MAP_UNDER_PASSES=1 (experiment, round 5): the map on its own stream, the passes waiting only for the witness sort.  A
        // transform workgroup (38 KB of LDS, one 120-register wave per SIMD) fits beside the G1 kernel's waves -- not beside the G2
        // kernel's -- so the map would stall under the G2 pass and finish underneath the G1 launch, its ~50 % of idle issue slots filled.
        const bool overlap_map = !h_ext && getenv("G16_MAP_UNDER_PASSES") != nullptr && atoi(getenv("G16_MAP_UNDER_PASSES")) != 0;
        hipStream_t sm = overlap_map ? ctx->stream_wm : s1;
        Fr* d_h = nullptr;
        ScalarSort sort_h, sort_z, sort_l;
        if (overlap_map && !chunked) G16_HIP_TRY(hipStreamWaitEvent(sm, ctx->ev_z, 0));   // behind the staged copy / the caller's position on stream 1
        G16_TRY(ctx->t_wm.start(sm));
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
            RoctxRange rr("R1CS to QAP witness map");                                                    // prover.rs:36
            G16_TRY((witness_map_device<C>(ck, d_z, d_h, ctx->arena, sm, ctx->t_ntt, chunked ? &up : nullptr)));
            if (chunked) G16_HIP_TRY(hipEventRecord(ctx->ev_z, ctx->stream_h2d));   // behind the last piece: the whole assignment is in HBM
        }
        G16_TRY(ctx->t_wm.stop(sm));
        G16_HIP_TRY(hipEventRecord(ctx->ev_wm, sm));

        // ---- stream 2, beside the witness map: assignment = full_assignment[1..] (prover.rs:80-85), ONE digit/sort
        // pass for a, b_g1, b_g2 (and l)
        if (prepared) {
            sort_z = prepared_sort;                      // ev_z was recorded on stream 2 behind the sort by the prepare call
            ctx->t_prep_z.used = false;
        } else {
            G16_HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_z, 0));
            G16_TRY(ctx->t_prep_z.start(s2));
            G16_TRY((sort_scalars<C>(d_z + 1 + pk->a_start, pk->a_count, pk->c_z, ctx->arena, s2, &sort_z, pk->shard_n, pk->shard_r)));
            G16_TRY(ctx->t_prep_z.stop(s2));
            G16_HIP_TRY(hipEventRecord(ctx->ev_z, s2));   // (re-recorded: now also covers the witness sort)
        }
        G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_z, 0));
        // ---- stream 3: h's digit/sort pass, underneath the first bucket pass (stream 2 stays free for the reductions)
        G16_HIP_TRY(hipStreamWaitEvent(s3, ctx->ev_wm, 0));
        G16_TRY(ctx->t_prep_h.start(s3));
        G16_TRY((sort_scalars<C>(d_h + pk->h_start, pk->h_count, pk->c_h, ctx->arena, s3, &sort_h, pk->shard_n, pk->shard_r)));
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
        // Bucket passes back to back on stream 1: the G2 MSM first -- its reduction (the longest chain: ~3x a G1 one) runs on its own
        // stream underneath the G1 passes -- then the G1 MSMs that are ready together as ONE launch (l, a, b_g1; round 5), then h, whose
        // sort has run underneath that launch.  The G1 reductions are NOT started one by one underneath the following pass (a reduction
        // is a few hundred waves of dependent additions that hold register slots for milliseconds and slowed every pass they ran
        // under: round 3) but run TOGETHER, one launch per stage for all of them (msm_reduce_batch), after the last pass -- except their
        // FIRST stage (heavy buckets, then one sum per bucket), which goes underneath the next pass as one launch per kernel.
        const bool short_passes = (uint64_t)pk->a_count * (uint64_t)sort_z.plan.W / (uint64_t)pk->shard_n < 20000000ull;
        // Timestamps: ONE event per boundary between back-to-back passes (the end of pass k is the begin of pass k + 1) instead of a
        // start / stop / done / span-start record around every pass -- each record is a barrier packet the command processor retires
        // before it starts the next kernel, and the four of them cost ~0.13 ms of idle GPU between two passes (kernel trace of round 3).
        hipEvent_t pass_begin[5] = {}, pass_end[5] = {}, last_end = nullptr;
        int n_edge = 0;
        auto mark = [&](hipEvent_t* ev, hipStream_t on = nullptr) -> int {
            if (n_edge >= 8) return G16_ERR_INTERNAL;
            *ev = ctx->ev_edge[n_edge++];
            G16_HIP_TRY(hipEventRecord(*ev, on ? on : s1));
            return G16_OK;
        };
        // The G2 pass (and its reduction) on its own queue beside the G1 passes of stream 1, so that the dispatcher fills the tail of one
        // kernel with workgroups of the other: on for the SHORT passes of a sharded proof, where a tail is 10-15 % of a pass (8-way share
        // at 2^22, same box: 11.04 -> 10.91 ms bucket-space, 12.31 -> 12.23 base ranges); off for whole-key proofs, where it only delays
        // the G2 reduction towards the end of the proof (67.8 / 68.2 vs 68.3 ms).  G16_PASS_CONCURRENT=0 / 1 forces.
        bool concurrent = short_passes;
        if (const char* e = getenv("G16_PASS_CONCURRENT")) concurrent = atoi(e) != 0;
        auto run_pass = [&](int k, auto* bases, int64_t shift, uint64_t count, const ScalarSort& ss, auto* buf, hipStream_t on = nullptr) -> int {
            if (last_end && !on) pass_begin[k] = last_end;
            else G16_TRY(mark(&pass_begin[k], on));
            G16_TRY((msm_bucket_pass(bases, shift, count, ss, ctx->arena, on ? on : s1, buf, nullptr)));
            G16_TRY(mark(&pass_end[k], on));
            if (!on) last_end = pass_end[k];
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
            RoctxRange rr("Compute B in G2");                                                                 // prover.rs:111
            if (concurrent) {
                G16_HIP_TRY(hipStreamWaitEvent(sr, ctx->ev_z, 0));
                // never underneath the witness map (it would starve).  (Sharded proof, round 6: the G2 pass started behind the witness sort
                // only, the distributed map's remaining links beside it, with four-wave and with one-wave transform workgroups:
                // 9.93 - 10.2 ms per rank in every arrangement, profiles/r06_ab_g2_before_dist_map.txt -- the share is work, not waiting.)
                G16_HIP_TRY(hipStreamWaitEvent(sr, ctx->ev_wm, 0));
                G16_TRY(run_pass(4, pk->b_g2, 0, pk->b_g2_count, sort_z, &buf_b2, sr));
            } else {
                G16_TRY(run_pass(4, pk->b_g2, 0, pk->b_g2_count, sort_z, &buf_b2));                           // prover.rs:113
                G16_HIP_TRY(hipStreamWaitEvent(sr, pass_end[4], 0));
            }
            G16_TRY((msm_reduce(buf_b2, sort_z, sr)));
            G16_TRY(copy_out(4, buf_b2, sort_z, sr));
        }
        struct G1Job { int k; MsmBuffers<Fq>* buf; const ScalarSort* ss; };
        G1Job jobs[4];
        int njobs = 0;
        // The G1 passes that are ready together go as ONE launch (msm_bucket_pass_batch: one tail instead of one per MSM): a and b_g1
        // always (one sort, equal counts), l when it reuses the witness sort, and h when its sort is certain to be done by then -- the
        // sharded proof, where the passes wait for the distributed map anyway and h's sort is a few hundred microseconds.  On one GPU
        // h's sort runs starved underneath the passes (stream 3) and finishes late: there the h pass stays a launch of its own at the end.
        PassJob<Fq> pj[4];
        int pk_of[4], npj = 0;
        double weight[5] = {0, 0, 0, 0, 0};   // a batch's time is shared out by the points of its members (one W: entries ~ points)
        auto plans_match = [](const MsmPlan& x, const MsmPlan& y) {
            return x.B == y.B && x.groups == y.groups && x.Lmax == y.Lmax && x.merged == y.merged && x.affine_levels == 0 && y.affine_levels == 0;
        };
        const bool batch_ok = sort_z.plan.affine_levels == 0 || sort_z.max_sorted == 0;
        // (sharded proof: h waits in the same launch -- the launch then waits ~0.4 ms for h's sort, which only gets going when the G2
        // pass drains, but a fourth launch with its own tail costs more: 8-way share 10.76 vs 11.03 ms, same box.  G16_PASS_H_IN_BATCH=0 / 1)
        bool h_in_batch = h_ext != nullptr && batch_ok && plans_match(sort_z.plan, sort_h.plan) && !getenv("G16_PASS_NO_BATCH");
        if (const char* e = getenv("G16_PASS_H_IN_BATCH")) h_in_batch = h_in_batch && atoi(e) != 0;
        int g1_launches = 0;
        bool last_batch = false, early[4] = {false, false, false, false};
        int early_ev[4] = {0, 1, 2, 3};   // which ev_heavy[] covers MSM k's first reduction stage
        auto run_batch = [&]() -> int {   // whatever is queued in pj[]
            if (!npj) return G16_OK;
            hipEvent_t b0, b1;
            if (last_end) b0 = last_end;
            else G16_TRY(mark(&b0));
            if (batch_ok && !getenv("G16_PASS_NO_BATCH")) {
                G16_TRY((msm_bucket_pass_batch<Fq>(pj, npj, ctx->arena, s1, nullptr)));
                ++g1_launches;
            } else {
                for (int i = 0; i < npj; ++i) {
                    G16_TRY((msm_bucket_pass<Fq>(pj[i].bases, pj[i].shift, pj[i].base_count, *pj[i].ss, ctx->arena, s1, pj[i].out, nullptr)));
                    ++g1_launches;
                }
            }
            G16_TRY(mark(&b1));
            last_end = b1;
            for (int i = 0; i < npj; ++i) { pass_begin[pk_of[i]] = b0; pass_end[pk_of[i]] = b1; }
            // the first stage of an MSM's reduction (heavy-bucket combine, then one sum per bucket) goes underneath the NEXT pass on the
            // MSM's own stream; after the last pass there is nothing to hide under, and the batched reduction runs that stage itself
            // -- as ONE launch per kernel for the whole batch, on ONE side stream: six launches on three streams shared hardware queues
            // with the G2 reduction's long chain and the last of them came 1 ms late (kernel trace of round 5)
            if (!last_batch) {
                const MsmBuffers<Fq>* eb[4];
                const ScalarSort* es[4];
                for (int i = 0; i < npj; ++i) { eb[i] = pj[i].out; es[i] = pj[i].ss; }
                const int k0 = pk_of[0];
                G16_HIP_TRY(hipStreamWaitEvent(ctx->red[k0], b1, 0));
                if (batch_ok) {
                    G16_TRY((msm_heavy_reduce_batch<Fq>(eb, es, npj, ctx->red[k0])));
                } else {
                    for (int i = 0; i < npj; ++i) G16_TRY((msm_heavy_reduce<Fq>(*eb[i], *es[i], ctx->red[k0])));
                }
                G16_HIP_TRY(hipEventRecord(ctx->ev_heavy[k0], ctx->red[k0]));
                for (int i = 0; i < npj; ++i) { early[pk_of[i]] = true; early_ev[pk_of[i]] = k0; }
            }
            npj = 0;
            return G16_OK;
        };
        auto queue_pass = [&](int k, const G1A* bases, int64_t shift, uint64_t count, const ScalarSort& ss, MsmBuffers<Fq>* buf) {
            pj[npj] = PassJob<Fq>{bases, shift, count, &ss, buf};
            pk_of[npj++] = k;
            weight[k] = (double)count;
            jobs[njobs++] = {k, buf, &ss};
        };
        if (l_covered) {
            const int64_t shift = (int64_t)pk->a_start - (int64_t)(nin - 1) - (int64_t)pk->l_start;
            queue_pass(1, pk->l, shift, pk->l_count, sort_z, &buf_l);
        } else {
            G16_TRY((sort_scalars<C>(d_z + nin + pk->l_start, pk->l_count, pk->c_z, ctx->arena, s1, &sort_l, pk->shard_n, pk->shard_r)));
            last_end = nullptr;   // the sort sits between the passes: this one gets its own begin mark
            queue_pass(1, pk->l, 0, pk->l_count, sort_l, &buf_l);
            G16_TRY(run_batch());
        }
        { RoctxRange rr("Compute A");                                                                        // prover.rs:89
        queue_pass(2, pk->a, 0, pk->a_count, sort_z, &buf_a); }                                              // prover.rs:92
        if (!skip_b_g1) {                                                                                    // prover.rs:98-108
            RoctxRange rr("Compute B in G1");                                                                 // prover.rs:99
            queue_pass(3, pk->b_g1, 0, pk->b_g1_count, sort_z, &buf_b1);
        }
        if (!h_in_batch) G16_TRY(run_batch());
        // ---- h_acc = msm(h_query, h) (prover.rs:63-66): needs the witness map
        G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_h, 0));
        if (!h_in_batch) last_end = nullptr;       // whatever stream 1 waits for here is not the h pass
        { RoctxRange rr("Compute C");                                                                        // prover.rs:62 (h_acc; l_acc above)
        queue_pass(0, pk->h, 0, pk->h_count, sort_h, &buf_h);
        last_batch = true;
        G16_TRY(run_batch()); }
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
            bool any_early = false;
            for (int q = 0; q < nb; ++q) any_early = any_early || early[jobs[idx[q]].k];
            for (int q = 0; q < nb; ++q) {
                if (early[jobs[idx[q]].k]) G16_HIP_TRY(hipStreamWaitEvent(s1, ctx->ev_heavy[early_ev[jobs[idx[q]].k]], 0));
                else if (any_early) G16_TRY((msm_heavy_reduce<Fq>(*bb[q], *sp[q], s1)));   // mixed group: this one's first stage here
            }
            G16_TRY((msm_reduce_batch<Fq>(bb, sp, nb, s1, /*heavy_done=*/any_early)));
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
        if (overlap_map) G16_HIP_TRY(hipStreamSynchronize(ctx->stream_wm));
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
            // MSMs that went as one launch share its interval: each gets the part its points are of the launch's
            double wsum = 0.0;
            for (int q = 0; q < 4; ++q) if (pass_begin[q] == pass_begin[i] && pass_end[q] == pass_end[i]) wsum += weight[q];
            if (i < 4 && wsum > 0.0) tm.bucket_ms[i] *= weight[i] / wsum;
            tm.bucket_pass_ms += tm.bucket_ms[i];
        }
        tm.g1_pass_launches = (double)g1_launches;
        tm.finish_ms = fold_ms;
        tm.total_ms = t_end - t_begin;
        tm.window_bits = sort_z.plan.c;
        tm.windows = sort_z.plan.W;
        return G16_OK;
    }

    // prover.rs:76-131 glue over the summed MSM results
};

}  // namespace
