#!/usr/bin/env python3
"""Staged GPU diagnostics: every stage runs in its own try/except and prints PASS/FAIL so that one
gpurun call yields as much information as possible.  Test infrastructure (uses the oracle)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402

import groth16_amd as g  # noqa: E402
from helpers import oracle  # noqa: E402

orc = oracle()
results = []


def stage(name):
    def deco(fn):
        t0 = time.time()
        try:
            msg = fn()
            results.append((name, "PASS", msg or "", time.time() - t0))
            print(f"[PASS] {name} {msg or ''} ({time.time() - t0:.2f}s)", flush=True)
        except Exception as e:  # noqa: BLE001
            results.append((name, "FAIL", repr(e), time.time() - t0))
            print(f"[FAIL] {name}: {e!r}", flush=True)
            traceback.print_exc()
        return fn
    return deco


def mats_of(ck):
    return g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])


def pk_of(pk):
    return g.ProvingKey(pk.curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                        pk.h_query, pk.l_query)


big = "--big" in sys.argv
for curve in ("bls12_381", "bn254"):
    prover = g.Groth16(curve, 0)

    for log_n in (0, 1, 2, 5, 10, 11, 12, 14, 17) + ((20,) if big else ()):
        @stage(f"{curve} ntt log_n={log_n}")
        def _():
            x = orc.rand_fr(curve, log_n + 1, 1 << log_n)
            bad = []
            for inv in (False, True):
                for cos in (False, True):
                    got = prover.ntt(x, inv, cos)
                    want = orc.ntt(curve, x, inv, cos)
                    if not (got == want).all():
                        bad.append((inv, cos, int((got != want).any(axis=1).sum())))
            assert not bad, f"mismatch (inverse, coset, #bad): {bad}"

    for k in (2, 3, 6, 10, 12, 15):
        @stage(f"{curve} witness_map SYN({k})")
        def _():
            ck = orc.syn_circuit(curve, k, 7)
            h = prover.witness_map_from_matrices(mats_of(ck), ck.num_inputs, ck.num_constraints, ck.z)
            want = orc.witness_map(ck)
            nbad = int((h != want).any(axis=1).sum())
            assert nbad == 0, f"{nbad} of {len(h)} differ"

    for g2 in (False, True):
        for n in (0, 1, 2, 33, 1000, 5000, 1 << 14) + ((1 << 18,) if (big and not g2) else ()):
            @stage(f"{curve} msm {'g2' if g2 else 'g1'} n={n}")
            def _():
                bases = orc.synth_bases(curve, g2, 3, max(n, 1))[:n]
                sc = orc.rand_fr(curve, 5 + n, max(n, 1))[:n]
                if n >= 33:
                    sc[0] = 0
                    bases[1] = 0
                    bases[3] = bases[2]
                    sc[3] = sc[2]
                t0 = time.time()
                got = prover.msm(bases, sc, g2)
                t1 = time.time()
                want = orc.msm(curve, g2, bases, sc) if n else np.zeros_like(got)
                assert (got == want).all(), "msm mismatch"
                return f"gpu {1e3 * (t1 - t0):.1f} ms, bucket {prover.timings()['bucket_pass_ms']:.2f} ms"

        @stage(f"{curve} msm {'g2' if g2 else 'g1'} all-equal scalars n=4096")
        def _():
            bases = orc.synth_bases(curve, g2, 3, 4096)
            sc = np.repeat(orc.rand_fr(curve, 9, 1), 4096, axis=0)
            assert (prover.msm(bases, sc, g2) == orc.msm(curve, g2, bases, sc)).all()

    for k in (3, 8, 12):
        @stage(f"{curve} prove SYN({k}) valid CRS + trapdoor")
        def _():
            ck = orc.syn_circuit(curve, k, 1)
            pk, ex = orc.setup(ck, 5)
            r, s = orc.rand_fr(curve, 11, 1)[0], orc.rand_fr(curve, 12, 1)[0]
            proof = prover.create_proof_with_reduction_and_matrices(pk_of(pk), r, s, mats_of(ck), ck.num_inputs, ck.num_constraints, ck.z)
            want, _ = orc.prove(pk, ck, r, s)
            assert (proof.flat() == want).all(), "proof != oracle"
            assert (orc.trapdoor_proof(ck, ex, orc.witness_map(ck), r, s) == want).all(), "oracle != trapdoor"

    for k in (14, 16) + ((18,) if big else ()):
        @stage(f"{curve} prove SYN({k}) synthetic bases")
        def _():
            ck = orc.syn_circuit(curve, k, 2)
            pk = orc.synth_pk(ck, 9)
            r, s = orc.rand_fr(curve, 21, 1)[0], orc.rand_fr(curve, 22, 1)[0]
            gm, gp = mats_of(ck), pk_of(pk)
            proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
            t0 = time.time()
            proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
            t1 = time.time()
            tm = prover.timings()
            t2 = time.time()
            want, otm = orc.prove(pk, ck, r, s)
            t3 = time.time()
            assert (proof.flat() == want).all(), "proof != oracle"
            return (f"gpu {1e3 * (t1 - t0):.1f} ms (wm {tm['witness_map_ms']:.2f} prep {tm['scalar_prep_ms']:.2f} h {tm['msm_h_ms']:.2f} "
                    f"l {tm['msm_l_ms']:.2f} a {tm['msm_a_ms']:.2f} b1 {tm['msm_b_g1_ms']:.2f} b2 {tm['msm_b_g2_ms']:.2f} "
                    f"bucket {tm['bucket_pass_ms']:.2f} finish {tm['finish_ms']:.2f}); cpu oracle {t3 - t2:.2f} s on {orc.threads} threads")

    prover.close()

nfail = sum(1 for r in results if r[1] == "FAIL")
print(f"\nSUMMARY: {len(results) - nfail} passed, {nfail} failed")
for name, st, msg, dt in results:
    if st == "FAIL":
        print("  FAIL", name, msg)
sys.exit(1 if nfail else 0)
