#!/usr/bin/env python3
"""Adversarial scalar distributions at scale (correctness + graceful degradation), GPU vs oracle."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import groth16_amd as g
from helpers import oracle, ints_to_mont
import pymodel as pm
orc = oracle(); orc.set_threads(32)
curve = "bls12_381"; cp = pm.BLS12_381
prover = g.Groth16(curve, 0)
n = 1 << 18
bases = np.tile(orc.synth_bases(curve, False, 2, 1 << 14), (n >> 14, 1))
bases2 = np.tile(orc.synth_bases(curve, True, 2, 1 << 12), (n >> 12, 1))
uni = np.tile(orc.rand_fr(curve, 1, 1 << 14), (n >> 14, 1))
cases = {
    "uniform": uni,
    "all-equal (benches/bench.rs:52-54 shape)": np.repeat(orc.rand_fr(curve, 3, 1), n, axis=0),
    "all-one": np.repeat(ints_to_mont([1], cp.r, 4), n, axis=0),
    "all r-1": np.repeat(ints_to_mont([cp.r - 1], cp.r, 4), n, axis=0),
    "two values": np.tile(orc.rand_fr(curve, 4, 2), (n // 2, 1)),
    "boolean 0/1 witness": np.ascontiguousarray(ints_to_mont([0, 1], cp.r, 4)[np.random.default_rng(0).integers(0, 2, n)]),
}
for name, sc in cases.items():
    for g2, b in ((False, bases), (True, bases2)):
        prover.msm(b, sc, g2)
        t0 = time.time(); got = prover.msm(b, sc, g2); dt = time.time() - t0
        want = orc.msm(curve, g2, b, sc)
        print(f"{'G2' if g2 else 'G1'} n=2^18 {name:45s} gpu {1e3*dt:8.1f} ms (bucket {prover.timings()['bucket_pass_ms']:7.2f})  match={bool((got==want).all())}", flush=True)
prover.close()
