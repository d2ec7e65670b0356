#!/usr/bin/env python3
"""Generates the golden fixtures in this directory from the big-int model (oracle/pymodel.py).

The reference (ark-groth16, Rust) holds no golden vectors and cannot be built or imported in this environment, so these
vectors are NOT reference outputs: they are frozen outputs of the independent big-int model, each additionally checked at
generation time against the known-trapdoor closed form (expected A, B, C as scalar * generator).  They pin the C++ oracle,
the product's host code and the GPU path against silent drift.  Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import pymodel as pm  # noqa: E402


def hx(v):
    return hex(v)


def pt1(P):
    return None if P is None else [hx(P[0]), hx(P[1])]


def pt2(P):
    return None if P is None else [[hx(P[0][0]), hx(P[0][1])], [hx(P[1][0]), hx(P[1][1])]]


def main():
    pm.selfcheck()
    for cp in (pm.BLS12_381, pm.BN254):
        cases = []
        for name, (cs, z), seed in (("syn3", pm.syn_circuit(cp, 3, 0), 7), ("syn5_dense", pm.syn_circuit(cp, 5, 1, dense=True), 8),
                                    ("mimc7", pm.mimc_circuit(cp, 7, 2), 9)):
            assert pm.is_satisfied(cs, z, cp.r)
            pk, td = pm.generate_parameters(cp, cs, seed)
            for tag, r, s in (("rs", pm.SplitMix64(seed + 100).field(cp.r), pm.SplitMix64(seed + 200).field(cp.r)), ("r0", 0, 5), ("s0", 9, 0)):
                parts = {}
                pr = pm.create_proof_with_reduction_and_matrices(cp, pk, r, s, cs, z, parts)
                ex = pm.trapdoor_expected_proof(cp, cs, td, z, r, s, parts["h"])
                assert (pr.a, pr.b, pr.c) == (ex.a, ex.b, ex.c), "trapdoor KAT failed"
                cases.append(dict(
                    name=f"{name}_{tag}", num_inputs=cs.num_inputs, num_witness=cs.num_witness,
                    a=[[[hx(c), i] for c, i in row] for row in cs.a], b=[[[hx(c), i] for c, i in row] for row in cs.b],
                    c=[[[hx(c), i] for c, i in row] for row in cs.c], z=[hx(v) for v in z], r=hx(r), s=hx(s),
                    pk=dict(alpha_g1=pt1(pk.alpha_g1), beta_g1=pt1(pk.beta_g1), beta_g2=pt2(pk.beta_g2), delta_g1=pt1(pk.delta_g1),
                            delta_g2=pt2(pk.delta_g2), a_query=[pt1(p) for p in pk.a_query], b_g1_query=[pt1(p) for p in pk.b_g1_query],
                            b_g2_query=[pt2(p) for p in pk.b_g2_query], h_query=[pt1(p) for p in pk.h_query],
                            l_query=[pt1(p) for p in pk.l_query]),
                    expect=dict(h=[hx(v) for v in parts["h"]], h_acc=pt1(parts["h_acc"]), l_acc=pt1(parts["l_acc"]), a_msm=pt1(parts["a_msm"]),
                                b1_msm=pt1(parts["b1_msm"]), b2_msm=pt2(parts["b2_msm"]), proof_a=pt1(pr.a), proof_b=pt2(pr.b),
                                proof_c=pt1(pr.c), proof_bytes_unverified_encoding=pm.proof_bytes(cp, pr).hex())))
        with open(os.path.join(HERE, f"{cp.name}.json"), "w") as f:
            json.dump(dict(curve=cp.name, generator="tests/golden/make_golden.py", cases=cases), f, separators=(",", ":"))
        print(cp.name, len(cases), "cases")


if __name__ == "__main__":
    main()
