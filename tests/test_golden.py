"""Frozen golden vectors (tests/golden/*.json, produced by tests/golden/make_golden.py from the big-int model and checked
there against the trapdoor closed form).  CPU: the C++ oracle reproduces them.  GPU (-m gpu): the HIP path reproduces them."""
import json
import os

import numpy as np
import pytest

import pymodel as pm
from helpers import (FlatCircuit, arr_to_g1, arr_to_g2, circuit_from_pymodel, ints_to_mont, mont_to_ints, pk_from_pymodel)

HERE = os.path.dirname(os.path.abspath(__file__))
CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}


def _i(x):
    return int(x, 16)


def _p1(P):
    return None if P is None else (_i(P[0]), _i(P[1]))


def _p2(P):
    return None if P is None else ((_i(P[0][0]), _i(P[0][1])), (_i(P[1][0]), _i(P[1][1])))


def load_cases(curve):
    d = json.load(open(os.path.join(HERE, "golden", f"{curve}.json")))
    cp = CP[curve]
    for c in d["cases"]:
        rows = lambda m: [[(_i(cf), idx) for cf, idx in row] for row in m]  # noqa: E731
        cs = pm.R1CS(c["num_inputs"], c["num_witness"], rows(c["a"]), rows(c["b"]), rows(c["c"]))
        z = [_i(v) for v in c["z"]]
        k = c["pk"]
        pk = pm.ProvingKey(_p1(k["alpha_g1"]), _p1(k["beta_g1"]), _p2(k["beta_g2"]), _p1(k["delta_g1"]), _p2(k["delta_g2"]), None, [],
                           [_p1(p) for p in k["a_query"]], [_p1(p) for p in k["b_g1_query"]], [_p2(p) for p in k["b_g2_query"]],
                           [_p1(p) for p in k["h_query"]], [_p1(p) for p in k["l_query"]])
        yield c["name"], cp, cs, z, _i(c["r"]), _i(c["s"]), pk, c["expect"]


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_oracle_reproduces_golden(orc, curve):
    n = 0
    for name, cp, cs, z, r, s, pk, ex in load_cases(curve):
        ck, fpk = circuit_from_pymodel(cp, cs, z), pk_from_pymodel(cp, pk)
        ra, sa = ints_to_mont([r], cp.r, 4)[0], ints_to_mont([s], cp.r, 4)[0]
        proof, h, parts, _ = orc.prove(fpk, ck, ra, sa, want_parts=True)
        L = cp.fq_limbs64
        assert mont_to_ints(h, cp.r) == [_i(v) for v in ex["h"]], name
        assert arr_to_g1(parts[: 2 * L], cp)[0] == _p1(ex["h_acc"]) and arr_to_g1(parts[2 * L: 4 * L], cp)[0] == _p1(ex["l_acc"])
        assert arr_to_g1(parts[4 * L: 6 * L], cp)[0] == _p1(ex["a_msm"]) and arr_to_g1(parts[6 * L: 8 * L], cp)[0] == _p1(ex["b1_msm"])
        assert arr_to_g2(parts[8 * L:], cp)[0] == _p2(ex["b2_msm"])
        assert arr_to_g1(proof[: 2 * L], cp)[0] == _p1(ex["proof_a"]), name
        assert arr_to_g2(proof[2 * L: 6 * L], cp)[0] == _p2(ex["proof_b"]), name
        assert arr_to_g1(proof[6 * L:], cp)[0] == _p1(ex["proof_c"]), name
        n += 1
    assert n == 9


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_gpu_reproduces_golden(curve):
    import groth16_amd as g

    with g.Groth16(curve, 0) as prover:
        for name, cp, cs, z, r, s, pk, ex in load_cases(curve):
            ck, fpk = circuit_from_pymodel(cp, cs, z), pk_from_pymodel(cp, pk)
            mats = g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])
            gpk = g.ProvingKey(curve, fpk.alpha_g1, fpk.beta_g1, fpk.delta_g1, fpk.beta_g2, fpk.delta_g2, fpk.a_query, fpk.b_g1_query,
                               fpk.b_g2_query, fpk.h_query, fpk.l_query)
            h = prover.witness_map_from_matrices(mats, ck.num_inputs, ck.num_constraints, ck.z)
            assert mont_to_ints(h, cp.r) == [_i(v) for v in ex["h"]], name
            proof = prover.create_proof_with_reduction_and_matrices(gpk, ints_to_mont([r], cp.r, 4)[0], ints_to_mont([s], cp.r, 4)[0], mats,
                                                                    ck.num_inputs, ck.num_constraints, ck.z)
            assert arr_to_g1(proof.a, cp)[0] == _p1(ex["proof_a"]) and arr_to_g2(proof.b, cp)[0] == _p2(ex["proof_b"]), name
            assert arr_to_g1(proof.c, cp)[0] == _p1(ex["proof_c"]), name
            l_acc = prover.msm(fpk.l_query, ck.z[ck.num_inputs:])
            assert arr_to_g1(l_acc, cp)[0] == _p1(ex["l_acc"]), name
