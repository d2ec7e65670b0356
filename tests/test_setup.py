"""SURVEY.md row f3 -- CRS generation (src/generator.rs:47-208).  The scalar half (instance_map_with_evaluation,
src/r1cs_to_qap.rs:120-170) runs on the host and is checked here without a GPU; the whole of g16_generate_parameters is
checked on the MI355X against the CPU oracle's setup on the same toxic waste, array by array and bit for bit, and then the
way the reference's own tests use a key (src/test.rs:45-73): generate, prove, verify, reject a wrong input."""
import ctypes as C

import numpy as np
import pytest

import pymodel as pm
from helpers import circuit_from_pymodel, ints_to_mont, mont_to_ints, ptr64

CURVES = [pm.BLS12_381, pm.BN254]
CURVE_ID = {"bls12_381": 0, "bn254": 1}


@pytest.fixture(scope="module")
def lb():
    from groth16_amd.binding import lib

    return lib()


def _views(ck):
    from groth16_amd.binding import CsrViewC, ptr32

    return (CsrViewC * 3)(*[CsrViewC(ptr64(m.row_ptr), ptr32(m.col), ptr64(m.val)) for m in ck.abc])


def _circuits(orc, cp):
    yield orc.syn_circuit(cp.name, 5, 3)
    yield circuit_from_pymodel(cp, *pm.mimc_circuit(cp, 5, 2))
    yield circuit_from_pymodel(cp, *pm.syn_circuit(cp, 3, 1, dense=True))   # multi-term rows, padded domain


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_qap_evaluations_match_oracle(lb, orc, cp):
    for ck in _circuits(orc, cp):
        _, ex = orc.setup(ck, 11)
        td = ex["trapdoor"]                       # alpha beta gamma delta t zt
        nv = ck.num_vars
        a, b, c = (np.zeros((nv, 4), dtype=np.uint64) for _ in range(3))
        zt = np.zeros(4, dtype=np.uint64)
        rc = lb.c.g16_host_qap_evaluations(CURVE_ID[cp.name], _views(ck), ck.num_inputs, ck.num_constraints, nv, ptr64(td[4].copy()),
                                           ptr64(a), ptr64(b), ptr64(c), ptr64(zt))
        assert rc == 0
        assert (a == ex["abc_t"][0]).all() and (b == ex["abc_t"][1]).all() and (c == ex["abc_t"][2]).all()
        assert (zt == td[5]).all()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_qap_evaluations_against_bigint_model(lb, cp):
    """independent of the C++ oracle: u_i = L_i(t) from pymodel's Domain, accumulated in Python integers"""
    cs, z = pm.mimc_circuit(cp, 3, 4)
    ck = circuit_from_pymodel(cp, cs, z)
    t = pm.SplitMix64(9).field(cp.r)
    dom = pm.Domain(cp, cs.num_constraints + cs.num_inputs)
    u = dom.lagrange_at(t)
    nv = ck.num_vars
    want = [[0] * nv for _ in range(3)]
    for j in range(cs.num_inputs):
        want[0][j] = u[cs.num_constraints + j]
    for which, rows in enumerate((cs.a, cs.b, cs.c)):
        for i, row in enumerate(rows):
            for coeff, idx in row:
                want[which][idx] = (want[which][idx] + u[i] * coeff) % cp.r
    a, b, c = (np.zeros((nv, 4), dtype=np.uint64) for _ in range(3))
    zt = np.zeros(4, dtype=np.uint64)
    rc = lb.c.g16_host_qap_evaluations(CURVE_ID[cp.name], _views(ck), ck.num_inputs, ck.num_constraints, nv,
                                       ptr64(ints_to_mont([t], cp.r, 4)[0].copy()), ptr64(a), ptr64(b), ptr64(c), ptr64(zt))
    assert rc == 0
    for got, w in zip((a, b, c), want):
        assert mont_to_ints(got, cp.r) == w
    assert mont_to_ints(zt[None, :], cp.r)[0] == (pow(t, dom.n, cp.r) - 1) % cp.r


def test_qap_evaluations_errors(lb, orc):
    cp = pm.BLS12_381
    ck = orc.syn_circuit(cp.name, 4, 1)
    nv = ck.num_vars
    bufs = [np.zeros((nv, 4), dtype=np.uint64) for _ in range(3)] + [np.zeros(4, dtype=np.uint64)]
    one = ints_to_mont([1], cp.r, 4)[0].copy()     # t = 1 = w^0 lies in the domain
    args = lambda ni, nc, t: (CURVE_ID[cp.name], _views(ck), ni, nc, nv, ptr64(t), *[ptr64(x) for x in bufs])  # noqa: E731
    assert lb.c.g16_host_qap_evaluations(*args(ck.num_inputs, ck.num_constraints, one)) == 3
    t = ints_to_mont([12345], cp.r, 4)[0].copy()
    assert lb.c.g16_host_qap_evaluations(*args(ck.num_inputs, ck.num_constraints, t)) == 0
    assert lb.c.g16_host_qap_evaluations(*args(0, ck.num_constraints, t)) == 2


# ---------------------------------------------------------------------------------------------------------------------
def _mats(g, ck):
    return g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])


@pytest.mark.gpu
@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_generate_parameters_matches_oracle(orc, cp):
    import groth16_amd as g

    with g.Groth16(cp.name, 0) as prover:
        for ck in list(_circuits(orc, cp)) + [orc.syn_circuit(cp.name, 10, 2)]:
            pk, ex = orc.setup(ck, 21)
            td = ex["trapdoor"]
            got = prover.generate_parameters_with_qap(_mats(g, ck), td[0], td[1], td[2], td[3], ex["g1gen"], ex["g2gen"], td[4])
            for name in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
                assert (getattr(got, name) == getattr(pk, name)).all(), name
            assert (got.gamma_g2.reshape(-1) == ex["gamma_g2"]).all()
            assert (got.gamma_abc_g1 == ex["gamma_abc"]).all()
        # SynthesisError::UnexpectedIdentity (generator.rs:110-111)
        zero = np.zeros(4, dtype=np.uint64)
        with pytest.raises(g.UnexpectedIdentity):
            prover.generate_parameters_with_qap(_mats(g, ck), td[0], td[1], zero, td[3], ex["g1gen"], ex["g2gen"], td[4])
        with pytest.raises(g.UnexpectedIdentity):
            prover.generate_parameters_with_qap(_mats(g, ck), td[0], td[1], td[2], zero, ex["g1gen"], ex["g2gen"], td[4])


@pytest.mark.gpu
@pytest.mark.parametrize("cp,k", [(pm.BLS12_381, 18), (pm.BN254, 16), (pm.BLS12_381, 22)], ids=["bls12_381-k18", "bn254-k16", "bls12_381-k22"])
def test_setup_prove_verify_on_gpu(orc, cp, k):
    """the reference's end-to-end pattern at sizes the CPU setup would need minutes for, up to BASELINE.json's full size
    (2^22 constraints: the merged-window MSM with 16 bucket classes, every kernel at its benchmark shape): key from
    g16_generate_parameters, proof from g16_prove with fresh r, s, accepted by the pairing verifier, rejected on a wrong
    public input"""
    import groth16_amd as g
    from test_verifier import _proof_from_flat
    from helpers import arr_to_g1, arr_to_g2

    ck = orc.syn_circuit(cp.name, k, 5)
    rnd = orc.rand_fr(cp.name, 77, 5)
    gens = orc.setup(orc.syn_circuit(cp.name, 2, 1), 3)[1]         # any pair of generators
    with g.Groth16(cp.name, 0) as prover:
        mats = _mats(g, ck)
        pk = prover.generate_parameters_with_qap(mats, rnd[0], rnd[1], rnd[2], rnd[3], gens["g1gen"], gens["g2gen"], rnd[4])
        proof = prover.create_random_proof_with_reduction(pk, mats, ck.num_inputs, ck.num_constraints, ck.z)
    vk = pm.ProvingKey(arr_to_g1(pk.alpha_g1, cp)[0], None, arr_to_g2(pk.beta_g2, cp)[0], None, arr_to_g2(pk.delta_g2, cp)[0],
                       arr_to_g2(pk.gamma_g2, cp)[0], arr_to_g1(pk.gamma_abc_g1, cp), [], [], [], [], [])
    public = mont_to_ints(ck.z[1: ck.num_inputs], cp.r)
    assert pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), public)
    assert not pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), [(public[0] + 1) % cp.r])


@pytest.mark.gpu
@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_generate_parameters_reproduces_golden_keys(cp):
    """the proving keys frozen in tests/golden/*.json (big-int model, checked there against the trapdoor closed form) come
    back array for array from g16_generate_parameters when it is given the model's toxic waste and generators"""
    import groth16_amd as g
    from helpers import g1_to_arr, g2_to_arr
    from test_golden import load_cases

    seeds = {"syn3": 7, "syn5_dense": 8, "mimc7": 9}      # tests/golden/make_golden.py
    fr = lambda v: ints_to_mont([v], cp.r, 4)[0]          # noqa: E731
    seen = set()
    with g.Groth16(cp.name, 0) as prover:
        for name, _, cs, z, _r, _s, pk, _ex in load_cases(cp.name):
            base = name.rsplit("_", 1)[0]
            if base in seen:
                continue
            seen.add(base)
            _, td = pm.generate_parameters(cp, cs, seeds[base])
            ck = circuit_from_pymodel(cp, cs, z)
            got = prover.generate_parameters_with_qap(_mats(g, ck), fr(td.alpha), fr(td.beta), fr(td.gamma), fr(td.delta),
                                                      g1_to_arr([td.g1], cp)[0], g2_to_arr([td.g2], cp)[0], fr(td.t))
            for field, conv in (("alpha_g1", g1_to_arr), ("beta_g1", g1_to_arr), ("delta_g1", g1_to_arr), ("beta_g2", g2_to_arr),
                                ("delta_g2", g2_to_arr)):
                assert (getattr(got, field).reshape(-1) == conv([getattr(pk, field)], cp)[0]).all(), (name, field)
            for field, conv in (("a_query", g1_to_arr), ("b_g1_query", g1_to_arr), ("b_g2_query", g2_to_arr), ("h_query", g1_to_arr),
                                ("l_query", g1_to_arr)):
                assert (getattr(got, field) == conv(getattr(pk, field), cp)).all(), (name, field)
    assert seen == set(seeds)
