"""Row a1: ``create_proof_with_reduction`` with host-side synthesis (src/prover.rs:173-217) and the reference's own
tests written against the product's API: src/test.rs:14-73 (MySillyCircuit: setup, prove, verify, reject a wrong input) and
tests/mimc.rs:64-229 (MiMC-322 preimage: BASELINE.json configs[0]).  The circuits below are the reference's circuits, statement
for statement, on groth16_amd.r1cs; verification uses the test infrastructure's pairing verifier (oracle/pymodel.py)."""
import random

import numpy as np
import pytest

import pymodel as pm
from helpers import arr_to_g1, arr_to_g2, mont_to_ints

CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}


class MySillyCircuit:
    """src/test.rs:14-43"""

    def __init__(self, a=None, b=None):
        self.a, self.b = a, b

    def generate_constraints(self, cs):
        from groth16_amd import lc

        a = cs.new_witness_variable(lambda: self.a)
        b = cs.new_witness_variable(lambda: self.b)
        c = cs.new_input_variable(lambda: None if self.a is None or self.b is None else self.a * self.b)
        for _ in range(6):
            cs.enforce_constraint(lc() + a, lc() + b, lc() + c)


MIMC_ROUNDS = 322


def mimc(xl, xr, constants, p):
    """tests/mimc.rs:46-62"""
    for c in constants:
        xl, xr = (xr + pow(xl + c, 3, p)) % p, xl
    return xl


class MiMCDemo:
    """tests/mimc.rs:64-143"""

    def __init__(self, xl, xr, constants, p):
        self.xl, self.xr, self.constants, self.p = xl, xr, constants, p

    def generate_constraints(self, cs):
        from groth16_amd import Variable, lc

        p = self.p
        xl_value, xr_value = self.xl, self.xr
        xl = cs.new_witness_variable(lambda: xl_value)
        xr = cs.new_witness_variable(lambda: xr_value)
        for i, ci in enumerate(self.constants):
            tmp_value = None if xl_value is None else pow(xl_value + ci, 2, p)
            tmp = cs.new_witness_variable(lambda: tmp_value)
            cs.enforce_constraint(lc() + xl + (ci, Variable.One), lc() + xl + (ci, Variable.One), lc() + tmp)
            new_xl_value = None if xl_value is None else ((xl_value + ci) * tmp_value + xr_value) % p
            if i == len(self.constants) - 1:
                new_xl = cs.new_input_variable(lambda: new_xl_value)
            else:
                new_xl = cs.new_witness_variable(lambda: new_xl_value)
            cs.enforce_constraint(lc() + tmp, lc() + xl + (ci, Variable.One), lc() + new_xl - xr)
            xr, xr_value = xl, xl_value
            xl, xl_value = new_xl, new_xl_value


def test_constraint_system_matches_the_model_on_mimc():
    """the product's synthesis layer produces the matrices and the assignment the big-int model builds for the same circuit"""
    from groth16_amd.r1cs import from_montgomery, synthesize

    cp = pm.BLS12_381
    rounds, seed = 17, 4
    cs_m, z_m = pm.mimc_circuit(cp, rounds, seed)
    rng = pm.SplitMix64(seed)
    consts = [rng.field(cp.r) for _ in range(rounds)]
    xl, xr = rng.field(cp.r), rng.field(cp.r)
    cs = synthesize(cp.name, MiMCDemo(xl, xr, consts, cp.r), setup_mode=False)
    assert cs.is_satisfied()
    assert (cs.num_instance_variables, cs.num_witness_variables, cs.num_constraints) == (cs_m.num_inputs, cs_m.num_witness, cs_m.num_constraints)
    assert cs.instance_assignment + cs.witness_assignment == z_m
    assert cs.instance_assignment[1] == mimc(xl, xr, consts, cp.r)
    m = cs.to_matrices()
    for (rp, col, val), rows in ((m.a, cs_m.a), (m.b, cs_m.b), (m.c, cs_m.c)):
        vals = from_montgomery(val, cp.r)
        for i, row in enumerate(rows):
            got = sorted((vals[k], int(col[k])) for k in range(int(rp[i]), int(rp[i + 1])))
            assert got == sorted((cf % cp.r, c) for cf, c in row), i
    # setup mode: same shape, closures never evaluated, no assignment
    cs0 = synthesize(cp.name, MiMCDemo(None, None, consts, cp.r), setup_mode=True)
    assert (cs0.num_instance_variables, cs0.num_witness_variables, cs0.num_constraints) == (cs.num_instance_variables, cs.num_witness_variables, cs.num_constraints)
    m0 = cs0.to_matrices()
    assert all((x == y).all() for a0, a1 in ((m0.a, m.a), (m0.b, m.b), (m0.c, m.c)) for x, y in zip(a0, a1))


def test_assignment_missing_and_unsatisfied():
    from groth16_amd import AssignmentMissing
    from groth16_amd.r1cs import synthesize

    with pytest.raises(AssignmentMissing):      # SynthesisError::AssignmentMissing, src/test.rs:24
        synthesize("bn254", MySillyCircuit(None, 3), setup_mode=False)
    cs = synthesize("bn254", MySillyCircuit(3, 5), setup_mode=False)
    assert cs.is_satisfied() and cs.instance_assignment == [1, 15]
    cs.instance_assignment[1] = 16
    assert not cs.is_satisfied()


def test_generator_constants_match_the_model():
    from groth16_amd.groth16 import _GENERATORS, _MODULUS_Q, _MODULUS_R

    for name, cp in CP.items():
        assert _GENERATORS[name] == (cp.g1, cp.g2) and _MODULUS_Q[name] == cp.q and _MODULUS_R[name] == cp.r


def _vk(cp, pk):
    return pm.ProvingKey(arr_to_g1(pk.alpha_g1, cp)[0], None, arr_to_g2(pk.beta_g2, cp)[0], None, arr_to_g2(pk.delta_g2, cp)[0],
                         arr_to_g2(pk.gamma_g2, cp)[0], arr_to_g1(pk.gamma_abc_g1, cp), [], [], [], [], [])


def _proof(cp, proof):
    return pm.Proof(arr_to_g1(proof.a[None, :], cp)[0], arr_to_g2(proof.b[None, :], cp)[0], arr_to_g1(proof.c[None, :], cp)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_prove_and_verify_my_silly_circuit(curve):
    """src/test.rs:45-73 (test_prove_and_verify): setup once, then fresh a, b, prove, verify([c]) and !verify([a])"""
    import groth16_amd as g

    cp = CP[curve]
    rng = random.Random(20240)
    with g.Groth16(curve, 0) as prover:
        pk, vk = prover.setup(MySillyCircuit(), rng)
        for _ in range(3):
            a, b = rng.randrange(cp.r), rng.randrange(cp.r)
            proof = prover.prove(pk, MySillyCircuit(a, b), rng)
            assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [a * b % cp.r])
            assert not pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [a])
        # the reference's own signatures on the reference's own names: (circuit, pk, rng) / (circuit, pk)
        proof = prover.create_random_proof_with_reduction(MySillyCircuit(3, 4), pk, rng)
        assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [12])
        assert len(prover._cks) == 1            # one upload of the matrices for all these proofs
        zk = prover.create_proof_with_reduction_no_zk(MySillyCircuit(7, 9), pk)
        assert zk == prover.create_proof_no_zk(MySillyCircuit(7, 9), pk)      # r = s = 0: deterministic (prover.rs:155-168)
        assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, zk), [63])


class MySillyCircuitTwice(MySillyCircuit):
    """another circuit over the same variables: twelve constraints instead of six"""

    def generate_constraints(self, cs):
        from groth16_amd import lc

        a = cs.new_witness_variable(lambda: self.a)
        b = cs.new_witness_variable(lambda: self.b)
        c = cs.new_input_variable(lambda: None if self.a is None or self.b is None else self.a * self.b)
        for _ in range(12):
            cs.enforce_constraint(lc() + a, lc() + b, lc() + c)


@pytest.mark.gpu
def test_circuit_id_names_one_circuit():
    """create_proof_with_reduction(..., circuit_id=...) skips hashing the matrices: the id IS the cache key.  The same id with another
    circuit must be refused (it would prove against the cached, wrong device matrices and return an invalid proof silently);
    the same id with the same circuit proves, uploads the matrices once, and verifies"""
    import groth16_amd as g

    cp = CP["bls12_381"]
    rng = random.Random(5)
    with g.Groth16("bls12_381", 0) as prover:
        pk, vk = prover.setup(MySillyCircuit(), rng)
        r, s = g.groth16._rand_fr("bls12_381", rng), g.groth16._rand_fr("bls12_381", rng)
        for a, b in ((3, 5), (7, 11)):
            proof = prover.create_proof_with_reduction(MySillyCircuit(a, b), pk, r, s, circuit_id="silly")
            assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [a * b])
        assert len(prover._cks) == 1
        with pytest.raises(ValueError, match="one id must mean one circuit"):
            prover.create_proof_with_reduction(MySillyCircuitTwice(3, 5), pk, r, s, circuit_id="silly")


@pytest.mark.gpu
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_rerandomize(curve):
    """src/test.rs:75-118 (test_rerandomize): a rerandomised proof (and its rerandomisation) verifies when the original does,
    fails on a wrong input, and the three differ as group elements"""
    import groth16_amd as g

    cp = CP[curve]
    rng = random.Random(99)
    with g.Groth16(curve, 0) as prover:
        pk, vk = prover.setup(MySillyCircuit(), rng)
        for _ in range(2):
            a, b = rng.randrange(cp.r), rng.randrange(cp.r)
            proof1 = prover.prove(pk, MySillyCircuit(a, b), rng)
            proof2 = prover.rerandomize_proof(vk, proof1, rng)
            proof3 = prover.rerandomize_proof(vk, proof2, rng)
            for pr in (proof1, proof2, proof3):
                assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, pr), [a * b % cp.r])
                assert not pm.verify_proof(cp, _vk(cp, vk), _proof(cp, pr), [a])
            assert not (proof1 == proof2) and not (proof1 == proof3) and not (proof2 == proof3)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_rerandomize_proof_of_an_oracle_proof(orc, curve):
    """prover.rs:223-250 through the product's host code on a CPU-oracle proof (no GPU): the result verifies, differs from the
    input, and rerandomising again still verifies (src/test.rs:98-116)"""
    import groth16_amd as g

    cp = CP[curve]
    ck = orc.syn_circuit(curve, 4, 21)
    pk, ex = orc.setup(ck, 2)
    flat, _ = orc.prove(pk, ck, orc.rand_fr(curve, 3, 1)[0], orc.rand_fr(curve, 4, 1)[0])
    L = cp.fq_limbs64
    proof1 = g.Proof(flat[: 2 * L].copy(), flat[2 * L: 6 * L].copy(), flat[6 * L:].copy())
    vk = g.ProvingKey(curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                      pk.h_query, pk.l_query, gamma_g2=ex["gamma_g2"], gamma_abc_g1=ex["gamma_abc"])
    rng = random.Random(5)
    proof2 = g.rerandomize_proof(curve, vk, proof1, rng)
    proof3 = g.rerandomize_proof(curve, vk, proof2, rng)
    public = mont_to_ints(ck.z[1: ck.num_inputs], cp.r)
    for pr in (proof1, proof2, proof3):
        assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, pr), public)
        assert not pm.verify_proof(cp, _vk(cp, vk), _proof(cp, pr), [(public[0] + 1) % cp.r])
    assert not (proof1 == proof2) and not (proof2 == proof3) and not (proof1 == proof3)


def test_rerandomize_host_math_against_the_model():
    """the same three formulas in the big-int model on a model proof (no GPU): rerandomised proofs verify"""
    cp = pm.BN254
    G1, G2 = pm.groups(cp)
    cs, z = pm.mimc_circuit(cp, 3, 2)
    pk, _ = pm.generate_parameters(cp, cs, 5)
    proof = pm.create_proof_with_reduction_and_matrices(cp, pk, 11, 13, cs, z)
    r1, r2 = 0x1234567, 0x7654321
    new = pm.Proof(G1.mul(proof.a, pow(r1, -1, cp.r)), G2.add(G2.mul(proof.b, r1), G2.mul(pk.delta_g2, r1 * r2 % cp.r)),
                   G1.add(proof.c, G1.mul(proof.a, r2)))
    assert pm.verify_proof(cp, pk, new, z[1: cs.num_inputs]) and new.a != proof.a


@pytest.mark.gpu
def test_mimc_groth16():
    """tests/mimc.rs:145-229 on BLS12-381 (BASELINE.json configs[0]): parameters from the circuit without values, proofs of
    fresh preimages, verification on the image, rejection of another image"""
    import groth16_amd as g

    cp = pm.BLS12_381
    rng = random.Random(7)
    constants = [rng.randrange(cp.r) for _ in range(MIMC_ROUNDS)]
    with g.Groth16(cp.name, 0) as prover:
        pk, vk = prover.setup(MiMCDemo(None, None, constants, cp.r), rng)
        assert len(pk.a_query) == 2 + 2 * MIMC_ROUNDS + 1 and len(pk.h_query) == 1023      # 644 constraints + 2 -> domain 1024
        for _ in range(2):
            xl, xr = rng.randrange(cp.r), rng.randrange(cp.r)
            image = mimc(xl, xr, constants, cp.r)
            proof = prover.prove(pk, MiMCDemo(xl, xr, constants, cp.r), rng)
            assert pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [image])
            assert not pm.verify_proof(cp, _vk(cp, vk), _proof(cp, proof), [(image + 1) % cp.r])


def test_block_distribution_indices_match_the_model():
    """the product's h_query gather order (dist_h_indices) is the block distribution the modelled distributed witness map leaves
    h in, and the shards tile [0, n) exactly"""
    from groth16_amd.groth16 import dist_h_indices

    cp = pm.BN254
    cs, z = pm.syn_circuit(cp, 6, 3)          # domain 64
    for N in (2, 4, 8):
        pieces, idx = pm.distributed_witness_map(cp, cs, z, N)
        h = pm.witness_map_from_matrices(cp, cs, z)
        seen = []
        for r in range(N):
            mine = dist_h_indices(64, r, N)
            assert list(mine) == idx[r]
            assert [h[i] for i in mine] == pieces[r]
            seen += list(mine)
        assert sorted(seen) == list(range(64))
