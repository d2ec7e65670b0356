"""SURVEY.md row f2 -- the reference's own test pattern (src/test.rs:45-73: prove, then assert verify(correct input) and
!verify(wrong input)) with a plain Tate-pairing verifier restating src/verifier.rs:13-77 in the big-int model.  Independent
of the trapdoor check: here the proof is accepted by the Groth16 verification equation itself."""
import numpy as np
import pytest

import pymodel as pm
from helpers import arr_to_g1, arr_to_g2, ints_to_mont, mont_to_ints

CURVES = [pm.BLS12_381, pm.BN254]


def _vk_from_oracle(cp, pk, ex):
    """ProvingKey carrying the vk fields verify_proof reads, from the oracle's flat setup output"""
    return pm.ProvingKey(arr_to_g1(pk.alpha_g1, cp)[0], None, arr_to_g2(pk.beta_g2, cp)[0], None, arr_to_g2(pk.delta_g2, cp)[0],
                         arr_to_g2(ex["gamma_g2"], cp)[0], arr_to_g1(ex["gamma_abc"], cp), [], [], [], [], [])


def _proof_from_flat(cp, flat):
    L = cp.fq_limbs64
    return pm.Proof(arr_to_g1(flat[: 2 * L], cp)[0], arr_to_g2(flat[2 * L: 6 * L], cp)[0], arr_to_g1(flat[6 * L:], cp)[0])


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_pairing_properties(cp):
    G1, G2 = pm.groups(cp)
    F = pm.Fq12(cp)
    e = pm.tate_pairing(cp, cp.g1, cp.g2)
    assert e != F.one and F.pow(e, cp.r) == F.one
    a, b = 0x1234567, 0x89ABCDE
    assert pm.tate_pairing(cp, G1.mul(cp.g1, a), G2.mul(cp.g2, b)) == F.pow(e, a * b % cp.r)
    assert pm.tate_pairing(cp, None, cp.g2) == F.one


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_pymodel_prove_and_verify(cp):
    cs, z = pm.mimc_circuit(cp, 4, 3)
    pk, _ = pm.generate_parameters(cp, cs, 5)
    r, s = pm.SplitMix64(1).field(cp.r), pm.SplitMix64(2).field(cp.r)
    proof = pm.create_proof_with_reduction_and_matrices(cp, pk, r, s, cs, z)
    public = z[1: cs.num_inputs]
    assert pm.verify_proof(cp, pk, proof, public)
    assert not pm.verify_proof(cp, pk, proof, [(public[0] + 1) % cp.r])       # src/test.rs:71
    assert not pm.verify_proof(cp, pk, pm.Proof(proof.a, proof.b, proof.a), public)
    with pytest.raises(ValueError):                                             # MalformedVerifyingKey, verifier.rs:29-31
        pm.verify_proof(cp, pk, proof, public + [1])


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_oracle_proof_verifies(orc, cp):
    ck = orc.syn_circuit(cp.name, 4, 9)
    pk, ex = orc.setup(ck, 3)
    r, s = orc.rand_fr(cp.name, 5, 1)[0], orc.rand_fr(cp.name, 6, 1)[0]
    flat, _ = orc.prove(pk, ck, r, s)
    vk = _vk_from_oracle(cp, pk, ex)
    public = mont_to_ints(ck.z[1: ck.num_inputs], cp.r)
    assert pm.verify_proof(cp, vk, _proof_from_flat(cp, flat), public)
    assert not pm.verify_proof(cp, vk, _proof_from_flat(cp, flat), [(public[0] + 1) % cp.r])


@pytest.mark.gpu
@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_gpu_proof_verifies(orc, cp):
    """the GPU prover's proof passes the pairing check on the right public input and fails on a wrong one"""
    import groth16_amd as g

    ck = orc.syn_circuit(cp.name, 7, 11)
    pk, ex = orc.setup(ck, 4)
    mats = g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])
    gpk = g.ProvingKey(cp.name, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                       pk.h_query, pk.l_query)
    with g.Groth16(cp.name, 0) as prover:
        proof = prover.create_random_proof_with_reduction(gpk, mats, ck.num_inputs, ck.num_constraints, ck.z)
        proof2 = prover.create_random_proof_with_reduction(gpk, mats, ck.num_inputs, ck.num_constraints, ck.z)
    assert not (proof.flat() == proof2.flat()).all()  # fresh r, s each time (prover.rs:146-147)
    vk = _vk_from_oracle(cp, pk, ex)
    public = mont_to_ints(ck.z[1: ck.num_inputs], cp.r)
    for pr in (proof, proof2):
        assert pm.verify_proof(cp, vk, _proof_from_flat(cp, pr.flat()), public)
    assert not pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), [(public[0] + 1) % cp.r])


@pytest.mark.gpu
def test_mimc322_prove_and_verify_gpu(orc):
    """BASELINE.json configs[0]: the reference's integration test shape (tests/mimc.rs:145-229 -- MiMC with 322 rounds, 644
    constraints, one public input; BLS12-377 there, BLS12-381 here as the config asks): setup, GPU proof with fresh r, s,
    pairing verification on the image, rejection of a wrong image; the proof also equals the CPU oracle's bit for bit."""
    import groth16_amd as g
    from helpers import circuit_from_pymodel

    cp = pm.BLS12_381
    cs, z = pm.mimc_circuit(cp, 322, 5)
    assert cs.num_constraints == 644 and cs.num_inputs == 2
    ck = circuit_from_pymodel(cp, cs, z)
    pk, ex = orc.setup(ck, 6)
    mats = g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])
    gpk = g.ProvingKey(cp.name, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                       pk.h_query, pk.l_query)
    r, s = orc.rand_fr(cp.name, 8, 1)[0], orc.rand_fr(cp.name, 9, 1)[0]
    with g.Groth16(cp.name, 0) as prover:
        proof = prover.create_proof_with_reduction_and_matrices(gpk, r, s, mats, ck.num_inputs, ck.num_constraints, ck.z)
    assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
    vk = _vk_from_oracle(cp, pk, ex)
    image = z[1:2]
    assert pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), image)
    assert not pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), [(image[0] + 1) % cp.r])
