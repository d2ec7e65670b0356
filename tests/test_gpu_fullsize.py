"""Bit-exact parity at BASELINE.json's sizes (configs #2, #3, #4) and the reference bench's own circuit shape.

A verifying proof is not a bit-exact proof: `north_star` asks for proof bytes equal to the reference CPU prover's on the same
ProvingKey and witness (src/prover.rs:54-132).  At 2^20 / 2^22 constraints the 16-class merged-window MSM, the u32 entry tags,
the multi-gigabyte arena and the three-sweep NTT exist only at these sizes, so they are checked here directly:

  * witness map: GPU h == oracle h, all 2^k coefficients (src/r1cs_to_qap.rs:172-235)
  * trapdoor closed form: the key comes from g16_generate_parameters with KNOWN toxic waste; the expected proof is
    A = (alpha + sum z_i a_i(t) + r delta) G, B = (beta + sum z_i b_i(t) + s delta) H, C = (...) G computed by the oracle with
    scalar arithmetic and three fixed-base multiplications -- no MSM, no NTT, no key -- and must equal the GPU proof
  * oracle proof (k = 20): orc_prove = the C++ restatement of create_proof_with_reduction_and_matrices on the SAME key
    (downloaded) and witness, proof and the five raw MSM results (prover.rs:66,74,92,105,113) compared one by one

All calls go through the C ABI (groth16_amd host mirror).  Run with -m gpu on the MI355X box.
"""
import ctypes as C

import numpy as np
import pytest

import pymodel as pm
from helpers import FlatCircuit, FlatPk, Csr, FQ_LIMBS, mont_to_ints, ptr64

pytestmark = pytest.mark.gpu

CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}
CURVE_ID = {"bls12_381": 0, "bn254": 1}


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import groth16_amd

    return groth16_amd


def mats_of(g, ck):
    return g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])


def flat_pk(curve, pk):
    cc = np.ascontiguousarray
    return FlatPk(curve, cc(pk.alpha_g1.reshape(1, -1)), cc(pk.beta_g1.reshape(1, -1)), cc(pk.delta_g1.reshape(1, -1)), cc(pk.beta_g2.reshape(1, -1)),
                  cc(pk.delta_g2.reshape(1, -1)), cc(pk.a_query), cc(pk.b_g1_query), cc(pk.b_g2_query), cc(pk.h_query), cc(pk.l_query))


def trapdoor_extras(g, curve, ck, toxic, g1gen, g2gen):
    """what Oracle.trapdoor_proof needs: (alpha beta gamma delta t Z(t)) and a_i(t), b_i(t), c_i(t) -- the latter from the
    library's HOST implementation of instance_map_with_evaluation (r1cs_to_qap.rs:120-170; checked against the oracle and the
    big-int model in test_setup.py), which shares nothing with the prover's MSM / NTT kernels"""
    from groth16_amd.binding import CsrViewC, lib, ptr32

    nv = ck.num_vars
    abc_t = np.zeros((3, nv, 4), dtype=np.uint64)
    zt = np.zeros(4, dtype=np.uint64)
    views = (CsrViewC * 3)(*[CsrViewC(ptr64(m.row_ptr), ptr32(m.col), ptr64(m.val)) for m in ck.abc])
    rc = lib().c.g16_host_qap_evaluations(CURVE_ID[curve], views, ck.num_inputs, ck.num_constraints, nv, ptr64(np.ascontiguousarray(toxic[4])),
                                          ptr64(abc_t[0]), ptr64(abc_t[1]), ptr64(abc_t[2]), ptr64(zt))
    assert rc == 0
    td = np.zeros((6, 4), dtype=np.uint64)
    td[:5] = toxic[:5]
    td[5] = zt
    return dict(trapdoor=td, abc_t=abc_t, g1gen=g1gen, g2gen=g2gen)


def xyzz_to_affine(cp, rec, g2):
    """one XYZZ record (standard Montgomery limbs, as g16_partial holds it) -> affine integers (None = identity)"""
    L = cp.fq_limbs64
    if not g2:
        x, y, zz, zzz = mont_to_ints(np.asarray(rec[: 4 * L], dtype=np.uint64).reshape(4, L), cp.q)
        if zz == 0:
            return None
        return (x * pow(zz, -1, cp.q) % cp.q, y * pow(zzz, -1, cp.q) % cp.q)
    v = mont_to_ints(np.asarray(rec[: 8 * L], dtype=np.uint64).reshape(8, L), cp.q)
    x, y, zz, zzz = (v[0], v[1]), (v[2], v[3]), (v[4], v[5]), (v[6], v[7])
    if zz == (0, 0):
        return None
    f2 = pm.groups(cp)[1].F
    return (f2.mul(x, f2.inv(zz)), f2.mul(y, f2.inv(zzz)))


def partial_to_parts(g, cp, part_bytes):
    """g16_partial bytes -> the five raw MSM results as affine integers, in the oracle's order h, l, a, b_g1, b_g2"""
    from groth16_amd.binding import PartialC

    p = PartialC.from_buffer_copy(part_bytes)
    return [xyzz_to_affine(cp, list(p.h), False), xyzz_to_affine(cp, list(p.l), False), xyzz_to_affine(cp, list(p.a), False),
            xyzz_to_affine(cp, list(p.b_g1), False), xyzz_to_affine(cp, list(p.b_g2), True)]


def oracle_parts(cp, parts):
    from helpers import arr_to_g1, arr_to_g2

    L = cp.fq_limbs64
    return [arr_to_g1(parts[2 * L * i: 2 * L * (i + 1)], cp)[0] for i in range(4)] + [arr_to_g2(parts[8 * L: 12 * L], cp)[0]]


@pytest.mark.parametrize("curve,k,full_oracle", [("bls12_381", 20, True), ("bn254", 20, True), ("bls12_381", 22, True), ("bn254", 22, True)],
                         ids=["bls12_381-k20", "bn254-k20", "bls12_381-k22", "bn254-k22"])
def test_full_size_proof_bit_exact(g, orc, curve, k, full_oracle):
    cp = CP[curve]
    ck = orc.syn_circuit(curve, k, 17 + k)
    toxic = orc.rand_fr(curve, 900 + k, 5)                      # alpha beta gamma delta t
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]         # a pair of generators (any subgroup points do: generator.rs:31-32)
    r, s = orc.rand_fr(curve, 61, 1)[0], orc.rand_fr(curve, 62, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        proof = prover.create_proof_with_reduction_and_matrices(pk, r, s, mats, ck.num_inputs, ck.num_constraints, ck.z)
        h_gpu = prover.witness_map_from_matrices(mats, ck.num_inputs, ck.num_constraints, ck.z)
        part = prover.prove_partial(pk, mats, ck.z, (0, 1)) if full_oracle else None
    # (1) witness map, every coefficient
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    assert not h_gpu[-1].any()
    # (2) trapdoor closed form (uses the ORACLE's h, just proven equal to the GPU's)
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    want = orc.trapdoor_proof(ck, ex, h_orc, r, s)
    assert (proof.flat() == want).all(), "GPU proof differs from the trapdoor closed form"
    # (3) the oracle's prover on the same key and witness: proof and the five raw MSM sums
    if full_oracle:
        o_proof, o_h, o_parts, _ = orc.prove(flat_pk(curve, pk), ck, r, s, want_parts=True)
        assert (o_h == h_gpu).all()
        assert (o_proof == proof.flat()).all(), "GPU proof differs from the oracle prover's"
        assert partial_to_parts(g, cp, part) == oracle_parts(cp, o_parts)


def dummy_circuit(orc, curve, num_variables, num_constraints, seed):
    """benches/bench.rs:41-64 DummyCircuit: witnesses a, b, then num_variables - 3 more witnesses all equal to a; one input
    c = a*b; num_constraints - 1 copies of a * b = c and a final empty constraint 0 * 0 = 0"""
    ab = orc.rand_fr(curve, seed, 2)
    a, b = ab[0], ab[1]
    c = orc.field_op(curve, 0, 2, a.copy(), b.copy())
    nw = 2 + (num_variables - 3)
    nv = 2 + nw
    z = np.zeros((nv, 4), dtype=np.uint64)
    z[0] = orc.field_op(curve, 0, 5, np.array([1, 0, 0, 0], dtype=np.uint64))   # from_canonical(1) = Montgomery one
    z[1] = c
    z[2] = a
    z[3] = b
    z[4:] = a
    nc = num_constraints
    rp = np.arange(nc + 1, dtype=np.uint64)
    rp[nc] = nc - 1                                                            # the last row is empty
    val = np.repeat(z[0:1], nc - 1, axis=0)
    mk = lambda col: Csr(rp, np.full(nc - 1, col, dtype=np.uint32), val)       # noqa: E731
    return FlatCircuit(curve, 2, nc, nv, [mk(2), mk(3), mk(1)], z)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_reference_bench_circuit_shape_2_20(g, orc, curve):
    """the reference's own benchmark instance (benches/bench.rs:19-20,41-64; NUM_CONSTRAINTS = NUM_VARIABLES = 2^20 - 100):
    every witness value equal, a_query / b_query / l_query identity almost everywhere, the last constraint empty -- end to
    end at 2^20: GPU key, GPU proof == trapdoor closed form == oracle prover on the same key"""
    nvar = ncons = (1 << 20) - 100
    ck = dummy_circuit(orc, curve, nvar, ncons, 5)
    assert ck.domain_size == 1 << 20
    toxic = orc.rand_fr(curve, 321, 5)
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]
    r, s = orc.rand_fr(curve, 71, 1)[0], orc.rand_fr(curve, 72, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        proof = prover.create_proof_with_reduction_and_matrices(pk, r, s, mats, ck.num_inputs, ck.num_constraints, ck.z)
        h_gpu = prover.witness_map_from_matrices(mats, ck.num_inputs, ck.num_constraints, ck.z)
    # the key has the reference bench's shape: b_query is the identity except at variable b (index 3)
    assert not pk.b_g1_query[4:].any() and pk.b_g1_query[3].any() and not pk.l_query[2:].any()
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    assert (proof.flat() == orc.trapdoor_proof(ck, ex, h_orc, r, s)).all()
    o_proof, _ = orc.prove(flat_pk(curve, pk), ck, r, s)
    assert (o_proof == proof.flat()).all()


def test_sharded_2_24_three_shards_trapdoor(g, orc):
    """BASELINE.json configs[4] (2^24 constraints, BLS12-381, MSM bases sharded) at reduced cost on ONE GPU: the key is cut
    into three shards, each shard's partial sums are computed on their own (tables loaded, proved, evicted), the records are
    folded by g16_prove_finalize, and the proof must equal the trapdoor closed form -- the five sums of prover.rs:66,74,92,
    105,113 are the same whatever the cut.  Exercises what exists only at this size: 2^24-point domains (four-sweep NTTs),
    entry tags point | window << 26 near their limits, multi-gigabyte arenas."""
    curve, k = "bls12_381", 24
    ck = orc.syn_circuit(curve, k, 3)
    toxic = orc.rand_fr(curve, 924, 5)
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]
    r, s = orc.rand_fr(curve, 81, 1)[0], orc.rand_fr(curve, 82, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        parts = []
        for i in range(3):
            parts.append(prover.prove_partial(pk, mats, ck.z, (i, 3)))
            if i < 2:
                prover.evict_pk(pk, (i, 3))
        proof = prover.prove_finalize(pk, ck.num_inputs, parts, r, s, (2, 3))
        h_gpu = prover.witness_map_from_matrices(mats, ck.num_inputs, ck.num_constraints, ck.z)
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    assert (proof.flat() == orc.trapdoor_proof(ck, ex, h_orc, r, s)).all()


def test_configs4_2_24_eight_ranks_distributed_map_block_h(g, orc):
    """BASELINE.json configs[4] AS SPECIFIED, every rank on the one visible GPU: 2^24 constraints, BLS12-381, world = 8.
    Distributed witness map (four stages per rank, the three all-to-all exchanges done by moving the chunks between the
    ranks' buffers) -> per rank g16_prove_partial_h over ITS key shard, whose h_query is gathered in the block order the map
    leaves h in (load 16 GB of window tables, prove, evict) -> g16_prove_finalize over the eight records.  What must not depend
    on the cut: h of src/r1cs_to_qap.rs:232 (all 2^24 coefficients against the oracle's) and the five sums of
    src/prover.rs:66,74,92,105,113 (the proof against the trapdoor closed form)."""
    from test_gpu_dist_wm import run_all_ranks

    curve, k, world = "bls12_381", 24, 8
    ck = orc.syn_circuit(curve, k, 4)
    toxic = orc.rand_fr(curve, 1924, 5)
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]
    r, s = orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        ranks, h_gpu = run_all_ranks(g, prover, mats, ck.z, world)
        parts = []
        for d in ranks:
            parts.append(prover.prove_partial_h(pk, mats, ck.z, (d.rank, world), d.h_local.data_ptr(), d.M))
            if d.rank < world - 1:
                prover.evict_pk(pk, (d.rank, world))
        proof = prover.prove_finalize(pk, ck.num_inputs, parts, r, s, (world - 1, world), dist_h=True)
        for d in ranks:
            d.close()
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    assert not h_gpu[-1].any()
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    assert (proof.flat() == orc.trapdoor_proof(ck, ex, h_orc, r, s)).all()


def test_configs4_2_24_eight_ranks_bucket_space_shards(g, orc):
    """BASELINE.json configs[4] under the OTHER cut of the MSMs (round 5, g16_pk_load_bucket_shard): 2^24 constraints, BLS12-381,
    world = 8, every rank on the one visible GPU.  The whole key is resident ONCE as window tables (126 GB -- what a rank of this mode
    holds) and re-labelled rank by rank (g16_pk_rebind_bucket_shard: the tables do not depend on the rank); h comes from the distributed
    witness map and is all-gathered (the ranks' blocks back to back, h_query loaded in that order); every rank runs
    g16_prove_partial_h over ALL points and its eighth of the buckets (b mod 8 == rank); g16_prove_finalize folds the eight records.
    Must not depend on the cut: h (all 2^24 coefficients against the oracle's) and the proof (against the trapdoor closed form)."""
    import torch
    from test_gpu_dist_wm import run_all_ranks

    curve, k, world = "bls12_381", 24, 8
    ck = orc.syn_circuit(curve, k, 4)
    toxic = orc.rand_fr(curve, 1924, 5)
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]
    r, s = orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        ranks, h_gpu = run_all_ranks(g, prover, mats, ck.z, world)
        h_all = torch.cat([d.h_local for d in ranks])   # what all_gather_into_tensor leaves on every rank
        for d in ranks:
            d.close()
        torch.cuda.synchronize()
        shard = (0, world, "bucket")
        dpk = prover._pk(pk, ck.num_inputs, shard, dist_h=True)
        info = dpk.info()
        assert info["table_fallback"] == 0 and info["window_bits_z"] == 20 and info["device_bytes"] > 100e9, info
        parts = []
        for i in range(world):
            dpk.rebind(i, world)
            parts.append(prover.prove_partial_h(pk, mats, ck.z, shard, h_all.data_ptr(), h_all.shape[0]))
        proof = prover.prove_finalize(pk, ck.num_inputs, parts, r, s, shard, dist_h=True)
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    assert (proof.flat() == orc.trapdoor_proof(ck, ex, h_orc, r, s)).all()


@pytest.mark.parametrize("curve,world", [("bls12_381", 8), ("bn254", 4)])
def test_headline_size_bucket_space_shards_trapdoor(g, orc, curve, world):
    """The configuration the projected multi-GPU numbers are quoted on: 2^22 constraints, bucket-space shards, every rank
    on the one GPU -- distributed witness map, h all-gathered, the resident tables re-labelled rank by rank, the SUM of the ranks' five
    partial sums equal to the single-GPU proof's (the same key proved whole on the same context) and the proof equal to the trapdoor
    closed form."""
    import torch
    from test_gpu_dist_wm import run_all_ranks

    k = 22
    ck = orc.syn_circuit(curve, k, 6)
    toxic = orc.rand_fr(curve, 2924, 5)
    gens = orc.setup(orc.syn_circuit(curve, 2, 1), 3)[1]
    r, s = orc.rand_fr(curve, 93, 1)[0], orc.rand_fr(curve, 94, 1)[0]
    with g.Groth16(curve, 0) as prover:
        mats = mats_of(g, ck)
        pk = prover.generate_parameters_with_qap(mats, toxic[0], toxic[1], toxic[2], toxic[3], gens["g1gen"], gens["g2gen"], toxic[4])
        whole = prover.create_proof_with_reduction_and_matrices(pk, r, s, mats, ck.num_inputs, ck.num_constraints, ck.z)
        prover.evict_pk(pk)
        ranks, h_gpu = run_all_ranks(g, prover, mats, ck.z, world)
        h_all = torch.cat([d.h_local for d in ranks])
        for d in ranks:
            d.close()
        torch.cuda.synchronize()
        shard = (0, world, "bucket")
        dpk = prover._pk(pk, ck.num_inputs, shard, dist_h=True)
        parts = []
        for i in range(world):
            dpk.rebind(i, world)
            parts.append(prover.prove_partial_h(pk, mats, ck.z, shard, h_all.data_ptr(), h_all.shape[0]))
        proof = prover.prove_finalize(pk, ck.num_inputs, parts, r, s, shard, dist_h=True)
    assert (proof.flat() == whole.flat()).all()
    h_orc = orc.witness_map(ck)
    assert (h_gpu == h_orc).all()
    ex = trapdoor_extras(g, curve, ck, toxic, gens["g1gen"], gens["g2gen"])
    assert (proof.flat() == orc.trapdoor_proof(ck, ex, h_orc, r, s)).all()
