"""Bucket-space shards on the GPU (g16_pk_load_bucket_shard): every rank holds the WHOLE key as window tables and owns
the buckets b mod world == rank of the five MSMs of /root/reference/src/prover.rs:66,74,262.  All ranks run one after the other on
the one visible MI355X; what must not depend on the cut is compared with the CPU oracle: each MSM (the ranks' shares add up to it),
the proof with the replicated witness map (any rank count) and with the distributed one (h all-gathered: the ranks' blocks back to
back, h_query loaded in that order), r = 0 included."""
import numpy as np
import pytest

import pymodel as pm
from helpers import ints_to_mont, oracle  # noqa: F401

pytestmark = pytest.mark.gpu

CURVES = ["bls12_381", "bn254"]
CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}
WORLDS = [2, 5, 8]


def mats_of(g, ck):
    return g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])


def pk_of(g, pk):
    return g.ProvingKey(pk.curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                        pk.h_query, pk.l_query)


@pytest.fixture(scope="module")
def g():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    import groth16_amd

    return groth16_amd


@pytest.fixture(scope="module", params=CURVES)
def env(request, g, orc):
    prover = g.Groth16(request.param, 0)
    yield request.param, prover
    prover.close()


def sum_of_shares(orc, prover, curve, g2, bases, sc, world):
    total = None
    for r in range(world):
        part = prover.msm_bucket_shard(bases, sc, r, world, g2)
        total = part if total is None else orc.group_op(curve, g2, 0, total, part)
    return total


@pytest.mark.parametrize("window", [None, "17", "20"])   # cost model (c = 9 ... at these sizes: one class), 2 classes, 16 classes before the cut
@pytest.mark.parametrize("world", WORLDS + [64])
@pytest.mark.parametrize("g2", [False, True])
def test_msm_bucket_shares_add_up(env, orc, g2, world, window, monkeypatch):
    if window:
        monkeypatch.setenv("G16_MSM_PRECOMP_WINDOW", window)
    curve, prover = env
    for n in (1, 257, 6000):
        bases = orc.synth_bases(curve, g2, 3, n)
        sc = orc.rand_fr(curve, 7 + n + world, n)
        assert (sum_of_shares(orc, prover, curve, g2, bases, sc, world) == orc.msm(curve, g2, bases, sc)).all(), n


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("g2", [False, True])
def test_msm_bucket_shares_adversarial(env, orc, g2, world):
    """all-equal scalars (benches/bench.rs:52-54: EVERY entry of a window lands in one bucket, i.e. on one rank), zero, one, r - 1,
    identity bases, a repeated base: the worst-case buffers of the sort are what holds these"""
    curve, prover = env
    cp = CP[curve]
    n = 3000
    bases = orc.synth_bases(curve, g2, 11, n)
    bases[10:500:7] = 0
    bases[21] = bases[20]
    sc = orc.rand_fr(curve, 13, n)
    sc[: 6] = ints_to_mont([0, 1, cp.r - 1, 2, (cp.r - 1) // 2, (cp.r + 1) // 2], cp.r, 4)
    sc[21] = sc[20]
    for scal in (sc, np.repeat(sc[40:41], n, axis=0), np.zeros_like(sc), np.repeat(ints_to_mont([1], cp.r, 4), n, axis=0)):
        assert (sum_of_shares(orc, prover, curve, g2, bases, scal, world) == orc.msm(curve, g2, bases, scal)).all()


@pytest.mark.parametrize("world", WORLDS)
def test_bucket_sharded_proof_replicated_map(env, orc, g, world):
    """every rank: g16_prove_partial over its bucket-space shard (the witness map replicated -- any rank count, 5 included);
    g16_prove_finalize over the records == the oracle's proof (prover.rs:54-132), r = 0 (B in G1 skipped, :98-108) too"""
    curve, prover = env
    ck = orc.syn_circuit(curve, 9, 3 + world)
    pk, _ = orc.setup(ck, 8)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    for r, s in ((orc.rand_fr(curve, 31, 1)[0], orc.rand_fr(curve, 32, 1)[0]), (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 33, 1)[0])):
        parts = [prover.prove_partial(gp, gm, ck.z, (i, world, "bucket"), skip_b_g1=not r.any()) for i in range(world)]
        proof = prover.prove_finalize(gp, ck.num_inputs, parts, r, s, (0, world, "bucket"))
        assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
    info = prover.pk_info(gp, ck.num_inputs, (1, world, "bucket"))
    assert info["bucket_shard_rank"] == 1 and info["bucket_shard_world"] == world and info["table_fallback"] == 0 and info["window_bits_z"] >= 9
    for i in range(world):
        prover.evict_pk(gp, (i, world, "bucket"))


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("k,world", [(8, 2), (11, 4), (13, 8)])
def test_bucket_sharded_proof_distributed_h(g, orc, curve, k, world):
    """distributed witness map -> all-gather of h (rank 0's block, rank 1's, ...) -> every rank g16_prove_partial_h over ALL of h
    and its 1 / world of the buckets, h_query loaded in the gathered order; one resident copy of the tables serves every rank
    (g16_pk_rebind_bucket_shard)"""
    import torch
    from test_gpu_dist_wm import run_all_ranks

    ck = orc.syn_circuit(curve, k, 5 + k)
    pk, _ = orc.setup(ck, 3)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    with g.Groth16(curve, 0) as prover:
        ranks, _ = run_all_ranks(g, prover, gm, ck.z, world)
        h_all = torch.cat([d.h_local for d in ranks])      # what all_gather_into_tensor leaves on every rank
        torch.cuda.synchronize()
        shard = (0, world, "bucket")
        dpk = prover._pk(gp, ck.num_inputs, shard, dist_h=True)
        z_dev = torch.from_numpy(np.ascontiguousarray(ck.z).view(np.int64)).to("cuda:0")
        for r, s in ((orc.rand_fr(curve, 61, 1)[0], orc.rand_fr(curve, 62, 1)[0]), (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 63, 1)[0])):
            parts = []
            for i in range(world):
                # a witness sort prepared for the key as it was labelled BEFORE (g16_prove_partial_prepare) must not survive the
                # re-labelling: the sort depends on the residue class
                prover.prove_partial_prepare(gp, gm, z_dev.data_ptr(), z_dev.shape[0], shard, dist_h=True)
                dpk.rebind(i, world)
                parts.append(prover.prove_partial_h(gp, gm, ck.z, shard, h_all.data_ptr(), h_all.shape[0], skip_b_g1=not r.any(),
                                                    z_dev_ptr=z_dev.data_ptr()))
            proof = prover.prove_finalize(gp, ck.num_inputs, parts, r, s, shard, dist_h=True)
            assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        for d in ranks:
            d.close()


def test_bucket_shard_load_refusals(g, orc, monkeypatch):
    """no silent fall-back in this mode: a base-range view, a key that may not have tables and a table that does not fit are errors;
    the plain load of the same key reports WHY it holds plain bases (g16_pk_get_info)"""
    curve = "bn254"
    ck = orc.syn_circuit(curve, 7, 2)
    pk, _ = orc.setup(ck, 4)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    with g.Groth16(curve, 0) as prover:
        monkeypatch.setenv("G16_MSM_PRECOMP", "0")
        with pytest.raises(g.G16Error):
            prover._pk(gp, ck.num_inputs, (0, 2, "bucket"))
        assert prover.pk_info(gp, ck.num_inputs)["table_fallback"] == 1
        prover.evict_pk(gp)
        monkeypatch.delenv("G16_MSM_PRECOMP")
        monkeypatch.setenv("G16_PK_TABLE_BUDGET_MB", "0.001")
        with pytest.raises(g.G16Error) as ei:
            prover._pk(gp, ck.num_inputs, (0, 2, "bucket"))
        assert ei.value.status == 5   # G16_ERR_OOM
        info = prover.pk_info(gp, ck.num_inputs)
        assert info["table_fallback"] == 3 and info["window_bits_z"] == 0
        # a proof over the fallen-back key is still right
        r, s = orc.rand_fr(curve, 1, 1)[0], orc.rand_fr(curve, 2, 1)[0]
        proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
        assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        prover.evict_pk(gp)
        monkeypatch.delenv("G16_PK_TABLE_BUDGET_MB")
        assert prover.pk_info(gp, ck.num_inputs)["table_fallback"] == 0
        # a base-range shard cannot be re-labelled as a bucket-space one
        with pytest.raises(g.G16Error):
            prover._pk(gp, ck.num_inputs, (1, 2)).rebind(0, 2)
