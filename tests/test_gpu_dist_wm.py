"""Distributed witness map on the GPU (g16_dwm_*): every rank's stages run on the one visible MI355X, the
all-to-all between the stages is done here by moving the chunks between the ranks' buffers, and the result is compared
bit-for-bit with the single-GPU witness map / the CPU oracle (src/r1cs_to_qap.rs:172-235); then the sharded proof over the
block-distributed h (g16_prove_partial_h) against the oracle's proof."""
import numpy as np
import pytest

import pymodel as pm
from helpers import circuit_from_pymodel, oracle  # noqa: F401

pytestmark = pytest.mark.gpu

CURVES = ["bls12_381", "bn254"]
CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}


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


def run_all_ranks(g, prover, gm, z, world):
    """the four stages on `world` simulated ranks; returns (dwm objects, h in natural order)"""
    import torch
    from groth16_amd.groth16 import DistributedWitnessMap, dist_h_indices

    dck = prover._ck(gm)
    n = dck.domain_size
    lb = prover._ctx.lib
    ranks = [DistributedWitnessMap(lb, prover._ctx.handle, dck.handle, r, world, "cuda:0") for r in range(world)]
    zc = np.ascontiguousarray(z, dtype=np.uint64)
    for s in range(4):
        for d in ranks:
            d.stage(s, zc.ctypes.data if s == 0 else 0, zc.shape[0] if s == 0 else 0, False)
        if s < 3:
            blk = ranks[0].blk
            for m in range(DistributedWitnessMap.arrays_after(s)):
                for dst in ranks:           # chunk `dst.rank` of every source's work array, stored by source rank
                    for src in ranks:
                        dst.recv[m][src.rank * blk:(src.rank + 1) * blk] = src.work[m][dst.rank * blk:(dst.rank + 1) * blk]
            torch.cuda.synchronize()
    h = np.zeros((n, 4), dtype=np.uint64)
    for d in ranks:
        h[dist_h_indices(n, d.rank, world)] = d.h_local.cpu().numpy().view(np.uint64)
    return ranks, h


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("k,world", [(2, 2), (4, 2), (4, 4), (6, 8), (10, 2), (10, 4), (10, 8), (12, 16), (16, 8), (20, 8), (22, 8)])
def test_distributed_witness_map_matches_oracle(g, orc, curve, k, world):
    """k = 20, 22: BASELINE.json's sizes -- the local transforms are multi-sweep (2^17 / 2^19-point domains) only there"""
    ck = orc.syn_circuit(curve, k, 70 + k + world)
    with g.Groth16(curve, 0) as prover:
        ranks, h = run_all_ranks(g, prover, mats_of(g, ck), ck.z, world)
        assert (h == orc.witness_map(ck)).all()
        for d in ranks:
            d.close()


@pytest.mark.parametrize("curve", CURVES)
def test_distributed_witness_map_padded_domain_and_dense_rows(g, orc, curve):
    """nc + num_inputs not a power of two (zero rows, the instance copies of r1cs_to_qap.rs:195-199 on different ranks) and
    rows with several terms"""
    cp = CP[curve]
    for (cs, z), world in ((pm.mimc_circuit(cp, 20, 4), 4), (pm.syn_circuit(cp, 5, 6, dense=True), 2), (pm.mimc_circuit(cp, 90, 5), 8)):
        ck = circuit_from_pymodel(cp, cs, z)
        with g.Groth16(curve, 0) as prover:
            gm = mats_of(g, ck)
            ranks, h = run_all_ranks(g, prover, gm, ck.z, world)
            assert (h == prover.witness_map_from_matrices(gm, ck.num_inputs, ck.num_constraints, ck.z)).all()
            assert (h == orc.witness_map(ck)).all()
            for d in ranks:
                d.close()


def test_dwm_rejects_bad_world(g, orc):
    from groth16_amd.groth16 import DistributedWitnessMap

    ck = orc.syn_circuit("bn254", 4, 1)
    with g.Groth16("bn254", 0) as prover:
        dck = prover._ck(mats_of(g, ck))
        for rank, world in ((0, 3), (2, 2), (0, 8), (0, 32)):   # not a power of two; rank out of range; world^2 > n; world > 16
            with pytest.raises(g.G16Error):
                DistributedWitnessMap(prover._ctx.lib, prover._ctx.handle, dck.handle, rank, world, "cuda:0")


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("k,world", [(8, 2), (11, 4), (13, 8)])
def test_sharded_proof_with_distributed_h(g, orc, curve, k, world):
    """every rank: its block of h from the distributed map -> g16_prove_partial_h over the key shard whose h_query is gathered
    in the same block order; the folded proof equals the oracle's (prover.rs:54-132), r = 0 included"""
    ck = orc.syn_circuit(curve, k, 5 + k)
    pk, _ = orc.setup(ck, 3)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    with g.Groth16(curve, 0) as prover:
        ranks, _ = run_all_ranks(g, prover, gm, ck.z, world)
        for r, s in ((orc.rand_fr(curve, 61, 1)[0], orc.rand_fr(curve, 62, 1)[0]), (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 63, 1)[0])):
            parts = [prover.prove_partial_h(gp, gm, ck.z, (d.rank, world), d.h_local.data_ptr(), d.M, skip_b_g1=not r.any()) for d in ranks]
            proof = prover.prove_finalize(gp, ck.num_inputs, parts, r, s, (0, world), dist_h=True)
            assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        for d in ranks:
            d.close()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("k", [6, 13])
def test_async_stages_one_rank_then_proof(g, orc, curve, k):
    """g16_dwm_stage_async: the four stages (world = 1: the exchanges are device copies) enqueued on the library's witness-map
    stream through torch's ExternalStream with NO host synchronisation, g16_prove_partial_h right behind them -- its h MSM must
    order itself after that stream, its witness sort must not need it.  Proof == oracle's, then h == oracle's."""
    import torch
    from groth16_amd.groth16 import DistributedWitnessMap

    ck = orc.syn_circuit(curve, k, 23 + k)
    pk, _ = orc.setup(ck, 3)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    r, s = orc.rand_fr(curve, 71, 1)[0], orc.rand_fr(curve, 72, 1)[0]
    with g.Groth16(curve, 0) as prover:
        dck = prover._ck(gm)
        d = DistributedWitnessMap(prover._ctx.lib, prover._ctx.handle, dck.handle, 0, 1, "cuda:0")
        z_dev = torch.from_numpy(np.ascontiguousarray(ck.z).view(np.int64)).to("cuda:0")
        torch.cuda.synchronize()
        for it in range(4):   # repeated: stale events / buffers of an earlier round must not satisfy a later one
            # rounds 1 and 3: the witness sort enqueued ahead of the map (g16_prove_partial_prepare) and consumed by the partial
            # call on the same device assignment; round 2: prepared, but the partial call passes the HOST assignment -- the
            # prepared sort must be dropped, not used; round 0: no prepare
            if it:
                prover.prove_partial_prepare(gp, gm, z_dev.data_ptr(), z_dev.shape[0], (0, 1))
            h = d.run(z_dev.data_ptr(), z_dev.shape[0], True, None)
            part = prover.prove_partial_h(gp, gm, ck.z, (0, 1), h.data_ptr(), d.M, z_dev_ptr=z_dev.data_ptr() if it in (1, 3) else 0)
            proof = prover.prove_finalize(gp, ck.num_inputs, [part], r, s, (0, 1), dist_h=True)
            assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        torch.cuda.synchronize()
        assert (h.cpu().numpy().view(np.uint64) == orc.witness_map(ck)).all()
        d.close()
