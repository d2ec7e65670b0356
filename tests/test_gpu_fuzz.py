"""Randomised GPU-vs-oracle checks: odd sizes, mixed scalar populations (zeros, ones, small, repeated, full-width), identity
bases, odd shard counts.  Seeds are fixed, so failures reproduce."""
import numpy as np
import pytest

import pymodel as pm
from helpers import ints_to_mont

pytestmark = pytest.mark.gpu
CP = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}


def _mats(g, ck):
    return g.ConstraintMatrices(ck.num_inputs, ck.num_vars - ck.num_inputs, ck.num_constraints, *[(m.row_ptr, m.col, m.val) for m in ck.abc])


def _pk(g, pk):
    return g.ProvingKey(pk.curve, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                        pk.h_query, pk.l_query)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_msm_fuzz(orc, curve):
    import groth16_amd as g

    cp = CP[curve]
    rng = np.random.default_rng(2024)
    pool1 = orc.synth_bases(curve, False, 7, 6000)
    pool2 = orc.synth_bases(curve, True, 8, 3000)
    full = orc.rand_fr(curve, 5, 6000)
    special = ints_to_mont([0, 1, 2, cp.r - 1, cp.r - 2, (1 << 16) - 1, 1 << 16, (1 << 240) % cp.r, 12345], cp.r, 4)
    with g.Groth16(curve, 0) as prover:
        for it in range(14):
            g2 = bool(it % 2)
            pool = pool2 if g2 else pool1
            n = int(rng.integers(1, len(pool)))
            bases = pool[rng.integers(0, len(pool), n)].copy()          # repeated bases are likely
            kind = it % 4
            if kind == 0:
                sc = full[rng.integers(0, len(full), n)].copy()
            elif kind == 1:                                               # mostly special values
                sc = special[rng.integers(0, len(special), n)].copy()
            elif kind == 2:                                               # few distinct values -> heavy buckets
                sc = full[rng.integers(0, 3, n)].copy()
            else:                                                         # mixture
                sc = np.where((rng.random(n) < 0.5)[:, None], special[rng.integers(0, len(special), n)], full[rng.integers(0, len(full), n)])
                sc = np.ascontiguousarray(sc)
            bases[rng.random(n) < 0.05] = 0                               # identity bases
            got = prover.msm(bases, sc, g2)
            want = orc.msm(curve, g2, bases, sc)
            assert (got == want).all(), (it, n, g2, kind)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("world", [2, 5, 7])
def test_sharded_proof_fuzz(orc, curve, world):
    import groth16_amd as g

    ck = orc.syn_circuit(curve, 7, 40 + world)
    pk, _ = orc.setup(ck, 17)
    r, s = orc.rand_fr(curve, 3, 1)[0], orc.rand_fr(curve, 4, 1)[0]
    want, _ = orc.prove(pk, ck, r, s)
    with g.Groth16(curve, 0) as prover:
        gm, gp = _mats(g, ck), _pk(g, pk)
        parts = [prover.prove_partial(gp, gm, ck.z, (i, world)) for i in range(world)]
        proof = prover.prove_finalize(gp, ck.num_inputs, parts, r, s, (0, world))
        assert (proof.flat() == want).all()
        # the host-only aggregator path gives the same proof
        assert (g.finalize_host(curve, gp, parts, r, s).flat() == want).all()
        # r = 0: B in G1 is skipped on every rank (prover.rs:98)
        z4 = np.zeros(4, dtype=np.uint64)
        parts0 = [prover.prove_partial(gp, gm, ck.z, (i, world), skip_b_g1=True) for i in range(world)]
        assert (prover.prove_finalize(gp, ck.num_inputs, parts0, z4, s, (0, world)).flat() == orc.prove(pk, ck, z4, s)[0]).all()
