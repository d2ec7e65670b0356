"""CPU-only checks of the product library: it loads without a GPU, exports every symbol that
include/g16_mi355x.h declares, fails loudly without a device, and its host-compiled field / group /
bucket-method code (the same headers the kernels use) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pymodel as pm
from helpers import ROOT, g1_to_arr, g2_to_arr, ints_to_mont, mont_to_ints
from groth16_amd import binding
from groth16_amd.binding import CURVE_ID, ptr32, ptr64

CURVES = [pm.BLS12_381, pm.BN254]


@pytest.fixture(scope="module")
def lb():
    return binding.lib()


def test_exports_match_header(lb):
    hdr = open(os.path.join(ROOT, "include", "g16_mi355x.h")).read()
    declared = set(re.findall(r"\b(g16_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(binding.EXPORTS), declared ^ set(binding.EXPORTS)
    for name in declared:
        assert hasattr(lb.c, name), name
    assert "gfx950" in lb.version()


def test_no_gpu_fails_loudly(lb):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lb.c.g16_ctx_create(0, 0, C.byref(h))
    assert rc != 0 and not h
    with pytest.raises(binding.G16Error):
        lb.check(rc)


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_host_field_ops(lb, orc, cp):
    rng = pm.SplitMix64(23)
    cid = CURVE_ID[cp.name]
    for which, p, nl in ((0, cp.r, 4), (1, cp.q, cp.fq_limbs64)):
        xs = [rng.field(p) for _ in range(200)] + [0, 1, p - 1, p - 1, (1 << (64 * nl)) % p]
        ys = [rng.field(p) for _ in range(200)] + [p - 1, 0, p - 1, 1, (1 << (64 * nl)) % p]
        for x, y in zip(xs, ys):
            a, b = ints_to_mont([x], p, nl)[0], ints_to_mont([y], p, nl)[0]
            out = np.zeros(nl, dtype=np.uint64)
            for op in (0, 1, 2):
                assert lb.c.g16_host_field_op(cid, which, op, ptr64(a), ptr64(b), ptr64(out)) == 0
                assert (out == orc.field_op(cp.name, which, op, a, b)).all(), (which, op, x, y)
            for op in (4, 5):
                assert lb.c.g16_host_field_op(cid, which, op, ptr64(a), None, ptr64(out)) == 0
                assert (out == orc.field_op(cp.name, which, op, a)).all()
        for x in xs[:10] + [1, p - 1]:
            if x == 0:
                continue
            a = ints_to_mont([x], p, nl)[0]
            out = np.zeros(nl, dtype=np.uint64)
            assert lb.c.g16_host_field_op(cid, which, 3, ptr64(a), None, ptr64(out)) == 0
            assert mont_to_ints(out, p)[0] == pow(x, p - 2, p)


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_host_group_ops(lb, orc, cp):
    G1, G2 = pm.groups(cp)
    rng = pm.SplitMix64(31)
    cid = CURVE_ID[cp.name]
    for g2, G, gen, conv in ((0, G1, cp.g1, g1_to_arr), (1, G2, cp.g2, g2_to_arr)):
        P = G.mul(gen, rng.field(cp.r))
        Q = G.mul(gen, rng.field(cp.r))
        pa, qa = conv([P], cp)[0], conv([Q], cp)[0]
        ident = np.zeros_like(pa)
        negp = conv([G.neg(P)], cp)[0]
        out = np.zeros_like(pa)
        cases = [(pa, qa), (pa, pa), (pa, negp), (pa, ident), (ident, qa), (ident, ident)]
        for a, b in cases:
            for op in (0, 2):
                assert lb.c.g16_host_group_op(cid, g2, op, ptr64(a), ptr64(b), ptr64(out)) == 0
                assert (out == orc.group_op(cp.name, bool(g2), 0, a, b)).all(), (g2, op)
        for k in (0, 1, 2, cp.r - 1, rng.field(cp.r)):
            kb = np.array([(k >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)
            assert lb.c.g16_host_group_op(cid, g2, 1, ptr64(pa), ptr64(kb), ptr64(out)) == 0
            assert (out == orc.group_op(cp.name, bool(g2), 1, pa, kb)).all(), k


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("c_win", [0, 3, 5, 9, 13, 16, -9, -12, -16, -17, -20])
def test_host_msm_model(lb, orc, cp, c_win):
    """the signed-digit / bucket / chunked-reduction / fold scheme the kernels implement, run on the
    CPU with the kernels' own helper code, against the oracle's ark-ec style Pippenger.  Negative c_win: the merged-window
    scheme of the proving-key path (window tables 2^(cj) P, one bucket set cut into classes of 2^15, fold over classes)."""
    cid = CURVE_ID[cp.name]
    n = 70 if abs(c_win) >= 13 else 300
    for g2 in (0, 1):
        if g2 and c_win in (9, 16, -12, -16, -20):
            continue
        bases = orc.synth_bases(cp.name, bool(g2), 5, n)
        sc = orc.rand_fr(cp.name, 77 + c_win, n)
        p = cp.r
        specials = ints_to_mont([0, 1, p - 1, 2, (1 << 255) % p, (p - 1) // 2, (p + 1) // 2], p, 4)
        sc[: len(specials)] = specials
        bases[9] = 0  # identity base
        bases[11] = bases[10]  # repeated base
        sc[11] = sc[10]
        out = np.zeros(bases.shape[1], dtype=np.uint64)
        assert lb.c.g16_host_msm_model(cid, g2, ptr64(bases), ptr64(sc), n, c_win, ptr64(out)) == 0
        assert (out == orc.msm(cp.name, bool(g2), bases, sc)).all(), (g2, c_win)


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("G", [4, 16, 32])
@pytest.mark.parametrize("c_win", [7, -12, -17])
def test_host_msm_model_buckets_per_lane(lb, orc, cp, c_win, G, monkeypatch):
    """the bucket reduction's chunk size G (buckets per lane; 8 by default, 16 for 2^19-bucket sets since round 3) enters the
    recombination P_0 + G sum_k 2^k P_(2+k) and the number of bit planes: the CPU model of the scheme against the oracle for
    other chunk sizes, per-window and merged (two and several classes)"""
    monkeypatch.setenv("G16_MSM_REDUCE_G", str(G))
    n = 120
    bases = orc.synth_bases(cp.name, False, 6, n)
    sc = orc.rand_fr(cp.name, 300 + G, n)
    sc[5:40] = sc[5]
    out = np.zeros(bases.shape[1], dtype=np.uint64)
    assert lb.c.g16_host_msm_model(CURVE_ID[cp.name], 0, ptr64(bases), ptr64(sc), n, c_win, ptr64(out)) == 0
    assert (out == orc.msm(cp.name, False, bases, sc)).all()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("world", [2, 3, 5, 8, 64])
@pytest.mark.parametrize("c_win", [-9, -12, -17, -20])
def test_host_msm_model_bucket_shards_add_up(lb, orc, cp, c_win, world):
    """bucket-space shards (g16_pk_load_bucket_shard): rank r keeps the entries whose bucket b has b mod world == r, re-indexes them
    by k = b / world, reduces its LOCAL buckets and turns  sum_k (k+1) S_k, sum_k S_k  into its share
    world * sum_k (k+1) S_k + (r + 1 - world) * sum_k S_k.  The CPU model of exactly that code (plan, filter, fold) for every rank;
    the shares must add up to the oracle's MSM -- one class (c = 9), several (17, 20), rank counts that do not divide 2^(c-1)."""
    cid = CURVE_ID[cp.name]
    n = 60 if c_win <= -17 else 200
    for g2 in (0, 1):
        if g2 and (c_win != -12 or world not in (2, 5)):
            continue
        bases = orc.synth_bases(cp.name, bool(g2), 5, n)
        sc = orc.rand_fr(cp.name, 177 + world - c_win, n)
        p = cp.r
        specials = ints_to_mont([0, 1, p - 1, 2, (p - 1) // 2, (p + 1) // 2], p, 4)
        sc[: len(specials)] = specials
        sc[20:50] = sc[20]   # one bucket per window holds 30 entries: all on one rank
        bases[9] = 0
        total = None
        out = np.zeros(bases.shape[1], dtype=np.uint64)
        for r in range(world):
            assert lb.c.g16_host_msm_model_shard(cid, g2, ptr64(bases), ptr64(sc), n, c_win, r, world, ptr64(out)) == 0
            total = out.copy() if total is None else orc.group_op(cp.name, bool(g2), 0, total, out)
        assert (total == orc.msm(cp.name, bool(g2), bases, sc)).all(), (g2, c_win, world)
    # per-window plans have no bucket-space shard; rank out of range
    assert lb.c.g16_host_msm_model_shard(cid, 0, ptr64(bases), ptr64(sc), n, 5, 0, 2, ptr64(out)) != 0
    assert lb.c.g16_host_msm_model_shard(cid, 0, ptr64(bases), ptr64(sc), n, -9, 2, 2, ptr64(out)) != 0


def test_host_msm_model_bucket_shards_fuzz(lb, orc):
    """property, over random rank counts, window sizes, sizes and scalar shapes (uniform, small, all equal, mostly zero): the ranks'
    bucket-space shares -- each from the product's plan / filter / re-index / fold code -- add up to the oracle's MSM.  Deterministic
    draws (hypothesis with a fixed seed and no example database), BN254 G1 for speed."""
    from hypothesis import HealthCheck, given, seed, settings
    from hypothesis import strategies as st

    cp = pm.BN254
    cid = CURVE_ID[cp.name]
    bases_all = orc.synth_bases(cp.name, False, 17, 96)
    pool = orc.rand_fr(cp.name, 4242, 96)
    small = ints_to_mont(list(range(96)), cp.r, 4)

    @seed(20260926)
    @settings(max_examples=25, deadline=None, database=None, suppress_health_check=list(HealthCheck))
    @given(world=st.integers(2, 64), c=st.integers(9, 20), n=st.integers(1, 96), shape=st.sampled_from(["uniform", "small", "equal", "sparse"]),
           rot=st.integers(0, 95))
    def prop(world, c, n, shape, rot):
        bases = np.ascontiguousarray(np.roll(bases_all, rot, axis=0)[:n])
        if shape == "uniform":
            sc = np.roll(pool, rot, axis=0)[:n].copy()
        elif shape == "small":
            sc = np.roll(small, rot, axis=0)[:n].copy()
        elif shape == "equal":
            sc = np.repeat(pool[rot:rot + 1], n, axis=0)
        else:
            sc = np.zeros((n, 4), dtype=np.uint64)
            sc[:: 7] = pool[rot]
        sc = np.ascontiguousarray(sc)
        total = None
        out = np.zeros(bases.shape[1], dtype=np.uint64)
        for r in range(world):
            assert lb.c.g16_host_msm_model_shard(cid, 0, ptr64(bases), ptr64(sc), n, -c, r, world, ptr64(out)) == 0
            total = out.copy() if total is None else orc.group_op(cp.name, False, 0, total, out)
        assert (total == orc.msm(cp.name, False, bases, sc)).all(), (world, c, n, shape, rot)

    prop()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_host_selftest_fp30(lb, cp):
    """reduced-radix lazy arithmetic of the G1 bucket kernel (fp30.hpp) vs the standard field / group code"""
    for seed in (1, 2, 3):
        assert lb.c.g16_host_selftest(CURVE_ID[cp.name], seed, 400) == 0


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_synth_circuit_matches_oracle(lb, orc, cp):
    k = 6
    nc = (1 << k) - 2
    z = np.zeros((nc + 3, 4), dtype=np.uint64)
    rp = np.zeros(nc + 1, dtype=np.uint64)
    cols = [np.zeros(nc, dtype=np.uint32) for _ in range(3)]
    val = np.zeros((nc, 4), dtype=np.uint64)
    assert lb.c.g16_synth_circuit(CURVE_ID[cp.name], k, 3, ptr64(z), ptr64(rp), ptr32(cols[0]), ptr32(cols[1]), ptr32(cols[2]),
                                  ptr64(val)) == 0
    ck = orc.syn_circuit(cp.name, k, 3)
    assert (z == ck.z).all() and (rp == ck.abc[0].row_ptr).all() and (val == ck.abc[0].val).all()
    for i in range(3):
        assert (cols[i] == ck.abc[i].col).all()
