"""GPU parity tests (run on the MI355X box with -m gpu): every call goes through the C ABI of
libg16_mi355x.so via the groth16_amd host mirror and is compared bit-for-bit with the CPU oracle on
the same seeded inputs, plus size-independent properties at larger sizes."""
import numpy as np
import pytest

import pymodel as pm
from helpers import (arr_to_g1, arr_to_g2, circuit_from_pymodel, ints_to_mont, mont_to_ints, oracle, pk_from_pymodel)

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


@pytest.fixture(scope="module", params=CURVES)
def env(request, g, orc):
    prover = g.Groth16(request.param, 0)
    yield request.param, prover
    prover.close()


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 7, 11, 12, 13, 16, 18])
def test_ntt_matches_oracle(env, orc, log_n):
    curve, prover = env
    x = orc.rand_fr(curve, 100 + log_n, 1 << log_n)
    for inverse in (False, True):
        for coset in (False, True):
            got = prover.ntt(x, inverse, coset)
            assert (got == orc.ntt(curve, x, inverse, coset)).all(), (inverse, coset)


def test_ntt_roundtrip_2_22(env, orc):
    """size-independent property at BASELINE.json's full domain size: ifft(fft(x)) == x, coset too"""
    curve, prover = env
    n = 1 << 22
    x = np.tile(orc.rand_fr(curve, 5, 1 << 12), (n >> 12, 1))
    x[:, 0] ^= np.arange(n, dtype=np.uint64) & np.uint64(0xFFFF)  # distinct, still < modulus (low limb tweak)
    y = prover.ntt(x, False, True)
    assert not (y == x).all()
    assert (prover.ntt(y, True, True) == x).all()


@pytest.mark.parametrize("k", [2, 3, 5, 9, 11, 13, 16])
def test_witness_map_matches_oracle(env, orc, g, k):
    curve, prover = env
    ck = orc.syn_circuit(curve, k, 40 + k)
    h = prover.witness_map_from_matrices(mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
    assert (h == orc.witness_map(ck)).all()
    assert not h[-1].any()  # deg h <= n-2 for a satisfied system


def test_witness_map_padded_domain_and_dense_rows(env, orc, g):
    """nc + num_inputs not a power of two (zero padding rows) and rows with several terms"""
    curve, prover = env
    cp = CP[curve]
    for cs, z in (pm.mimc_circuit(cp, 20, 4), pm.syn_circuit(cp, 5, 6, dense=True)):
        ck = circuit_from_pymodel(cp, cs, z)
        h = prover.witness_map_from_matrices(mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
        assert mont_to_ints(h, cp.r) == pm.witness_map_from_matrices(cp, cs, z)


@pytest.fixture(params=["per_window", "merged", "merged_c17", "merged_c20", "per_window_affine2", "merged_affine3", "merged_c17_affine4",
                        "merged_c20_affine1"])
def msm_path(request, monkeypatch):
    """which bucket scheme g16_msm_* runs: per-window buckets (ad-hoc bases, the default of that entry point) or the
    proving-key scheme -- window tables + merged windows -- with the window size from the cost model or forced so that the
    bucket set is cut into 2 / 16 classes (the shape a 2^22 key uses) even at test sizes.  The library reads the
    variables at call time."""
    name = request.param
    # "..._affineR": R levels of batched-affine pairwise additions in front of the XYZZ bucket pass.  OPT-IN and OFF by default
    # (make_msm_plan sets affine_levels = 0 unless G16_MSM_AFFINE_LEVELS is set: the levels measured slower);
    # forced here so that holes, tangents, cancellations and identity bases keep running through the path at test sizes
    monkeypatch.setenv("G16_MSM_AFFINE_LEVELS", name[-1] if "_affine" in name else "0")
    name = name.split("_affine")[0]
    if name != "per_window":
        monkeypatch.setenv("G16_MSM_API_PRECOMP", "1")
    if name.startswith("merged_c"):
        monkeypatch.setenv("G16_MSM_PRECOMP_WINDOW", name[len("merged_c"):])
    return request.param


@pytest.mark.parametrize("g2", [False, True])
@pytest.mark.parametrize("n", [0, 1, 2, 31, 32, 257, 4096, 20000])
def test_msm_matches_oracle(env, orc, g2, n, msm_path):
    curve, prover = env
    bases = orc.synth_bases(curve, g2, 3, max(n, 1))[:n]
    sc = orc.rand_fr(curve, 7 + n, max(n, 1))[:n]
    got = prover.msm(bases, sc, g2)
    want = orc.msm(curve, g2, bases, sc) if n else np.zeros_like(got)
    assert (got == want).all()


@pytest.mark.parametrize("seg", [8, 19, 60, 70, 139])
@pytest.mark.parametrize("g2", [False, True])
@pytest.mark.parametrize("merged", [False, True])
def test_msm_any_segment_length(env, orc, g2, merged, seg, monkeypatch):
    """the bucket pass walks segments of ANY length (round 3: the kernels divide by the length instead of shifting; the plan itself
    keeps 64 / 32 / 16 after the A/B of profiles/r03_ab_segment_length.txt): odd lengths, lengths above and below the mean bucket
    load, with repeated scalars so that buckets span many segments (heavy-bucket path) -- result == oracle"""
    monkeypatch.setenv("G16_MSM_SEGMENT", str(seg))
    if merged:
        monkeypatch.setenv("G16_MSM_API_PRECOMP", "1")
    curve, prover = env
    n = 5000
    bases = orc.synth_bases(curve, g2, 5, n)
    sc = orc.rand_fr(curve, 40 + seg, n)
    sc[100:2100] = sc[100]          # 2000 equal scalars: every window gets one bucket of 2000 entries
    bases[50:400:9] = 0
    got = prover.msm(bases, sc, g2)
    assert (got == orc.msm(curve, g2, bases, sc)).all()


@pytest.mark.parametrize("G", [4, 16, 32])
@pytest.mark.parametrize("g2", [False, True])
@pytest.mark.parametrize("merged", [False, True])
def test_msm_buckets_per_lane_of_the_reduction(env, orc, g2, merged, G, monkeypatch):
    """G16_MSM_REDUCE_G: the chunk size of the bucket reduction (8 by default, 16 for the 2^19-bucket sets of whole keys) changes
    the chunk count, the number of bit planes and the host's recombination -- result == oracle for every size"""
    monkeypatch.setenv("G16_MSM_REDUCE_G", str(G))
    if merged:
        monkeypatch.setenv("G16_MSM_API_PRECOMP", "1")
    curve, prover = env
    n = 6000
    bases = orc.synth_bases(curve, g2, 8, n)
    sc = orc.rand_fr(curve, 60 + G, n)
    sc[200:900] = sc[200]
    assert (prover.msm(bases, sc, g2) == orc.msm(curve, g2, bases, sc)).all()


@pytest.mark.parametrize("g2", [False, True])
def test_msm_adversarial_inputs(env, orc, g2, msm_path):
    """zero scalars, r-1, all-equal scalars (benches/bench.rs:52-54 shape), identity bases, repeated
    bases, P and -P in the same bucket, all-zero, all-one"""
    curve, prover = env
    cp = CP[curve]
    n = 3000
    bases = orc.synth_bases(curve, g2, 11, n)
    sc = orc.rand_fr(curve, 13, n)
    sp = ints_to_mont([0, 1, cp.r - 1, 2, (cp.r - 1) // 2, (cp.r + 1) // 2], cp.r, 4)
    sc[: len(sp)] = sp
    bases[10:500:7] = 0  # identity bases
    bases[21] = bases[20]
    sc[21] = sc[20]  # same point twice in every bucket it lands in -> doubling branch
    neg = bases[30].copy()
    L = cp.fq_limbs64
    conv, back = (None, None)
    from helpers import g1_to_arr, g2_to_arr
    if g2:
        P = arr_to_g2(bases[30], cp)[0]
        neg = g2_to_arr([pm.groups(cp)[1].neg(P)], cp)[0]
    else:
        P = arr_to_g1(bases[30], cp)[0]
        neg = g1_to_arr([pm.groups(cp)[0].neg(P)], cp)[0]
    bases[31] = neg
    sc[31] = sc[30]  # P and -P with equal scalars cancel
    for scal in (sc, np.repeat(sc[40:41], n, axis=0), np.zeros_like(sc), np.repeat(ints_to_mont([1], cp.r, 4), n, axis=0)):
        assert (prover.msm(bases, scal, g2) == orc.msm(curve, g2, bases, scal)).all()


@pytest.mark.parametrize("path", ["per_window", "merged"])
def test_msm_linearity_large(env, orc, path, monkeypatch):
    """size-independent property at 2^20 points: msm(b, s1) + msm(b, s2) == msm(b, s1 + s2)"""
    if path == "merged":
        monkeypatch.setenv("G16_MSM_API_PRECOMP", "1")
    curve, prover = env
    cp = CP[curve]
    n = 1 << 20
    bases = np.tile(orc.synth_bases(curve, False, 2, 1 << 12), (n >> 12, 1))
    s1 = np.tile(orc.rand_fr(curve, 1, 1 << 12), (n >> 12, 1))
    s2 = np.tile(orc.rand_fr(curve, 2, 1 << 12), (n >> 12, 1))
    r1, r2 = prover.msm(bases, s1), prover.msm(bases, s2)
    s12 = np.tile(np.stack([orc.field_op(curve, 0, 0, a, b) for a, b in zip(s1[: 1 << 12], s2[: 1 << 12])]), (n >> 12, 1))
    r12 = prover.msm(bases, s12)
    assert (orc.group_op(curve, False, 0, r1, r2) == r12).all()
    # and against the oracle on the folded problem: sum over 256 repeats = 256 * msm(base block, s)
    small = orc.msm(curve, False, bases[: 1 << 12], s1[: 1 << 12])
    k = np.array([n >> 12, 0, 0, 0], dtype=np.uint64)
    assert (orc.group_op(curve, False, 1, small, k) == r1).all()


@pytest.mark.parametrize("k", [3, 6, 10])
def test_proof_valid_crs_and_trapdoor(env, orc, g, k):
    """the reference's own test pattern (src/test.rs:45-73: prove, then check) with a stronger
    check: GPU proof == oracle proof == proof computed from the trapdoor"""
    curve, prover = env
    ck = orc.syn_circuit(curve, k, 1)
    pk, ex = orc.setup(ck, 5)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    h = orc.witness_map(ck)
    for r, s in ((orc.rand_fr(curve, 11, 1)[0], orc.rand_fr(curve, 12, 1)[0]),
                 (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 13, 1)[0]),  # r == 0 skips B in G1 (prover.rs:98)
                 (orc.rand_fr(curve, 14, 1)[0], np.zeros(4, dtype=np.uint64))):
        proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
        want, _ = orc.prove(pk, ck, r, s)
        assert (proof.flat() == want).all()
        assert (orc.trapdoor_proof(ck, ex, h, r, s) == want).all()
    nozk = prover.create_proof_with_reduction_no_zk(gp, gm, ck.num_inputs, ck.num_constraints, ck.z)
    z4 = np.zeros(4, dtype=np.uint64)
    assert (nozk.flat() == orc.prove(pk, ck, z4, z4)[0]).all()


def test_pipelined_prover_two_contexts_one_key(env, orc, g):
    """throughput mode: two contexts on the one GPU over ONE device-resident key and circuit (a g16_pk / g16_circuit may serve every
    single-device context of its GPU), two worker threads, proofs in flight side by side -- every proof must equal the oracle's, the
    key and the circuit must have been loaded once, and the plain C call with a key owned by ANOTHER context must work as well"""
    import ctypes as C

    from groth16_amd.binding import ProofC, ptr64

    curve, prover = env
    ck = orc.syn_circuit(curve, 11, 2)
    pk, _ = orc.setup(ck, 9)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    rs = [(orc.rand_fr(curve, 300 + i, 1)[0], orc.rand_fr(curve, 400 + i, 1)[0]) for i in range(8)]
    rs[3] = (np.zeros(4, dtype=np.uint64), rs[3][1])     # r == 0 among them
    with g.PipelinedProver(curve, 0) as pp:
        futs = [pp.submit(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z) for r, s in rs]
        proofs = [f.result(timeout=300) for f in futs]
        assert len(pp._owner._pks) == 1 and len(pp._owner._cks) == 1 and not pp._second._pks and not pp._second._cks
        # the C ABI directly: the existing fixture context proves with the PipelinedProver's key and circuit handles
        dpk, dck = pp._owner._pk(gp, ck.num_inputs), pp._owner._ck(gm)
        out = ProofC()
        lb = prover._ctx.lib
        z = np.ascontiguousarray(ck.z)
        lb.check(lb.c.g16_prove(prover._ctx.handle, dpk.handle, dck.handle, z.ctypes.data, z.shape[0], 0, ptr64(rs[0][0]), ptr64(rs[0][1]),
                                C.byref(out)))
        L = len(proofs[0].a) // 2
        direct = np.concatenate([np.array(out.a[: 2 * L], dtype=np.uint64), np.array(out.b[: 4 * L], dtype=np.uint64),
                                 np.array(out.c[: 2 * L], dtype=np.uint64)])
    for (r, s), proof in zip(rs, proofs):
        want, _ = orc.prove(pk, ck, r, s)
        assert (proof.flat() == want).all()
    assert (direct == orc.prove(pk, ck, *rs[0])[0]).all()


def test_proof_valid_crs_2_16_trapdoor_and_pairing(env, orc, g):
    """a VALID proving key at 2^16 constraints (oracle setup, a few seconds): the GPU proof equals the closed form
    computed from the trapdoor (scalar arithmetic and three fixed-base multiplications, no MSM) and passes the pairing verifier"""
    from test_verifier import _proof_from_flat, _vk_from_oracle

    curve, prover = env
    cp = CP[curve]
    ck = orc.syn_circuit(curve, 16, 21)
    pk, ex = orc.setup(ck, 9)
    r, s = orc.rand_fr(curve, 31, 1)[0], orc.rand_fr(curve, 32, 1)[0]
    proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, pk), r, s, mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
    assert (orc.trapdoor_proof(ck, ex, orc.witness_map(ck), r, s) == proof.flat()).all()
    public = mont_to_ints(ck.z[1: ck.num_inputs], cp.r)
    vk = _vk_from_oracle(cp, pk, ex)
    assert pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), public)
    assert not pm.verify_proof(cp, vk, _proof_from_flat(cp, proof.flat()), [(public[0] + 1) % cp.r])


@pytest.mark.parametrize("scheme", [{"G16_MSM_PRECOMP": "0"}, {"G16_MSM_PRECOMP_WINDOW": "16"}, {"G16_MSM_PRECOMP_WINDOW": "17"},
                                    {"G16_MSM_PRECOMP_WINDOW": "20"}, {"G16_PK_TABLE_BUDGET_MB": "1.0"},
                                    {"G16_MSM_PRECOMP": "0", "G16_MSM_AFFINE_LEVELS": "2"}, {"G16_MSM_AFFINE_LEVELS": "3"},
                                    {"G16_MSM_PRECOMP_WINDOW": "10", "G16_MSM_AFFINE_LEVELS": "4"}],
                         ids=["plain_bases", "tables_c16", "tables_c17", "tables_c20", "tables_do_not_fit", "plain_bases_affine2", "tables_affine3",
                              "tables_c10_affine4"])
def test_proof_bucket_schemes(env, orc, g, scheme, monkeypatch):
    """g16_pk_load decides how the key is held (plain bases + per-window buckets, or window tables + merged windows at
    the window size of the cost model); every choice must give the oracle's proof, including the 16-class shape of a
    2^22 key, a sharded key whose l range needs its own sort, and the fall-back to plain bases when a table cannot be
    allocated (here: a 1 MB cap that the G2 table of this 512-constraint key exceeds -- 511 points x 192 / 128 B x >= 13 rows --
    after the G1 tables were built)"""
    curve, prover = env
    for k_, v in scheme.items():
        monkeypatch.setenv(k_, v)
    ck = orc.syn_circuit(curve, 9, 5)
    pk, _ = orc.setup(ck, 6)
    gm = mats_of(g, ck)
    r, s = orc.rand_fr(curve, 41, 1)[0], orc.rand_fr(curve, 42, 1)[0]
    want, _ = orc.prove(pk, ck, r, s)
    proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, pk), r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
    assert (proof.flat() == want).all()
    gp = pk_of(g, pk)
    parts = [prover.prove_partial(gp, gm, ck.z, (i, 3)) for i in range(3)]
    assert (prover.prove_finalize(gp, ck.num_inputs, parts, r, s, (0, 3)).flat() == want).all()
    # the key says how it is held -- and WHY, when it is the slower plain-bases form (g16_pk_get_info.table_fallback)
    info = prover.pk_info(gp, ck.num_inputs, (1, 3))
    if scheme.get("G16_MSM_PRECOMP") == "0":
        assert info["table_fallback"] == 1 and info["window_bits_z"] == 0, info
    elif "G16_PK_TABLE_BUDGET_MB" in scheme:
        monkeypatch.setenv("G16_PK_TABLE_BUDGET_MB", "0.01")   # (whatever window the cost model picks: no table of this key is that small)
        whole = prover.pk_info(gp, ck.num_inputs)
        assert whole["table_fallback"] == 3 and whole["window_bits_z"] == 0 and "did not fit" in whole["held_as"], whole
    else:
        assert info["table_fallback"] == 0 and info["window_bits_z"] == int(scheme.get("G16_MSM_PRECOMP_WINDOW", info["window_bits_z"])), info
        assert info["device_bytes"] > 0 and info["bucket_shard_world"] == 1


def test_proof_golden_pymodel(env, orc, g):
    """MiMC and dense-row circuits generated by the big-int model (BASELINE config #1 shape)"""
    curve, prover = env
    cp = CP[curve]
    for cs, z in (pm.mimc_circuit(cp, 6, 2), pm.syn_circuit(cp, 4, 1, dense=True)):
        pk, td = pm.generate_parameters(cp, cs, 7)
        r, s = pm.SplitMix64(1).field(cp.r), pm.SplitMix64(2).field(cp.r)
        pr = pm.create_proof_with_reduction_and_matrices(cp, pk, r, s, cs, z)
        ck, fpk = circuit_from_pymodel(cp, cs, z), pk_from_pymodel(cp, pk)
        proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, fpk), ints_to_mont([r], cp.r, 4)[0], ints_to_mont([s], cp.r, 4)[0],
                                                                mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
        assert arr_to_g1(proof.a, cp)[0] == pr.a and arr_to_g2(proof.b, cp)[0] == pr.b and arr_to_g1(proof.c, cp)[0] == pr.c


@pytest.mark.parametrize("k", [14, 16])
def test_proof_synthetic_bases(env, orc, g, k):
    curve, prover = env
    ck = orc.syn_circuit(curve, k, 2)
    pk = orc.synth_pk(ck, 9)
    r, s = orc.rand_fr(curve, 21, 1)[0], orc.rand_fr(curve, 22, 1)[0]
    proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, pk), r, s, mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
    assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()


def test_proof_domain_2_and_4(env, orc, g):
    """SURVEY.md 8(d) edge domains: n = 2 (one constraint, no public input: only the constant-one instance variable) and
    n = 4; the NTT kernels' smallest shapes and empty input_assignment slices (prover.rs:44-45)"""
    curve, prover = env
    cp = CP[curve]
    p = cp.r
    w0, w1 = pm.SplitMix64(3).field(p), pm.SplitMix64(4).field(p)
    tiny = (pm.R1CS(1, 3, [[(1, 1)]], [[(1, 2)]], [[(1, 3)]]), [1, w0, w1, w0 * w1 % p])
    for cs, z in (tiny, pm.syn_circuit(cp, 2, 9)):
        pk, _ = pm.generate_parameters(cp, cs, 7)
        r, s = pm.SplitMix64(1).field(p), pm.SplitMix64(2).field(p)
        pr = pm.create_proof_with_reduction_and_matrices(cp, pk, r, s, cs, z)
        ck, fpk = circuit_from_pymodel(cp, cs, z), pk_from_pymodel(cp, pk)
        assert ck.domain_size == (2 if cs is tiny[0] else 4)
        gm = mats_of(g, ck)
        h = prover.witness_map_from_matrices(gm, ck.num_inputs, ck.num_constraints, ck.z)
        assert (h == orc.witness_map(ck)).all()
        proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, fpk), ints_to_mont([r], p, 4)[0], ints_to_mont([s], p, 4)[0], gm,
                                                                ck.num_inputs, ck.num_constraints, ck.z)
        assert arr_to_g1(proof.a, cp)[0] == pr.a and arr_to_g2(proof.b, cp)[0] == pr.b and arr_to_g1(proof.c, cp)[0] == pr.c


@pytest.mark.parametrize("scheme", [{}, {"G16_MSM_PRECOMP": "0"}], ids=["tables", "plain_bases"])
def test_proof_half_identity_key(env, orc, g, scheme, monkeypatch):
    """SURVEY.md 8(d) robustness input: 50 % identity bases in every query of the key (real keys hold many:
    generator.rs:102-108), full proof against the oracle on the same key"""
    curve, prover = env
    for k_, v in scheme.items():
        monkeypatch.setenv(k_, v)
    ck = orc.syn_circuit(curve, 12, 4)
    pk = orc.synth_pk(ck, 5)
    rs = np.random.RandomState(7)
    for name in ("a_query", "b_g1_query", "b_g2_query", "h_query", "l_query"):
        q = getattr(pk, name)
        q[rs.rand(len(q)) < 0.5] = 0
    r, s = orc.rand_fr(curve, 51, 1)[0], orc.rand_fr(curve, 52, 1)[0]
    proof = prover.create_proof_with_reduction_and_matrices(pk_of(g, pk), r, s, mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z)
    assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()


@pytest.mark.parametrize("order", ["natural", "scattered", "reversed"])
def test_host_witness_in_pieces_whatever_columns_the_rows_read(orc, g, order):
    """A host assignment of 2^17 + 1 variables is uploaded in four pieces and the sparse mat-vec's row blocks follow them, each as soon as
    the piece holding the LAST column it reads has landed (need_col, computed at g16_circuit_load; witness_map_device / ZUpload).  The
    product chain reads its columns in row order -- the case the overlap is made for; here the variables are also renumbered at random
    and in reverse (row block 0 then reads the last piece: no overlap, same proof).  Synthetic-bases key; proof == the oracle's on the
    same renumbered instance, twice (events and buffers of the first proof must not satisfy the second)."""
    from helpers import Csr, FlatCircuit

    curve, k = "bn254", 17
    ck = orc.syn_circuit(curve, k, 29)
    nv = ck.num_vars
    perm = np.arange(nv)
    if order == "scattered":
        rs = np.random.RandomState(5)
        perm[2:] = 2 + rs.permutation(nv - 2)          # old witness variable j -> new index perm[j]; the two instance variables stay
    elif order == "reversed":
        perm[2:] = np.arange(nv - 1, 1, -1)
    z = np.zeros_like(ck.z)
    z[perm] = ck.z
    abc = [Csr(m.row_ptr, perm[m.col].astype(np.uint32), m.val) for m in ck.abc]
    ck2 = FlatCircuit(curve, ck.num_inputs, ck.num_constraints, nv, abc, z)
    pk = orc.synth_pk(ck2, 77)
    r, s = orc.rand_fr(curve, 51, 1)[0], orc.rand_fr(curve, 52, 1)[0]
    want, _ = orc.prove(pk, ck2, r, s)
    with g.Groth16(curve, 0) as prover:
        gm, gp = mats_of(g, ck2), pk_of(g, pk)
        for _ in range(2):
            proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck2.num_inputs, ck2.num_constraints, ck2.z)
            assert (proof.flat() == want).all()
        assert (prover.witness_map_from_matrices(gm, ck2.num_inputs, ck2.num_constraints, ck2.z) == orc.witness_map(ck)).all()   # h does not depend on the numbering


def test_sharded_proof_equals_single(env, orc, g):
    """MSM base sharding: 3 shards on one GPU, partial records combined by prove_finalize"""
    curve, prover = env
    ck = orc.syn_circuit(curve, 9, 3)
    pk, _ = orc.setup(ck, 8)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    r, s = orc.rand_fr(curve, 31, 1)[0], orc.rand_fr(curve, 32, 1)[0]
    parts = [prover.prove_partial(gp, gm, ck.z, (i, 3)) for i in range(3)]
    proof = prover.prove_finalize(gp, ck.num_inputs, parts, r, s, (0, 3))
    assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()


@pytest.mark.parametrize("curve", CURVES)
@pytest.mark.parametrize("mode", ["base", "bucket"])
@pytest.mark.parametrize("n_dev", [3, 2, 4])
def test_multi_device_context_single_call(orc, g, curve, n_dev, mode, monkeypatch):
    """g16_ctx_create_multi: ONE g16_prove over a context of several devices -- here the one visible GPU
    n_dev times -- shards the key inside the library, runs a host thread per device and folds the partial records; the
    proof equals the oracle's, r = 0 included (prover.rs:98-108), and the per-device form is refused on such a context.
    n_dev = 3 replicates the witness map; a power of two distributes it (four stages per device, the all-to-all as peer
    copies between the devices' buffers, h_query sharded in block order).  mode (G16_MULTI_SHARD_MODE): the key cut by base ranges,
    or -- round 5 -- in bucket space: every device holds the whole key's tables and owns the buckets b mod n_dev == device index; with
    the distributed map every device then pulls ALL blocks of h (the all-gather as peer copies)."""
    monkeypatch.setenv("G16_MULTI_SHARD_MODE", mode)
    ck = orc.syn_circuit(curve, 11, 6)
    pk, _ = orc.setup(ck, 4)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    with g.Groth16(curve, [0] * n_dev) as prover:
        assert prover._ctx.num_devices == n_dev
        for r, s in ((orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]), (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 93, 1)[0])):
            proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
            assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        assert prover.timings()["total_ms"] > 0
        info = prover.pk_info(gp, ck.num_inputs)
        assert info["n_devices"] == n_dev and info["bucket_shard_world"] == (n_dev if mode == "bucket" else 1), info
        assert (prover.witness_map_from_matrices(gm, ck.num_inputs, ck.num_constraints, ck.z) == orc.witness_map(ck)).all()
        with pytest.raises(g.G16Error):
            prover.prove_partial(gp, gm, ck.z, (0, 1))
    with g.Groth16(curve, [0]) as prover:      # n_dev == 1 is the plain context
        assert prover._ctx.num_devices == 1
        r, s = orc.rand_fr(curve, 94, 1)[0], orc.rand_fr(curve, 95, 1)[0]
        proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
        assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()


@pytest.mark.parametrize("curve", CURVES)
def test_multi_device_context_distinct_gpus(orc, g, curve):
    """the multi-device context over DISTINCT physical GPUs (peer access, hipMemcpyPeerAsync between devices, cross-device event
    waits, concurrent per-device key loads) -- skipped on a one-GPU box, which is all the build pool offers: until this has passed
    on real hardware the multi-device context is EXPERIMENTAL (include/g16_mi355x.h says so)."""
    import torch

    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs >= 2 visible GPUs")
    ids = list(range(min(ndev, 8)))
    ids = ids[: 1 << (len(ids).bit_length() - 1)]          # a power of two: the distributed witness map
    ck = orc.syn_circuit(curve, 12, 16)
    pk, _ = orc.setup(ck, 4)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    for devs in (ids, ids[:3] if len(ids) > 2 else ids):     # distributed map; replicated map (3 devices)
        with g.Groth16(curve, devs) as prover:
            for r, s in ((orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]), (np.zeros(4, dtype=np.uint64), orc.rand_fr(curve, 93, 1)[0])):
                proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
                assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()


def test_multi_device_context_peer_access_refusal(orc, g, monkeypatch):
    """g16_ctx_create_multi probes peer access per device pair instead of assuming it.  One physical GPU listed twice has no
    distinct pair, so the refusal is injected (G16_MULTI_FAKE_NO_PEER=1): with G16_MULTI_REQUIRE_PEER=1 the context is REFUSED with
    its own status (10, G16_ERR_NO_PEER_ACCESS) -- a loud failure instead of a slow prover; without it the context comes up, reports
    the pair as host-staged (g16_ctx_peer_access == 0) and still proves correctly."""
    curve = "bn254"
    ck = orc.syn_circuit(curve, 8, 6)
    pk, _ = orc.setup(ck, 4)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    r, s = orc.rand_fr(curve, 91, 1)[0], orc.rand_fr(curve, 92, 1)[0]
    with g.Groth16(curve, [0, 0]) as prover:   # same physical device twice: "direct" by definition
        lb = prover._ctx.lib
        assert lb.c.g16_ctx_peer_access(prover._ctx.handle, 0, 1) == 1 and lb.c.g16_ctx_peer_access(prover._ctx.handle, 0, 2) == -1
    monkeypatch.setenv("G16_MULTI_FAKE_NO_PEER", "1")
    with g.Groth16(curve, [0, 0]) as prover:
        lb = prover._ctx.lib
        assert lb.c.g16_ctx_peer_access(prover._ctx.handle, 0, 1) == 0 and lb.c.g16_ctx_peer_access(prover._ctx.handle, 1, 1) == 1
        proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
        assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
    monkeypatch.setenv("G16_MULTI_REQUIRE_PEER", "1")
    with pytest.raises(g.G16Error) as ei:
        g.Groth16(curve, [0, 0])
    assert ei.value.status == 10 and "no peer access" in str(ei.value)


def test_multi_device_context_padded_domain(orc, g):
    """a circuit whose domain is padded (MiMC: nc + num_inputs not a power of two) through the distributed path of the
    multi-device context, and a key that does not belong to the circuit's domain is refused"""
    curve, cp = "bls12_381", CP["bls12_381"]
    cs, z = pm.mimc_circuit(cp, 40, 9)
    ck = circuit_from_pymodel(cp, cs, z)
    pk, _ = orc.setup(ck, 6)
    gm, gp = mats_of(g, ck), pk_of(g, pk)
    r, s = orc.rand_fr(curve, 96, 1)[0], orc.rand_fr(curve, 97, 1)[0]
    with g.Groth16(curve, [0, 0, 0, 0]) as prover:
        proof = prover.create_proof_with_reduction_and_matrices(gp, r, s, gm, ck.num_inputs, ck.num_constraints, ck.z)
        assert (proof.flat() == orc.prove(pk, ck, r, s)[0]).all()
        ck2 = orc.syn_circuit(curve, 10, 2)     # another domain size than the key's
        with pytest.raises(g.G16Error):
            prover.create_proof_with_reduction_and_matrices(gp, r, s, mats_of(g, ck2), ck2.num_inputs, ck2.num_constraints, ck2.z)


def test_error_paths(env, orc, g):
    curve, prover = env
    ck = orc.syn_circuit(curve, 4, 3)
    pk = orc.synth_pk(ck, 1)
    r = orc.rand_fr(curve, 1, 1)[0]
    with pytest.raises(g.G16Error) as e:  # assignment too short: the reference would panic on the slice (prover.rs:44-45)
        prover.create_proof_with_reduction_and_matrices(pk_of(g, pk), r, r, mats_of(g, ck), ck.num_inputs, ck.num_constraints, ck.z[:-1])
    assert e.value.status == 2
    if curve == "bn254":  # 2-adicity 28: a 2^29 domain is PolynomialDegreeTooLarge (r1cs_to_qap.rs:178-179)
        big = g.ConstraintMatrices(2, 1, (1 << 28) + 5, *[(np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint32),
                                                           np.zeros((0, 4), dtype=np.uint64))] * 3)
        with pytest.raises(g.PolynomialDegreeTooLarge):
            prover.witness_map_from_matrices(big, 2, (1 << 28) + 5, ck.z[:3])


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_two_ranks_one_gpu(launcher):
    """bench.py's N > 1 path (sharded key, all-gather of partial records, finalize on every rank, max-over-ranks timing)
    with two ranks on the one visible GPU (gloo carries the exchange; RCCL refuses two ranks per device).  The bench
    itself asserts that all ranks produce the same proof; here we also require it to equal the single-rank proof.
    launcher = "self": exactly `python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE in the environment (bench.py
    starts its own ranks); "torchrun": the contract's torch.distributed.run command line.  The configs[4] leg (a second, larger
    instance timed after the headline one, reported as a field) runs here at 2^13 instead of 2^24."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(G16_BENCH_BACKEND="gloo", G16_BENCH_FORCE_DEVICE0="1", G16_BENCH_PRINT_PROOF="1")
    port = 29000 + os.getpid() % 2000
    tail = ["--gpus", "2", "--steps", "2", "--warmup", "1", "--log2", "12", "--no-cpu-baseline", "--configs4", "on", "--configs4-log2", "13"]
    # the two ways of cutting the MSMs over the ranks: "self" leaves the default (auto -> bucket-space shards: the whole key's tables
    # fit easily at this size), "torchrun" forces base-range shards
    want_mode = "bucket" if launcher == "self" else "base"
    if launcher != "self":
        tail += ["--shard-mode", "base"]
    if launcher == "self":
        cmd2 = [sys.executable, os.path.join(root, "bench.py")] + tail
    else:
        cmd2 = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(root, "bench.py")] + tail
    out2 = subprocess.run(cmd2, env=env, capture_output=True, text=True, timeout=600)
    assert out2.returncode == 0, out2.stderr[-2000:]
    lines2 = [l for l in out2.stdout.splitlines() if l.startswith("{")]
    assert len(lines2) == 1, "exactly one JSON line, from rank 0"
    d2 = json.loads(lines2[-1])
    base1 = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    out1 = subprocess.run(base1 + ["--log2", "12"], env=env, capture_output=True, text=True, timeout=600)
    assert out1.returncode == 0, out1.stderr[-2000:]
    d1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["n_gpus"] == 2 and d1["n_gpus"] == 1 and d2["scaling"] == "strong"
    assert d2["proof_sha256"] == d1["proof_sha256"]
    assert d2["config"]["shard_mode"] == want_mode and d2["config"]["pk"]["table_fallback"] == 0
    assert d2["config"]["pk"]["bucket_shard_world"] == (2 if want_mode == "bucket" else 1)
    assert d2["configs4"]["shard_mode"] == want_mode
    # round 6: the collectives of a sharded proof, each timed alone (the terms a measured N > 1 line is compared with the projection by)
    cm = d2["collective_ms"]
    assert cm["record_all_gather"] > 0 and (d2["config"].get("parallelism", "").find("distributed witness map") < 0 or cm["all_to_all_one_array"] > 0), cm
    # the single-GPU line carries the projected per-rank shares at 2 / 4 / 8 ranks under both cuts (measured on this GPU), and the
    # key is whole again afterwards
    ps = d1["projected_scaling"]
    assert "error" not in ps and ps["headline_proof_unchanged_after"], ps
    assert sorted((q["shard_mode"], q["n_gpus"]) for q in ps["points"]) == sorted((m, n) for m in ("base", "bucket") for n in (2, 4, 8))
    assert all("error" not in q and q["rank_share_ms"] > 0 for q in ps["points"]), ps["points"]
    # the line proves its ranks: world size from the process group, one record per rank gathered through the collective
    assert d2["rccl_world"] == 2 and [r["rank"] for r in d2["ranks"]] == [0, 1] and len({r["pid"] for r in d2["ranks"]}) == 2
    assert "rccl_world" not in d1
    c4 = d2["configs4"]
    assert "error" not in c4, c4
    assert c4["log2_domain"] == 13 and c4["n_gpus"] == 2 and c4["ranks_agree_on_proof"] and c4["value"] > 0
    out13 = subprocess.run(base1 + ["--log2", "13", "--key", "synthetic"], env=env, capture_output=True, text=True, timeout=600)   # the leg's key
    assert out13.returncode == 0, out13.stderr[-2000:]
    d13 = json.loads([l for l in out13.stdout.splitlines() if l.startswith("{")][-1])
    assert c4["proof_sha256"] == d13["proof_sha256"]


def test_bench_rccl_single_rank():
    """the collective path of bench.py over RCCL itself (backend "nccl"): one rank on the one visible GPU -- process group
    bound to the device, all-gather of the partial records as device tensors, all-reduce of the timing, barrier.  Proof
    must equal the plain single-GPU run's."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 31000 + os.getpid() % 2000
    base = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log2", "12", "--no-cpu-baseline"]
    env = dict(os.environ, G16_BENCH_PRINT_PROOF="1")
    envd = dict(env, G16_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    outd = subprocess.run(base, env=envd, capture_output=True, text=True, timeout=600)
    assert outd.returncode == 0, outd.stderr[-3000:]
    out1 = subprocess.run(base, env=env, capture_output=True, text=True, timeout=600)
    assert out1.returncode == 0, out1.stderr[-3000:]
    dd = json.loads([l for l in outd.stdout.splitlines() if l.startswith("{")][-1])
    d1 = json.loads([l for l in out1.stdout.splitlines() if l.startswith("{")][-1])
    assert dd["proof_sha256"] == d1["proof_sha256"] and dd["n_gpus"] == 1
    # the distributed witness map's RCCL path with the one rank a one-GPU box has: stages enqueued by g16_dwm_stage_async on the
    # library's stream, all_to_all_single issued on that same stream through torch's ExternalStream, g16_prove_partial_h behind them
    envw = dict(envd, G16_BENCH_FORCE_DWM="1", G16_DWM_FORCE_COLLECTIVE="1", MASTER_PORT=str(port + 1))
    outw = subprocess.run(base, env=envw, capture_output=True, text=True, timeout=600)
    assert outw.returncode == 0, outw.stderr[-3000:]
    dw = json.loads([l for l in outw.stdout.splitlines() if l.startswith("{")][-1])
    assert dw["proof_sha256"] == d1["proof_sha256"]
    assert "dist_witness_map_enqueue_ms" in dw["phases_ms_per_step"], "the distributed-map path was not taken"
