"""SURVEY.md row f1 -- CanonicalSerialize of the reference's data types (src/data_structures.rs:8,31,125).  The product's
encoder/decoder (csrc/serialize.hip + groth16_amd/serialize.py, CPU) against the big-int model's independent encoder, the
frozen proof bytes of the golden fixtures and the one external known answer available (the IETF/zcash compressed BLS12-381
G1 generator); round trips; rejection of malformed input.  The formats themselves are restated from published definitions
and are NOT verified against arkworks (no toolchain here)."""
import numpy as np
import pytest

import pymodel as pm
from helpers import arr_to_g1, arr_to_g2, g1_to_arr, g2_to_arr

CURVES = [pm.BLS12_381, pm.BN254]


def _model_bytes(cp, P, g2):
    if cp.name == "bls12_381":
        return pm.compress_g2_bls(P) if g2 else pm.compress_g1_bls(P)
    return pm.compress_g2_bn(P) if g2 else pm.compress_g1_bn(P)


def _points(cp, g2, n, seed):
    G1, G2 = pm.groups(cp)
    G, gen = (G2, cp.g2) if g2 else (G1, cp.g1)
    rng = pm.SplitMix64(seed)
    pts = [G.mul(gen, rng.field(cp.r - 1) + 1) for _ in range(n)]
    pts[1] = None                      # identity
    pts[2] = G.neg(pts[3])             # a pair (P, -P): both signs of y
    return pts


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("g2", [False, True])
def test_compressed_points_match_model_and_round_trip(cp, g2):
    import groth16_amd.serialize as ser

    pts = _points(cp, g2, 12, 5 + g2)
    arr = (g2_to_arr if g2 else g1_to_arr)(pts, cp)
    data = ser.serialize_points(cp.name, arr, g2, compressed=True)
    sz = ser.point_size(cp.name, g2, True)
    assert sz == (cp.fq_limbs64 * 8) * (2 if g2 else 1)
    for i, P in enumerate(pts):
        assert data[i * sz: (i + 1) * sz] == _model_bytes(cp, P, g2), i
    for validate in (0, 1, 2):
        back = ser.deserialize_points(cp.name, data, len(pts), g2, True, validate)
        assert (back == arr).all()
    raw = ser.serialize_points(cp.name, arr, g2, compressed=False)
    assert len(raw) == 2 * len(data)
    assert (ser.deserialize_points(cp.name, raw, len(pts), g2, False, 2) == arr).all()


def test_zcash_generator_known_answer():
    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    data = ser.serialize_points(cp.name, g1_to_arr([cp.g1], cp), False, True)
    assert data.hex() == ("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
                          "6c55e83ff97a1aeffb3af00adb22c6bb")
    assert arr_to_g1(ser.deserialize_points(cp.name, data, 1, False, True, 2), cp)[0] == cp.g1


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_proof_bytes_match_golden(cp):
    import groth16_amd as g
    import groth16_amd.serialize as ser
    from test_golden import _p1, _p2, load_cases

    n = 0
    for name, _, _cs, _z, _r, _s, _pk, ex in load_cases(cp.name):
        proof = g.Proof(g1_to_arr([_p1(ex["proof_a"])], cp)[0], g2_to_arr([_p2(ex["proof_b"])], cp)[0], g1_to_arr([_p1(ex["proof_c"])], cp)[0])
        data = ser.proof_to_bytes(cp.name, proof, compressed=True)
        assert data.hex() == ex["proof_bytes_unverified_encoding"], name
        assert ser.proof_from_bytes(cp.name, data) == proof
        assert ser.proof_from_bytes(cp.name, ser.proof_to_bytes(cp.name, proof, compressed=False), compressed=False) == proof
        n += 1
    assert n == 9


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
@pytest.mark.parametrize("compressed", [True, False])
def test_proving_key_round_trip(orc, cp, compressed):
    import groth16_amd as g
    import groth16_amd.serialize as ser

    ck = orc.syn_circuit(cp.name, 4, 2)
    pk, ex = orc.setup(ck, 3)
    key = g.ProvingKey(cp.name, pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query, pk.b_g1_query, pk.b_g2_query,
                       pk.h_query, pk.l_query, ex["gamma_g2"][None, :], ex["gamma_abc"])
    data = ser.proving_key_to_bytes(cp.name, key, compressed)
    L, n, nv, ni = cp.fq_limbs64 * 8, ck.domain_size, ck.num_vars, ck.num_inputs
    g1s, g2s = (L, 2 * L) if compressed else (2 * L, 4 * L)
    # vk: 1 G1 + 3 G2 + Vec<G1>(ni); then 2 G1 + Vec(nv) + Vec(nv) + Vec<G2>(nv) + Vec(n - 1) + Vec(nv - ni)
    assert len(data) == (3 + ni + 2 * nv + (n - 1) + (nv - ni)) * g1s + (3 + nv) * g2s + 6 * 8
    back, used = ser.proving_key_from_bytes(cp.name, data, compressed, validate=2)
    assert used == len(data)
    for f in ("alpha_g1", "beta_g1", "delta_g1", "beta_g2", "delta_g2", "a_query", "b_g1_query", "b_g2_query", "h_query", "l_query",
              "gamma_abc_g1"):
        assert (np.asarray(getattr(back, f)).reshape(-1) == np.asarray(getattr(key, f)).reshape(-1)).all(), f
    assert (back.gamma_g2.reshape(-1) == ex["gamma_g2"]).all()


@pytest.mark.parametrize("cp", CURVES, ids=lambda c: c.name)
def test_malformed_input_is_rejected(cp):
    import groth16_amd as g
    import groth16_amd.serialize as ser
    from groth16_amd.binding import InvalidData

    G1, _ = pm.groups(cp)
    # an x with no point on the curve
    x = 1
    while pow((x ** 3 + cp.b1) % cp.q, (cp.q - 1) // 2, cp.q) == 1:
        x += 1
    L = cp.fq_limbs64 * 8
    if cp.name == "bls12_381":
        bad_x = bytearray(x.to_bytes(L, "big"))
        bad_x[0] |= 0x80
        not_canonical = bytearray((cp.q + 1).to_bytes(L, "big"))
        not_canonical[0] |= 0x80
        wrong_mode = bytearray(ser.serialize_points(cp.name, g1_to_arr([cp.g1], cp), False, True))
        wrong_mode[0] &= 0x7f      # compression flag cleared on a compressed encoding
    else:
        bad_x = bytearray(x.to_bytes(L, "little"))
        not_canonical = bytearray((cp.q + 1).to_bytes(L, "little"))
        wrong_mode = bytearray(L)
        wrong_mode[-1] = 0xC0      # infinity and sign flag together
    for data in (bad_x, not_canonical, wrong_mode):
        with pytest.raises(InvalidData):
            ser.deserialize_points(cp.name, bytes(data), 1, False, True, 2)
    # an off-curve uncompressed point passes Validate::No and fails the on-curve check
    P = G1.mul(cp.g1, 5)
    arr = g1_to_arr([(P[0], (P[1] + 1) % cp.q)], cp)
    raw = ser.serialize_points(cp.name, arr, False, compressed=False)
    assert (ser.deserialize_points(cp.name, raw, 1, False, False, 0) == arr).all()
    with pytest.raises(InvalidData):
        ser.deserialize_points(cp.name, raw, 1, False, False, 1)
    with pytest.raises(ValueError):
        ser.deserialize_points(cp.name, raw[:-1], 1, False, False, 0)
    assert g.Proof  # the type the round trips return


def test_bulk_round_trip_uses_threads(orc):
    """65 536 points (the threaded path of g16_serialize_points / g16_deserialize_points), compressed, Validate::No"""
    import time

    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    n = 1 << 16
    pts = np.tile(orc.synth_bases(cp.name, False, 4, 1 << 10), (n >> 10, 1))
    pts[12345] = 0                                 # an identity in the middle
    t0 = time.time()
    data = ser.serialize_points(cp.name, pts, False, True)
    back = ser.deserialize_points(cp.name, data, n, False, True, validate=1)
    assert (back == pts).all()
    assert time.time() - t0 < 120
    bad = bytearray(data)
    bad[48 * 40000] ^= 0x1f                        # corrupt one x: with overwhelming probability no longer canonical / on the curve
    from groth16_amd.binding import InvalidData
    try:
        out = ser.deserialize_points(cp.name, bytes(bad), n, False, True, validate=1)
        assert not (out[40000] == pts[40000]).all()   # (the corrupted x happened to be another valid point)
    except InvalidData:
        pass
