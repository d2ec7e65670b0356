"""Known answers that come from OUTSIDE this repository (published specifications), so that the byte formats and the group
law are pinned to something other than the builder's own models:
  * IETF pairing-friendly-curves draft / zcash BLS12-381 encoding: G1 and G2 generators, compressed and uncompressed, and the
    encodings of the identity;
  * EIP-196 (alt_bn128 = BN254): the doubling of the generator (1, 2);
  * ark-serialize's SWFlags layout for BN254 (little-endian, flags in the two top bits of the last byte:
    0x80 = y is the larger of (y, -y), 0x40 = identity) on the points whose encodings can be written down by hand.
Everything else about the Proof / VerifyingKey / ProvingKey containers stays labelled unverified."""
import numpy as np
import pytest

import pymodel as pm
from helpers import CURVE_ID, arr_to_g1, g1_to_arr, g2_to_arr, oracle, ptr64  # noqa: F401

G1_GEN_X = "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
G1_GEN_Y = "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"
G2_GEN_X_C1 = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e"
G2_GEN_X_C0 = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
G2_GEN_Y_C1 = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be"
G2_GEN_Y_C0 = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"


def test_bls12_381_generators_zcash_encoding():
    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    g1, g2 = g1_to_arr([cp.g1], cp), g2_to_arr([cp.g2], cp)
    # compressed: flag 0x80 on the first byte; neither generator's y is the larger root, so 0x20 stays clear
    assert ser.serialize_points(cp.name, g1, False, True).hex() == "97" + G1_GEN_X[2:]
    assert ser.serialize_points(cp.name, g2, True, True).hex() == "93" + G2_GEN_X_C1[2:] + G2_GEN_X_C0
    # uncompressed: x | y, Fq2 as c1 | c0, no flag bits
    assert ser.serialize_points(cp.name, g1, False, False).hex() == G1_GEN_X + G1_GEN_Y
    assert ser.serialize_points(cp.name, g2, True, False).hex() == G2_GEN_X_C1 + G2_GEN_X_C0 + G2_GEN_Y_C1 + G2_GEN_Y_C0
    # and back, with full validation (on curve + prime-order subgroup)
    for arr, is_g2 in ((g1, False), (g2, True)):
        for comp in (True, False):
            data = ser.serialize_points(cp.name, arr, is_g2, comp)
            assert (ser.deserialize_points(cp.name, data, 1, is_g2, comp, 2) == arr).all()


def test_bls12_381_negated_generator_sets_the_sort_flag():
    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    G1, G2 = pm.groups(cp)
    assert ser.serialize_points(cp.name, g1_to_arr([G1.neg(cp.g1)], cp), False, True).hex() == "b7" + G1_GEN_X[2:]
    assert ser.serialize_points(cp.name, g2_to_arr([G2.neg(cp.g2)], cp), True, True).hex() == "b3" + G2_GEN_X_C1[2:] + G2_GEN_X_C0


def test_bls12_381_identity_encodings():
    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    z1, z2 = np.zeros((1, 12), dtype=np.uint64), np.zeros((1, 24), dtype=np.uint64)
    assert ser.serialize_points(cp.name, z1, False, True).hex() == "c0" + "00" * 47
    assert ser.serialize_points(cp.name, z2, True, True).hex() == "c0" + "00" * 95
    assert ser.serialize_points(cp.name, z1, False, False).hex() == "40" + "00" * 95
    assert ser.serialize_points(cp.name, z2, True, False).hex() == "40" + "00" * 191
    assert not ser.deserialize_points(cp.name, bytes.fromhex("c0" + "00" * 47), 1, False, True, 2).any()


def test_bn254_ark_serialize_flags_on_hand_written_points():
    import groth16_amd.serialize as ser

    cp = pm.BN254
    G1, _ = pm.groups(cp)
    one = "01" + "00" * 31
    gen, neg = g1_to_arr([cp.g1], cp), g1_to_arr([G1.neg(cp.g1)], cp)
    # (1, 2): y = 2 < q - 2, "positive" -> no flag; the negation carries 0x80 on the last byte; little-endian x
    assert ser.serialize_points(cp.name, gen, False, True).hex() == one
    assert ser.serialize_points(cp.name, neg, False, True).hex() == "01" + "00" * 30 + "80"
    assert ser.serialize_points(cp.name, gen, False, False).hex() == one + "02" + "00" * 31
    z1 = np.zeros((1, 8), dtype=np.uint64)
    assert ser.serialize_points(cp.name, z1, False, True).hex() == "00" * 31 + "40"
    assert ser.serialize_points(cp.name, z1, False, False).hex() == "00" * 63 + "40"
    for arr in (gen, neg):
        assert (ser.deserialize_points(cp.name, ser.serialize_points(cp.name, arr, False, True), 1, False, True, 2) == arr).all()


EIP196_2G = (0x030644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD3, 0x15ED738C0E0A7C92E7845F96B2AE9C0A68A6A449E3538FC7FF3EBF7A5A18A2C4)


def test_bn254_generator_doubling_eip196():
    """2 * (1, 2) on alt_bn128 as published with EIP-196: pins the group law of the model, the oracle and the product's host code"""
    import groth16_amd as g

    cp = pm.BN254
    G1, _ = pm.groups(cp)
    assert G1.add(cp.g1, cp.g1) == EIP196_2G
    assert G1.mul(cp.g1, 2) == EIP196_2G
    want = g1_to_arr([EIP196_2G], cp)[0]
    gen = g1_to_arr([cp.g1], cp)[0]
    orc = oracle()
    assert (orc.group_op(cp.name, False, 0, gen, gen) == want).all()
    out = np.zeros_like(gen)
    lb = g.lib()
    two = np.array([2, 0, 0, 0], dtype=np.uint64)
    assert lb.c.g16_host_group_op(CURVE_ID[cp.name], 0, 0, ptr64(gen), ptr64(gen), ptr64(out)) == 0 and (out == want).all()
    assert lb.c.g16_host_group_op(CURVE_ID[cp.name], 0, 1, ptr64(gen), ptr64(two), ptr64(out)) == 0 and (out == want).all()


@pytest.mark.gpu
def test_bn254_generator_doubling_eip196_gpu_msm():
    """the same known answer through the GPU: an MSM of the generator with scalars summing to 2 (tangent case included)"""
    import groth16_amd as g

    cp = pm.BN254
    gen = g1_to_arr([cp.g1, cp.g1], cp)
    from helpers import ints_to_mont

    with g.Groth16(cp.name, 0) as prover:
        for scalars in ([1, 1], [2, 0], [cp.r - 1, 3]):
            out = prover.msm(gen, ints_to_mont(scalars, cp.r, 4))
            assert arr_to_g1(out[None, :], cp)[0] == EIP196_2G
