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


# ---- round 3: more published constants -------------------------------------------------------------------------------------
# Ethereum consensus-layer (eth2) BLS public keys of the secret keys 2 and 3 = compressed 2 G1, 3 G1 in the zcash / IETF encoding
# (secret key 1 gives the generator's encoding above).  They pin doubling, addition and the y-sort flag of BLS12-381 G1.
ETH2_PUBKEY_SK2 = "a572cbea904d67468808c8eb50a9450c9721db309128012543902d0ac358a62ae28f75bb8f1c7c42c39a8c5529bf0f4e"
ETH2_PUBKEY_SK3 = "89ece308f9d1f0131765212deca99697b112d61f9be9a5f1f3780a51335b3ff981747a0b2ca2179b96d2c0c9024e5224"
# zkcrypto/bls12_381 scalar.rs: ROOT_OF_UNITY (= GENERATOR^t with GENERATOR = 7, a primitive 2^32-th root of unity) and R = 2^256 mod r,
# both as four little-endian u64 limbs IN MONTGOMERY FORM -- the in-memory form of ark_ff::Fp and of every Fr that crosses this C ABI.
ZKCRYPTO_ROOT_OF_UNITY = [0xB9B58D8C5F0E466A, 0x5B1B4C801819D7EC, 0x0AF53AE352A31E64, 0x5BF3ADDA19E9B27B]
ZKCRYPTO_R = [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]
# EIP-196 (alt_bn128) test vectors: 3 * (1, 2)
EIP196_3G = (3353031288059533942658390886683067124040920775575537747144343083137631628272,
             19321533766552368860946552437480515441416830039777911637913418824951667761761)


def test_bls12_381_eth2_pubkeys_of_sk_2_and_3():
    """model, oracle, the product's host group code and the product's serialiser against the published encodings"""
    import groth16_amd as g
    import groth16_amd.serialize as ser

    cp = pm.BLS12_381
    G1, _ = pm.groups(cp)
    gen = g1_to_arr([cp.g1], cp)[0]
    lb, orc = g.lib(), oracle()
    for k, want_hex in ((2, ETH2_PUBKEY_SK2), (3, ETH2_PUBKEY_SK3)):
        assert pm.compress_g1_bls(G1.mul(cp.g1, k)).hex() == want_hex
        kb = np.array([k, 0, 0, 0], dtype=np.uint64)
        out = np.zeros_like(gen)
        assert lb.c.g16_host_group_op(CURVE_ID[cp.name], 0, 1, ptr64(gen), ptr64(kb), ptr64(out)) == 0
        assert ser.serialize_points(cp.name, out[None, :], False, True).hex() == want_hex
        assert ser.serialize_points(cp.name, orc.group_op(cp.name, False, 1, gen, kb)[None, :], False, True).hex() == want_hex
    # 2 G + G through the affine addition of the host code
    two, out = np.zeros_like(gen), np.zeros_like(gen)
    assert lb.c.g16_host_group_op(CURVE_ID[cp.name], 0, 0, ptr64(gen), ptr64(gen), ptr64(two)) == 0
    assert lb.c.g16_host_group_op(CURVE_ID[cp.name], 0, 0, ptr64(two), ptr64(gen), ptr64(out)) == 0
    assert ser.serialize_points(cp.name, out[None, :], False, True).hex() == ETH2_PUBKEY_SK3


def test_bls12_381_scalar_field_montgomery_constants_zkcrypto():
    """the 2^32-th root of unity the NTT domains are built from and the Montgomery radix, as another library publishes them limb
    for limb: 7^((r - 1) / 2^32) * 2^256 mod r and 2^256 mod r -- through the model and through the product's own Fr code
    (from_canonical = the conversion every caller-supplied scalar has undergone on the Rust side)"""
    import groth16_amd as g

    cp = pm.BLS12_381
    root = pow(7, (cp.r - 1) >> 32, cp.r)
    assert pow(root, 1 << 32, cp.r) == 1 and pow(root, 1 << 31, cp.r) != 1
    limbs = lambda v: [(v >> (64 * i)) & (2**64 - 1) for i in range(4)]  # noqa: E731
    assert limbs(root * (1 << 256) % cp.r) == ZKCRYPTO_ROOT_OF_UNITY
    assert limbs((1 << 256) % cp.r) == ZKCRYPTO_R
    lb = g.lib()
    out = np.zeros(4, dtype=np.uint64)
    for canonical, want in ((root, ZKCRYPTO_ROOT_OF_UNITY), (1, ZKCRYPTO_R)):
        a = np.array(limbs(canonical), dtype=np.uint64)
        assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 5, ptr64(a), None, ptr64(out)) == 0      # from_canonical
        assert [int(x) for x in out] == want
        back = np.zeros(4, dtype=np.uint64)
        assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 4, ptr64(out.copy()), None, ptr64(back)) == 0   # to_canonical
        assert [int(x) for x in back] == limbs(canonical)


def test_bn254_generator_tripling_eip196():
    import groth16_amd as g

    cp = pm.BN254
    G1, _ = pm.groups(cp)
    assert G1.mul(cp.g1, 3) == EIP196_3G and G1.add(EIP196_2G, cp.g1) == EIP196_3G
    gen, want = g1_to_arr([cp.g1], cp)[0], g1_to_arr([EIP196_3G], cp)[0]
    out = np.zeros_like(gen)
    three = np.array([3, 0, 0, 0], dtype=np.uint64)
    assert g.lib().c.g16_host_group_op(CURVE_ID[cp.name], 0, 1, ptr64(gen), ptr64(three), ptr64(out)) == 0 and (out == want).all()
    assert (oracle().group_op(cp.name, False, 1, gen, three) == want).all()


@pytest.mark.gpu
def test_published_multiples_through_gpu_msm():
    """the eth2 public keys (BLS12-381) and EIP-196's 3 G (BN254) as GPU MSM results, per-window and merged-window paths"""
    import os

    import groth16_amd as g
    import groth16_amd.serialize as ser
    from helpers import ints_to_mont

    for precomp in ("0", "1"):
        os.environ["G16_MSM_API_PRECOMP"] = precomp
        try:
            cp = pm.BLS12_381
            gen = g1_to_arr([cp.g1, cp.g1, cp.g1], cp)
            with g.Groth16(cp.name, 0) as prover:
                for scalars, want_hex in (([1, 1, 0], ETH2_PUBKEY_SK2), ([2, 0, 0], ETH2_PUBKEY_SK2), ([1, 1, 1], ETH2_PUBKEY_SK3), ([cp.r - 1, 4, 0], ETH2_PUBKEY_SK3)):
                    out = prover.msm(gen, ints_to_mont(scalars, cp.r, 4))
                    assert ser.serialize_points(cp.name, out[None, :], False, True).hex() == want_hex
            cp = pm.BN254
            gen = g1_to_arr([cp.g1, cp.g1, cp.g1], cp)
            with g.Groth16(cp.name, 0) as prover:
                for scalars in ([1, 1, 1], [3, 0, 0], [cp.r - 2, 2, 3]):
                    assert arr_to_g1(prover.msm(gen, ints_to_mont(scalars, cp.r, 4))[None, :], cp)[0] == EIP196_3G
        finally:
            os.environ.pop("G16_MSM_API_PRECOMP", None)


def test_bn254_scalar_field_two_adic_root_published():
    """the primitive 2^28-th root of unity of the BN254 scalar field that libff / snarkjs / gnark publish for alt_bn128, which is
    5^((r - 1) / 2^28) -- arkworks' TWO_ADIC_ROOT_OF_UNITY for GENERATOR = 5; through the model and the product's Fr code"""
    import groth16_amd as g

    cp = pm.BN254
    published = 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(5, (cp.r - 1) >> 28, cp.r) == published
    assert pow(published, 1 << 28, cp.r) == 1 and pow(published, 1 << 27, cp.r) == cp.r - 1
    # the product's Fr: from_canonical then 28 squarings (g16_host_field_op mul) give one; 27 give -1
    lb = g.lib()
    limbs = lambda v: np.array([(v >> (64 * i)) & (2**64 - 1) for i in range(4)], dtype=np.uint64)  # noqa: E731
    x = np.zeros(4, dtype=np.uint64)
    assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 5, ptr64(limbs(published)), None, ptr64(x)) == 0
    one, minus_one = np.zeros(4, dtype=np.uint64), np.zeros(4, dtype=np.uint64)
    assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 5, ptr64(limbs(1)), None, ptr64(one)) == 0
    assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 5, ptr64(limbs(cp.r - 1)), None, ptr64(minus_one)) == 0
    for i in range(28):
        if i == 27:
            assert (x == minus_one).all()
        y = np.zeros(4, dtype=np.uint64)
        assert lb.c.g16_host_field_op(CURVE_ID[cp.name], 0, 2, ptr64(x.copy()), ptr64(x.copy()), ptr64(y)) == 0
        x = y
    assert (x == one).all()


# Montgomery "one" (R mod p) of the four fields as other libraries publish it limb for limb (zkcrypto/bls12_381 fp.rs `R`, `R2`;
# bellman / ff-derived BN256 Fq and Fr `R`): pins the little-endian u64 Montgomery limb form in which coordinates and scalars cross
# this C ABI (= ark_ff::Fp's in-memory form) for BOTH curves, independently of the repository's own models.
PUBLISHED_R = {
    ("bls12_381", 1): [0x760900000002FFFD, 0xEBF4000BC40C0002, 0x5F48985753C758BA, 0x77CE585370525745, 0x5C071A97A256EC6D, 0x15F65EC3FA80E493],
    ("bls12_381", 0): ZKCRYPTO_R,
    ("bn254", 1): [0xD35D438DC58F0D9D, 0x0A78EB28F5C70B3D, 0x666EA36F7879462C, 0x0E0A77C19A07DF2F],
    ("bn254", 0): [0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F],
}
ZKCRYPTO_FQ_R2 = [0xF4DF1F341C341746, 0x0A76E6A609D104F1, 0x8DE5476C4C95B6D5, 0x67EB88A9939D83C0, 0x9A793E85B519952D, 0x11988FE592CAE3AA]


@pytest.mark.parametrize("curve,which", sorted(PUBLISHED_R))
def test_montgomery_one_matches_published_limbs(curve, which):
    import groth16_amd as g

    cp = {"bls12_381": pm.BLS12_381, "bn254": pm.BN254}[curve]
    p = cp.q if which else cp.r
    want = PUBLISHED_R[(curve, which)]
    nl = len(want)
    assert [(((1 << (64 * nl)) % p) >> (64 * i)) & (2**64 - 1) for i in range(nl)] == want
    one = np.zeros(nl, dtype=np.uint64)
    one[0] = 1
    out = np.zeros(nl, dtype=np.uint64)
    lb = g.lib()
    assert lb.c.g16_host_field_op(CURVE_ID[curve], which, 5, ptr64(one), None, ptr64(out)) == 0      # from_canonical(1)
    assert [int(x) for x in out] == want
    if (curve, which) == ("bls12_381", 1):   # and R^2: from_canonical(R) (what a "to Montgomery" conversion multiplies by)
        r_can = np.array(want, dtype=np.uint64)
        assert lb.c.g16_host_field_op(CURVE_ID[curve], which, 5, ptr64(r_can), None, ptr64(out)) == 0
        assert [int(x) for x in out] == ZKCRYPTO_FQ_R2
