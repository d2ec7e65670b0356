"""CanonicalSerialize / CanonicalDeserialize for the reference's data types.

Containers as ark-serialize derives them -- a struct is its fields in declaration order, ``Vec<T>`` a little-endian ``u64``
length followed by the elements -- over the point encodings of ``g16_serialize_points`` (csrc/serialize.hip):
  Proof          a, b, c                                                        src/data_structures.rs:8-16
  VerifyingKey   alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1: Vec       src/data_structures.rs:31-44
  ProvingKey     vk, beta_g1, delta_g1, a_query, b_g1_query, b_g2_query, h_query, l_query (all Vec)   :125-143
The byte formats are restated from the published ark-serialize / zcash definitions and cannot be checked against the
reference in this environment (no Rust toolchain, no fixtures in the reference): treat them as unverified.
"""
from __future__ import annotations

import ctypes as C
import struct
from typing import Tuple

import numpy as np

from .binding import CURVE_ID, FQ_LIMBS, lib, ptr64
from .groth16 import Proof, ProvingKey


def point_size(curve: str, g2: bool, compressed: bool) -> int:
    return int(lib().c.g16_serialized_point_size(CURVE_ID[curve], int(g2), int(compressed)))


def serialize_points(curve: str, points: np.ndarray, g2: bool, compressed: bool = True) -> bytes:
    words = (4 if g2 else 2) * FQ_LIMBS[curve]
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, words)
    n = pts.shape[0]
    out = C.create_string_buffer(max(1, n * point_size(curve, g2, compressed)))
    lb = lib()
    lb.check(lb.c.g16_serialize_points(CURVE_ID[curve], int(g2), int(compressed), ptr64(pts.reshape(-1)) if n else None, n, out))
    return out.raw[: n * point_size(curve, g2, compressed)]


def deserialize_points(curve: str, data: bytes, n: int, g2: bool, compressed: bool = True, validate: int = 2) -> np.ndarray:
    """validate: 0 = Validate::No, 1 = on-curve, 2 = on-curve and prime-order subgroup (Validate::Yes, the default of
    CanonicalDeserialize::deserialize_compressed)"""
    words = (4 if g2 else 2) * FQ_LIMBS[curve]
    sz = point_size(curve, g2, compressed)
    if len(data) < n * sz:
        raise ValueError("not enough bytes")
    out = np.zeros((n, words), dtype=np.uint64)
    lb = lib()
    lb.check(lb.c.g16_deserialize_points(CURVE_ID[curve], int(g2), int(compressed), bytes(data[: n * sz]), n, validate,
                                         ptr64(out.reshape(-1)) if n else None))
    return out


def _vec(curve, pts, g2, compressed) -> bytes:
    words = (4 if g2 else 2) * FQ_LIMBS[curve]
    n = np.asarray(pts).reshape(-1, words).shape[0]
    return struct.pack("<Q", n) + serialize_points(curve, pts, g2, compressed)


class _Reader:
    def __init__(self, curve, data, compressed, validate):
        self.curve, self.data, self.pos, self.compressed, self.validate = curve, data, 0, compressed, validate

    def point(self, g2) -> np.ndarray:
        return self.points(1, g2)

    def points(self, n, g2) -> np.ndarray:
        sz = point_size(self.curve, g2, self.compressed)
        out = deserialize_points(self.curve, self.data[self.pos: self.pos + n * sz], n, g2, self.compressed, self.validate)
        self.pos += n * sz
        return out

    def vec(self, g2) -> np.ndarray:
        if self.pos + 8 > len(self.data):
            raise ValueError("not enough bytes")
        (n,) = struct.unpack_from("<Q", self.data, self.pos)
        self.pos += 8
        return self.points(n, g2)


def proof_to_bytes(curve: str, proof: Proof, compressed: bool = True) -> bytes:
    return (serialize_points(curve, proof.a, False, compressed) + serialize_points(curve, proof.b, True, compressed) +
            serialize_points(curve, proof.c, False, compressed))


def proof_from_bytes(curve: str, data: bytes, compressed: bool = True, validate: int = 2) -> Proof:
    r = _Reader(curve, data, compressed, validate)
    return Proof(r.point(False)[0], r.point(True)[0], r.point(False)[0])


def verifying_key_to_bytes(curve: str, pk: ProvingKey, compressed: bool = True) -> bytes:
    if pk.gamma_g2 is None or pk.gamma_abc_g1 is None:
        raise ValueError("this ProvingKey carries no gamma_g2 / gamma_abc_g1")
    return (serialize_points(curve, pk.alpha_g1, False, compressed) + serialize_points(curve, pk.beta_g2, True, compressed) +
            serialize_points(curve, pk.gamma_g2, True, compressed) + serialize_points(curve, pk.delta_g2, True, compressed) +
            _vec(curve, pk.gamma_abc_g1, False, compressed))


def proving_key_to_bytes(curve: str, pk: ProvingKey, compressed: bool = True) -> bytes:
    return (verifying_key_to_bytes(curve, pk, compressed) + serialize_points(curve, pk.beta_g1, False, compressed) +
            serialize_points(curve, pk.delta_g1, False, compressed) + _vec(curve, pk.a_query, False, compressed) +
            _vec(curve, pk.b_g1_query, False, compressed) + _vec(curve, pk.b_g2_query, True, compressed) +
            _vec(curve, pk.h_query, False, compressed) + _vec(curve, pk.l_query, False, compressed))


def proving_key_from_bytes(curve: str, data: bytes, compressed: bool = True, validate: int = 2) -> Tuple[ProvingKey, int]:
    """returns (key, bytes consumed)"""
    r = _Reader(curve, data, compressed, validate)
    alpha_g1, beta_g2, gamma_g2, delta_g2 = r.point(False), r.point(True), r.point(True), r.point(True)
    gamma_abc = r.vec(False)
    beta_g1, delta_g1 = r.point(False), r.point(False)
    a_q, b1_q, b2_q, h_q, l_q = r.vec(False), r.vec(False), r.vec(True), r.vec(False), r.vec(False)
    return ProvingKey(curve, alpha_g1, beta_g1, delta_g1, beta_g2, delta_g2, a_q, b1_q, b2_q, h_q, l_q, gamma_g2, gamma_abc), r.pos
