"""Test-side helpers: ctypes binding of the CPU oracle (oracle/libg16_oracle.so),
limb <-> integer conversion, and builders that turn oracle / pymodel objects into the
flat numpy arrays the C ABIs take.  TEST INFRASTRUCTURE: not imported by the product.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)

import pymodel  # noqa: E402

CURVE_ID = {"bls12_381": 0, "bn254": 1}
FQ_LIMBS = {"bls12_381": 6, "bn254": 4}

u64p = C.POINTER(C.c_uint64)
u32p = C.POINTER(C.c_uint32)


def ptr64(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


def ptr32(a: np.ndarray):
    assert a.dtype == np.uint32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u32p)


class CsrViewC(C.Structure):
    _fields_ = [("row_ptr", u64p), ("col", u32p), ("val", u64p)]


class PkViewC(C.Structure):
    _fields_ = [
        ("alpha_g1", u64p), ("beta_g1", u64p), ("delta_g1", u64p),
        ("beta_g2", u64p), ("delta_g2", u64p),
        ("a_query", u64p), ("a_len", C.c_uint64),
        ("b_g1_query", u64p), ("b_g1_len", C.c_uint64),
        ("b_g2_query", u64p), ("b_g2_len", C.c_uint64),
        ("h_query", u64p), ("h_len", C.c_uint64),
        ("l_query", u64p), ("l_len", C.c_uint64),
    ]


# ---------------------------------------------------------------------------------
# integer <-> Montgomery limb arrays
# ---------------------------------------------------------------------------------


def ints_to_mont(vals: Sequence[int], p: int, nl: int) -> np.ndarray:
    out = np.zeros((len(vals), nl), dtype=np.uint64)
    R = 1 << (64 * nl)
    for i, v in enumerate(vals):
        m = (v % p) * R % p
        for k in range(nl):
            out[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return out


def mont_to_ints(arr: np.ndarray, p: int) -> List[int]:
    arr = np.asarray(arr, dtype=np.uint64)
    nl = arr.shape[-1]
    flat = arr.reshape(-1, nl)
    rinv = pow(1 << (64 * nl), p - 2, p)
    out = []
    for row in flat:
        v = 0
        for k in range(nl):
            v |= int(row[k]) << (64 * k)
        out.append(v * rinv % p)
    return out


def g1_to_arr(points, cp) -> np.ndarray:
    """list of affine (x,y)/None -> [n, 2*L] uint64 Montgomery; identity = all zero."""
    L = cp.fq_limbs64
    out = np.zeros((len(points), 2 * L), dtype=np.uint64)
    for i, P in enumerate(points):
        if P is not None:
            out[i] = ints_to_mont([P[0], P[1]], cp.q, L).reshape(-1)
    return out


def g2_to_arr(points, cp) -> np.ndarray:
    L = cp.fq_limbs64
    out = np.zeros((len(points), 4 * L), dtype=np.uint64)
    for i, P in enumerate(points):
        if P is not None:
            out[i] = ints_to_mont([P[0][0], P[0][1], P[1][0], P[1][1]], cp.q, L).reshape(-1)
    return out


def arr_to_g1(arr: np.ndarray, cp):
    L = cp.fq_limbs64
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 2, L)
    out = []
    for row in a:
        if not row.any():
            out.append(None)
        else:
            x, y = mont_to_ints(row, cp.q)
            out.append((x, y))
    return out


def arr_to_g2(arr: np.ndarray, cp):
    L = cp.fq_limbs64
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4, L)
    out = []
    for row in a:
        if not row.any():
            out.append(None)
        else:
            v = mont_to_ints(row, cp.q)
            out.append(((v[0], v[1]), (v[2], v[3])))
    return out


# ---------------------------------------------------------------------------------
# flat circuit / key containers
# ---------------------------------------------------------------------------------


@dataclass
class Csr:
    row_ptr: np.ndarray  # uint64 [nc+1]
    col: np.ndarray  # uint32 [nnz]
    val: np.ndarray  # uint64 [nnz,4] Montgomery

    def view(self) -> CsrViewC:
        return CsrViewC(ptr64(self.row_ptr), ptr32(self.col), ptr64(self.val))


@dataclass
class FlatCircuit:
    curve: str
    num_inputs: int
    num_constraints: int
    num_vars: int
    abc: List[Csr]
    z: np.ndarray  # uint64 [num_vars,4] Montgomery full assignment

    @property
    def domain_size(self) -> int:
        n = 1
        while n < self.num_constraints + self.num_inputs:
            n <<= 1
        return n


@dataclass
class FlatPk:
    curve: str
    alpha_g1: np.ndarray
    beta_g1: np.ndarray
    delta_g1: np.ndarray
    beta_g2: np.ndarray
    delta_g2: np.ndarray
    a_query: np.ndarray
    b_g1_query: np.ndarray
    b_g2_query: np.ndarray
    h_query: np.ndarray
    l_query: np.ndarray

    def view(self) -> PkViewC:
        return PkViewC(
            ptr64(self.alpha_g1), ptr64(self.beta_g1), ptr64(self.delta_g1), ptr64(self.beta_g2),
            ptr64(self.delta_g2),
            ptr64(self.a_query), len(self.a_query),
            ptr64(self.b_g1_query), len(self.b_g1_query),
            ptr64(self.b_g2_query), len(self.b_g2_query),
            ptr64(self.h_query), len(self.h_query),
            ptr64(self.l_query), len(self.l_query),
        )


def rows_to_csr(rows, cp) -> Csr:
    row_ptr = np.zeros(len(rows) + 1, dtype=np.uint64)
    cols, vals = [], []
    for i, row in enumerate(rows):
        for cf, idx in row:
            cols.append(idx)
            vals.append(cf)
        row_ptr[i + 1] = len(cols)
    col = np.array(cols, dtype=np.uint32)
    val = ints_to_mont(vals, cp.r, 4) if vals else np.zeros((0, 4), dtype=np.uint64)
    return Csr(row_ptr, col, val)


def circuit_from_pymodel(cp, cs: "pymodel.R1CS", z: Sequence[int]) -> FlatCircuit:
    return FlatCircuit(
        cp.name, cs.num_inputs, cs.num_constraints, len(z),
        [rows_to_csr(cs.a, cp), rows_to_csr(cs.b, cp), rows_to_csr(cs.c, cp)],
        ints_to_mont(z, cp.r, 4),
    )


def pk_from_pymodel(cp, pk: "pymodel.ProvingKey") -> FlatPk:
    return FlatPk(
        cp.name,
        g1_to_arr([pk.alpha_g1], cp), g1_to_arr([pk.beta_g1], cp), g1_to_arr([pk.delta_g1], cp),
        g2_to_arr([pk.beta_g2], cp), g2_to_arr([pk.delta_g2], cp),
        g1_to_arr(pk.a_query, cp), g1_to_arr(pk.b_g1_query, cp), g2_to_arr(pk.b_g2_query, cp),
        g1_to_arr(pk.h_query, cp), g1_to_arr(pk.l_query, cp),
    )


# ---------------------------------------------------------------------------------
# the oracle library
# ---------------------------------------------------------------------------------


class Oracle:
    def __init__(self):
        # G16_ORACLE_LIB: another build of the oracle (the sanitizer tier, tools/sanitize.sh, loads an ASan / UBSan build)
        path = os.environ.get("G16_ORACLE_LIB", os.path.join(ROOT, "oracle", "libg16_oracle.so"))
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.lib = C.CDLL(path)
        self.lib.orc_domain_size.restype = C.c_uint64
        self.lib.orc_domain_size.argtypes = [C.c_uint64]

    @property
    def threads(self) -> int:
        return self.lib.orc_num_threads()

    def set_threads(self, n: int):
        self.lib.orc_set_num_threads(n)

    def field_op(self, curve: str, which: int, op: int, a: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
        out = np.zeros_like(a)
        rc = self.lib.orc_field_op(CURVE_ID[curve], which, op, ptr64(a), ptr64(b), ptr64(out))
        assert rc == 0
        return out

    def constant(self, curve: str, which: int) -> np.ndarray:
        L = FQ_LIMBS[curve]
        out = np.zeros({0: 4, 1: 4, 2: 2 * L, 3: 4 * L}[which], dtype=np.uint64)
        assert self.lib.orc_constant(CURVE_ID[curve], which, ptr64(out)) == 0
        return out

    def group_op(self, curve: str, g2: bool, op: int, p: np.ndarray, q_or_k: np.ndarray) -> np.ndarray:
        out = np.zeros_like(p)
        self.lib.orc_group_op(CURVE_ID[curve], int(g2), op, ptr64(p), ptr64(q_or_k), ptr64(out))
        return out

    def ntt(self, curve: str, data: np.ndarray, inverse: bool, coset: bool) -> np.ndarray:
        d = np.ascontiguousarray(data.copy())
        n = d.shape[0]
        log_n = n.bit_length() - 1
        assert 1 << log_n == n
        rc = self.lib.orc_ntt(CURVE_ID[curve], ptr64(d), log_n, int(inverse), int(coset))
        assert rc == 0
        return d

    def witness_map(self, ck: FlatCircuit, want_abc: bool = False):
        n = ck.domain_size
        h = np.zeros((n, 4), dtype=np.uint64)
        abc = np.zeros((3, n, 4), dtype=np.uint64) if want_abc else None
        views = (CsrViewC * 3)(*[m.view() for m in ck.abc])
        rc = self.lib.orc_witness_map(CURVE_ID[ck.curve], views, C.c_uint64(ck.num_inputs),
                                      C.c_uint64(ck.num_constraints), ptr64(ck.z), ptr64(h), ptr64(abc))
        if rc:
            raise ValueError("PolynomialDegreeTooLarge")
        return (h, abc) if want_abc else h

    def msm(self, curve: str, g2: bool, bases: np.ndarray, scalars: np.ndarray) -> np.ndarray:
        L = FQ_LIMBS[curve]
        n = min(len(bases), len(scalars))
        out = np.zeros((4 if g2 else 2) * L, dtype=np.uint64)
        fn = self.lib.orc_msm_g2 if g2 else self.lib.orc_msm_g1
        fn(CURVE_ID[curve], ptr64(np.ascontiguousarray(bases)), ptr64(np.ascontiguousarray(scalars)), C.c_uint64(n),
           ptr64(out))
        return out

    def prove(self, pk: FlatPk, ck: FlatCircuit, r: np.ndarray, s: np.ndarray, want_parts: bool = False):
        L = FQ_LIMBS[ck.curve]
        proof = np.zeros(8 * L, dtype=np.uint64)
        n = ck.domain_size
        h = np.zeros((n, 4), dtype=np.uint64) if want_parts else None
        parts = np.zeros(12 * L, dtype=np.uint64) if want_parts else None
        times = (C.c_double * 6)()
        views = (CsrViewC * 3)(*[m.view() for m in ck.abc])
        pkv = pk.view()
        rc = self.lib.orc_prove(CURVE_ID[ck.curve], C.byref(pkv), views, C.c_uint64(ck.num_inputs),
                                C.c_uint64(ck.num_constraints), ptr64(ck.z), C.c_uint64(ck.num_vars), ptr64(r), ptr64(s),
                                ptr64(proof), ptr64(h), ptr64(parts), times)
        if rc:
            raise ValueError("PolynomialDegreeTooLarge")
        tnames = ["witness_map", "compute_c", "compute_a", "compute_b_g1", "compute_b_g2", "finish_c"]
        tm = dict(zip(tnames, list(times)))
        if want_parts:
            return proof, h, parts, tm
        return proof, tm

    def setup(self, ck: FlatCircuit, seed: int):
        """generate_parameters_with_qap with a known trapdoor.  Returns (FlatPk, extras)."""
        L = FQ_LIMBS[ck.curve]
        n = ck.domain_size
        m = ck.num_vars - 1
        w = ck.num_vars - ck.num_inputs
        g1o = np.zeros((4, 2 * L), dtype=np.uint64)
        g2o = np.zeros((4, 4 * L), dtype=np.uint64)
        a_q = np.zeros((m + 1, 2 * L), dtype=np.uint64)
        b1_q = np.zeros((m + 1, 2 * L), dtype=np.uint64)
        b2_q = np.zeros((m + 1, 4 * L), dtype=np.uint64)
        h_q = np.zeros((n - 1, 2 * L), dtype=np.uint64)
        l_q = np.zeros((w, 2 * L), dtype=np.uint64)
        gabc = np.zeros((ck.num_inputs, 2 * L), dtype=np.uint64)
        td = np.zeros((6, 4), dtype=np.uint64)
        abc_t = np.zeros((3, m + 1, 4), dtype=np.uint64)
        views = (CsrViewC * 3)(*[mm.view() for mm in ck.abc])
        rc = self.lib.orc_setup(CURVE_ID[ck.curve], views, C.c_uint64(ck.num_inputs), C.c_uint64(ck.num_constraints),
                                C.c_uint64(ck.num_vars), C.c_uint64(seed), ptr64(g1o), ptr64(g2o), ptr64(a_q), ptr64(b1_q),
                                ptr64(b2_q), ptr64(h_q), ptr64(l_q), ptr64(gabc), ptr64(td), ptr64(abc_t))
        assert rc == 0
        pk = FlatPk(ck.curve, g1o[0:1].copy(), g1o[1:2].copy(), g1o[2:3].copy(), g2o[0:1].copy(), g2o[1:2].copy(),
                    a_q, b1_q, b2_q, h_q, l_q)
        extras = dict(g1gen=g1o[3].copy(), g2gen=g2o[3].copy(), gamma_g2=g2o[2].copy(), gamma_abc=gabc, trapdoor=td,
                      abc_t=abc_t)
        return pk, extras

    def trapdoor_proof(self, ck: FlatCircuit, extras, h: np.ndarray, r: np.ndarray, s: np.ndarray) -> np.ndarray:
        L = FQ_LIMBS[ck.curve]
        out = np.zeros(8 * L, dtype=np.uint64)
        self.lib.orc_trapdoor_proof(CURVE_ID[ck.curve], ptr64(extras["trapdoor"]), ptr64(extras["abc_t"]),
                                    C.c_uint64(ck.num_vars - 1), C.c_uint64(ck.num_inputs), C.c_uint64(ck.domain_size),
                                    ptr64(extras["g1gen"]), ptr64(extras["g2gen"]), ptr64(ck.z),
                                    ptr64(np.ascontiguousarray(h)), ptr64(r), ptr64(s), ptr64(out))
        return out

    def syn_circuit(self, curve: str, k: int, seed: int) -> FlatCircuit:
        nc = (1 << k) - 2
        z = np.zeros((nc + 3, 4), dtype=np.uint64)
        row_ptr = np.zeros(nc + 1, dtype=np.uint64)
        cols = [np.zeros(nc, dtype=np.uint32) for _ in range(3)]
        val = np.zeros((nc, 4), dtype=np.uint64)
        self.lib.orc_syn_circuit(CURVE_ID[curve], k, C.c_uint64(seed), ptr64(z), ptr64(row_ptr), ptr32(cols[0]),
                                 ptr32(cols[1]), ptr32(cols[2]), ptr64(val))
        return FlatCircuit(curve, 2, nc, nc + 3, [Csr(row_ptr, cols[i], val) for i in range(3)], z)

    def synth_bases(self, curve: str, g2: bool, seed: int, n: int) -> np.ndarray:
        L = FQ_LIMBS[curve]
        out = np.zeros((n, (4 if g2 else 2) * L), dtype=np.uint64)
        self.lib.orc_synth_bases(CURVE_ID[curve], int(g2), C.c_uint64(seed), C.c_uint64(n), ptr64(out))
        return out

    def synth_pk(self, ck: FlatCircuit, seed: int) -> FlatPk:
        """synthetic-bases proving key: distinct non-identity points,
        NOT a valid CRS; used for bit-exact GPU-vs-oracle parity at sizes where setup is slow."""
        n = ck.domain_size
        m = ck.num_vars - 1
        w = ck.num_vars - ck.num_inputs
        g1 = self.synth_bases(ck.curve, False, seed, 3 + 2 * (m + 1) + (n - 1) + w)
        g2 = self.synth_bases(ck.curve, True, seed + 1, 2 + (m + 1))
        o = 3
        a_q = g1[o:o + m + 1]; o += m + 1
        b1_q = g1[o:o + m + 1]; o += m + 1
        h_q = g1[o:o + n - 1]; o += n - 1
        l_q = g1[o:o + w]
        cc = np.ascontiguousarray
        return FlatPk(ck.curve, cc(g1[0:1]), cc(g1[1:2]), cc(g1[2:3]), cc(g2[0:1]), cc(g2[1:2]), cc(a_q), cc(b1_q),
                      cc(g2[2:]), cc(h_q), cc(l_q))

    def on_curve(self, curve: str, g2: bool, p: np.ndarray) -> bool:
        return self.lib.orc_on_curve(CURVE_ID[curve], int(g2), ptr64(np.ascontiguousarray(p))) == 0

    def rand_fr(self, curve: str, seed: int, n: int) -> np.ndarray:
        out = np.zeros((n, 4), dtype=np.uint64)
        self.lib.orc_rand_fr(CURVE_ID[curve], C.c_uint64(seed), C.c_uint64(n), ptr64(out))
        return out


_ORACLE = None


def oracle() -> Oracle:
    global _ORACLE
    if _ORACLE is None:
        _ORACLE = Oracle()
    return _ORACLE
