"""Host-side mirror of ark-groth16's prover interface over the C ABI (see package docstring).

Data is held exactly as the Rust side holds it: numpy uint64 arrays of little-endian Montgomery
limbs (``Fr`` = 4 limbs; ``Fq`` = 6 / 4 limbs for BLS12-381 / BN254), affine points as
``x|y`` (G1) or ``x.c0 x.c1 y.c0 y.c1`` (G2), the identity as all-zero limbs.
"""
from __future__ import annotations

import ctypes as C
import secrets
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .binding import (CURVE_ID, FQ_LIMBS, CsrViewC, ParamsViewC, PartialC, PkInfoC, PkViewC, ProofC, QueryC, TimingsC, ToxicWasteC, lib,
                      ptr32, ptr64, u64p)

_MODULUS_R = {
    "bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
}

_MODULUS_Q = {
    "bls12_381": 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    "bn254": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}
# the curves' standard generators (x, y; G2 coordinates as (c0, c1)): IETF pairing-friendly-curves draft / EIP-197
_GENERATORS = {
    "bls12_381": (
        (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
         0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
        ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
          0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
         (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
          0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))),
    "bn254": (
        (1, 2),
        ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
          11559732032986387107991004021392285783925812861821192530917403151452391805634),
         (8495653923123431417604973247489272438418190587263600148770280649306958101930,
          4082367875863433681332203403145435568316851327593401208105741076214120093531))),
}


def _c(a: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _limbs(v: int, n: int) -> List[int]:
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def _fq_mont(curve: str, coords: Sequence[int]) -> np.ndarray:
    """integers mod q -> concatenated Montgomery limbs (the in-memory form of an affine point's coordinates)"""
    L, q = FQ_LIMBS[curve], _MODULUS_Q[curve]
    return np.array([w for x in coords for w in _limbs((x << (64 * L)) % q, L)], dtype=np.uint64)


def _rand_fr(curve: str, rng=None, nonzero: bool = False) -> np.ndarray:
    """F::rand(rng) as Montgomery limbs; rng: anything with getrandbits (random.Random), default the OS source"""
    p = _MODULUS_R[curve]
    while True:
        v = (rng.getrandbits(512) if rng is not None else secrets.randbits(512)) % p
        if v or not nonzero:
            return np.array(_limbs((v << 256) % p, 4), dtype=np.uint64)


@dataclass
class ConstraintMatrices:
    """ark_relations::r1cs::ConstraintMatrices flattened to CSR (row_ptr u64, col u32, val Fr)."""

    num_instance_variables: int
    num_witness_variables: int
    num_constraints: int
    a: Tuple[np.ndarray, np.ndarray, np.ndarray]
    b: Tuple[np.ndarray, np.ndarray, np.ndarray]
    c: Tuple[np.ndarray, np.ndarray, np.ndarray]

    @staticmethod
    def from_rows(curve: str, num_instance: int, num_witness: int, rows_a, rows_b, rows_c) -> "ConstraintMatrices":
        """rows: Vec<Vec<(F, usize)>> with F given as 4-limb Montgomery arrays."""
        def conv(rows):
            rp = np.zeros(len(rows) + 1, dtype=np.uint64)
            cols, vals = [], []
            for i, row in enumerate(rows):
                for coeff, idx in row:
                    cols.append(idx)
                    vals.append(np.asarray(coeff, dtype=np.uint64))
                rp[i + 1] = len(cols)
            val = np.stack(vals) if vals else np.zeros((0, 4), dtype=np.uint64)
            return rp, np.asarray(cols, dtype=np.uint32), _c(val)
        return ConstraintMatrices(num_instance, num_witness, len(rows_a), conv(rows_a), conv(rows_b), conv(rows_c))


@dataclass
class ProvingKey:
    """src/data_structures.rs:125-143 (vk fields the prover reads are flattened in)."""

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
    # the rest of the VerifyingKey (data_structures.rs:28-41); the prover never reads these
    gamma_g2: Optional[np.ndarray] = None
    gamma_abc_g1: Optional[np.ndarray] = None


@dataclass
class Proof:
    """src/data_structures.rs:8-16; affine Montgomery limbs."""

    a: np.ndarray
    b: np.ndarray
    c: np.ndarray

    def __eq__(self, o):
        return (self.a == o.a).all() and (self.b == o.b).all() and (self.c == o.c).all()

    def flat(self) -> np.ndarray:
        return np.concatenate([self.a, self.b, self.c])


def shard_ranges(m: int, w: int, h_len: int, num_inputs: int, idx: int, cnt: int):
    """Contiguous shard `idx` of `cnt` of the five MSM base arrays.  a / b_g1 / b_g2 (index space query[1..], length m)
    are cut identically; l (length w, index j <-> a-index j + num_inputs - 1) is cut at the matching places so that
    the library can reuse the witness bucket sort for it; h (length h_len) is cut evenly."""
    a_lo, a_hi = m * idx // cnt, m * (idx + 1) // cnt
    l_lo = min(w, max(0, a_lo - (num_inputs - 1)))
    l_hi = min(w, max(0, a_hi - (num_inputs - 1)))
    h_lo, h_hi = h_len * idx // cnt, h_len * (idx + 1) // cnt
    return dict(a=(a_lo, a_hi), l=(l_lo, l_hi), h=(h_lo, h_hi))


class _Ctx:
    def __init__(self, curve: str, device):
        """device: one HIP device id, or a sequence of them (g16_ctx_create_multi: the key is sharded over the devices inside
        the library and one g16_prove call uses all of them)"""
        self.lib = lib()
        self.curve = curve
        self.handle = C.c_void_p()
        if isinstance(device, (list, tuple)):
            ids = (C.c_int * len(device))(*[int(d) for d in device])
            self.lib.check(self.lib.c.g16_ctx_create_multi(CURVE_ID[curve], ids, len(device), C.byref(self.handle)))
        else:
            self.lib.check(self.lib.c.g16_ctx_create(CURVE_ID[curve], device, C.byref(self.handle)))
        self.num_devices = int(self.lib.c.g16_ctx_num_devices(self.handle))

    def close(self):
        if self.handle:
            self.lib.c.g16_ctx_destroy(self.handle)
            self.handle = C.c_void_p()


class _DevicePk:
    """g16_pk handle.  shard = (index, count) splits every MSM base array into contiguous ranges (base-range shards);
    shard = (index, count, "bucket") is the bucket-space shard: the whole key on every rank, rank `index` owning the buckets
    b mod count == index (g16_pk_load_bucket_shard)."""

    def __init__(self, ctx: _Ctx, pk: ProvingKey, num_inputs: int, shard=(0, 1), dist_h: bool = False):
        """dist_h: the h_query shard is this rank's block of the distributed witness map (dist_h_indices) instead of a
        contiguous range -- the form g16_prove_partial_h expects; bucket-space shard: the WHOLE h_query in the order the
        all-gathered blocks arrive in (bucket_h_indices)"""
        self.ctx = ctx
        self.handle = C.c_void_p()
        self._keep = []
        idx, cnt = shard[0], shard[1]
        bucket = len(shard) > 2 and shard[2] == "bucket"
        rg = shard_ranges(len(pk.a_query) - 1, len(pk.l_query), len(pk.h_query), num_inputs, 0 if bucket else idx, 1 if bucket else cnt)
        (a_lo, a_hi), (l_lo, l_hi), (h_lo, h_hi) = rg["a"], rg["l"], rg["h"]
        h_query = pk.h_query
        if dist_h:
            n_dom = len(pk.h_query) + 1                                  # domain size n = len(h_query) + 1 (generator.rs:168)
            sel = bucket_h_indices(n_dom, cnt) if bucket else dist_h_indices(n_dom, idx, cnt)
            h_query = pk.h_query[sel[sel < len(pk.h_query)]]            # only the very last index of the last rank is n - 1
            h_lo, h_hi = 0, len(h_query)

        def q(arr: np.ndarray, skip: int, lo: int, hi: int) -> QueryC:
            sl = _c(arr[skip + lo: skip + hi])
            self._keep.append(sl)
            return QueryC(sl.ctypes.data if len(sl) else None, hi - lo, 0 if (dist_h and arr is h_query) else lo)

        keep = [_c(x).reshape(-1) for x in (pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query[0],
                                            pk.b_g1_query[0], pk.b_g2_query[0])]
        self._keep += keep
        view = PkViewC(*[ptr64(k) for k in keep], q(pk.a_query, 1, a_lo, a_hi), q(pk.b_g1_query, 1, a_lo, a_hi),
                       q(pk.b_g2_query, 1, a_lo, a_hi), q(h_query, 0, h_lo, h_hi), q(pk.l_query, 0, l_lo, l_hi), 0)
        if bucket:
            ctx.lib.check(ctx.lib.c.g16_pk_load_bucket_shard(ctx.handle, C.byref(view), idx, cnt, C.byref(self.handle)))
        else:
            ctx.lib.check(ctx.lib.c.g16_pk_load(ctx.handle, C.byref(view), C.byref(self.handle)))
        self._keep = []  # the library copied everything

    def rebind(self, rank: int, world: int):
        """g16_pk_rebind_bucket_shard: the same resident tables as rank `rank` of `world` (tests, --sim-shards)"""
        self.ctx.lib.check(self.ctx.lib.c.g16_pk_rebind_bucket_shard(self.handle, rank, world))

    def info(self) -> dict:
        """g16_pk_get_info: window sizes, why the key is held as plain bases if it is, bucket-shard rank / world, HBM bytes"""
        out = PkInfoC()
        self.ctx.lib.check(self.ctx.lib.c.g16_pk_get_info(self.handle, C.byref(out)))
        return out.as_dict()

    def close(self):
        if self.handle:
            self.ctx.lib.c.g16_pk_free(self.handle)
            self.handle = C.c_void_p()

def dist_h_indices(n: int, rank: int, world: int) -> np.ndarray:
    """h coefficients rank `rank` of `world` holds after the distributed witness map (g16_dwm_*): the BLOCK indices
    (rank * blk + j) + M * k1, j < blk = M / world, k1 < world, M = n / world, in the order [k1][j] -- the order its shard of
    h_query has to be gathered in (include/g16_mi355x.h)."""
    M = n // world
    blk = M // world
    return ((rank * blk + np.arange(blk, dtype=np.int64))[None, :] + (M * np.arange(world, dtype=np.int64))[:, None]).reshape(-1)


def bucket_h_indices(n: int, world: int) -> np.ndarray:
    """h coefficients in the order an all-gather of the ranks' blocks of the distributed witness map leaves them in: rank 0's
    block (dist_h_indices), rank 1's, ... -- the order a bucket-space shard's h_query is loaded in, since there every rank
    runs the h MSM over ALL coefficients (and 1 / world of the buckets)"""
    return np.concatenate([dist_h_indices(n, q, world) for q in range(world)])


def _csr_views(m: "ConstraintMatrices"):
    """CsrViewC[3] over the three matrices plus the arrays the views point into.  A ctypes structure keeps only the raw
    address: when a matrix arrives with another dtype or layout (e.g. scipy's int64 indptr) the normalised copy made here
    is what the library reads, so the caller must hold the returned list until the C call has returned."""
    keep = [(np.ascontiguousarray(x[0], dtype=np.uint64), np.ascontiguousarray(x[1], dtype=np.uint32), _c(x[2])) for x in (m.a, m.b, m.c)]
    views = (CsrViewC * 3)(*[CsrViewC(ptr64(rp), ptr32(col), ptr64(val)) for rp, col, val in keep])
    return views, keep


class _DeviceCircuit:
    def __init__(self, ctx: _Ctx, m: ConstraintMatrices):
        self.ctx = ctx
        self.handle = C.c_void_p()
        self.num_variables = m.num_instance_variables + m.num_witness_variables
        views, self._keep = _csr_views(m)
        ctx.lib.check(ctx.lib.c.g16_circuit_load(ctx.handle, views, m.num_instance_variables, m.num_constraints, self.num_variables,
                                                 C.byref(self.handle)))

    @property
    def domain_size(self) -> int:
        return int(self.ctx.lib.c.g16_circuit_domain_size(self.handle))

    def close(self):
        if self.handle:
            self.ctx.lib.c.g16_circuit_free(self.handle)
            self.handle = C.c_void_p()


class LibsnarkReduction:
    """R1CSToQAP default reduction (src/r1cs_to_qap.rs:123-248), witness side only."""

    @staticmethod
    def witness_map_from_matrices(prover: "Groth16", matrices: ConstraintMatrices, num_inputs: int, num_constraints: int,
                                  full_assignment: np.ndarray) -> np.ndarray:
        return prover.witness_map_from_matrices(matrices, num_inputs, num_constraints, full_assignment)


class Groth16:
    """``Groth16::<E, LibsnarkReduction>`` prover methods for E in {Bls12_381, Bn254} on one MI355X (``device=0``) or on
    several GPUs of the node behind the same calls (``device=[0, 1, ...]``: the library shards the key and folds the partial sums).

    Device-resident copies of proving keys and constraint matrices are cached per object
    (they are per-circuit constants, like ``&pk`` in the reference)."""

    def __init__(self, curve: str = "bls12_381", device=0):
        if curve not in CURVE_ID:
            raise ValueError(f"unsupported curve {curve}")
        self.curve = curve
        self._ctx = _Ctx(curve, device)
        self._pks: Dict[Tuple[int, Tuple[int, int]], _DevicePk] = {}
        self._cks: Dict[int, _DeviceCircuit] = {}
        self._circuits_by_content: Dict[bytes, ConstraintMatrices] = {}   # create_proof_with_reduction: one upload per circuit
        self._owner: Optional["Groth16"] = None

    def share_device_data_of(self, owner: "Groth16"):
        """use the device-resident keys and circuits of another prover object on the SAME GPU instead of loading copies (the C ABI
        allows it: a g16_pk / g16_circuit is read-only during a proof and may serve every single-device context of its GPU).  Two
        objects like this, driven from two threads, are the throughput mode (`PipelinedProver`).  The owner must outlive this
        object's proofs; keys and circuits are loaded -- and evicted -- through the owner only."""
        if owner.curve != self.curve:
            raise ValueError("the two provers are for different curves")
        self._owner = owner

    # -- handles -------------------------------------------------------------------------
    def _pk(self, pk: ProvingKey, num_inputs: int, shard=(0, 1), dist_h: bool = False) -> _DevicePk:
        if self._owner is not None:
            return self._owner._pk(pk, num_inputs, shard, dist_h)
        key = (id(pk), shard) if not dist_h else (id(pk), shard, "dist_h")
        if key not in self._pks:
            if pk.curve != self.curve:
                raise ValueError("proving key is for another curve")
            # the cache entry holds a reference to pk so that id(pk) cannot be recycled while it lives
            self._pks[key] = (pk, _DevicePk(self._ctx, pk, num_inputs, shard, dist_h))
        return self._pks[key][1]

    def _ck(self, m: ConstraintMatrices) -> _DeviceCircuit:
        if self._owner is not None:
            return self._owner._ck(m)
        if id(m) not in self._cks:
            self._cks[id(m)] = (m, _DeviceCircuit(self._ctx, m))
        return self._cks[id(m)][1]

    def evict_pk(self, pk: ProvingKey, shard=(0, 1)):
        """drop the device-resident copies of one key shard -- the contiguous cut and the block-order cut of the distributed
        witness map alike (their window tables are 13x the shard)"""
        for key in ((id(pk), shard), (id(pk), shard, "dist_h")):
            ent = self._pks.pop(key, None)
            if ent is not None:
                ent[1].close()

    def evict(self):
        """drop every cached device-resident key / circuit (frees their HBM) and the by-content circuit index"""
        for _, p in self._pks.values():
            p.close()
        for _, c in self._cks.values():
            c.close()
        self._pks, self._cks = {}, {}
        self._circuits_by_content = {}

    def close(self):
        self.evict()
        self._ctx.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- generator.rs:47-208 ----------------------------------------------------------------
    def generate_parameters_with_qap(self, matrices: ConstraintMatrices, alpha: np.ndarray, beta: np.ndarray, gamma: np.ndarray,
                                     delta: np.ndarray, g1_generator: np.ndarray, g2_generator: np.ndarray, t: np.ndarray) -> ProvingKey:
        """Groth16::generate_parameters_with_qap on the matrices of an already synthesised circuit.  Arguments as the reference's
        (toxic waste as Fr limbs, generators as affine points); `t` is what the reference draws from its rng
        (domain.sample_element_outside_domain, generator.rs:90)."""
        L = FQ_LIMBS[self.curve]
        ni, nv = matrices.num_instance_variables, matrices.num_instance_variables + matrices.num_witness_variables
        need, n = matrices.num_constraints + ni, 1
        while n < need:
            n <<= 1
        g1 = lambda k: np.zeros((k, 2 * L), dtype=np.uint64)  # noqa: E731
        g2 = lambda k: np.zeros((k, 4 * L), dtype=np.uint64)  # noqa: E731
        pk = ProvingKey(self.curve, g1(1), g1(1), g1(1), g2(1), g2(1), g1(nv), g1(nv), g2(nv), g1(n - 1), g1(nv - ni), g2(1), g1(ni))
        tw = ToxicWasteC()
        for name, v in (("alpha", alpha), ("beta", beta), ("gamma", gamma), ("delta", delta), ("t", t)):
            getattr(tw, name)[:] = [int(x) for x in _c(v).reshape(4)]
        vp = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
        out = ParamsViewC(ptr64(pk.alpha_g1), ptr64(pk.beta_g1), ptr64(pk.delta_g1), ptr64(pk.beta_g2), ptr64(pk.delta_g2), ptr64(pk.gamma_g2),
                          ptr64(pk.gamma_abc_g1), vp(pk.a_query), vp(pk.b_g1_query), vp(pk.b_g2_query), vp(pk.h_query), vp(pk.l_query), 0)
        views, keep = _csr_views(matrices)   # `keep` owns the (possibly converted) arrays until the call returns
        lb = self._ctx.lib
        lb.check(lb.c.g16_generate_parameters(self._ctx.handle, views, ni, matrices.num_constraints, nv, C.byref(tw),
                                              ptr64(_c(g1_generator).reshape(-1)), ptr64(_c(g2_generator).reshape(-1)), C.byref(out)))
        return pk

    # -- generator.rs:20-45 (+ lib.rs:63-74 circuit_specific_setup) -------------------------------
    def generate_random_parameters_with_reduction(self, circuit, rng=None) -> ProvingKey:
        """Groth16::generate_random_parameters_with_reduction: synthesise `circuit` in setup mode on the host, draw the toxic
        waste, random generators (a random multiple of the curve's standard generator, as E::G1::rand does up to distribution)
        and t outside the domain from `rng`, and build the key on the GPU (generate_parameters_with_qap)."""
        from .r1cs import synthesize

        cs = synthesize(self.curve, circuit, setup_mode=True)
        matrices = cs.to_matrices()
        lb = self._ctx.lib
        g1s, g2s = _GENERATORS[self.curve]
        L = FQ_LIMBS[self.curve]

        def rand_generator(g2: bool) -> np.ndarray:
            k = _rand_fr(self.curve, rng, nonzero=True)   # read as a plain integer below r: any non-zero multiple will do
            base = _fq_mont(self.curve, [g2s[0][0], g2s[0][1], g2s[1][0], g2s[1][1]] if g2 else list(g1s))
            out = np.zeros((4 if g2 else 2) * L, dtype=np.uint64)
            lb.check(lb.c.g16_host_group_op(CURVE_ID[self.curve], int(g2), 1, ptr64(base), ptr64(k), ptr64(out)))
            return out

        need, n = matrices.num_constraints + matrices.num_instance_variables, 1
        while n < need:
            n <<= 1
        p = _MODULUS_R[self.curve]
        rinv = pow(1 << 256, -1, p)
        while True:   # domain.sample_element_outside_domain(rng), generator.rs:90
            t = _rand_fr(self.curve, rng, nonzero=True)
            tv = int.from_bytes(t.tobytes(), "little") * rinv % p
            if pow(tv, n, p) != 1:
                break
        alpha, beta, gamma, delta = (_rand_fr(self.curve, rng, nonzero=True) for _ in range(4))
        return self.generate_parameters_with_qap(matrices, alpha, beta, gamma, delta, rand_generator(False), rand_generator(True), t)

    def setup(self, circuit, rng=None) -> Tuple[ProvingKey, ProvingKey]:
        """SNARK::circuit_specific_setup (lib.rs:63-74): (pk, vk); the vk is the key's own alpha_g1 / beta_g2 / gamma_g2 / delta_g2
        / gamma_abc_g1 fields (data_structures.rs:28-41), returned as the same object"""
        pk = self.generate_random_parameters_with_reduction(circuit, rng)
        return pk, pk

    # -- prover.rs:173-217 ------------------------------------------------------------------------
    _MAX_CIRCUITS_BY_CONTENT = 64

    def create_proof_with_reduction(self, circuit, pk: ProvingKey, r: np.ndarray, s: np.ndarray, circuit_id=None) -> Proof:
        """Groth16::create_proof_with_reduction: host-side synthesis exactly as prover.rs:185-204 (fresh constraint system,
        generate_constraints, matrices, full_assignment = instance ++ witness), then the pure-data call on the GPU.  The
        device copy of the matrices is cached, so proving the same circuit again uploads only the assignment: by
        `circuit_id` (any hashable the caller vouches for: same id = same constraint matrices; no hashing at all) or, without
        one, by a SHA-1 of the matrices' content (bounded: the oldest entry and its device copy are dropped past 64 circuits --
        which also invalidates any DistributedWitnessMap still holding that circuit's handle: create those after the circuits)."""
        import hashlib

        from .r1cs import synthesize

        cs = synthesize(self.curve, circuit, setup_mode=False)
        m = cs.to_matrices()
        if circuit_id is not None:
            key = ("id", circuit_id)
        else:
            h = hashlib.sha1()
            for mat in (m.a, m.b, m.c):
                for arr in mat:
                    h.update(np.ascontiguousarray(arr).tobytes())
            h.update(repr((m.num_instance_variables, m.num_witness_variables, m.num_constraints)).encode())
            key = h.digest()
        known = self._circuits_by_content.get(key)
        if known is None:
            if len(self._circuits_by_content) >= self._MAX_CIRCUITS_BY_CONTENT:
                old_key = next(iter(self._circuits_by_content))
                old = self._circuits_by_content.pop(old_key)
                ent = self._cks.pop(id(old), None)
                if ent is not None:
                    ent[1].close()
            self._circuits_by_content[key] = m
        else:
            if circuit_id is not None:
                # the caller vouches that one id means one circuit; the cheap invariants are checked anyway -- a reused id would
                # otherwise prove against the WRONG device matrices and hand back an invalid proof without any error
                def shape(mm):
                    return (mm.num_instance_variables, mm.num_witness_variables, mm.num_constraints,
                            tuple(int(np.asarray(mat[0])[-1]) for mat in (mm.a, mm.b, mm.c)))   # row_ptr[-1] = nnz of A, B, C

                if shape(known) != shape(m):
                    raise ValueError(f"circuit_id {circuit_id!r} was first used for a circuit of shape {shape(known)} (instance, witness, "
                                     f"constraints, nnz) and is now passed with one of shape {shape(m)}: one id must mean one circuit")
            m = known
        return self.create_proof_with_reduction_and_matrices(pk, r, s, m, cs.num_instance_variables, cs.num_constraints, cs.full_assignment())

    def prove(self, pk: ProvingKey, circuit, rng=None) -> Proof:
        """SNARK::prove (lib.rs:76-82) = create_random_proof_with_reduction(circuit, pk, rng) (prover.rs:138-150)"""
        return self.create_proof_with_reduction(circuit, pk, _rand_fr(self.curve, rng), _rand_fr(self.curve, rng))

    def create_proof_no_zk(self, circuit, pk: ProvingKey) -> Proof:
        """prover.rs:155-168 on a circuit: r = s = 0 (same as create_proof_with_reduction_no_zk(circuit, pk))"""
        return self.create_proof_with_reduction_no_zk(circuit, pk)

    # -- prover.rs:223-250 ----------------------------------------------------------------------
    def rerandomize_proof(self, vk: ProvingKey, proof: Proof, rng=None) -> Proof:
        return rerandomize_proof(self.curve, vk, proof, rng)

    # -- prover.rs:26-51 -------------------------------------------------------------------
    def create_proof_with_reduction_and_matrices(self, pk: ProvingKey, r: np.ndarray, s: np.ndarray, matrices: ConstraintMatrices,
                                                 num_inputs: int, num_constraints: int, full_assignment: np.ndarray) -> Proof:
        assert num_inputs == matrices.num_instance_variables and num_constraints == matrices.num_constraints
        L = FQ_LIMBS[self.curve]
        dpk, dck = self._pk(pk, num_inputs), self._ck(matrices)
        z = _c(full_assignment)
        out = ProofC()
        lb = self._ctx.lib
        lb.check(lb.c.g16_prove(self._ctx.handle, dpk.handle, dck.handle, z.ctypes.data, z.shape[0], 0, ptr64(_c(r)), ptr64(_c(s)),
                                C.byref(out)))
        return Proof(np.array(out.a[: 2 * L], dtype=np.uint64), np.array(out.b[: 4 * L], dtype=np.uint64),
                     np.array(out.c[: 2 * L], dtype=np.uint64))

    # -- prover.rs:155-168 -----------------------------------------------------------------
    def create_proof_with_reduction_no_zk(self, *args) -> Proof:
        """the reference's signature ``(circuit, pk)`` (prover.rs:155-168), or the pure-data form
        ``(pk, matrices, num_inputs, num_constraints, full_assignment)``: r = s = 0"""
        zero = np.zeros(4, dtype=np.uint64)
        if len(args) == 2 and not isinstance(args[0], ProvingKey):
            return self.create_proof_with_reduction(args[0], args[1], zero, zero)
        pk, matrices, num_inputs, num_constraints, full_assignment = args
        return self.create_proof_with_reduction_and_matrices(pk, zero, zero, matrices, num_inputs, num_constraints, full_assignment)

    # -- prover.rs:138-150 -----------------------------------------------------------------
    def create_random_proof_with_reduction(self, *args, rng=None) -> Proof:
        """the reference's signature ``(circuit, pk, rng)`` (prover.rs:138-150), or the pure-data form
        ``(pk, matrices, num_inputs, num_constraints, full_assignment[, rng])``: fresh r, s from rng"""
        if not isinstance(args[0], ProvingKey):
            circuit, pk = args[0], args[1]
            rng = args[2] if len(args) > 2 else rng
            return self.create_proof_with_reduction(circuit, pk, _rand_fr(self.curve, rng), _rand_fr(self.curve, rng))
        pk, matrices, num_inputs, num_constraints, full_assignment = args[:5]
        rng = args[5] if len(args) > 5 else rng
        return self.create_proof_with_reduction_and_matrices(pk, _rand_fr(self.curve, rng), _rand_fr(self.curve, rng), matrices, num_inputs,
                                                             num_constraints, full_assignment)

    # -- r1cs_to_qap.rs:172-235 --------------------------------------------------------------
    def witness_map_from_matrices(self, matrices: ConstraintMatrices, num_inputs: int, num_constraints: int,
                                  full_assignment: np.ndarray) -> np.ndarray:
        dck = self._ck(matrices)
        z = _c(full_assignment)
        h = np.zeros((dck.domain_size, 4), dtype=np.uint64)
        lb = self._ctx.lib
        lb.check(lb.c.g16_witness_map(self._ctx.handle, dck.handle, z.ctypes.data, z.shape[0], 0, ptr64(h)))
        return h

    # -- VariableBaseMSM::msm (scalars in Montgomery form; into_bigint happens on the GPU) ----
    def msm(self, bases: np.ndarray, scalars: np.ndarray, g2: bool = False) -> np.ndarray:
        L = FQ_LIMBS[self.curve]
        n = min(len(bases), len(scalars))  # msm_bigint truncates to the shorter input
        b, s = _c(bases[:n]), _c(scalars[:n])
        out = np.zeros((4 if g2 else 2) * L, dtype=np.uint64)
        lb = self._ctx.lib
        fn = lb.c.g16_msm_g2 if g2 else lb.c.g16_msm_g1
        lb.check(fn(self._ctx.handle, ptr64(b) if n else None, ptr64(s) if n else None, n, ptr64(out)))
        return out

    def msm_bucket_shard(self, bases: np.ndarray, scalars: np.ndarray, rank: int, world: int, g2: bool = False) -> np.ndarray:
        """g16_msm_bucket_shard: rank's bucket-space share of msm(bases, scalars) -- the `world` results add up to the MSM"""
        L = FQ_LIMBS[self.curve]
        n = min(len(bases), len(scalars))
        b, s = _c(bases[:n]), _c(scalars[:n])
        out = np.zeros((4 if g2 else 2) * L, dtype=np.uint64)
        lb = self._ctx.lib
        lb.check(lb.c.g16_msm_bucket_shard(self._ctx.handle, int(g2), ptr64(b) if n else None, ptr64(s) if n else None, n, rank, world, ptr64(out)))
        return out

    # -- EvaluationDomain::{fft, ifft, coset variants}, natural order ---------------------------
    def ntt(self, data: np.ndarray, inverse: bool = False, coset: bool = False) -> np.ndarray:
        d = _c(data).copy()
        n = d.shape[0]
        log_n = n.bit_length() - 1
        if 1 << log_n != n:
            raise ValueError("length must be a power of two")
        lb = self._ctx.lib
        lb.check(lb.c.g16_ntt(self._ctx.handle, ptr64(d), log_n, int(inverse), int(coset)))
        return d

    def pk_info(self, pk: ProvingKey, num_inputs: int, shard=(0, 1), dist_h: bool = False) -> dict:
        """how the device-resident copy of (this shard of) the key is held: g16_pk_get_info"""
        return self._pk(pk, num_inputs, shard, dist_h).info()

    def timings(self) -> dict:
        t = TimingsC()
        self._ctx.lib.check(self._ctx.lib.c.g16_get_timings(self._ctx.handle, C.byref(t)))
        return t.as_dict()

    # -- sharded form --------------------------------------------------------------------------
    def prove_partial(self, pk: ProvingKey, matrices: ConstraintMatrices, full_assignment: np.ndarray, shard: Tuple[int, int],
                      skip_b_g1: bool = False) -> bytes:
        dpk, dck = self._pk(pk, matrices.num_instance_variables, shard), self._ck(matrices)
        z = _c(full_assignment)
        part = PartialC()
        lb = self._ctx.lib
        lb.check(lb.c.g16_prove_partial(self._ctx.handle, dpk.handle, dck.handle, z.ctypes.data, z.shape[0], 0, int(skip_b_g1),
                                        C.byref(part)))
        return bytes(part)

    def prove_partial_prepare(self, pk: ProvingKey, matrices: ConstraintMatrices, z_dev_ptr: int, n_assign: int, shard: Tuple[int, int],
                              dist_h: bool = True):
        """g16_prove_partial_prepare: enqueue the witness digit/sort pass of the next prove_partial[_h] over this key shard and this
        DEVICE assignment now -- before the distributed witness map's stages -- so that the two run side by side"""
        dpk, dck = self._pk(pk, matrices.num_instance_variables, shard, dist_h=dist_h), self._ck(matrices)
        lb = self._ctx.lib
        lb.check(lb.c.g16_prove_partial_prepare(self._ctx.handle, dpk.handle, dck.handle, C.c_void_p(z_dev_ptr), n_assign))

    def prove_finalize_prepare(self, pk: ProvingKey, num_inputs: int, r: np.ndarray, s: np.ndarray, shard: Tuple[int, int],
                               dist_h: bool = False):
        """g16_prove_finalize_prepare: start the (r, s)-only half of the glue of prover.rs:76-131 on a host thread now, so that it
        runs while the GPU computes this rank's partial sums; the next `finalize` over the same (key shard, r, s) picks it up"""
        dpk = self._pk(pk, num_inputs, shard, dist_h=dist_h)
        lb = self._ctx.lib
        lb.check(lb.c.g16_prove_finalize_prepare(self._ctx.handle, dpk.handle, ptr64(np.ascontiguousarray(r, dtype=np.uint64)),
                                                 ptr64(np.ascontiguousarray(s, dtype=np.uint64))))

    def prove_partial_h(self, pk: ProvingKey, matrices: ConstraintMatrices, full_assignment, shard: Tuple[int, int],
                        h_dev_ptr: int, h_len: int, skip_b_g1: bool = False, z_dev_ptr: int = 0) -> bytes:
        """g16_prove_partial_h: the shard's five partial sums with h supplied by the caller (device memory: this rank's block of
        the distributed witness map); the key shard is the one gathered in that block order (dist_h).  z_dev_ptr != 0: the
        assignment is taken from device memory (`full_assignment` then only gives the length)"""
        dpk, dck = self._pk(pk, matrices.num_instance_variables, shard, dist_h=True), self._ck(matrices)
        part = PartialC()
        lb = self._ctx.lib
        if z_dev_ptr:
            n_assign = matrices.num_instance_variables + matrices.num_witness_variables
            lb.check(lb.c.g16_prove_partial_h(self._ctx.handle, dpk.handle, dck.handle, C.c_void_p(z_dev_ptr), n_assign, 1, C.c_void_p(h_dev_ptr),
                                              h_len, int(skip_b_g1), C.byref(part)))
            return bytes(part)
        z = _c(full_assignment)
        lb.check(lb.c.g16_prove_partial_h(self._ctx.handle, dpk.handle, dck.handle, z.ctypes.data, z.shape[0], 0, C.c_void_p(h_dev_ptr), h_len,
                                          int(skip_b_g1), C.byref(part)))
        return bytes(part)

    def prove_finalize(self, pk: ProvingKey, num_inputs: int, parts: Sequence[bytes], r: np.ndarray, s: np.ndarray,
                       shard: Tuple[int, int] = (0, 1), dist_h: bool = False) -> Proof:
        L = FQ_LIMBS[self.curve]
        dpk = self._pk(pk, num_inputs, shard, dist_h)   # any shard carries the eight fixed points the glue reads
        arr = (PartialC * len(parts))(*[PartialC.from_buffer_copy(p) for p in parts])
        out = ProofC()
        lb = self._ctx.lib
        lb.check(lb.c.g16_prove_finalize(self._ctx.handle, dpk.handle, arr, len(parts), ptr64(_c(r)), ptr64(_c(s)), C.byref(out)))
        return Proof(np.array(out.a[: 2 * L], dtype=np.uint64), np.array(out.b[: 4 * L], dtype=np.uint64),
                     np.array(out.c[: 2 * L], dtype=np.uint64))


def rerandomize_proof(curve: str, vk: "ProvingKey", proof: "Proof", rng=None) -> "Proof":
    """Groth16::rerandomize_proof (prover.rs:223-250): A' = (1/r1) A, B' = r1 B + r1 r2 delta_g2, C' = C + r2 A for fresh
    non-zero r1, r2 (figure 1 of BKSV20).  Three scalar multiplications: host work, through the library's host field / group
    code -- no GPU context needed."""
    lb, cid, L = lib(), CURVE_ID[curve], FQ_LIMBS[curve]
    r1, r2 = _rand_fr(curve, rng, nonzero=True), _rand_fr(curve, rng, nonzero=True)

    def fr(op, a, b=None):   # g16_host_field_op on Fr: 2 mul, 3 inverse, 4 to_canonical
        out = np.zeros(4, dtype=np.uint64)
        lb.check(lb.c.g16_host_field_op(cid, 0, op, ptr64(_c(a)), ptr64(_c(b)) if b is not None else None, ptr64(out)))
        return out

    def grp(g2, op, p, q):   # g16_host_group_op: 0 p + q, 1 k * p (k canonical)
        out = np.zeros((4 if g2 else 2) * L, dtype=np.uint64)
        lb.check(lb.c.g16_host_group_op(cid, int(g2), op, ptr64(_c(p).reshape(-1)), ptr64(_c(q).reshape(-1)), ptr64(out)))
        return out

    canon = lambda x: fr(4, x)  # noqa: E731
    new_a = grp(False, 1, proof.a, canon(fr(3, r1)))
    new_b = grp(True, 0, grp(True, 1, proof.b, canon(r1)), grp(True, 1, vk.delta_g2, canon(fr(2, r1, r2))))
    new_c = grp(False, 0, proof.c, grp(False, 1, proof.a, canon(r2)))
    return Proof(new_a, new_b, new_c)


def fixed_points_view(pk: "ProvingKey"):
    """g16_pk_view with only the eight fixed points filled (what g16_finalize_host reads); returns (view, keepalive)"""
    keep = [_c(x).reshape(-1) for x in (pk.alpha_g1, pk.beta_g1, pk.delta_g1, pk.beta_g2, pk.delta_g2, pk.a_query[0], pk.b_g1_query[0],
                                        pk.b_g2_query[0])]
    empty = QueryC(None, 0, 0)
    return PkViewC(*[ptr64(k) for k in keep], empty, empty, empty, empty, empty, 0), keep


def finalize_host(curve: str, pk: "ProvingKey", parts: Sequence[bytes], r: np.ndarray, s: np.ndarray) -> "Proof":
    """g16_finalize_host: N-way EC fold of shard records + the prover.rs:76-131 glue, no GPU needed"""
    L = FQ_LIMBS[curve]
    view, keep = fixed_points_view(pk)
    arr = (PartialC * len(parts))(*[PartialC.from_buffer_copy(p) for p in parts])
    out = ProofC()
    lb = lib()
    lb.check(lb.c.g16_finalize_host(CURVE_ID[curve], C.byref(view), arr, len(parts), ptr64(_c(r)), ptr64(_c(s)), C.byref(out)))
    return Proof(np.array(out.a[: 2 * L], dtype=np.uint64), np.array(out.b[: 4 * L], dtype=np.uint64), np.array(out.c[: 2 * L], dtype=np.uint64))


class DistributedWitnessMap:
    """g16_dwm_*: one rank's side of the distributed witness map (h = witness_map_from_matrices, r1cs_to_qap.rs:172-235, with
    every n-point transform cut into a local n/world-point transform, a twiddle, ONE all-to-all and a local world-point
    transform).  The object owns the rank's device buffers (torch tensors: work[3], recv[3], h_local, M = n / world Fr each);
    the caller owns the exchange -- `run` does it with torch.distributed.all_to_all_single (RCCL over xGMI for backend
    "nccl"), the single-process tests move the chunks themselves between the ranks' objects."""

    def __init__(self, lib_, ctx_handle, circuit_handle, rank: int, world: int, device):
        import torch

        self.lib, self.ctx, self.rank, self.world = lib_, ctx_handle, rank, world
        self.handle = C.c_void_p()
        lib_.check(lib_.c.g16_dwm_create(ctx_handle, circuit_handle, rank, world, C.byref(self.handle)))
        self.M = int(lib_.c.g16_dwm_local_size(self.handle))
        self.blk = self.M // world
        mk = lambda: torch.empty((self.M, 4), dtype=torch.int64, device=device)  # noqa: E731
        self.work, self.recv, self.h_local = [mk() for _ in range(3)], [mk() for _ in range(3)], mk()
        self.h_full = None   # all_gather_h: the n-element buffer of a bucket-space shard
        self._wp = (C.c_void_p * 3)(*[t.data_ptr() for t in self.work])
        self._rp = (C.c_void_p * 3)(*[t.data_ptr() for t in self.recv])

    def stage(self, s: int, z_ptr: int = 0, n_assign: int = 0, on_device: bool = False):
        """stage 0: full_assignment -> work[0..2];  1: recv[0..2] -> work[0..2];  2: recv[0..2] -> work[0];  3: recv[0] -> h_local.
        Returns with the device work finished."""
        self.lib.check(self.lib.c.g16_dwm_stage(self.ctx, self.handle, s, C.c_void_p(z_ptr) if z_ptr else None, n_assign, int(on_device),
                                                self._wp, self._rp, C.c_void_p(self.h_local.data_ptr())))

    @staticmethod
    def arrays_after(s: int) -> int:
        """arrays exchanged after stage s (a, b, c after stages 0 and 1; the quotient after stage 2)"""
        return 3 if s < 2 else 1

    def stage_async(self, s: int, z_dev_ptr: int = 0, n_assign: int = 0):
        """g16_dwm_stage_async: the stage is enqueued on the context's witness-map stream and the call returns"""
        self.lib.check(self.lib.c.g16_dwm_stage_async(self.ctx, self.handle, s, C.c_void_p(z_dev_ptr) if z_dev_ptr else None, n_assign,
                                                      self._wp, self._rp, C.c_void_p(self.h_local.data_ptr())))

    def run(self, z_ptr: int, n_assign: int, on_device: bool, dist):
        """all four stages with the exchanges over `dist` (torch.distributed); returns h_local.

        Device-resident assignment + RCCL (or a single rank): NOTHING here waits on the host -- stages (g16_dwm_stage_async) and
        exchanges (all_to_all_single: chunk p of work -> rank p, the chunk from rank q -> position q of recv) are enqueued on the
        library's witness-map stream, which torch adopts as an external stream; the three arrays of an exchange step go
        back to back.  A following prove_partial_h orders only its h sort / h MSM after this stream.  Otherwise (host assignment, or
        gloo in the one-GPU / CPU tests, which has no all-to-all): stage by stage with host synchronisation."""
        import torch

        import os

        # G16_DWM_FORCE_COLLECTIVE (test-only): issue the collective with one rank too, so that a one-GPU box exercises
        # all_to_all_single over RCCL on the adopted stream exactly as the N > 1 runs do
        multi = dist is not None and (self.world > 1 or bool(os.environ.get("G16_DWM_FORCE_COLLECTIVE")))
        if on_device and (not multi or dist.get_backend() == "nccl"):
            ext = torch.cuda.ExternalStream(int(self.lib.c.g16_ctx_wm_stream(self.ctx)), device=self.h_local.device)
            with torch.cuda.stream(ext):
                for s in range(4):
                    self.stage_async(s, z_ptr if s == 0 else 0, n_assign if s == 0 else 0)
                    if s < 3:
                        for m in range(self.arrays_after(s)):
                            if multi:
                                dist.all_to_all_single(self.recv[m], self.work[m])
                            else:
                                self.recv[m].copy_(self.work[m])
            return self.h_local
        for s in range(4):
            self.stage(s, z_ptr if s == 0 else 0, n_assign if s == 0 else 0, on_device)
            if s < 3:
                for m in range(self.arrays_after(s)):
                    src, dst = self.work[m], self.recv[m]
                    if multi:
                        if dist.get_backend() == "nccl":
                            dist.all_to_all_single(dst, src)      # chunk p of src -> rank p; chunk from rank q -> position q
                        else:   # gloo (one-GPU / CPU tests) has no all-to-all: gather everything on the host, keep my chunks
                            mine = src.cpu()
                            every = [torch.empty_like(mine) for _ in range(self.world)]
                            dist.all_gather(every, mine)
                            blk = self.blk
                            dst.copy_(torch.cat([e[self.rank * blk:(self.rank + 1) * blk] for e in every]))
                    else:
                        dst.copy_(src)
                torch.cuda.synchronize()   # the next stage runs on the library's own stream
        return self.h_local

    def all_gather_h(self, dist=None, simulate: bool = False):
        """h WHOLE on this rank, for a bucket-space shard (every rank runs the h MSM over all coefficients and 1 / world of the
        buckets): ONE all-gather of the ranks' blocks (n / world Fr each -- 16 MiB per rank at 2^22 / 8) into an n-element buffer, in
        rank order -- the order `bucket_h_indices` lists and the key's h_query is loaded in, so nothing is un-permuted.  With RCCL
        (or one rank, or `simulate`: this rank's block copied `world` times -- the same bytes written, for the one-GPU share
        measurement) the gather is enqueued on the witness-map stream behind the map's last stage, with no host synchronisation;
        a following prove_partial_h orders itself after that stream.  gloo (tests): through the host."""
        import torch

        if getattr(self, "h_full", None) is None:
            self.h_full = torch.empty((self.M * self.world, 4), dtype=torch.int64, device=self.h_local.device)
        multi = dist is not None and self.world > 1 and not simulate
        if multi and dist.get_backend() != "nccl":
            torch.cuda.synchronize()
            mine = self.h_local.cpu()
            every = [torch.empty_like(mine) for _ in range(self.world)]
            dist.all_gather(every, mine)
            self.h_full.copy_(torch.cat(every))
            torch.cuda.synchronize()
            return self.h_full
        ext = torch.cuda.ExternalStream(int(self.lib.c.g16_ctx_wm_stream(self.ctx)), device=self.h_local.device)
        with torch.cuda.stream(ext):
            if multi:
                dist.all_gather_into_tensor(self.h_full, self.h_local)
            else:
                for q in range(self.world):
                    self.h_full[q * self.M:(q + 1) * self.M].copy_(self.h_local, non_blocking=True)
        return self.h_full

    def close(self):
        if self.handle:
            self.lib.c.g16_dwm_free(self.handle)
            self.handle = C.c_void_p()
        self.h_full = None


class PipelinedProver:
    """Throughput mode on one GPU: two contexts over ONE device-resident key / circuit, two worker threads.  `submit` returns a
    `concurrent.futures.Future` of the proof; with two proofs in flight the witness map / sort of one and the reductions / host glue of
    the other run under each other's bucket passes (bench.py reports the effect as `pipelined`: +3.9 % proofs per second at 2^22).
    Latency per proof roughly doubles -- use a plain `Groth16` when that matters."""

    def __init__(self, curve: str = "bls12_381", device: int = 0):
        import queue
        import threading

        self._owner = Groth16(curve, device)
        self._second = Groth16(curve, device)
        self._second.share_device_data_of(self._owner)
        self._lock = threading.Lock()           # key / circuit loads go through the owner's caches: one at a time
        self._jobs: "queue.Queue" = queue.Queue()
        self._threads = [threading.Thread(target=self._work, args=(p,), daemon=True) for p in (self._owner, self._second)]
        for t in self._threads:
            t.start()

    def _work(self, prover: Groth16):
        while True:
            job = self._jobs.get()
            if job is None:
                return
            fut, args = job
            if not fut.set_running_or_notify_cancel():
                continue
            try:
                pk, r, s, matrices, num_inputs, num_constraints, z = args
                with self._lock:     # make sure the handles exist (first use loads them) before the unlocked, concurrent proof
                    prover._pk(pk, num_inputs)
                    prover._ck(matrices)
                fut.set_result(prover.create_proof_with_reduction_and_matrices(pk, r, s, matrices, num_inputs, num_constraints, z))
            except BaseException as e:  # noqa: BLE001 -- delivered through the future
                fut.set_exception(e)

    def submit(self, pk: ProvingKey, r: np.ndarray, s: np.ndarray, matrices: ConstraintMatrices, num_inputs: int, num_constraints: int,
               full_assignment: np.ndarray):
        """Groth16::create_proof_with_reduction_and_matrices (prover.rs:26-51), asynchronously"""
        from concurrent.futures import Future

        fut: Future = Future()
        self._jobs.put((fut, (pk, r, s, matrices, num_inputs, num_constraints, full_assignment)))
        return fut

    def close(self):
        for _ in self._threads:
            self._jobs.put(None)
        for t in self._threads:
            t.join()
        self._second.close()
        self._owner.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class ShardedProver:
    """One process per GPU: every rank holds a contiguous shard of each MSM base array, computes its
    partial sums, and the fixed-size records are exchanged with ONE all-gather (RCCL over xGMI when
    the process group backend is "nccl"; gloo in the CPU tests).  Group addition is not an
    element-wise sum, so the collective is an all-gather of ~1.2 KB per rank followed by a local
    N-way EC addition inside g16_prove_finalize -- not an all-reduce.  The witness map is
    replicated (it is ~5 % of a single-GPU proof)."""

    PARTIAL_BYTES = C.sizeof(PartialC)

    def __init__(self, prover: Groth16, pk: ProvingKey, matrices: ConstraintMatrices, rank: int, world_size: int, mode: str = "base"):
        """mode "base": contiguous ranges of the bases per rank (1 / world of the key's memory per GPU); "bucket": the whole key
        on every GPU, the BUCKETS divided (b mod world == rank) -- the bucket reductions then shrink with the rank count too
        (g16_pk_load_bucket_shard)"""
        if mode not in ("base", "bucket"):
            raise ValueError("mode is 'base' or 'bucket'")
        self.prover, self.pk, self.matrices = prover, pk, matrices
        self.rank, self.world, self.mode = rank, world_size, mode
        self.shard = (rank, world_size, "bucket") if mode == "bucket" and world_size > 1 else (rank, world_size)

    def local_partial(self, full_assignment: np.ndarray, r: np.ndarray) -> bytes:
        return self.prover.prove_partial(self.pk, self.matrices, full_assignment, self.shard, skip_b_g1=not np.asarray(r).any())

    @staticmethod
    def exchange(local: bytes, dist, device=None) -> List[bytes]:
        """all-gather of the partial records through torch.distributed"""
        import torch

        t = torch.frombuffer(bytearray(local), dtype=torch.uint8)
        if device is not None:
            t = t.to(device)
        outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
        dist.all_gather(outs, t)
        return [bytes(o.cpu().numpy().tobytes()) for o in outs]

    def prove(self, full_assignment: np.ndarray, r: np.ndarray, s: np.ndarray, dist=None, device=None) -> Proof:
        local = self.local_partial(full_assignment, r)
        parts = [local] if (dist is None or self.world == 1) else self.exchange(local, dist, device)
        return self.prover.prove_finalize(self.pk, self.matrices.num_instance_variables, parts, r, s, self.shard)
