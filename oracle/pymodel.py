"""Slow, obviously-correct big-int model of the Groth16 prover hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (groth16_amd/, bench.py's
timed region) may import this file; it exists to pin the C++ oracle
(oracle/g16_oracle.cpp) and to generate the golden fixtures under tests/golden/.

PARITY STATUS: "parity unpinned" at the byte level -- the reference
(/root/reference, ark-groth16 0.5.0) ships no golden vectors and cannot be compiled
here (no Rust toolchain, un-vendored crates).  This model is pinned instead by
  (1) structural KATs (curve membership, r*G = O, w^n = 1, zcash G1 generator bytes),
  (2) the known-trapdoor check (expected A,B,C computed as scalar*generator with no
      MSM / NTT code), and
  (3) the QAP divisibility identity a(t)b(t)-c(t) = h(t)Z(t).

Reference anchors (file:line relative to /root/reference):
  evaluate_constraint            src/r1cs_to_qap.rs:28-67
  witness_map_from_matrices      src/r1cs_to_qap.rs:172-235
  instance_map_with_evaluation   src/r1cs_to_qap.rs:128-170
  h_query_scalars                src/r1cs_to_qap.rs:237-247
  create_proof_with_assignment   src/prover.rs:54-132
  calculate_coeff                src/prover.rs:252-270
  generate_parameters_with_qap   src/generator.rs:47-208
  prepare_inputs / verify_proof  src/verifier.rs:25-76   (plain Tate pairing)
External algebra (ark-ff/ec/poly 0.5.0, not vendored) is restated from its published
definitions: Radix2EvaluationDomain (natural order in/out, ifft scales by 1/n, coset
fft pre-multiplies coefficient k by g^k), short-Weierstrass a=0 group law.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

# ----------------------------------------------------------------------------------
# curve parameter tables  (every constant re-checked in
# selfcheck() below)
# ----------------------------------------------------------------------------------


@dataclass(frozen=True)
class CurveParams:
    name: str
    q: int  # base field modulus
    r: int  # scalar field modulus
    fr_generator: int  # Fr::GENERATOR (multiplicative generator)
    two_adicity: int
    b1: int  # G1: y^2 = x^3 + b1
    b2: Tuple[int, int]  # G2: y^2 = x^3 + b2 over Fq2 = Fq[u]/(u^2+1)
    g1: Tuple[int, int]
    g2: Tuple[Tuple[int, int], Tuple[int, int]]
    fq_limbs64: int
    fr_limbs64: int = 4

    @property
    def two_adic_root(self) -> int:
        return pow(self.fr_generator, (self.r - 1) >> self.two_adicity, self.r)


BLS12_381 = CurveParams(
    name="bls12_381",
    q=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
    r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    fr_generator=7,
    two_adicity=32,
    b1=4,
    b2=(4, 4),
    g1=(
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    ),
    g2=(
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    ),
    fq_limbs64=6,
)

_BN_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN254 = CurveParams(
    name="bn254",
    q=_BN_Q,
    r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
    fr_generator=5,
    two_adicity=28,
    b1=3,
    # b2 = 3 / (9 + u)
    b2=(
        19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690,
    ),
    g1=(1, 2),
    g2=(
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    ),
    fq_limbs64=4,
)

CURVES = {"bls12_381": BLS12_381, "bn254": BN254}

# ----------------------------------------------------------------------------------
# fields
# ----------------------------------------------------------------------------------


class Fq1:
    """Prime field ops on python ints."""

    def __init__(self, p: int):
        self.p = p
        self.zero = 0
        self.one = 1

    def add(self, a, b):
        return (a + b) % self.p

    def sub(self, a, b):
        return (a - b) % self.p

    def neg(self, a):
        return (-a) % self.p

    def mul(self, a, b):
        return a * b % self.p

    def sqr(self, a):
        return a * a % self.p

    def inv(self, a):
        assert a % self.p != 0
        return pow(a, self.p - 2, self.p)

    def is_zero(self, a):
        return a % self.p == 0

    def from_int(self, k):
        return k % self.p


class Fq2:
    """Fq[u]/(u^2+1), elements are (c0, c1) tuples."""

    def __init__(self, p: int):
        self.p = p
        self.zero = (0, 0)
        self.one = (1, 0)

    def add(self, a, b):
        return ((a[0] + b[0]) % self.p, (a[1] + b[1]) % self.p)

    def sub(self, a, b):
        return ((a[0] - b[0]) % self.p, (a[1] - b[1]) % self.p)

    def neg(self, a):
        return ((-a[0]) % self.p, (-a[1]) % self.p)

    def mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def sqr(self, a):
        return self.mul(a, a)

    def inv(self, a):
        p = self.p
        n = (a[0] * a[0] + a[1] * a[1]) % p
        assert n != 0
        ni = pow(n, p - 2, p)
        return (a[0] * ni % p, (-a[1]) * ni % p)

    def is_zero(self, a):
        return a[0] % self.p == 0 and a[1] % self.p == 0

    def from_int(self, k):
        return (k % self.p, 0)


# ----------------------------------------------------------------------------------
# short Weierstrass, a = 0.  Affine points are (x, y) or None for the identity;
# internal arithmetic is Jacobian (X, Y, Z), Z == zero  <=>  identity.
# ----------------------------------------------------------------------------------


class Group:
    def __init__(self, F, b, r: int):
        self.F = F
        self.b = b
        self.r = r

    # -- affine helpers
    def on_curve(self, P) -> bool:
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sub(F.sqr(y), F.add(F.mul(F.sqr(x), x), self.b)) == F.zero

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    # -- jacobian
    def jac_identity(self):
        return (self.F.one, self.F.one, self.F.zero)

    def to_jac(self, P):
        return self.jac_identity() if P is None else (P[0], P[1], self.F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.sqr(zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def jdouble(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return self.jac_identity()
        A = F.sqr(X)
        B = F.sqr(Y)
        C = F.sqr(B)
        t = F.sub(F.sub(F.sqr(F.add(X, B)), A), C)
        D = F.add(t, t)
        E = F.add(F.add(A, A), A)
        Fv = F.sqr(E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C)
        C8 = F.add(C8, C8)
        C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def jadd(self, P, Q):
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        if F.is_zero(Z1):
            return Q
        if F.is_zero(Z2):
            return P
        Z1Z1 = F.sqr(Z1)
        Z2Z2 = F.sqr(Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.jdouble(P)
            return self.jac_identity()
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.sqr(H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.sqr(R), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def jmul(self, J, k: int):
        k %= self.r
        acc = self.jac_identity()
        for bit in bin(k)[2:] if k else "":
            acc = self.jdouble(acc)
            if bit == "1":
                acc = self.jadd(acc, J)
        return acc

    # -- affine API
    def add(self, P, Q):
        return self.to_affine(self.jadd(self.to_jac(P), self.to_jac(Q)))

    def mul(self, P, k: int):
        return self.to_affine(self.jmul(self.to_jac(P), k))

    def msm(self, bases: Sequence, scalars: Sequence[int]):
        """msm_bigint semantics: truncates to the shorter input (ark-ec 0.5.0;
        relied on at src/prover.rs:66 where |h| = n and |h_query| = n-1)."""
        n = min(len(bases), len(scalars))
        if n == 0:
            return None
        # simple bucket method, window 4, unsigned digits: independent of every
        # production algorithm in this repo
        c = 4
        nb = (max(1, max(int(s) for s in scalars[:n]).bit_length()) + c - 1) // c
        total = self.jac_identity()
        for w in reversed(range(nb)):
            for _ in range(c):
                total = self.jdouble(total)
            buckets = [self.jac_identity() for _ in range(1 << c)]
            for P, s in zip(bases[:n], scalars[:n]):
                d = (int(s) >> (w * c)) & ((1 << c) - 1)
                if d and P is not None:
                    buckets[d] = self.jadd(buckets[d], self.to_jac(P))
            run = self.jac_identity()
            acc = self.jac_identity()
            for d in range((1 << c) - 1, 0, -1):
                run = self.jadd(run, buckets[d])
                acc = self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)

    def msm_naive(self, bases, scalars):
        n = min(len(bases), len(scalars))
        acc = self.jac_identity()
        for P, s in zip(bases[:n], scalars[:n]):
            if P is not None:
                acc = self.jadd(acc, self.jmul(self.to_jac(P), int(s)))
        return self.to_affine(acc)


def groups(cp: CurveParams) -> Tuple[Group, Group]:
    return Group(Fq1(cp.q), cp.b1, cp.r), Group(Fq2(cp.q), cp.b2, cp.r)


# ----------------------------------------------------------------------------------
# deterministic PRNG shared with the C++ side (SplitMix64)
# ----------------------------------------------------------------------------------


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def field(self, p: int) -> int:
        """512 bits reduced mod p (bias < 2^-250)."""
        v = 0
        for _ in range(8):
            v = (v << 64) | self.next()
        return v % p


# ----------------------------------------------------------------------------------
# Radix-2 evaluation domain (ark-poly 0.5.0 semantics, used at
# src/r1cs_to_qap.rs:178-232)
# ----------------------------------------------------------------------------------


class Domain:
    def __init__(self, cp: CurveParams, num_coeffs: int):
        n = 1
        while n < num_coeffs:
            n <<= 1
        log_n = n.bit_length() - 1
        if log_n > cp.two_adicity:
            raise ValueError("PolynomialDegreeTooLarge")
        self.cp = cp
        self.n = n
        self.log_n = log_n
        self.p = cp.r
        w = cp.two_adic_root
        for _ in range(cp.two_adicity - log_n):
            w = w * w % cp.r
        self.omega = w
        self.omega_inv = pow(w, cp.r - 2, cp.r)
        self.n_inv = pow(n, cp.r - 2, cp.r)

    def _ntt(self, a: List[int], w: int) -> List[int]:
        n = len(a)
        if n == 1:
            return a[:]
        p = self.p
        e = self._ntt(a[0::2], w * w % p)
        o = self._ntt(a[1::2], w * w % p)
        out = [0] * n
        t = 1
        for k in range(n // 2):
            x = t * o[k] % p
            out[k] = (e[k] + x) % p
            out[k + n // 2] = (e[k] - x) % p
            t = t * w % p
        return out

    def fft(self, a):
        assert len(a) == self.n
        return self._ntt(list(a), self.omega)

    def ifft(self, a):
        assert len(a) == self.n
        return [x * self.n_inv % self.p for x in self._ntt(list(a), self.omega_inv)]

    def coset_fft(self, a, g):
        p = self.p
        out, t = [], 1
        for x in a:
            out.append(x * t % p)
            t = t * g % p
        return self.fft(out)

    def coset_ifft(self, a, g):
        p = self.p
        gi = pow(g, p - 2, p)
        out, t = [], 1
        for x in self.ifft(a):
            out.append(x * t % p)
            t = t * gi % p
        return out

    def vanishing(self, tau: int) -> int:
        return (pow(tau, self.n, self.p) - 1) % self.p

    def lagrange_at(self, tau: int) -> List[int]:
        """evaluate_all_lagrange_coefficients(tau)."""
        p = self.p
        z = self.vanishing(tau)
        if z == 0:
            out, t = [], 1
            for _ in range(self.n):
                out.append(1 if t == tau % p else 0)
                t = t * self.omega % p
            return out
        out, t = [], 1
        for _ in range(self.n):
            out.append(z * self.n_inv % p * t % p * pow((tau - t) % p, p - 2, p) % p)
            t = t * self.omega % p
        return out


# ----------------------------------------------------------------------------------
# R1CS
# ----------------------------------------------------------------------------------

Row = List[Tuple[int, int]]  # [(coeff, column)]


@dataclass
class R1CS:
    """ConstraintMatrices + counts (ark-relations); columns: instance then witness."""

    num_inputs: int  # = num_instance_variables, includes the constant 1
    num_witness: int
    a: List[Row]
    b: List[Row]
    c: List[Row]

    @property
    def num_constraints(self) -> int:
        return len(self.a)


def evaluate_constraint(row: Row, z: Sequence[int], p: int) -> int:
    # src/r1cs_to_qap.rs:28-67 (the coeff==1 fast path has no arithmetic effect)
    return sum(cf * z[i] for cf, i in row) % p


def syn_circuit(cp: CurveParams, k: int, seed: int, dense: bool = False) -> Tuple[R1CS, List[int]]:
    """SYN(k, curve, seed): Fibonacci product chain, n_c = 2^k - 2, l = 2, domain 2^k
    (SURVEY.md section 8(d)).  z = [1, x, u_0 .. u_{n_c}]."""
    p = cp.r
    rng = SplitMix64(seed)
    nc = (1 << k) - 2
    u = [rng.field(p), rng.field(p)]
    for i in range(nc):
        u.append(u[i] * u[i + 1] % p)
    x = u[nc + 1]
    z = [1, x] + u[: nc + 1]

    def col(j):  # column of u_j
        return 1 if j == nc + 1 else 2 + j

    A, B, C = [], [], []
    for i in range(nc):
        ra, rb, rc = [(1, col(i))], [(1, col(i + 1))], [(1, col(i + 2))]
        if dense:
            # two extra terms that cancel: +k*u_j - k*u_j split over two entries
            kf = rng.field(p)
            j = rng.next() % (nc + 1)
            ra += [(kf, col(j)), ((-kf) % p, col(j))]
        A.append(ra)
        B.append(rb)
        C.append(rc)
    return R1CS(2, nc + 1, A, B, C), z


def mimc_circuit(cp: CurveParams, rounds: int, seed: int) -> Tuple[R1CS, List[int]]:
    """MiMC (LongsightF, x^3 Feistel) preimage circuit: tests/mimc.rs:46-143.
    2 constraints per round; public input = final xl."""
    p = cp.r
    rng = SplitMix64(seed)
    consts = [rng.field(p) for _ in range(rounds)]
    xl, xr = rng.field(p), rng.field(p)
    wit = [xl, xr]  # witness columns
    inst = [1]
    nin = 2
    A, B, C = [], [], []

    def wcol(j):
        return nin + j

    xl_col, xr_col = wcol(0), wcol(1)
    xl_v, xr_v = xl, xr
    for i in range(rounds):
        tmp = (xl_v + consts[i]) ** 2 % p
        wit.append(tmp)
        tmp_col = wcol(len(wit) - 1)
        # (xl + c) * (xl + c) = tmp
        lc = [(1, xl_col), (consts[i], 0)]
        A.append(list(lc))
        B.append(list(lc))
        C.append([(1, tmp_col)])
        new_xl = (xr_v + tmp * (xl_v + consts[i])) % p
        if i == rounds - 1:
            inst.append(new_xl)
            new_col = 1
        else:
            wit.append(new_xl)
            new_col = wcol(len(wit) - 1)
        # tmp * (xl + c) = new_xl - xr
        A.append([(1, tmp_col)])
        B.append(list(lc))
        C.append([(1, new_col), ((-1) % p, xr_col)])
        xr_v, xr_col = xl_v, xl_col
        xl_v, xl_col = new_xl, new_col
    return R1CS(nin, len(wit), A, B, C), inst + wit


def is_satisfied(cs: R1CS, z: Sequence[int], p: int) -> bool:
    return all(
        evaluate_constraint(cs.a[i], z, p) * evaluate_constraint(cs.b[i], z, p) % p
        == evaluate_constraint(cs.c[i], z, p)
        for i in range(cs.num_constraints)
    )


# ----------------------------------------------------------------------------------
# witness map (src/r1cs_to_qap.rs:172-235)
# ----------------------------------------------------------------------------------


def witness_map_from_matrices(cp: CurveParams, cs: R1CS, z: Sequence[int], want_abc: bool = False):
    p = cp.r
    nc, nin = cs.num_constraints, cs.num_inputs
    dom = Domain(cp, nc + nin)
    n = dom.n
    a = [0] * n
    b = [0] * n
    for i in range(nc):  # :186-193
        a[i] = evaluate_constraint(cs.a[i], z, p)
        b[i] = evaluate_constraint(cs.b[i], z, p)
    for j in range(nin):  # :195-199
        a[nc + j] = z[j]
    c = [0] * n
    for i in range(nc):  # :213-218
        c[i] = evaluate_constraint(cs.c[i], z, p)
    abc_evals = (a[:], b[:], c[:])
    g = cp.fr_generator
    a = dom.coset_fft(dom.ifft(a), g)  # :201,206
    b = dom.coset_fft(dom.ifft(b), g)  # :202,207
    ab = [x * y % p for x, y in zip(a, b)]  # :209
    c = dom.coset_fft(dom.ifft(c), g)  # :220-221
    zinv = pow(dom.vanishing(g), p - 2, p)  # :223-226
    ab = [(x - y) * zinv % p for x, y in zip(ab, c)]  # :227-230
    h = dom.coset_ifft(ab, g)  # :232
    return (h, abc_evals) if want_abc else h


# ----------------------------------------------------------------------------------
# Distributed witness map: the algorithm the next round's multi-GPU path will run, modelled
# here with N simulated ranks so that its index maps are pinned before any kernel exists.  n = N * M.
#   residue distribution: rank r holds x[r + N*i2]           for i2 < M
#   block distribution:   rank r holds x[k2 + M*k1]          for k2 in [r*M/N, (r+1)*M/N), k1 < N
# An n-point transform is one local pass, one twiddle, ONE all-to-all and one more local pass (the 4-step FFT):
#   type 1 (residue -> block): M-point NTT over i2, times w^(i1 k2), exchange, N-point NTT over i1
#   type 2 (block -> residue): N-point NTT over k1, times w^(k2 ka), exchange, M-point NTT over k2
# so ifft -> coset fft -> pointwise -> coset ifft alternates 1, 2, 1 and needs no re-distribution in between; each rank ends
# with the h coefficients of its block distribution, which is then also how h_query has to be cut.
# ----------------------------------------------------------------------------------


def _ntt_small(x: Sequence[int], w: int, p: int) -> List[int]:
    """X[k] = sum_i x[i] w^(ik): by definition for short inputs, radix-2 recursion otherwise (power-of-two lengths)"""
    n = len(x)
    if n <= 16:
        return [sum(x[i] * pow(w, i * k, p) for i in range(n)) % p for k in range(n)]
    e, o = _ntt_small(x[0::2], w * w % p, p), _ntt_small(x[1::2], w * w % p, p)
    out, t = [0] * n, 1
    for k in range(n // 2):
        v = t * o[k] % p
        out[k], out[k + n // 2] = (e[k] + v) % p, (e[k] - v) % p
        t = t * w % p
    return out


def dist_transform_type1(ranks: List[List[int]], w: int, p: int) -> List[List[int]]:
    """ranks[r][i2] = x[r + N*i2]  ->  out[r][k1*(M//N) + j] = X[(r*M//N + j) + M*k1],  X[k] = sum_i x[i] w^(ik)"""
    N, M = len(ranks), len(ranks[0])
    wM, wN, blk = pow(w, N, p), pow(w, M, p), M // N
    send = []
    for i1 in range(N):
        y = _ntt_small(ranks[i1], wM, p)
        send.append([y[k2] * pow(w, i1 * k2, p) % p for k2 in range(M)])
    out = []
    for r in range(N):          # the all-to-all: rank r receives, from every i1, the slice k2 in its block
        cols = [[send[i1][r * blk + j] for i1 in range(N)] for j in range(blk)]
        res = [_ntt_small(c, wN, p) for c in cols]           # res[j][k1]
        out.append([res[j][k1] for k1 in range(N) for j in range(blk)])
    return out


def dist_transform_type2(ranks: List[List[int]], w: int, p: int) -> List[List[int]]:
    """ranks[r][k1*(M//N) + j] = x[(r*M//N + j) + M*k1]  ->  out[ka][kb] = X[ka + N*kb]"""
    N = len(ranks)
    blk = len(ranks[0]) // N
    M = blk * N
    wM, wN = pow(w, N, p), pow(w, M, p)
    send = [[None] * M for _ in range(N)]                    # send[ka][ib]
    for r in range(N):
        for j in range(blk):
            ib = r * blk + j
            z = _ntt_small([ranks[r][ia * blk + j] for ia in range(N)], wN, p)   # over ia, index i = ia*M + ib
            for ka in range(N):
                send[ka][ib] = z[ka] * pow(w, ib * ka, p) % p
    return [_ntt_small(send[ka], wM, p) for ka in range(N)]


def distributed_witness_map(cp: CurveParams, cs: R1CS, z: Sequence[int], N: int):
    """witness_map_from_matrices on N simulated ranks.  Returns (pieces, index_sets): pieces[r][t] = h[index_sets[r][t]]."""
    p = cp.r
    nc, nin = cs.num_constraints, cs.num_inputs
    dom = Domain(cp, nc + nin)
    n = dom.n
    assert n % (N * N) == 0, "needs N^2 | n"
    M = n // N
    blk = M // N
    g = cp.fr_generator
    ginv = pow(g, p - 2, p)

    def rows(mat, extra_inputs):
        out = []
        for r in range(N):     # rank r evaluates the constraints i = r (mod N): any row split works for the sparse products
            v = []
            for i2 in range(M):
                i = r + N * i2
                if i < nc:
                    v.append(evaluate_constraint(mat[i], z, p))
                elif extra_inputs and i < nc + nin:
                    v.append(z[i - nc])
                else:
                    v.append(0)
            out.append(v)
        return out

    def block_index(r, t):
        k1, j = divmod(t, blk)
        return (r * blk + j) + M * k1

    def coset_evals(ev):
        co = dist_transform_type1(ev, dom.omega_inv, p)                                               # ifft (x n)
        co = [[co[r][t] * dom.n_inv % p * pow(g, block_index(r, t), p) % p for t in range(M)] for r in range(N)]
        return dist_transform_type2(co, dom.omega, p)                                                  # coset fft

    a, b, c = coset_evals(rows(cs.a, True)), coset_evals(rows(cs.b, False)), coset_evals(rows(cs.c, False))
    zinv = pow(dom.vanishing(g), p - 2, p)
    q = [[(a[r][t] * b[r][t] - c[r][t]) * zinv % p for t in range(M)] for r in range(N)]              # residue distribution
    h = dist_transform_type1(q, dom.omega_inv, p)
    idx = [[block_index(r, t) for t in range(M)] for r in range(N)]
    h = [[h[r][t] * dom.n_inv % p * pow(ginv, idx[r][t], p) % p for t in range(M)] for r in range(N)]
    return h, idx


# ----------------------------------------------------------------------------------
# keys, setup with a known trapdoor (src/generator.rs:47-208)
# ----------------------------------------------------------------------------------


@dataclass
class Trapdoor:
    alpha: int
    beta: int
    gamma: int
    delta: int
    t: int
    g1: tuple
    g2: tuple
    # QAP evaluations at t (kept for the trapdoor KAT)
    a_t: List[int] = field(default_factory=list)
    b_t: List[int] = field(default_factory=list)
    c_t: List[int] = field(default_factory=list)
    zt: int = 0


@dataclass
class ProvingKey:
    """src/data_structures.rs:125-143 (+ the vk fields the prover reads)."""

    alpha_g1: tuple
    beta_g1: tuple
    beta_g2: tuple
    delta_g1: tuple
    delta_g2: tuple
    gamma_g2: tuple
    gamma_abc_g1: list
    a_query: list
    b_g1_query: list
    b_g2_query: list
    h_query: list
    l_query: list


def instance_map_with_evaluation(cp: CurveParams, cs: R1CS, t: int):
    # src/r1cs_to_qap.rs:128-170
    p = cp.r
    nc, nin = cs.num_constraints, cs.num_inputs
    dom = Domain(cp, nc + nin)
    zt = dom.vanishing(t)
    u = dom.lagrange_at(t)
    m = (nin - 1) + cs.num_witness
    a = [0] * (m + 1)
    b = [0] * (m + 1)
    c = [0] * (m + 1)
    for j in range(nin):
        a[j] = u[nc + j]
    for i in range(nc):
        for cf, idx in cs.a[i]:
            a[idx] = (a[idx] + u[i] * cf) % p
        for cf, idx in cs.b[i]:
            b[idx] = (b[idx] + u[i] * cf) % p
        for cf, idx in cs.c[i]:
            c[idx] = (c[idx] + u[i] * cf) % p
    return a, b, c, zt, m, dom.n


def generate_parameters(cp: CurveParams, cs: R1CS, seed: int) -> Tuple[ProvingKey, Trapdoor]:
    p = cp.r
    G1, G2 = groups(cp)
    rng = SplitMix64(seed ^ 0x5E7)
    alpha, beta, gamma, delta = (rng.field(p - 1) + 1 for _ in range(4))
    # the reference draws random generators (generator.rs:31-32)
    g1 = G1.mul(cp.g1, rng.field(p - 1) + 1)
    g2 = G2.mul(cp.g2, rng.field(p - 1) + 1)
    dom = Domain(cp, cs.num_constraints + cs.num_inputs)
    while True:  # sample_element_outside_domain (generator.rs:90)
        t = rng.field(p)
        if dom.vanishing(t) != 0:
            break
    a, b, c, zt, m, n = instance_map_with_evaluation(cp, cs, t)
    nin = cs.num_inputs
    gi = pow(gamma, p - 2, p)
    di = pow(delta, p - 2, p)
    gamma_abc = [(beta * a[i] + alpha * b[i] + c[i]) * gi % p for i in range(nin)]  # :113-117
    l = [(beta * a[i] + alpha * b[i] + c[i]) * di % p for i in range(nin, m + 1)]  # :119-123
    h_scalars = [zt * di % p * pow(t, i, p) % p for i in range(n - 1)]  # r1cs_to_qap.rs:243-245

    def bm(G, gen, scalars):
        return [G.mul(gen, s) if s % p else None for s in scalars]

    pk = ProvingKey(
        alpha_g1=G1.mul(g1, alpha),
        beta_g1=G1.mul(g1, beta),
        beta_g2=G2.mul(g2, beta),
        delta_g1=G1.mul(g1, delta),
        delta_g2=G2.mul(g2, delta),
        gamma_g2=G2.mul(g2, gamma),
        gamma_abc_g1=bm(G1, g1, gamma_abc),
        a_query=bm(G1, g1, a),
        b_g1_query=bm(G1, g1, b),
        b_g2_query=bm(G2, g2, b),
        h_query=bm(G1, g1, h_scalars),
        l_query=bm(G1, g1, l),
    )
    td = Trapdoor(alpha, beta, gamma, delta, t, g1, g2, a, b, c, zt)
    return pk, td


# ----------------------------------------------------------------------------------
# prover (src/prover.rs:26-132, 252-270)
# ----------------------------------------------------------------------------------


@dataclass
class Proof:
    a: tuple
    b: tuple
    c: tuple


def calculate_coeff(G: Group, initial, query, vk_param, assignment):
    # src/prover.rs:252-270
    acc = G.msm(query[1:], assignment)
    res = G.add(initial, query[0])
    res = G.add(res, acc)
    return G.add(res, vk_param)


def create_proof_with_assignment(cp, pk: ProvingKey, r: int, s: int, h, input_assignment, aux_assignment, parts=None):
    G1, G2 = groups(cp)
    p = cp.r
    h_acc = G1.msm(pk.h_query, h)  # :66
    l_aux_acc = G1.msm(pk.l_query, aux_assignment)  # :74
    r_s_delta_g1 = G1.mul(pk.delta_g1, r * s % p)  # :76
    assignment = list(input_assignment) + list(aux_assignment)  # :80-85
    r_g1 = G1.mul(pk.delta_g1, r)
    g_a = calculate_coeff(G1, r_g1, pk.a_query, pk.alpha_g1, assignment)  # :90-92
    s_g_a = G1.mul(g_a, s)  # :94
    if r % p != 0:  # :98-108
        s_g1 = G1.mul(pk.delta_g1, s)
        g1_b = calculate_coeff(G1, s_g1, pk.b_g1_query, pk.beta_g1, assignment)
    else:
        g1_b = None
    s_g2 = G2.mul(pk.delta_g2, s)
    g2_b = calculate_coeff(G2, s_g2, pk.b_g2_query, pk.beta_g2, assignment)  # :112-113
    r_g1_b = G1.mul(g1_b, r)
    g_c = G1.add(s_g_a, r_g1_b)  # :119-124
    g_c = G1.add(g_c, G1.neg(r_s_delta_g1))
    g_c = G1.add(g_c, l_aux_acc)
    g_c = G1.add(g_c, h_acc)
    if parts is not None:
        parts.update(
            h_acc=h_acc,
            l_acc=l_aux_acc,
            a_msm=G1.msm(pk.a_query[1:], assignment),
            b1_msm=G1.msm(pk.b_g1_query[1:], assignment),
            b2_msm=G2.msm(pk.b_g2_query[1:], assignment),
        )
    return Proof(g_a, g2_b, g_c)


def create_proof_with_reduction_and_matrices(cp, pk, r, s, cs: R1CS, full_assignment, parts=None):
    # src/prover.rs:26-51
    h = witness_map_from_matrices(cp, cs, full_assignment)
    if parts is not None:
        parts["h"] = h
    nin = cs.num_inputs
    return create_proof_with_assignment(cp, pk, r, s, h, full_assignment[1:nin], full_assignment[nin:], parts)


def trapdoor_expected_proof(cp, cs: R1CS, td: Trapdoor, z, r, s, h) -> Proof:
    """Expected proof as scalar*generator, sharing no MSM/NTT code with the prover
    (pin (2)).  h enters only through h(t) = sum h_i t^i."""
    p = cp.r
    G1, G2 = groups(cp)
    m = len(td.a_t) - 1
    nin = cs.num_inputs
    n = Domain(cp, cs.num_constraints + nin).n
    A = (td.alpha + sum(z[i] * td.a_t[i] for i in range(m + 1)) + r * td.delta) % p
    B = (td.beta + sum(z[i] * td.b_t[i] for i in range(m + 1)) + s * td.delta) % p
    di = pow(td.delta, p - 2, p)
    lsum = sum(z[i] * (td.beta * td.a_t[i] + td.alpha * td.b_t[i] + td.c_t[i]) for i in range(nin, m + 1)) % p
    ht = sum(h[i] * pow(td.t, i, p) for i in range(n - 1)) % p
    C = (lsum * di + ht * td.zt % p * di + s * A + r * B - r * s % p * td.delta) % p
    return Proof(G1.mul(td.g1, A), G2.mul(td.g2, B), G1.mul(td.g1, C))


# ----------------------------------------------------------------------------------
# Montgomery limb conversion + point encodings
# ----------------------------------------------------------------------------------


def to_mont_limbs(x: int, p: int, nlimbs64: int) -> List[int]:
    v = (x << (64 * nlimbs64)) % p
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(nlimbs64)]


def from_mont_limbs(limbs: Sequence[int], p: int) -> int:
    n = len(limbs)
    v = sum(int(l) << (64 * i) for i, l in enumerate(limbs))
    return v * pow(1 << (64 * n), p - 2, p) % p


def compress_g1_bls(P) -> bytes:
    """zcash/IETF BLS12-381 compressed G1 (ark-bls12-381 override) [EXT-MEM]."""
    q = BLS12_381.q
    if P is None:
        return bytes([0xC0] + [0] * 47)
    x, y = P
    b = bytearray(x.to_bytes(48, "big"))
    b[0] |= 0x80
    if y > (q - 1) // 2:
        b[0] |= 0x20
    return bytes(b)


def compress_g2_bls(P) -> bytes:
    q = BLS12_381.q
    if P is None:
        return bytes([0xC0] + [0] * 95)
    (x0, x1), (y0, y1) = P
    b = bytearray(x1.to_bytes(48, "big") + x0.to_bytes(48, "big"))
    b[0] |= 0x80
    neg = (q - y0) % q, (q - y1) % q
    largest = (y1, y0) > (neg[1], neg[0])
    if largest:
        b[0] |= 0x20
    return bytes(b)


def compress_g1_bn(P) -> bytes:
    """arkworks default SW compressed encoding (LE x, flags in the top bits of the
    last byte: bit7 = y is 'negative' (y > -y), bit6 = infinity) [EXT-MEM]."""
    q = BN254.q
    if P is None:
        b = bytearray(32)
        b[-1] |= 0x40
        return bytes(b)
    x, y = P
    b = bytearray(x.to_bytes(32, "little"))
    if y > (q - y) % q:
        b[-1] |= 0x80
    return bytes(b)


def compress_g2_bn(P) -> bytes:
    q = BN254.q
    if P is None:
        b = bytearray(64)
        b[-1] |= 0x40
        return bytes(b)
    (x0, x1), (y0, y1) = P
    b = bytearray(x0.to_bytes(32, "little") + x1.to_bytes(32, "little"))
    n0, n1 = (q - y0) % q, (q - y1) % q
    if (y1, y0) > (n1, n0):
        b[-1] |= 0x80
    return bytes(b)


def proof_bytes(cp: CurveParams, pr: Proof) -> bytes:
    """Proof serialises a || b || c (src/data_structures.rs:8-16)."""
    if cp.name == "bls12_381":
        return compress_g1_bls(pr.a) + compress_g2_bls(pr.b) + compress_g1_bls(pr.c)
    return compress_g1_bn(pr.a) + compress_g2_bn(pr.b) + compress_g1_bn(pr.c)


# ----------------------------------------------------------------------------------
# structural KATs
# ----------------------------------------------------------------------------------


def selfcheck() -> None:
    for cp in (BLS12_381, BN254):
        G1, G2 = groups(cp)
        assert G1.on_curve(cp.g1) and G2.on_curve(cp.g2), cp.name
        assert G1.mul(cp.g1, cp.r) is None and G2.mul(cp.g2, cp.r) is None, cp.name
        assert G1.mul(cp.g1, cp.r - 1) == G1.neg(cp.g1)
        w = cp.two_adic_root
        assert pow(w, 1 << cp.two_adicity, cp.r) == 1 and pow(w, 1 << (cp.two_adicity - 1), cp.r) != 1
        assert (cp.r - 1) % (1 << cp.two_adicity) == 0 and ((cp.r - 1) >> cp.two_adicity) % 2 == 1
        d = Domain(cp, 8)
        assert pow(d.omega, 8, cp.r) == 1 and pow(d.omega, 4, cp.r) != 1
        x = [SplitMix64(1).field(cp.r) for _ in range(8)]
        assert d.ifft(d.fft(x)) == x and d.coset_ifft(d.coset_fft(x, 7), 7) == x
    assert BLS12_381.two_adic_root == 10238227357739495823651030575849232062558860180284477541189508159991286009131
    assert BN254.two_adic_root == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    # BN254 b2 = 3/(9+u)
    F2 = Fq2(BN254.q)
    assert F2.mul(BN254.b2, (9, 1)) == (3, 0)
    # zcash compressed generator KAT
    assert compress_g1_bls(BLS12_381.g1).hex() == (
        "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac58"
        "6c55e83ff97a1aeffb3af00adb22c6bb"
    )
    # Montgomery constants quoted in SURVEY.md 8(c)
    assert (-pow(BLS12_381.q, -1, 1 << 64)) % (1 << 64) == 0x89F3FFFCFFFCFFFD
    assert (-pow(BLS12_381.r, -1, 1 << 64)) % (1 << 64) == 0xFFFFFFFEFFFFFFFF
    assert (-pow(BN254.q, -1, 1 << 64)) % (1 << 64) == 0x87D20782E4866389
    assert (-pow(BN254.r, -1, 1 << 64)) % (1 << 64) == 0xC2E1F593EFFFFFFF
    assert (1 << 256) % BLS12_381.r == 0x1824B159ACC5056F998C4FEFECBC4FF55884B7FA0003480200000001FFFFFFFE


if __name__ == "__main__":
    selfcheck()
    print("pymodel selfcheck OK")


# ----------------------------------------------------------------------------------
# verifier (src/verifier.rs:13-77) -- SURVEY.md row f2.  A deliberately plain reduced Tate pairing
# e(P, Q) = f_{r,P}(Q)^((q^12-1)/r) over Fq12 = Fq[w]/(w^12 - c6 w^6 - c0): any non-degenerate bilinear pairing
# decides the Groth16 equation, so this need not be arkworks' optimal ate pairing.  Slow (about a second per pairing).
# ----------------------------------------------------------------------------------

_TOWER = {
    # xi = s + u is the sextic non-residue (Fq6 = Fq2[v]/(v^3 - xi), Fq12 = Fq6[w]/(w^2 - v)); w^6 = xi and u^2 = -1 give
    # (w^6 - s)^2 = -1, i.e. w^12 = 2 s w^6 - (s^2 + 1).  twist: 'M' (b' = b xi, BLS12-381) or 'D' (b' = b / xi, BN254).
    "bls12_381": dict(s=1, twist="M"),
    "bn254": dict(s=9, twist="D"),
}


class Fq12:
    def __init__(self, cp: CurveParams):
        self.p = cp.q
        s = _TOWER[cp.name]["s"]
        self.s = s
        self.c6 = 2 * s % self.p
        self.c0 = (-(s * s + 1)) % self.p
        self.one = [1] + [0] * 11
        self.zero = [0] * 12

    def add(self, a, b):
        return [(x + y) % self.p for x, y in zip(a, b)]

    def sub(self, a, b):
        return [(x - y) % self.p for x, y in zip(a, b)]

    def scale(self, a, k):
        return [x * k % self.p for x in a]

    def mul(self, a, b):
        p = self.p
        t = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    t[i + j] += x * y
        for i in range(22, 11, -1):
            v = t[i] % p
            if v:
                t[i - 6] += self.c6 * v
                t[i - 12] += self.c0 * v
        return [x % p for x in t[:12]]

    def pow(self, a, e):
        r = self.one
        for bit in bin(e)[2:]:
            r = self.mul(r, r)
            if bit == "1":
                r = self.mul(r, a)
        return r

    def from_fq2(self, a):  # a0 + a1 u with u = w^6 - s
        c = [0] * 12
        c[0] = (a[0] - self.s * a[1]) % self.p
        c[6] = a[1] % self.p
        return c

    def inv(self, a):
        return self.pow(a, self.p ** 12 - 2)


def untwist(cp: CurveParams, Q):
    """G2 affine point over Fq2 -> point of E(Fq12)"""
    F = Fq12(cp)
    x, y = F.from_fq2(Q[0]), F.from_fq2(Q[1])
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    if _TOWER[cp.name]["twist"] == "D":
        return F.mul(x, w2), F.mul(y, w3)
    return F.mul(x, F.inv(w2)), F.mul(y, F.inv(w3))


def tate_pairing(cp: CurveParams, P, Q):
    """reduced Tate pairing of P in G1 (affine over Fq) and Q in G2 (affine over Fq2); identity inputs give 1"""
    F = Fq12(cp)
    if P is None or Q is None:
        return F.one
    p = cp.q
    xq, yq = untwist(cp, Q)
    xp, yp = P
    tx, ty = xp, yp
    f = F.one

    def line(lam, x0, y0):  # (yQ - y0) - lam (xQ - x0), lam in Fq
        v = F.sub(yq, F.scale(xq, lam))
        v[0] = (v[0] + lam * x0 - y0) % p
        return v

    bits = bin(cp.r)[3:]
    for i, bit in enumerate(bits):
        lam = 3 * tx * tx * pow(2 * ty, p - 2, p) % p
        f = F.mul(F.mul(f, f), line(lam, tx, ty))
        nx = (lam * lam - 2 * tx) % p
        ty = (lam * (tx - nx) - ty) % p
        tx = nx
        if bit == "1":
            if tx == xp:  # T = -P at the very last step: vertical line, killed by the final exponentiation
                assert (ty + yp) % p == 0 and i == len(bits) - 1
                continue
            lam = (yp - ty) * pow((xp - tx) % p, p - 2, p) % p
            f = F.mul(f, line(lam, tx, ty))
            nx = (lam * lam - tx - xp) % p
            ty = (lam * (tx - nx) - ty) % p
            tx = nx
    return F.pow(f, (p ** 12 - 1) // cp.r)


def prepare_inputs(cp: CurveParams, gamma_abc_g1, public_inputs):
    # src/verifier.rs:25-39
    if len(public_inputs) + 1 != len(gamma_abc_g1):
        raise ValueError("MalformedVerifyingKey")
    G1, _ = groups(cp)
    acc = gamma_abc_g1[0]
    for x, b in zip(public_inputs, gamma_abc_g1[1:]):
        acc = G1.add(acc, G1.mul(b, x))
    return acc


def verify_proof(cp: CurveParams, pk: ProvingKey, proof: Proof, public_inputs) -> bool:
    """src/verifier.rs:44-76: e(A,B) e(IC,-gamma) e(C,-delta) == e(alpha, beta), on the vk carried by `pk`"""
    G1, _ = groups(cp)
    F = Fq12(cp)
    ic = prepare_inputs(cp, pk.gamma_abc_g1, public_inputs)
    lhs = tate_pairing(cp, proof.a, proof.b)
    lhs = F.mul(lhs, tate_pairing(cp, G1.neg(ic), pk.gamma_g2))
    lhs = F.mul(lhs, tate_pairing(cp, G1.neg(proof.c), pk.delta_g2))
    return lhs == tate_pairing(cp, pk.alpha_g1, pk.beta_g2)
