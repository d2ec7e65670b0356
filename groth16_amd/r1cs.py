"""Host-side circuit synthesis in front of the GPU boundary: the part of ``Groth16::create_proof_with_reduction``
(/root/reference/src/prover.rs:173-217) and of ``generate_random_parameters_with_reduction`` (src/generator.rs:20-45, 47-88) that
runs BEFORE the pure-data call -- a ``ConstraintSynthesizer`` fills a constraint system, the system is flattened to
``ConstraintMatrices`` plus the full assignment, and those go to ``create_proof_with_reduction_and_matrices`` /
``generate_parameters_with_qap`` on the MI355X.

The reference takes this layer from ark-relations (un-vendored, 0.5); what is restated here is the subset its own tests use
(src/test.rs:14-43, tests/mimc.rs:64-143): ``new_input_variable`` / ``new_witness_variable`` with value closures that are not
evaluated in setup mode, ``enforce_constraint`` over linear combinations built with ``lc() + var``, ``+ (coeff, var)``,
``- var``, ``to_matrices`` with instance columns first (column 0 = the constant one) and ``SynthesisError::AssignmentMissing``.
Field values are Python integers mod r; limbs appear only in ``to_matrices`` / ``full_assignment``."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple, Union

import numpy as np

MODULUS_R = {
    "bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
}


class SynthesisError(Exception):
    """ark_relations::r1cs::SynthesisError (the variants synthesis itself can raise)"""


class AssignmentMissing(SynthesisError):
    pass


@dataclass(frozen=True)
class Variable:
    """ark_relations::r1cs::Variable: One, Instance(i), Witness(i)"""

    kind: str   # "one" | "instance" | "witness"
    index: int = 0

    def __add__(self, other):
        return lc() + self + other

    def __sub__(self, other):
        return lc() + self - other


Variable.One = Variable("one", 0)   # type: ignore[attr-defined]
Term = Tuple[int, Variable]


class LinearCombination:
    """Vec<(F, Variable)>; ``lc() + a + (c, Variable.One) - b`` as with ark's ``lc!()`` macro"""

    def __init__(self, terms: Optional[List[Term]] = None):
        self.terms: List[Term] = list(terms or [])

    @staticmethod
    def _terms(x, sign: int) -> List[Term]:
        if isinstance(x, Variable):
            return [(sign, x)]
        if isinstance(x, LinearCombination):
            return [(sign * c, v) for c, v in x.terms]
        if isinstance(x, tuple) and len(x) == 2 and isinstance(x[1], Variable):
            return [(sign * int(x[0]), x[1])]
        raise TypeError(f"cannot add {type(x).__name__} to a linear combination")

    def __add__(self, x):
        return LinearCombination(self.terms + self._terms(x, 1))

    def __sub__(self, x):
        return LinearCombination(self.terms + self._terms(x, -1))


def lc() -> LinearCombination:
    return LinearCombination()


class ConstraintSystem:
    """ark_relations::r1cs::ConstraintSystem: instance variable 0 is the constant one (so num_instance_variables starts at 1,
    as in ark), witness variables follow the instance variables in the matrices' column order."""

    def __init__(self, curve: str, setup_mode: bool = False):
        if curve not in MODULUS_R:
            raise ValueError(f"unsupported curve {curve}")
        self.curve, self.p, self.setup_mode = curve, MODULUS_R[curve], setup_mode
        self.instance_assignment: List[int] = [1]
        self.witness_assignment: List[int] = []
        self.num_instance_variables, self.num_witness_variables = 1, 0
        self._rows: Tuple[List[List[Term]], List[List[Term]], List[List[Term]]] = ([], [], [])

    # -- allocation (closures are not evaluated in setup mode: SynthesisMode::Setup) -------------------------------------
    def _value(self, f: Callable[[], Optional[int]]) -> int:
        v = f()
        if v is None:
            raise AssignmentMissing("a value closure returned None in proving mode")
        return int(v) % self.p

    def new_input_variable(self, f: Callable[[], Optional[int]]) -> Variable:
        if not self.setup_mode:
            self.instance_assignment.append(self._value(f))
        self.num_instance_variables += 1
        return Variable("instance", self.num_instance_variables - 1)

    def new_witness_variable(self, f: Callable[[], Optional[int]]) -> Variable:
        if not self.setup_mode:
            self.witness_assignment.append(self._value(f))
        self.num_witness_variables += 1
        return Variable("witness", self.num_witness_variables - 1)

    def enforce_constraint(self, a, b, c) -> None:
        for rows, x in zip(self._rows, (a, b, c)):
            rows.append(LinearCombination._terms(x, 1) if not isinstance(x, LinearCombination) else list(x.terms))

    @property
    def num_constraints(self) -> int:
        return len(self._rows[0])

    # -- flattening --------------------------------------------------------------------------------------------------
    def _column(self, v: Variable) -> int:
        if v.kind == "one":
            return 0
        return v.index if v.kind == "instance" else self.num_instance_variables + v.index

    def _flat_rows(self, rows: List[List[Term]]) -> List[List[Tuple[int, int]]]:
        """per row: (coefficient mod r, column), like terms merged, zero coefficients dropped (ark's `compactify`
        leaves the row order alone; the witness map is linear in each row, so merging changes no result)"""
        out = []
        for row in rows:
            acc = {}
            for cf, v in row:
                col = self._column(v)
                acc[col] = (acc.get(col, 0) + cf) % self.p
            out.append([(cf, col) for col, cf in sorted(acc.items()) if cf])
        return out

    def to_matrices(self):
        """cs.to_matrices(): the three matrices as the CSR form the C ABI takes, coefficients as Montgomery limbs"""
        from .groth16 import ConstraintMatrices

        def conv(rows):
            flat = self._flat_rows(rows)
            rp = np.zeros(len(flat) + 1, dtype=np.uint64)
            cols, vals = [], []
            for i, row in enumerate(flat):
                for cf, col in row:
                    cols.append(col)
                    vals.append(cf)
                rp[i + 1] = len(cols)
            return rp, np.asarray(cols, dtype=np.uint32), to_montgomery(vals, self.p)

        return ConstraintMatrices(self.num_instance_variables, self.num_witness_variables, self.num_constraints, conv(self._rows[0]),
                                  conv(self._rows[1]), conv(self._rows[2]))

    def full_assignment(self) -> np.ndarray:
        """[instance_assignment, witness_assignment].concat() (prover.rs:199-204) as Montgomery limbs"""
        if self.setup_mode:
            raise AssignmentMissing("a setup-mode constraint system has no assignment")
        return to_montgomery(self.instance_assignment + self.witness_assignment, self.p)

    def is_satisfied(self) -> bool:
        """debug_assert!(cs.is_satisfied().unwrap()) of prover.rs:193"""
        z = self.instance_assignment + self.witness_assignment
        flat = [self._flat_rows(r) for r in self._rows]
        ev = lambda row: sum(cf * z[col] for cf, col in row) % self.p  # noqa: E731
        return all(ev(a) * ev(b) % self.p == ev(c) for a, b, c in zip(*flat))


def to_montgomery(vals: Sequence[int], p: int) -> np.ndarray:
    """integers mod p -> (n, 4) uint64 limbs of v * 2^256 mod p (the arkworks in-memory form of Fr)"""
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    mask = (1 << 64) - 1
    for i, v in enumerate(vals):
        m = ((int(v) % p) << 256) % p
        out[i] = [m & mask, (m >> 64) & mask, (m >> 128) & mask, (m >> 192) & mask]
    return out


def from_montgomery(arr: np.ndarray, p: int) -> List[int]:
    rinv = pow(1 << 256, -1, p)
    a = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [(sum(int(a[i, k]) << (64 * k) for k in range(4)) * rinv) % p for i in range(a.shape[0])]


class ConstraintSynthesizer:
    """trait ConstraintSynthesizer<F>: implement ``generate_constraints(self, cs)``; any object with that method is accepted"""

    def generate_constraints(self, cs: ConstraintSystem) -> None:   # pragma: no cover - interface
        raise NotImplementedError


Synth = Union[ConstraintSynthesizer, object]


def synthesize(curve: str, circuit: Synth, setup_mode: bool) -> ConstraintSystem:
    """prover.rs:185-198 / generator.rs:57-68: a fresh system, the optimisation goal (Constraints: no rewriting of the rows),
    generate_constraints, finalize"""
    cs = ConstraintSystem(curve, setup_mode)
    circuit.generate_constraints(cs)
    return cs
