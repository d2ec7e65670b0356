"""groth16_amd -- MI355X-native Groth16 prover hot path behind ark-groth16's prover API.

Host-side mirror (Python, because no Rust toolchain exists in this image) of the reference's
prover interface for the accelerated path; everything below the method signatures goes through
the C ABI of ``libg16_mi355x.so`` (include/g16_mi355x.h) -- there is no CPU fallback.

Reference interface mirrored (paths relative to /root/reference):
  Groth16.create_proof_with_reduction_and_matrices   src/prover.rs:26-51
  Groth16.create_proof_with_reduction_no_zk          src/prover.rs:155-168 (matrices form)
  Groth16.create_random_proof_with_reduction         src/prover.rs:138-150 (matrices form)
  LibsnarkReduction.witness_map_from_matrices        src/r1cs_to_qap.rs:172-235
  ProvingKey / Proof / ConstraintMatrices            src/data_structures.rs:8-16,125-143
  SynthesisError.PolynomialDegreeTooLarge            src/r1cs_to_qap.rs:178-179
  Groth16.generate_parameters_with_qap               src/generator.rs:47-208 (matrices form)
  rerandomize_proof / Groth16.rerandomize_proof      src/prover.rs:223-250 (host: three scalar multiplications)
  Groth16.create_proof_with_reduction / prove / setup src/prover.rs:173-217, src/lib.rs:63-82, src/generator.rs:20-45 -- host-side
      synthesis (groth16_amd.r1cs: ConstraintSystem, Variable, lc, ConstraintSynthesizer) in front of the GPU calls
"""
from .binding import (G16Error, Lib, PolynomialDegreeTooLarge, SynthesisError, UnexpectedIdentity, lib, FQ_LIMBS, CURVE_ID)  # noqa: F401
from .groth16 import (ConstraintMatrices, Groth16, LibsnarkReduction, PipelinedProver, Proof, ProvingKey, ShardedProver, finalize_host,  # noqa: F401
                      rerandomize_proof, shard_ranges)
from .r1cs import AssignmentMissing, ConstraintSynthesizer, ConstraintSystem, LinearCombination, Variable, lc  # noqa: F401

__all__ = [
    "Groth16", "LibsnarkReduction", "ConstraintMatrices", "ProvingKey", "Proof", "ShardedProver", "PipelinedProver", "G16Error", "SynthesisError",
    "PolynomialDegreeTooLarge", "UnexpectedIdentity", "lib", "ConstraintSystem", "ConstraintSynthesizer", "Variable", "LinearCombination",
    "lc", "AssignmentMissing",
]
