#!/bin/bash
# builds the micro-benchmarks in three code-shape variants (field-mul inlined / outlined by reference / by value)
set -e
cd "$(dirname "$0")/.."
CS=groth16_amd/csrc
mkdir -p tools/bin
build() { # name, flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$CS -I include $2 -DG16_VARIANT="\"$1\"" tools/ubench.hip $CS/synth.hip -o tools/bin/ubench_$1 &
}
build outlined_ref "-DG16_UBENCH_OPS"
build outlined_val "-DG16_MUL_BYVAL"
build inlined "-DG16_NOINLINE_MUL_LIMBS=99"
wait
ls -la tools/bin
