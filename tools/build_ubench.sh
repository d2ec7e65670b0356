#!/bin/bash
# builds the micro-benchmarks in three code-shape variants (field-mul inlined / outlined by reference / by value)
set -e
cd "$(dirname "$0")/.."
CS=groth16_amd/csrc
mkdir -p tools/bin
build() { # name, flags
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$CS -I include $2 -DG16_VARIANT="\"$1\"" tools/ubench.hip $CS/synth.hip -o tools/bin/ubench_$1 &
}
rm -f tools/bin/ubench_*
build default ""
build fp2inline "-DG16_FP2X30_INLINE"
wait
ls -la tools/bin
