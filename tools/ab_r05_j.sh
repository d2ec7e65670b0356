#!/bin/bash
# Round 5 A/B (one box): the G1 first-stage reduction kernels held to 128 registers (fit beside the G1 pass) vs unconstrained
# (libg16_fs2.so = tools/build_variant.sh fs2 "-DG16_FIRST_STAGE_WAVES=2")
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
V=$PWD/groth16_amd/libg16_fs2.so
single w4_a G16_NOOP=1
single w2_a G16_LIB=$V
single w4_b G16_NOOP=1
single w2_b G16_LIB=$V
shard bucket_w4 bucket G16_NOOP=1
shard bucket_w2 bucket G16_LIB=$V
shard bucket_w4b bucket G16_NOOP=1
shard bucket_w2b bucket G16_LIB=$V
