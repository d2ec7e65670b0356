#!/bin/bash
# Round 6 A/B (one box): product-scanning field products (main) vs the round-5 operand-scanning forms (libg16_nofips.so =
# tools/build_variant.sh nofips "-DG16_NO_FIPS" msm.o msm_bn254.o ntt.o), and the bucket pass's workgroups per compute unit
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
V=$PWD/groth16_amd/libg16_nofips.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
single fips_a G16_NOOP=1
single nofips_a G16_LIB=$V
single fips_wg8 G16_PASS_WG_PER_CU=8
single fips_wg10 G16_PASS_WG_PER_CU=10
single fips_b G16_NOOP=1
single nofips_b G16_LIB=$V
shard bucket_fips bucket G16_NOOP=1
shard bucket_nofips bucket G16_LIB=$V
shard bucket_fips_wg8 bucket G16_PASS_WG_PER_CU=8
