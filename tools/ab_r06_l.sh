#!/bin/bash
# Round 6 (one box): the h MSM inside the one G1 batch on a single GPU (G16_PASS_H_IN_BATCH=2) against the default (h pass first, then l/a/b_g1 batch)
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
for r in a b c; do
  single base_$r G16_NOOP=1
  single hbatch_$r G16_PASS_H_IN_BATCH=2
done
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
G16_PASS_H_IN_BATCH=2 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not 2_24" 2>&1 | tail -3
