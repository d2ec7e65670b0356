#!/bin/bash
# Round 6 (one box): buckets per lane of the bucket reduction (G16_MSM_REDUCE_G; default 8), whole proof and 8-way bucket-space share
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
for r in a b c; do
  single g8_$r G16_NOOP=1
  single g16_$r G16_MSM_REDUCE_G=16
  single g32_$r G16_MSM_REDUCE_G=32
done
shard bucket_g8 bucket G16_NOOP=1
shard bucket_g16 bucket G16_MSM_REDUCE_G=16
shard bucket_g8b bucket G16_NOOP=1
shard bucket_g16b bucket G16_MSM_REDUCE_G=16
