#!/bin/bash
# Round 5 A/B (one box): sort classes of 2^13 buckets (fit beside the passes) vs 2^15 (128 KB histograms)
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
single c13_a G16_NOOP=1
single c15_a G16_SORT_CLASS_LOG=15
single c13_b G16_NOOP=1
single c15_b G16_SORT_CLASS_LOG=15
shard bucket_c13 bucket G16_NOOP=1
shard bucket_c15 bucket G16_SORT_CLASS_LOG=15
