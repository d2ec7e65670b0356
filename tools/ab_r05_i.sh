#!/bin/bash
# Round 5 A/B (one box): the witness map on its own stream underneath the passes (G16_MAP_UNDER_PASSES=1) vs first and alone
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
single first_a G16_NOOP=1
single under_a G16_MAP_UNDER_PASSES=1
single first_b G16_NOOP=1
single under_b G16_MAP_UNDER_PASSES=1
