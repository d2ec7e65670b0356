#!/bin/bash
# Round 6 A/B (one box): transform workgroups of ONE wave (2^8-element tiles, 9.5 KB of LDS: libg16_ntt8.so) so that they compete for
# slots like a bucket-pass workgroup does, with the map first (as always) and with the map underneath the passes
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
V=$PWD/groth16_amd/libg16_ntt8.so
single main_first_a G16_NOOP=1
single ntt8_first_a G16_LIB=$V
single ntt8_under_a G16_LIB=$V G16_MAP_UNDER_PASSES=1
single main_first_b G16_NOOP=1
single ntt8_first_b G16_LIB=$V
single ntt8_under_b G16_LIB=$V G16_MAP_UNDER_PASSES=1
timeout 200 rocprofv3 --kernel-trace -d $O/prof_trace -o tr --output-format csv -- env G16_LIB=$V G16_MAP_UNDER_PASSES=1 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-pipelined --no-projection > $O/trace_under.json 2> $O/trace_under.err
find $O/prof_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_under.csv; rm -rf $O/prof_trace
python tools/trace_timeline.py $O/kernel_trace_under.csv > $O/timeline_under.txt; grep -E "ntt30|spmv|bitrev|bucket_accumulate|class_count|window_reduce" $O/timeline_under.txt | tail -26
