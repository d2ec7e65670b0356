#!/bin/bash
# Round 6 A/B (one box): the witness map on its own stream UNDERNEATH the passes (G16_MAP_UNDER_PASSES=1), re-tested because a transform
# workgroup (91-98 registers since the assembly products) now fits beside the lane-pair kernel's waves too; + the MFMA glue probe again
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
tools/bin/probe_mfma > $O/probe_mfma.txt 2>&1; cat $O/probe_mfma.txt
single first_a G16_NOOP=1
single under_a G16_MAP_UNDER_PASSES=1
single first_b G16_NOOP=1
single under_b G16_MAP_UNDER_PASSES=1
timeout 200 rocprofv3 --kernel-trace -d $O/prof_trace -o tr --output-format csv -- env G16_MAP_UNDER_PASSES=1 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-pipelined --no-projection > $O/trace_under.json 2> $O/trace_under.err
find $O/prof_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_under.csv; rm -rf $O/prof_trace
python tools/trace_timeline.py $O/kernel_trace_under.csv > $O/timeline_under.txt; grep -E "ntt30|spmv|bitrev|bucket_accumulate|class_count" $O/timeline_under.txt | tail -24
