#!/bin/bash
# Round 6: SQ counters of the bucket kernels, product-scanning assembly (main) vs round 5's forms (libg16_nofips.so), one box
O=gpurun_out/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/sq_counters.txt
V=$PWD/groth16_amd/libg16_nofips.so
run() {  # tag, counters..., env
  tag=$1; shift
  ctrs=$1; shift
  timeout 400 env "$@" rocprofv3 --pmc $ctrs -d $O/pmc_$tag -o pmc --output-format csv -- \
     python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-projection > $O/bench_$tag.json 2> $O/bench_$tag.err
  echo "$tag rc=$?"
  python - $O/pmc_$tag "$ctrs" $tag >> $O/summary.txt <<'PY'
import sys, json, subprocess
d, ctrs, tag = sys.argv[1], sys.argv[2].split(), sys.argv[3]
for c in ctrs:
    out = json.loads(subprocess.run([sys.executable, "tools/pmc_summary.py", "raw", d, c], capture_output=True, text=True).stdout or "{}")
    for k, v in out.items():
        if "bucket_accumulate30" in k or "ntt30" in k:
            print(tag, c, k[:70], v)
PY
  rm -rf $O/pmc_$tag
}
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
B="SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"
Cc="SQ_IFETCH SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC"
run asm_a "$A" G16_NOOP=1
run old_a "$A" G16_LIB=$V
run asm_b "$B" G16_NOOP=1
run old_b "$B" G16_LIB=$V
run asm_c "$Cc" G16_NOOP=1
run old_c "$Cc" G16_LIB=$V
cat $O/summary.txt
