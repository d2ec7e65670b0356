#!/bin/bash
# Round 6 A/B (one box): token pins (no wait states; G1 kernel 248 registers, two waves per SIMD) vs the round-5 forms
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
V=$PWD/groth16_amd/libg16_nofips.so
single tok_a G16_NOOP=1
single nofips_a G16_LIB=$V
single tok_b G16_NOOP=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r06_ab_b/single_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); print(f, d["phases_ms_per_step"].get("bucket_ms"), d["roofline"]["valu_bound"])
PY
