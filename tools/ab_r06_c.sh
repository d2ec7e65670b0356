#!/bin/bash
# Round 6 A/B (one box): product-scanning field products as generated assembly (main) vs the round-5 operand-scanning forms
# (libg16_nofips.so = tools/build_variant.sh nofips "-DG16_NO_FIPS" msm.o msm_bn254.o ntt.o), bucket-pass workgroups per CU
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
V=$PWD/groth16_amd/libg16_nofips.so
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
single asm_a G16_NOOP=1
single nofips_a G16_LIB=$V
single asm_wg8 G16_PASS_WG_PER_CU=8
single asm_b G16_NOOP=1
single nofips_b G16_LIB=$V
shard bucket_asm bucket G16_NOOP=1
shard bucket_nofips bucket G16_LIB=$V
python - <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/single_*.json" % sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r06_ab_c/single_*.json")):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); p=d["phases_ms_per_step"]; print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in p.items()}, round(d["roofline"]["valu_bound"]["achieved_Tmad_s"],2), round(d["roofline"]["valu_bound"]["g2"]["achieved_Tmad_s"],2))
PY
