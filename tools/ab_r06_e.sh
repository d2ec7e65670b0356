#!/bin/bash
# Round 6 A/B (one box): subtractions fused into the product blocks (main) vs the build before (libg16_pre.so: assembly products, stand-alone subtractions)
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or proof or bucket" > $O/pytest_parity.log 2>&1; tail -3 $O/pytest_parity.log
single new_a G16_NOOP=1
single pre_a G16_LIB=$PWD/groth16_amd/libg16_pre.so
single new_b G16_NOOP=1
single pre_b G16_LIB=$PWD/groth16_amd/libg16_pre.so
shard bucket_new bucket G16_NOOP=1
shard bucket_pre bucket G16_LIB=$PWD/groth16_amd/libg16_pre.so
python - $1 <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/single_*.json" % sys.argv[1])):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); p=d["phases_ms_per_step"]; print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in p.items() if k in("msm_b_g2_ms","msm_l_ms","msm_h_ms","witness_map_ms","total_ms")}, round(d["roofline"]["valu_bound"]["achieved_Tmad_s"],2), round(d["roofline"]["valu_bound"]["g2"]["achieved_Tmad_s"],2))
PY
