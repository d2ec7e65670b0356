#!/bin/bash
# Round 6 (one box): the whole GPU tier on the tree with fused subtractions, init-free DPP moves, component broadcasts and the
# sign-tracking mixed addition; then main vs libg16_pre.so (assembly products only) vs libg16_nofips.so (round 5's arithmetic)
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
timeout 2400 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
single new_a G16_NOOP=1
single pre_a G16_LIB=$PWD/groth16_amd/libg16_pre.so
single old_a G16_LIB=$PWD/groth16_amd/libg16_nofips.so
single new_b G16_NOOP=1
single pre_b G16_LIB=$PWD/groth16_amd/libg16_pre.so
single old_b G16_LIB=$PWD/groth16_amd/libg16_nofips.so
shard bucket_new bucket G16_NOOP=1
shard bucket_old bucket G16_LIB=$PWD/groth16_amd/libg16_nofips.so
python - $1 <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/single_*.json" % sys.argv[1])):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); p=d["phases_ms_per_step"]; print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in p.items() if k in("msm_b_g2_ms","msm_l_ms","msm_h_ms","witness_map_ms","total_ms")}, round(d["roofline"]["valu_bound"]["achieved_Tmad_s"],2), round(d["roofline"]["valu_bound"]["g2"]["achieved_Tmad_s"],2))
PY
