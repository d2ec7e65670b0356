#!/bin/bash
# Round 6 A/B (one box): the lane-pair (G2) bucket kernel held to 168 registers = three waves per SIMD (scratch only in the cold
# doubling branch), with the fused four-sweep Y3 (w3f) and with Y3 as two two-sweep products (w3s), against the main build (185 / two waves)
O=gpurun_out/$1; mkdir -p $O
source tools/ab_lib.sh
single main_a G16_NOOP=1
single w3f_a G16_LIB=$PWD/groth16_amd/libg16_w3f.so
single w3s_a G16_LIB=$PWD/groth16_amd/libg16_w3s.so
single main_b G16_NOOP=1
single w3f_b G16_LIB=$PWD/groth16_amd/libg16_w3f.so
single w3s_b G16_LIB=$PWD/groth16_amd/libg16_w3s.so
python - $1 <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/single_*.json" % sys.argv[1])):
    l=[x for x in open(f) if x.startswith("{")]
    if l:
        d=json.loads(l[-1]); p=d["phases_ms_per_step"]; print(f.split('/')[-1], {k:(round(v,2) if isinstance(v,float) else v) for k,v in p.items() if k in("msm_b_g2_ms","msm_l_ms","msm_h_ms","witness_map_ms","total_ms")}, round(d["roofline"]["valu_bound"]["achieved_Tmad_s"],2), round(d["roofline"]["valu_bound"]["g2"]["achieved_Tmad_s"],2))
PY
