#!/bin/bash
# A/B on ONE box: buckets per lane of the bucket reduction (G16_MSM_REDUCE_G) for the full proof at 2^22.  usage: ab_reduce_g.sh <tag>
O=gpurun_out/$1; mkdir -p $O
for v in 8 plan 8 plan 32; do
  if [ $v = plan ]; then unset G16_MSM_REDUCE_G; else export G16_MSM_REDUCE_G=$v; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/redg_$v.json 2> $O/redg_$v.err
  python - $O/redg_$v.json $v <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]
print(f"reduce_G {sys.argv[2]:5s} proof {d['ms_per_step']:.2f} ms  passes {p['bucket_pass_ms']:.2f}  finish {p['finish_ms']:.2f}  h span {p['msm_h_ms']:.2f}")
PY
done
