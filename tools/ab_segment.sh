#!/bin/bash
# A/B on ONE box: bucket-pass segment length (G16_MSM_SEGMENT; unset = the plan's own choice) for the full proof at 2^22 and for
# the per-rank share of an 8-way sharded proof.  usage: ab_segment.sh <tag>
O=gpurun_out/$1; mkdir -p $O
single() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/seg_single_$name.json 2> $O/seg_single_$name.err
  python - $O/seg_single_$name.json $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]; r = d["roofline"]
print(f"single {sys.argv[2]:10s} {d['ms_per_step']:.2f} ms  G1 pass {r['avg_launch_ms']:.3f}  G2 pass {r['g2_bucket_avg_ms']:.2f}  passes {p['bucket_pass_ms']:.2f}  wm {p['witness_map_ms']:.2f}  peak {r['valu_bound']['measured_peak_Tmad_s']:.2f} Tmad/s")
PY
}
shard() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --sim-shards 8 --log2 22 --steps 8 --warmup 3 > $O/seg_sim8_$name.json 2> $O/seg_sim8_$name.err
  python - $O/seg_sim8_$name.json $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases"]
print(f"sim8   {sys.argv[2]:10s} partial {d['partial_ms']:.2f} finalize {d['finalize_ms']:.2f} buckets {[round(x, 2) for x in p['bucket_ms']]}")
PY
}
single plan G16_NOOP=1
single L64 G16_MSM_SEGMENT=64
single L60 G16_MSM_SEGMENT=60
single L84 G16_MSM_SEGMENT=84
single L104 G16_MSM_SEGMENT=104
single L139 G16_MSM_SEGMENT=139
single plan2 G16_NOOP=1
shard plan G16_NOOP=1
shard L16 G16_MSM_SEGMENT=16
shard L19 G16_MSM_SEGMENT=19
shard L28 G16_MSM_SEGMENT=28
shard plan2 G16_NOOP=1
