#!/bin/bash
# A/B on ONE box: the per-rank share of an 8-way sharded proof (bench.py --sim-shards 8 --log2 22) under the bucket-pass segment
# length (G16_MSM_SEGMENT) and the window size of the key's tables (G16_MSM_PRECOMP_WINDOW).  usage: ab_shards.sh <tag>
O=gpurun_out/$1; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --sim-shards 8 --log2 22 --steps 8 --warmup 3 > $O/sim8_$name.json 2> $O/sim8_$name.err
  python - $O/sim8_$name.json $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases"]
print(f"{sys.argv[2]:14s} partial {d['partial_ms']:.2f} finalize {d['finalize_ms']:.2f} buckets {[round(x, 2) for x in p['bucket_ms']]} c={p['window_bits']:.0f} W={p['windows']:.0f} finish {p['finish_ms']:.2f}")
PY
}
run default G16_NOOP=1
run seg32 G16_MSM_SEGMENT=32
run seg64 G16_MSM_SEGMENT=64
run c18 G16_MSM_PRECOMP_WINDOW=18
run c18_seg32 G16_MSM_PRECOMP_WINDOW=18 G16_MSM_SEGMENT=32
run c18_seg64 G16_MSM_PRECOMP_WINDOW=18 G16_MSM_SEGMENT=64
run c20_seg64 G16_MSM_PRECOMP_WINDOW=20 G16_MSM_SEGMENT=64
run default2 G16_NOOP=1
