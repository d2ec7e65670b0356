#!/bin/bash
# Round 5 A/B (one box): h in the G1 batch or after it, G2 on its own queue or not -- sharded share and single proof
O=gpurun_out/$1; mkdir -p $O
shard() {
  name=$1; mode=$2; shift 2
  env "$@" timeout 300 python bench.py --sim-shards 8 --shard-mode $mode --log2 22 --steps 10 --warmup 3 > $O/sim_${name}.json 2> $O/sim_${name}.err
  python - $O/sim_${name}.json "$name" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases"]
print(f"sim {sys.argv[2]:26s} partial {d['partial_ms']:6.2f} finalize {d['finalize_ms']:.2f} passes {p['bucket_pass_ms']:.2f} wm {p['witness_map_ms']:.2f} prep {p['scalar_prep_ms']:.2f} buckets {[round(x, 2) for x in p['bucket_ms']]}")
PY
}
single() {
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-projection > $O/single_$name.json 2> $O/single_$name.err
  python - $O/single_$name.json $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]; r = d["roofline"]; h = d.get("value_incl_h2d") or {}
print(f"single {sys.argv[2]:14s} {d['ms_per_step']:.2f} ms  passes {p['bucket_pass_ms']:.2f}  wm {p['witness_map_ms']:.2f} prep {p['scalar_prep_ms']:.2f}  8d pinned {h.get('pinned', {}).get('ms_per_step', 0):.2f} pageable {h.get('pageable', {}).get('ms_per_step', 0):.2f}  peak {r['valu_bound']['measured_peak_Tmad_s']:.2f} T")
PY
}
shard bucket_default bucket G16_NOOP=1
shard bucket_h_in_batch bucket G16_PASS_H_IN_BATCH=1
shard bucket_noconc bucket G16_PASS_CONCURRENT=0
shard bucket_noconc_hin bucket G16_PASS_CONCURRENT=0 G16_PASS_H_IN_BATCH=1
shard base_default base G16_NOOP=1
single default G16_NOOP=1
single concurrent G16_PASS_CONCURRENT=1
single one_copy G16_UPLOAD_CHUNKED=0
single default2 G16_NOOP=1
