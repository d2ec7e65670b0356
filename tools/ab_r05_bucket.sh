#!/bin/bash
# Round 5 A/B on ONE box: the per-rank share of an 8-way sharded 2^22 proof under the two cuts (base ranges / bucket space), with the
# G1 passes batched into one launch or not, the G2 pass on its own queue or not, and forced segment lengths; then the single-GPU
# proof under the same knobs.  usage: ab_r05_bucket.sh <tag> [quick]
O=gpurun_out/$1; mkdir -p $O
shard() {  # name mode env...
  name=$1; mode=$2; shift 2
  env "$@" timeout 300 python bench.py --sim-shards ${SH:-8} --shard-mode $mode --log2 ${K:-22} --steps 10 --warmup 3 > $O/sim_${name}.json 2> $O/sim_${name}.err
  python - $O/sim_${name}.json "$name" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases"]
print(f"sim {sys.argv[2]:26s} partial {d['partial_ms']:6.2f} finalize {d['finalize_ms']:.2f} passes {p['bucket_pass_ms']:.2f} buckets {[round(x, 2) for x in p['bucket_ms']]} c={int(p['window_bits'])} launches={int(p.get('g1_pass_launches', 0))}")
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
d = json.loads(l[-1]); p = d["phases_ms_per_step"]; r = d["roofline"]
print(f"single {sys.argv[2]:14s} {d['ms_per_step']:.2f} ms  G1 {r['avg_launch_ms']:.3f} x{r['launches_per_step']}  G2 {r['g2_bucket_avg_ms']:.2f}  passes {p['bucket_pass_ms']:.2f}  wm {p['witness_map_ms']:.2f}  8d {d.get('value_incl_h2d', {}).get('pinned', {}).get('ms_per_step', 0):.2f}  peak {r['valu_bound']['measured_peak_Tmad_s']:.2f} T")
PY
}
shard bucket_default bucket G16_NOOP=1
shard bucket_concurrent bucket G16_PASS_CONCURRENT=1
shard base_default base G16_NOOP=1
shard base_concurrent base G16_PASS_CONCURRENT=1
if [ "${2:-}" = sweep ]; then
  shard bucket_nobatch bucket G16_PASS_NO_BATCH=1
  for L in ${LS:-16 24 48 64}; do shard bucket_L$L bucket G16_MSM_SEGMENT=$L; done
  shard base_nobatch base G16_PASS_NO_BATCH=1
fi
if [ "${2:-}" != quick ]; then
  single default G16_NOOP=1
  single nobatch G16_PASS_NO_BATCH=1
  single concurrent G16_PASS_CONCURRENT=1
  single default2 G16_NOOP=1
fi
