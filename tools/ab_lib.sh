# shared by the round-5 A/B scripts (sourced; $O = output directory): one line per run
shard() {  # name mode env...   -- one rank's share of an 8-way sharded 2^22 proof
  name=$1; mode=$2; shift 2
  env "$@" timeout 300 python bench.py --sim-shards ${SH:-8} --shard-mode $mode --log2 ${K:-22} --steps 10 --warmup 3 > $O/sim_${name}.json 2> $O/sim_${name}.err
  python - $O/sim_${name}.json "$name" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases"]
print(f"sim {sys.argv[2]:26s} partial {d['partial_ms']:6.2f} finalize {d['finalize_ms']:.2f} passes {p['bucket_pass_ms']:.2f} wm {p['witness_map_ms']:.2f} prep {p['scalar_prep_ms']:.2f} buckets {[round(x, 2) for x in p['bucket_ms']]}")
PY
}
single() {  # name env...   -- the whole 2^22 proof on one GPU
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --no-projection > $O/single_$name.json 2> $O/single_$name.err
  python - $O/single_$name.json $name <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]; r = d["roofline"]; h = d.get("value_incl_h2d") or {}
print(f"single {sys.argv[2]:14s} {d['ms_per_step']:.2f} ms  passes {p['bucket_pass_ms']:.2f}  wm {p['witness_map_ms']:.2f}  8d pinned {h.get('pinned', {}).get('ms_per_step', 0):.2f} pageable {h.get('pageable', {}).get('ms_per_step', 0):.2f}  peak {r['valu_bound']['measured_peak_Tmad_s']:.2f} T")
PY
}
