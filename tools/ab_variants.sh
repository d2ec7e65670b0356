#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh): bench.py at 2^22 for each name -- proof time, mean G1 / G2 bucket-pass
# time, NTT time, the box's measured multiply-add peak, and the proof's hash (must be identical across variants).
#   usage: ab_variants.sh <tag> <variant|main> ...      (main = the shipped library; list it first and last to see drift)
O=gpurun_out/$1; shift; mkdir -p $O
i=0
for v in "$@"; do
  i=$((i+1))
  if [ $v = main ]; then unset G16_LIB; else export G16_LIB=$PWD/groth16_amd/libg16_$v.so; fi
  # parity: every variant must produce the SAME proof bytes over the same key / witness / r, s (hash in the JSON line); the full
  # parity tier is run on the variant that is kept (a per-variant pytest subset cost 140 s of box time each in round 4)
  par=""
  G16_BENCH_PRINT_PROOF=1 timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline $AB_BENCH_ARGS > $O/ab_${i}_$v.json 2> $O/ab_${i}_$v.err
  python - $O/ab_${i}_$v.json $v "$par" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "BENCH FAILED |", sys.argv[3]); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]; r = d["roofline"]
print(f"{sys.argv[2]:8s} proof {d['ms_per_step']:.2f} ms  G1 pass {r['avg_launch_ms']:.3f}  G2 pass {r['g2_bucket_avg_ms']:.3f}  ntt {p['ntt_ms']:.3f}  wm {p['witness_map_ms']:.3f}  "
      f"pipelined {d.get('pipelined', {}).get('ms_per_proof', 0):.2f}  finish {p['finish_ms']:.2f}  peak {r['valu_bound']['measured_peak_Tmad_s']:.2f} T/s  frac {r['valu_bound']['frac']:.3f} | proof {d.get('proof_sha256', '?')[:12]}")
PY
done
