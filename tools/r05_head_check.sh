#!/bin/bash
# Round 5 HEAD check (one box, one call): the GPU tier, smoke(), the default bench line and fresh PMC traffic on the round's LAST tree.
#   gpurun --timeout 1700 -- "G16_GIT_COMMIT=<short sha> bash tools/r05_head_check.sh r05_head"
set -u
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-pipelined --no-projection"
timeout 1300 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $O/pmc_$c
  timeout 200 rocprofv3 --pmc $c -d $O/pmc_$c/calib -o pmc --output-format csv -- tools/bin/calib > $O/calib_$c.jsonl 2> $O/calib_$c.err
  timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c/bench -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 $NOX > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
  echo "pmc $c rc=$?"
done
python tools/pmc_summary.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/calib_FETCH_SIZE.jsonl > $O/pmc_traffic_calibrated.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python tools/make_pmc_traffic.py $O/pmc_traffic_calibrated.json bls12_381 22 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bin/calib and of bench.py --steps 2 --warmup 1, tools/r05_head_check.sh" > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json   # (on the box: so that the bench below reads the counters of ITS tree)
timeout 600 python bench.py > $O/bench_k22_cpu_k22.json 2> $O/bench_k22_cpu_k22.err; echo "bench rc=$?"
python - $O/bench_k22_cpu_k22.json <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "value_survey_8d")}, "traffic", r["traffic"], r["traffic_source"])
print("G1 launch", r["avg_launch_ms"], "x", r["launches_per_step"], "frac", r["frac"], "valu", r["valu_bound"]["frac"], "peak", r["valu_bound"]["measured_peak_Tmad_s"])
print([(q["shard_mode"], q["n_gpus"], q["rank_share_ms"], q["projected_speedup"]) for q in d["projected_scaling"]["points"]])
print("cpu", d["cpu_baseline"].get("seconds"), d["cpu_baseline"].get("gpu_proof_matches_cpu"), "pipelined", d["pipelined"].get("ms_per_proof"))
PY
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 $NOX > $O/bench_stats.json 2> $O/bench_stats.err; echo "stats rc=$?"
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_k22_kernel_stats.csv; rm -rf $O/prof_stats; head -4 $O/bench_k22_kernel_stats.csv | cut -c1-200
