#!/bin/bash
# Round 6 follow-up (one box, one call): the MFMA glue probe, the calibrated PMC traffic of the tree (tools/bin/calib inside the same
# --pmc passes) and the default bench line reading it.   gpurun -- "G16_GIT_COMMIT=<sha> bash tools/r06_head_check.sh r06_head"
set -u
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-pipelined --no-projection"
if [ "${2:-}" = "full" ]; then   # the GPU tier and smoke() first (the round's LAST tree)
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -9 $O/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
else
  timeout 120 tools/bin/probe_mfma > $O/probe_mfma.txt 2>&1; cat $O/probe_mfma.txt
fi
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $O/pmc_$c
  timeout 200 rocprofv3 --pmc $c -d $O/pmc_$c/calib -o pmc --output-format csv -- tools/bin/calib > $O/calib_$c.jsonl 2> $O/calib_$c.err
  timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c/bench -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 $NOX > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
  echo "pmc $c rc=$?"
done
python tools/pmc_summary.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/calib_FETCH_SIZE.jsonl > $O/pmc_traffic_calibrated.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python tools/make_pmc_traffic.py $O/pmc_traffic_calibrated.json bls12_381 22 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bin/calib and of bench.py --steps 2 --warmup 1, tools/r06_head_check.sh" > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json
python - $O/pmc_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("G1 launch", round(d["hbm_bytes_per_launch"] / 1e9, 2), "GB  G2", round(d["g2_bucket_pass"]["hbm_bytes_per_launch"] / 1e9, 2), "GB  NTT/proof", round(d["ntt_hbm_bytes_per_step"] / 1e9, 2), "GB  tree", d["kernel_source_sha16"])
PY
timeout 600 python bench.py > $O/bench_k22_cpu_k22.json 2> $O/bench_k22_cpu_k22.err; echo "bench rc=$?"
python - $O/bench_k22_cpu_k22.json <<'PY'
import json, sys
d = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
r = d["roofline"]
print({k: d[k] for k in ("value", "ms_per_step", "value_survey_8d")}, "traffic", r["traffic"], r["traffic_calibration"])
print("G1 launch", r["avg_launch_ms"], "x", r["launches_per_step"], "frac", r["frac"], "valu", r["valu_bound"]["frac"], "peak", r["valu_bound"]["measured_peak_Tmad_s"])
print([(q["shard_mode"], q["n_gpus"], q["rank_share_ms"], q["projected_speedup"]) for q in d["projected_scaling"]["points"]])
PY
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 $NOX > $O/bench_stats.json 2> $O/bench_stats.err; echo "stats rc=$?"
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_k22_kernel_stats.csv; rm -rf $O/prof_stats; head -4 $O/bench_k22_kernel_stats.csv | cut -c1-200
