#!/bin/bash
# Round 6 evidence set, one box, one gpurun call: every number DESIGN.md / BASELINE.md quote for the round's last tree.
#   usage: gpurun --timeout 1500 -- "G16_GIT_COMMIT=$(git rev-parse --short HEAD) bash tools/r06_evidence.sh r06_final"
#   (outputs under gpurun_out/<tag>/, copied to profiles/r06_*; the commit id travels in the environment: the GPU box has no .git)
set -u
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
NOX="--no-cpu-baseline --no-pipelined --no-projection"
last() { python - "$1" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if l:
    d = json.loads(l[-1])
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k in ("value", "ms_per_step", "partial_ms", "finalize_ms", "value_survey_8d", "shard_mode")})
PY
}
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 $NOX > $O/bench_stats.json 2> $O/bench_stats.err; echo "stats rc=$?"
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_k22_kernel_stats.csv; rm -rf $O/prof_stats; head -8 $O/bench_k22_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $O/pmc_$c
  timeout 200 rocprofv3 --pmc $c -d $O/pmc_$c/calib -o pmc --output-format csv -- tools/bin/calib > $O/calib_$c.jsonl 2> $O/calib_$c.err
  timeout 400 rocprofv3 --pmc $c -d $O/pmc_$c/bench -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 $NOX > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
  echo "pmc $c rc=$?"
done
python tools/pmc_summary.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/calib_FETCH_SIZE.jsonl > $O/pmc_traffic_calibrated.json
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
python tools/make_pmc_traffic.py $O/pmc_traffic_calibrated.json bls12_381 22 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bin/calib and of bench.py --steps 2 --warmup 1, tools/r06_evidence.sh" > $O/pmc_traffic.json
cp $O/pmc_traffic.json profiles/pmc_traffic.json   # (on the box: so that the bench below reads the counters of ITS tree)
timeout 600 python bench.py > $O/bench_k22_cpu_k22.json 2> $O/bench_k22_cpu_k22.err; echo "bench rc=$?"; last $O/bench_k22_cpu_k22.json
python - $O/pmc_traffic.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("G1 launch", round(d["hbm_bytes_per_launch"] / 1e9, 2), "GB  G2", round(d["g2_bucket_pass"]["hbm_bytes_per_launch"] / 1e9, 2), "GB  NTT/proof", round(d["ntt_hbm_bytes_per_step"] / 1e9, 2), "GB  tree", d["kernel_source_sha16"])
PY
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 $NOX > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err; echo "pmcsq rc=$?"
python - $O <<'PY'
import json, subprocess, sys
O = sys.argv[1]
names = ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY")
d = {c: json.loads(subprocess.run([sys.executable, "tools/pmc_summary.py", "raw", O + "/pmc_sq", c], capture_output=True, text=True).stdout or "{}") for c in names}
out = {}
for k in d["SQ_WAVE_CYCLES"]:
    out[k] = {c: d[c].get(k, {}).get("avg_counter") for c in names}
    out[k]["launches"] = d["SQ_WAVE_CYCLES"][k]["launches"]
json.dump(out, open(O + "/pmc_sq.json", "w"), indent=1)
for k, v in out.items():
    if "bucket_accumulate30" in k or "ntt30" in k:
        print(k[:70], {c[3:]: v[c] for c in names})
PY
rm -rf $O/pmc_sq
timeout 300 rocprofv3 --kernel-trace -d $O/prof_trace -o tr --output-format csv -- python bench.py --steps 2 --warmup 2 $NOX > $O/trace_single.json 2> $O/trace_single.err; echo "trace rc=$?"
find $O/prof_trace -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_single_k22.csv; rm -rf $O/prof_trace
python tools/trace_timeline.py $O/kernel_trace_single_k22.csv > $O/timeline_single_k22.txt
for m in bucket base; do
  timeout 300 python bench.py --sim-shards 8 --shard-mode $m --log2 22 --steps 10 --warmup 3 > $O/sim_shards8_k22_$m.json 2> $O/sim_shards8_k22_$m.err; echo "sim8 22 $m rc=$?"; last $O/sim_shards8_k22_$m.json
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sim -o st --output-format csv -- python bench.py --sim-shards 8 --shard-mode bucket --log2 22 --steps 4 --warmup 2 > $O/trace_sim8.json 2> $O/trace_sim8.err
find $O/prof_sim -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_sim_shards8_k22_bucket.csv
find $O/prof_sim -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sim_shards8_k22_bucket_kernel_stats.csv; rm -rf $O/prof_sim
python tools/trace_timeline.py $O/kernel_trace_sim_shards8_k22_bucket.csv > $O/timeline_sim_shards8_k22_bucket.txt
timeout 300 python bench.py --curve bn254 --no-cpu-baseline > $O/bench_bn254_k22.json 2> $O/bench_bn254_k22.err; echo "bn254 rc=$?"; last $O/bench_bn254_k22.json
timeout 500 python bench.py --log2 24 --no-cpu-baseline --steps 4 > $O/bench_k24_single_gpu.json 2> $O/bench_k24_single_gpu.err; echo "k24 rc=$?"; last $O/bench_k24_single_gpu.json
for m in bucket base; do
  timeout 400 python bench.py --sim-shards 8 --shard-mode $m --log2 24 --steps 5 --warmup 2 > $O/sim_shards8_k24_$m.json 2> $O/sim_shards8_k24_$m.err; echo "sim8 24 $m rc=$?"; last $O/sim_shards8_k24_$m.json
done
timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bucket_space" > $O/pytest_k24_bucket.log 2>&1; echo "k24 bucket test rc=$?"; tail -2 $O/pytest_k24_bucket.log
