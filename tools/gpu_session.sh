#!/bin/bash
# The ONE script behind every measurement under profiles/ (round 3 on): run on the GPU box through
#     gpurun --timeout <s> -- 'bash tools/gpu_session.sh <tag> <step> [<step> ...]'
# Every step writes into gpurun_out/<tag>/; what is kept as evidence is copied from there into profiles/ (see profiles/README.md).
# Steps (each bounded by its own `timeout`, none combines --pmc with a trace domain):
#   tests [pytest args]   python -m pytest tests -m gpu -q <args>            -> pytest.log
#   newtests              the round-3 full-size parity cases only                   -> pytest_new.log
#   fasttests             the GPU tier without the minute-long full-size cases      -> pytest_fast.log
#   bench [bench args]    python bench.py <args>                               -> bench.json / bench.err
#   stats                 rocprofv3 --kernel-trace --stats of bench.py --steps 5     -> kernel_stats.csv
#   pmc                   tools/bin/calib + bench.py --steps 2 inside ONE rocprofv3 --pmc FETCH_SIZE pass and ONE --pmc WRITE_SIZE
#                         pass; tools/pmc_summary.py traffic -> pmc_traffic_calibrated.json
#   pmcsq                 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY of bench.py --steps 2
#   sim8 [log2]           bench.py --sim-shards 8 --log2 <22>; with rocprofv3 kernel stats
#   trace [single|sim8]   rocprofv3 --kernel-trace of a few proofs -> kernel_trace_<what>.csv + the last proof's timeline
#   ubench                tools/bin/ubench_* (instruction issue rates, accumulate probes)
set -u
TAG=$1; shift
O=gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
last_json() { python - "$1" <<'PY'
import json, sys
lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
if lines:
    d = json.loads(lines[-1])
    print({k: d.get(k) for k in ("value", "ms_per_step", "partial_ms", "finalize_ms") if k in d}, d.get("phases_ms_per_step", d.get("phases")))
PY
}
while [ $# -gt 0 ]; do
  step=$1; shift
  case $step in
    tests)
      args=""; while [ $# -gt 0 ] && [[ "$1" == -* || "$1" == tests/* ]]; do args="$args $1"; shift; done
      timeout 1500 python -m pytest tests -m gpu -q -x $args > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log; tail -5 $O/pytest.log ;;
    fasttests)   # everything but the minute-long full-size cases (those run in `tests` / `newtests`)
      timeout 1200 python -m pytest tests -m gpu -q -x --durations=10 -k "not k22 and not configs4 and not 2_24 and not 22-8 and not 20-8" > $O/pytest_fast.log 2>&1
      echo "fasttests rc=$?" | tee -a $O/pytest_fast.log; tail -16 $O/pytest_fast.log ;;
    newtests)
      timeout 1200 python -m pytest -m gpu -q -x --durations=8 \
        "tests/test_gpu_fullsize.py::test_configs4_2_24_eight_ranks_distributed_map_block_h" \
        "tests/test_gpu_fullsize.py::test_full_size_proof_bit_exact" \
        "tests/test_gpu_dist_wm.py::test_distributed_witness_map_matches_oracle" > $O/pytest_new.log 2>&1
      echo "newtests rc=$?" | tee -a $O/pytest_new.log; tail -15 $O/pytest_new.log ;;
    bench)
      args=""; while [ $# -gt 0 ] && [[ "$1" == -* || "$1" =~ ^[0-9]+$ || "$1" == bn254 || "$1" == bls12_381 ]]; do args="$args $1"; shift; done
      name=bench$(echo "$args" | tr -d ' ' | tr -c 'a-zA-Z0-9\n' '_')
      timeout 900 python bench.py $args > $O/$name.json 2> $O/$name.err; echo "bench$args rc=$?"; last_json $O/$name.json ;;
    stats)
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pipelined \
        > $O/bench_stats.json 2> $O/bench_stats.err; echo "stats rc=$?"
      find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv; rm -rf $O/prof_stats
      head -12 $O/kernel_stats.csv ;;
    pmc)
      # the known-bytes kernels and the bench under the SAME counter configuration, back to back on this box (two rocprofv3
      # processes: one output directory each, their CSVs share a file name)
      for c in FETCH_SIZE WRITE_SIZE; do
        mkdir -p $O/pmc_$c
        timeout 200 rocprofv3 --pmc $c -d $O/pmc_$c/calib -o pmc --output-format csv -- tools/bin/calib > $O/calib_$c.jsonl 2> $O/calib_$c.err
        timeout 600 rocprofv3 --pmc $c -d $O/pmc_$c/bench -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined \
          > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
        echo "pmc $c rc=$?"
      done
      python tools/pmc_summary.py traffic $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/calib_FETCH_SIZE.jsonl > $O/pmc_traffic_calibrated.json
      rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
      python tools/make_pmc_traffic.py $O/pmc_traffic_calibrated.json bls12_381 22 "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of tools/bin/calib and of bench.py --steps 2 --warmup 1, tools/gpu_session.sh pmc" > $O/pmc_traffic.json
      python - $O/pmc_traffic_calibrated.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d["calibration"].items():
    print(k, "fetch_factor", v["fetch_factor"], "write_factor", v["write_factor"], "GB/s", v["achieved_GBps"])
for k, v in d["kernels"].items():
    if "bucket_accumulate" in k or "ntt30" in k or "quotient" in k or "spmv" in k:
        print(k[:80], "raw", round(v["fetch_raw_bytes"] / 1e6), round(v["write_raw_bytes"] / 1e6), "MB -> calibrated", round(v["hbm_bytes_per_launch"] / 1e6), "MB")
PY
      ;;
    pmcsq)
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- \
        python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err; echo "pmcsq rc=$?"
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
        print(k[:80], {c: v[c] for c in names})
PY
      rm -rf $O/pmc_sq ;;
    sim8)
      k=22; if [ $# -gt 0 ] && [[ "$1" =~ ^[0-9]+$ ]]; then k=$1; shift; fi
      timeout 600 python bench.py --sim-shards 8 --log2 $k --steps 8 --warmup 3 > $O/sim8_k$k.json 2> $O/sim8_k$k.err; echo "sim8 rc=$?"; last_json $O/sim8_k$k.json
      timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_sim -o st --output-format csv -- python bench.py --sim-shards 8 --log2 $k --steps 8 --warmup 3 \
        > $O/sim8_k${k}_prof.json 2> $O/sim8_k${k}_prof.err
      find $O/prof_sim -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sim8_k${k}_kernel_stats.csv; rm -rf $O/prof_sim ;;
    trace)   # per-kernel start / end timestamps of a few proofs (kernel trace only): tools/trace_timeline.py prints the timeline
      what=single; if [ $# -gt 0 ] && [[ "$1" == sim8 || "$1" == single ]]; then what=$1; shift; fi
      if [ $what = sim8 ]; then cmd="python bench.py --sim-shards 8 --log2 22 --steps 3 --warmup 2"; else cmd="python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-pipelined"; fi
      timeout 600 rocprofv3 --kernel-trace -d $O/prof_trace_$what -o tr --output-format csv -- $cmd > $O/trace_$what.json 2> $O/trace_$what.err; echo "trace $what rc=$?"
      find $O/prof_trace_$what -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} $O/kernel_trace_$what.csv; rm -rf $O/prof_trace_$what
      python tools/trace_timeline.py $O/kernel_trace_$what.csv | tail -120 ;;
    ubench)
      for b in tools/bin/ubench_*; do timeout 200 $b > $O/$(basename $b).txt 2>&1; grep -h "PROBE" $O/$(basename $b).txt | head -12; done ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
