set -x
O=gpurun_out/r2i
mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_circuit_api.py tests/test_external_kats.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "pytest_new rc=$?" >> $O/pytest_new.log
tail -5 $O/pytest_new.log
timeout 400 python bench.py --steps 10 --warmup 3 > $O/bench_k22.json 2> $O/bench_k22.err; echo "bench rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o st --output-format csv -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_k22_prof.json 2> $O/bench_k22_prof.err; echo "prof rc=$?"
find $O/prof_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_k22_kernel_stats.csv
find $O/prof_stats -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $c -d $O/pmc_$c -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err; echo "pmc $c rc=$?"
  python tools/pmc_summary.py $O/pmc_$c $c > $O/pmc_$c.json
  rm -rf $O/pmc_$c
done
for k in 22 24; do
  timeout 600 python bench.py --sim-shards 8 --log2 $k --steps 5 --warmup 2 > $O/sim8_k${k}.json 2> $O/sim8_k${k}.err; echo "sim k=$k rc=$?"
  tail -c 900 $O/sim8_k${k}.json
done
timeout 600 python bench.py --log2 24 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_k24_single.json 2> $O/bench_k24_single.err; echo "k24 rc=$?"
timeout 300 python bench.py --curve bn254 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_bn254_k22.json 2> $O/bench_bn254_k22.err; echo "bn254 rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2i/bench_*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(f, round(d["ms_per_step"],2), round(d["value"]/1e6,2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"].get("ntt_ms"), (d.get("cpu_baseline") or {}).get("seconds"))
    except Exception as e: print(f, "ERR", e)
PY
