set -x
O=gpurun_out/r2m
mkdir -p $O
export TMPDIR=/tmp
for v in main ntt12 main ntt12; do
  if [ $v = main ]; then unset G16_LIB; else export G16_LIB=$PWD/groth16_amd/libg16_$v.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2m/bench_$v.json") if l.startswith("{")][-1])
print("$v", round(d["ms_per_step"],2), "ntt", d["phases_ms_per_step"]["ntt_ms"], "wm", d["phases_ms_per_step"]["witness_map_ms"])
PY
done
export G16_LIB=$PWD/groth16_amd/libg16_ntt12.so
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_wm.py -m gpu -x -q -k "ntt or witness_map" > $O/pytest_ntt12.log 2>&1; echo "pytest rc=$?" >> $O/pytest_ntt12.log
tail -4 $O/pytest_ntt12.log
G16_BENCH_LOG2=20 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_ntt12_k20.json 2>/dev/null
unset G16_LIB
G16_BENCH_LOG2=20 timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_main_k20.json 2>/dev/null
python - <<'PY'
import json
for v in ("main","ntt12"):
    d=json.loads([l for l in open(f"gpurun_out/r2m/bench_{v}_k20.json") if l.startswith("{")][-1]); print(v,"k20", round(d["ms_per_step"],2), "ntt", d["phases_ms_per_step"]["ntt_ms"])
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_sim -o st --output-format csv -- python bench.py --sim-shards 8 --log2 22 --steps 6 --warmup 2 > $O/sim8_prof.json 2> $O/sim8_prof.err
find $O/prof_sim -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/sim8_k22_kernel_stats.csv
rm -rf $O/prof_sim
