set -x
mkdir -p gpurun_out/r2e
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_dist_wm.py -m gpu -x -q > gpurun_out/r2e/dist_wm.log 2>&1; echo "dist_wm rc=$?" >> gpurun_out/r2e/dist_wm.log
tail -15 gpurun_out/r2e/dist_wm.log
for v in relax fullnorm; do
  if [ $v = fullnorm ]; then export G16_LIB=$PWD/groth16_amd/libg16_fullnorm.so; fi
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r2e/bench_$v.json 2> gpurun_out/r2e/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2e/bench_$v.json") if l.startswith("{")][-1])
print("$v", round(d["ms_per_step"],2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"])
PY
done
unset G16_LIB
for k in 22 24; do for dw in 1 0; do
  G16_BENCH_DIST_WM=$dw timeout 600 python bench.py --sim-shards 8 --log2 $k --steps 5 --warmup 2 > gpurun_out/r2e/sim8_k${k}_dwm$dw.json 2> gpurun_out/r2e/sim8_k${k}_dwm$dw.err; echo "sim k=$k dwm=$dw rc=$?"
  tail -c 1200 gpurun_out/r2e/sim8_k${k}_dwm$dw.json
done; done
timeout 900 python -X faulthandler -m pytest tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -s > gpurun_out/r2e/suite_s.log 2>&1; echo "suite rc=$?" >> gpurun_out/r2e/suite_s.log
grep -v "^  File" gpurun_out/r2e/suite_s.log | tail -30
