set -x
mkdir -p gpurun_out/r2f
export TMPDIR=/tmp
for v in main nomulsub pair1 main; do
  if [ $v = main ]; then unset G16_LIB; else export G16_LIB=$PWD/groth16_amd/libg16_$v.so; fi
  timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r2f/bench_$v.json 2> gpurun_out/r2f/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2f/bench_$v.json") if l.startswith("{")][-1])
print("$v", round(d["ms_per_step"],2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"])
PY
done
unset G16_LIB
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_external_kats.py -m gpu -x -q -s > gpurun_out/r2f/parity_s.log 2>&1; echo "parity rc=$?" >> gpurun_out/r2f/parity_s.log
grep -v "^  File" gpurun_out/r2f/parity_s.log | tail -25
