set -x
mkdir -p gpurun_out/r2d
export TMPDIR=/tmp
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "multi_device" > gpurun_out/r2d/multi.log 2>&1; echo "multi rc=$?" >> gpurun_out/r2d/multi.log
grep -v "^  File" gpurun_out/r2d/multi.log | tail -30
timeout 1200 python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_gpu_parity.py::test_multi_device_context_single_call > gpurun_out/r2d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2d/pytest.log
tail -25 gpurun_out/r2d/pytest.log
for v in mulsub nomulsub; do
  if [ $v = nomulsub ]; then export G16_LIB=$PWD/groth16_amd/libg16_nomulsub.so; fi
  G16_MSM_AFFINE_LEVELS=0 timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r2d/bench_$v.json 2> gpurun_out/r2d/bench_$v.err; echo "bench $v rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2d/bench_$v.json") if l.startswith("{")][-1])
print("$v", round(d["ms_per_step"],2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"])
PY
done
