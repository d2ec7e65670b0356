set -x
mkdir -p gpurun_out/r2b
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -x -q -k "msm or bucket_schemes or full_size or bench_circuit" > gpurun_out/r2b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2b/pytest.log
tail -12 gpurun_out/r2b/pytest.log
for lv in 0 1 2 3 4; do
  G16_MSM_AFFINE_LEVELS=$lv timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2b/bench_aff$lv.json 2> gpurun_out/r2b/bench_aff$lv.err; echo "bench aff$lv rc=$?"
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/r2b/bench_aff$lv.json") if l.startswith("{")][-1])
print("aff$lv", round(d["ms_per_step"],2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"])
PY
done
