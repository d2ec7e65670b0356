set -x
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q --durations=10 > gpurun_out/r2c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2c/pytest.log
tail -16 gpurun_out/r2c/pytest.log
for lv in 0 3; do
  G16_MSM_AFFINE_LEVELS=$lv timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r2c/prof_aff$lv -o aff$lv --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2c/bench_aff$lv.json 2> gpurun_out/r2c/bench_aff$lv.err; echo "prof aff$lv rc=$?"
  find gpurun_out/r2c/prof_aff$lv -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r2c/aff${lv}_kernel_stats.csv
  find gpurun_out/r2c/prof_aff$lv -name "*kernel_trace.csv" -delete
  head -25 gpurun_out/r2c/aff${lv}_kernel_stats.csv | cut -c1-200
done
