set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
tail -30 gpurun_out/r2a/pytest.log
python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench.json 2> gpurun_out/r2a/bench.err; echo "bench rc=$?"
python bench.py --steps 3 --warmup 1 --cpu-log2 22 > gpurun_out/r2a/bench_cpu22.json 2> gpurun_out/r2a/bench_cpu22.err; echo "bench22 rc=$?"
python bench.py --sim-shards 8 --log2 24 --steps 5 --warmup 2 > gpurun_out/r2a/sim8_k24.json 2> gpurun_out/r2a/sim8_k24.err; echo "sim24 rc=$?"
python bench.py --sim-shards 8 --log2 22 --steps 5 --warmup 2 > gpurun_out/r2a/sim8_k22.json 2> gpurun_out/r2a/sim8_k22.err; echo "sim22 rc=$?"
tail -c 600 gpurun_out/r2a/sim8_k24.err
cat gpurun_out/r2a/sim8_k24.json | head -c 1500
