set -x
O=gpurun_out/r2l
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -v "^  File" $O/pytest.log | tail -20
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 600 $O/bench_default.json
