set -x
mkdir -p gpurun_out/r2h
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --capture=sys --durations=12 > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2h/pytest.log
grep -v "^  File" gpurun_out/r2h/pytest.log | tail -40
