set -x
mkdir -p gpurun_out/r2g
export TMPDIR=/tmp
G16_DEBUG=1 timeout 150 python tools/gpu_debug_multi.py 2 > gpurun_out/r2g/multi2.log 2>&1; echo "multi2 rc=$?" >> gpurun_out/r2g/multi2.log
grep -v "^  File" gpurun_out/r2g/multi2.log | tail -40
timeout 400 python -m pytest tests/test_gpu_dist_wm.py -m gpu -x -q --timeout 120 > gpurun_out/r2g/dist_wm.log 2>&1; echo "dist_wm rc=$?" >> gpurun_out/r2g/dist_wm.log
tail -8 gpurun_out/r2g/dist_wm.log
