#!/bin/bash
# A/B on ONE box: workgroup size of the bucket pass (libg16_acc<T>.so = msm.hip built with ACC_THREADS = T) against the shipped library.
# usage (on the GPU box): bash tools/ab_acc_threads.sh <T> [<T> ...]
for v in main "$@" main; do
  if [ $v = main ]; then unset G16_LIB; else export G16_LIB=$PWD/groth16_amd/libg16_acc$v.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', round(d['ms_per_step'],2), 'G1', round(r['avg_launch_ms'],3), 'G2', round(r['g2_bucket_avg_ms'],2))"
done
