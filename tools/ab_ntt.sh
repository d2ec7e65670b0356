#!/bin/bash
# A/B on ONE box: NTT tile size / radix / occupancy variants (libg16_ntt<X>.so built with -DG16_NTT_TILE_LOG / _THREADS / _MAX_R /
# _MIN_WAVES, see ntt.hip) against the shipped library: parity of the NTT / witness-map tests, then ntt_ms of the full proof at 2^22
# and 2^20 and the 8-way shard.  usage: ab_ntt.sh <tag> <variant letters...>
O=gpurun_out/$1; shift; mkdir -p $O
for v in main "$@" main; do
  if [ $v = main ]; then unset G16_LIB; else export G16_LIB=$PWD/groth16_amd/libg16_ntt$v.so; fi
  if [ $v != main ]; then
    timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dist_wm.py -m gpu -x -q -k "ntt or witness_map or distributed_witness_map" > $O/pytest_ntt$v.log 2>&1
    echo "variant $v parity: $(tail -1 $O/pytest_ntt$v.log)"
  fi
  for k in 22 20; do
    timeout 300 python bench.py --log2 $k --steps 6 --warmup 2 --no-cpu-baseline > $O/ntt_${v}_k$k.json 2> $O/ntt_${v}_k$k.err
    python - $O/ntt_${v}_k$k.json $v $k <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print(sys.argv[2], "FAILED"); sys.exit()
d = json.loads(l[-1]); p = d["phases_ms_per_step"]
print(f"ntt {sys.argv[2]:5s} k={sys.argv[3]} ntt_ms {p['ntt_ms']:.3f} witness_map {p['witness_map_ms']:.3f} proof {d['ms_per_step']:.2f} peak {d['roofline']['valu_bound']['measured_peak_Tmad_s']:.2f}")
PY
  done
  timeout 300 python bench.py --sim-shards 8 --log2 22 --steps 8 --warmup 3 > $O/ntt_${v}_sim8.json 2> $O/ntt_${v}_sim8.err
  python - $O/ntt_${v}_sim8.json $v <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if l:
    d = json.loads(l[-1]); print(f"ntt {sys.argv[2]:5s} sim8 partial {d['partial_ms']:.2f}")
PY
done
