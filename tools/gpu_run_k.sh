set -x
O=gpurun_out/r2k
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "msm or proof_valid or bucket_schemes or golden or half_identity or sharded" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
timeout 300 python bench.py --sim-shards 8 --log2 22 --steps 6 --warmup 2 > $O/sim8_k22.json 2> $O/sim8_k22.err; echo "sim rc=$?"
timeout 400 python bench.py --sim-shards 8 --log2 24 --steps 4 --warmup 2 > $O/sim8_k24.json 2> $O/sim8_k24.err; echo "sim24 rc=$?"
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --curve bn254 --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_bn254.json 2> $O/bench_bn254.err; echo "bench rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2k/sim8_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); p=d["phases"]
    print(f, "partial", round(d["partial_ms"],2), "fin", round(d["finalize_ms"],2), "passes", round(p["bucket_pass_ms"],2), "g2span", round(p["msm_b_g2_ms"],2), "hspan", round(p["msm_h_ms"],2), "W", p["windows"], [round(x,2) for x in p["bucket_ms"]])
for f in ("bench","bench_bn254"):
    d=json.loads([l for l in open(f"gpurun_out/r2k/{f}.json") if l.startswith("{")][-1]); print(f, round(d["ms_per_step"],2), d["roofline"]["avg_launch_ms"], d["roofline"]["g2_bucket_avg_ms"], d["phases_ms_per_step"])
PY
