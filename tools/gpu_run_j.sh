set -x
O=gpurun_out/r2j
mkdir -p $O
export TMPDIR=/tmp
for c in 16 17 18 19; do
  G16_MSM_PRECOMP_WINDOW=$c timeout 300 python bench.py --sim-shards 8 --log2 22 --steps 6 --warmup 2 > $O/sim8_k22_c$c.json 2> $O/sim8_k22_c$c.err; echo "c=$c rc=$?"
done
for c in 18 19; do
  G16_MSM_PRECOMP_WINDOW=$c timeout 400 python bench.py --sim-shards 8 --log2 24 --steps 4 --warmup 2 > $O/sim8_k24_c$c.json 2> $O/sim8_k24_c$c.err; echo "c=$c rc=$?"
done
timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $O/bench_fixedbase.json 2> $O/bench_fixedbase.err; echo "bench rc=$?"
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2j/sim8_*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1]); p=d["phases"]
    print(f, "partial", round(d["partial_ms"],2), "fin", round(d["finalize_ms"],2), "passes", round(p["bucket_pass_ms"],2), "g2span", round(p["msm_b_g2_ms"],2), "hspan", round(p["msm_h_ms"],2), "W", p["windows"], [round(x,2) for x in p["bucket_ms"]])
d=json.loads([l for l in open("gpurun_out/r2j/bench_fixedbase.json") if l.startswith("{")][-1]); print("bench", round(d["ms_per_step"],2), d["phases_ms_per_step"])
PY
