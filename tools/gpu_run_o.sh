set -x
O=gpurun_out/r2o
mkdir -p $O
export TMPDIR=/tmp
timeout 120 tools/bin/ubench_relax > $O/ubench_relax.txt 2>&1
timeout 120 tools/bin/ubench_fullnorm > $O/ubench_fullnorm.txt 2>&1
cat $O/ubench_relax.txt $O/ubench_fullnorm.txt | grep PROBE
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $O/pmc_sq -o pmc --output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_pmc_sq.json 2> $O/bench_pmc_sq.err; echo "pmc rc=$?"
for c in SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY; do python tools/pmc_summary.py $O/pmc_sq $c > $O/pmc_$c.json; done
rm -rf $O/pmc_sq
python - <<'PY'
import json
d={c:json.load(open(f"gpurun_out/r2o/pmc_{c}.json")) for c in ("SQ_WAVE_CYCLES","SQ_ACTIVE_INST_VALU","SQ_WAIT_INST_ANY")}
for k in d["SQ_WAVE_CYCLES"]:
    if "bucket_accumulate30" in k or "ntt30" in k:
        w=d["SQ_WAVE_CYCLES"][k]["avg_counter"]; print(k[:90], "valu/wave_cycles", round(d["SQ_ACTIVE_INST_VALU"][k]["avg_counter"]/w,3), "wait/wave_cycles", round(d["SQ_WAIT_INST_ANY"][k]["avg_counter"]/w,3))
PY
