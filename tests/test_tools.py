"""The evidence tooling (tools/pmc_summary.py, tools/make_pmc_traffic.py, tools/trace_timeline.py) on small synthetic rocprofv3
outputs: the numbers under profiles/ are only as good as these scripts' arithmetic."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(*args):
    r = subprocess.run([sys.executable] + list(args), capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    return r.stdout


def counter_csv(path, rows):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write('"Kernel_Name","Counter_Name","Counter_Value"\n')
        for k, c, v in rows:
            f.write(f'"{k}","{c}",{v}\n')


def test_pmc_traffic_calibration(tmp_path):
    G1 = "void g16::bucket_accumulate30_kernel<g16::Fp30<g16::Bls12_381FqP>, false>(int)"
    G2 = "void g16::bucket_accumulate30_kernel<g16::Fp2p30<g16::Bls12_381FqP>, false>(int)"
    NTT = "void g16::ntt30_pass_kernel<g16::Bls12_381FrP, true>(int)"
    QUO = "void g16::bitrev_scale_kernel<g16::Bls12_381FrP>(int)"   # one launch per proof: how make_pmc_traffic.py counts the proofs
    # counters are in KiB; the coalesced-read calibration kernel reports half of its true bytes
    fetch = [("calib_read16(V16 const*, unsigned long, unsigned int*)", "FETCH_SIZE", 1000), ("calib_read32(R32 const*)", "FETCH_SIZE", 500),
             ("calib_gather96(R96 const*)", "FETCH_SIZE", 2000), ("calib_gather192p(R48 const*)", "FETCH_SIZE", 1000),
             (G1, "FETCH_SIZE", 4000), (G1, "FETCH_SIZE", 6000), (G2, "FETCH_SIZE", 3000), (NTT, "FETCH_SIZE", 100), (NTT, "FETCH_SIZE", 100),
             (QUO, "FETCH_SIZE", 10)]
    write = [("calib_write32(R32*)", "WRITE_SIZE", 1000), ("calib_write208(R208*)", "WRITE_SIZE", 2000), (G1, "WRITE_SIZE", 100),
             (G1, "WRITE_SIZE", 300), (G2, "WRITE_SIZE", 50), (NTT, "WRITE_SIZE", 100), (NTT, "WRITE_SIZE", 100), (QUO, "WRITE_SIZE", 10)]
    counter_csv(str(tmp_path / "f" / "x_counter_collection.csv"), fetch)
    counter_csv(str(tmp_path / "w" / "x_counter_collection.csv"), write)
    truth = tmp_path / "calib.jsonl"
    K = 1024
    truth.write_text("\n".join(json.dumps(d) for d in [
        {"kernel": "calib_read16", "read_bytes": 2000 * K, "write_bytes": 0, "ms": 1, "GBps": 1},
        {"kernel": "calib_read32", "read_bytes": 1000 * K, "write_bytes": 0, "ms": 1, "GBps": 1},
        {"kernel": "calib_write32", "read_bytes": 0, "write_bytes": 1000 * K, "ms": 1, "GBps": 1},
        {"kernel": "calib_gather96", "read_bytes": 2000 * K, "write_bytes": 0, "ms": 1, "GBps": 1},
        {"kernel": "calib_gather192p", "read_bytes": 1500 * K, "write_bytes": 0, "ms": 1, "GBps": 1},
        {"kernel": "calib_write208", "read_bytes": 0, "write_bytes": 1000 * K, "ms": 1, "GBps": 1}]))
    out = json.loads(run("tools/pmc_summary.py", "traffic", str(tmp_path / "f"), str(tmp_path / "w"), str(truth)))
    cal, ker = out["calibration"], out["kernels"]
    assert cal["calib_read16"]["fetch_factor"] == 2.0 and cal["calib_read32"]["fetch_factor"] == 2.0
    assert cal["calib_gather96"]["fetch_factor"] == 1.0 and cal["calib_gather192p"]["fetch_factor"] == 1.5
    assert cal["calib_write32"]["write_factor"] == 1.0 and cal["calib_write208"]["write_factor"] == 0.5
    g1 = next(v for k, v in ker.items() if "Fp30<" in k)
    assert g1["launches"] == 2 and g1["fetch_pattern"] == "calib_gather96" and g1["write_pattern"] == "calib_write208"
    assert g1["hbm_bytes_per_launch"] == 5000 * K * 1.0 + 200 * K * 0.5          # averages of the two launches, per-pattern factors
    g2 = next(v for k, v in ker.items() if "Fp2p30<" in k)
    assert g2["hbm_bytes_per_launch"] == 3000 * K * 1.5 + 50 * K * 0.5
    ntt = next(v for k, v in ker.items() if "ntt30_" in k)
    assert ntt["hbm_bytes_per_launch"] == 100 * K * 2.0 + 100 * K * 1.0          # coalesced 32-byte reads are doubled, writes are not
    assert not any(k.startswith("calib_") for k in ker)
    # bench.py's input
    cj = tmp_path / "cal.json"
    cj.write_text(json.dumps(out))
    pt = json.loads(run("tools/make_pmc_traffic.py", str(cj), "bls12_381", "22", "unit test"))
    assert pt["workload"] == {"curve": "bls12_381", "log2_domain": 22, "n_gpus": 1}
    assert pt["hbm_bytes_per_launch"] == g1["hbm_bytes_per_launch"] and pt["calibration"]["fetch"]["factor"] == 1.0
    assert pt["ntt_hbm_bytes_per_step"] == 2 * ntt["hbm_bytes_per_launch"]        # two launches, one proof (one un-permute launch)
    assert len(pt["kernel_source_sha16"]) == 16   # the tree the counters belong to (tools/tree_hash.py)


def test_trace_timeline(tmp_path):
    p = tmp_path / "kernel_trace.csv"
    rows = ['"Kernel_Name","Start_Timestamp","End_Timestamp","Queue_Id"']
    t = 1_000_000
    for proof in range(2):
        rows.append(f'"void g16::spmv3_kernel<int>(int)",{t},{t + 100_000},1')
        rows.append(f'"void g16::bucket_accumulate30_kernel<g16::Fp30<X>, false>(int)",{t + 150_000},{t + 9_150_000},1')
        rows.append(f'"void g16::bucket_reduce_kernel<g16::Fp30<X> >(int)",{t + 9_200_000},{t + 10_000_000},2')
        t += 12_000_000          # 2 ms of host time between the proofs
    p.write_text("\n".join(rows) + "\n")
    out = run("tools/trace_timeline.py", str(p))
    assert "# 2 proofs in the trace; the last one: 3 launches, 10.000 ms" in out
    lines = [l for l in out.splitlines() if not l.startswith("#")]
    assert lines[0].split()[0] == "0.000" and "spmv3_kernel" in lines[0]
    assert lines[1].split()[0] == "0.150" and "+" in lines[1] and "9000.0" in lines[1]
    assert lines[2].strip().endswith("bucket_reduce_kernel<Fp30<X> >") and "q1" in lines[2]
