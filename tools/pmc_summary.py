#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters from the counter_collection CSVs, with the FETCH_SIZE / WRITE_SIZE
calibration MI355X_MICROARCH.md's HBM section asks for.

  pmc_summary.py raw <dir written by rocprofv3 -d> <COUNTER>
        -> JSON {kernel: {launches, avg_counter}} on stdout
  pmc_summary.py traffic <fetch dir> <write dir> <calib.jsonl> [--steps N]
        -> JSON on stdout: calibration factors (true bytes / (counter * 1024)) of the known-bytes kernels of tools/calib.hip,
           which must have run inside the SAME --pmc passes, and raw + calibrated HBM bytes per launch of every other kernel.
           Each product kernel is corrected with the factor of the access pattern it reads / writes with (PATTERN below).

FETCH_SIZE / WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE is known to report 1/2 of a wide coalesced read; other
widths and WRITE_SIZE are uncalibrated -- hence the factors are measured, not assumed."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

# access pattern of each product kernel's dominant reads / writes -> the calibration kernel that reproduces it
PATTERN = {
    "bucket_accumulate30_kernel<Fp30": ("calib_gather96", "calib_write208"),
    "bucket_accumulate30_kernel<Fp2p30": ("calib_gather192p", "calib_write208"),
    "ntt30_": ("calib_read32", "calib_write32"),
    "quotient_kernel": ("calib_read32", "calib_write32"),
    "spmv3_kernel": ("calib_read32", "calib_write32"),
    "bitrev_scale_kernel": ("calib_read32", "calib_write32"),
    "dwm_": ("calib_read32", "calib_write32"),
}
DEFAULT_PATTERN = ("calib_read16", "calib_write32")


def collect(d, counter):
    acc = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = re.sub(r"g16::", "", r["Kernel_Name"])
            name = re.sub(r"\(.*", "", name)
            name = re.sub(r"^void ", "", name)
            a = acc[name]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "avg_counter": v[1] / v[0]} for k, v in acc.items()}


def main():
    mode = sys.argv[1]
    if mode == "raw":
        print(json.dumps(collect(sys.argv[2], sys.argv[3]), indent=1))
        return
    fetch, write = collect(sys.argv[2], "FETCH_SIZE"), collect(sys.argv[3], "WRITE_SIZE")
    truth = {}
    for line in open(sys.argv[4]):
        line = line.strip()
        if line.startswith("{"):
            j = json.loads(line)
            truth[j["kernel"]] = j
    factors = {}
    for k, t in truth.items():
        f, w = fetch.get(k), write.get(k)
        factors[k] = {
            "true_read_bytes": t["read_bytes"], "true_write_bytes": t["write_bytes"], "achieved_GBps": t["GBps"],
            "fetch_counter_bytes": f["avg_counter"] * 1024 if f else None, "write_counter_bytes": w["avg_counter"] * 1024 if w else None,
            "fetch_factor": (t["read_bytes"] / (f["avg_counter"] * 1024)) if f and t["read_bytes"] and f["avg_counter"] else None,
            "write_factor": (t["write_bytes"] / (w["avg_counter"] * 1024)) if w and t["write_bytes"] and w["avg_counter"] else None,
        }
    def factor(kernel, which):
        pat = DEFAULT_PATTERN
        for key, p in PATTERN.items():
            if key in kernel:
                pat = p
                break
        name = pat[0] if which == "fetch" else pat[1]
        v = factors.get(name, {}).get(which + "_factor")
        return (v if v else 1.0), name
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("calib_"):
            continue
        fr = fetch.get(k, {"avg_counter": 0.0, "launches": 0})
        wr = write.get(k, {"avg_counter": 0.0, "launches": 0})
        ff, fn = factor(k, "fetch")
        wf, wn = factor(k, "write")
        kernels[k] = {
            "launches": max(fr["launches"], wr["launches"]),
            "fetch_raw_bytes": fr["avg_counter"] * 1024, "write_raw_bytes": wr["avg_counter"] * 1024,
            "fetch_factor": ff, "fetch_pattern": fn, "write_factor": wf, "write_pattern": wn,
            "fetch_bytes": fr["avg_counter"] * 1024 * ff, "write_bytes": wr["avg_counter"] * 1024 * wf,
            "hbm_bytes_per_launch": fr["avg_counter"] * 1024 * ff + wr["avg_counter"] * 1024 * wf,
        }
    print(json.dumps({"calibration": factors, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
