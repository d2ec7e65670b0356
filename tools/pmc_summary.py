#!/usr/bin/env python3
"""Per-kernel averages of one rocprofv3 --pmc counter from its counter_collection CSV.

usage: pmc_summary.py <dir written by rocprofv3 -d> <COUNTER>   -> JSON on stdout
(bytes for FETCH_SIZE / WRITE_SIZE = counter * 1024, MI355X_MICROARCH.md's HBM section)"""
import csv, glob, json, re, sys
from collections import defaultdict

d, counter = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"g16::", "", r["Kernel_Name"])
        name = re.sub(r"\(.*", "", name)
        a = acc[name]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
print(json.dumps({k: {"launches": v[0], "avg_counter": v[1] / v[0], "avg_bytes_x1024": v[1] / v[0] * 1024} for k, v in acc.items()}, indent=1))
