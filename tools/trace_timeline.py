#!/usr/bin/env python3
"""Timeline of the LAST proof in a rocprofv3 --kernel-trace CSV: one line per kernel launch (start offset, duration, queue), so that
overlap, gaps and the tail of the schedule in api.hip::prove_partial can be read off.  A proof is delimited by its first kernel
(spmv3_kernel / class_count_kernel of the witness sort, whichever comes first after a gap of > 0.3 ms without launches).

usage: trace_timeline.py <kernel_trace.csv> [--all]"""
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void g16::", "").replace("g16::", "")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
# split into bursts separated by idle gaps > 0.3 ms (between proofs the host does the glue + the next call's setup)
bursts, cur, last_end = [], [], None
for s, e, n, q in rows:
    if last_end is not None and s - last_end > 300_000 and cur:
        bursts.append(cur)
        cur = []
    cur.append((s, e, n, q))
    last_end = max(last_end or 0, e)
if cur:
    bursts.append(cur)
proofs = [b for b in bursts if any("bucket_accumulate30" in k[2] for k in b)]
if not proofs:
    print("no proof found")
    sys.exit(0)
b = proofs[-1]
t0 = b[0][0]
print(f"# {len(proofs)} proofs in the trace; the last one: {len(b)} launches, {(max(k[1] for k in b) - t0) / 1e6:.3f} ms from first launch to last end")
queues = {q: i for i, q in enumerate(sorted({k[3] for k in b}))}
for s, e, n, q in b:
    print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e3:9.1f} us  q{queues[q]}  {n[:90]}")
