#!/usr/bin/env python3
"""How does the CPU oracle scale with OpenMP threads on this host?  (picks the cpu_baseline thread count)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from helpers import oracle
orc = oracle()
k = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ck = orc.syn_circuit("bls12_381", k, 2)
pk = orc.synth_pk(ck, 9)
r, s = orc.rand_fr("bls12_381", 21, 1)[0], orc.rand_fr("bls12_381", 22, 1)[0]
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
for t in (8, 16, 32, 64, 128, 256):
    if t > (os.cpu_count() or 8):
        break
    orc.set_threads(t)
    t0 = time.time(); _, tm = orc.prove(pk, ck, r, s); dt = time.time() - t0
    print(f"k={k} threads={t} prove {dt:.2f}s", {a: round(b, 2) for a, b in tm.items()}, flush=True)
