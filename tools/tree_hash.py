#!/usr/bin/env python3
"""sha256 (first 16 hex digits) over the kernel sources of the product library -- every *.hip / *.hpp under groth16_amd/csrc, in
name order (the public header is not part of it: its comments change without the kernels).  profiles/pmc_traffic.json records the hash of the tree its counters were collected on;
bench.py recomputes it at run time and reports roofline.traffic only when the two agree (the GPU box has no .git to ask)."""
import hashlib
import os
import sys


def kernel_source_sha16(root):
    d = os.path.join(root, "groth16_amd", "csrc")
    files = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp")))
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_sha16(sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
