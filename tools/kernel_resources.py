#!/usr/bin/env python3
"""Print VGPR / scratch / occupancy per kernel of one .hip translation unit (gfx950)."""
import re, subprocess, sys
src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "SGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "VGPR Spill"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    n = re.sub(r"g16::", "", r["name"])
    n = re.sub(r"\(.*", "", n)[:90]
    print(f"{n:90s} vgpr={r.get('VGPRs')} agpr={r.get('AGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}")
