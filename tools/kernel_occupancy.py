#!/usr/bin/env python3
"""Register / scratch / LDS budget of every kernel in a built HIP library, read from the code objects embedded in the .so
(clang offload bundles in .hip_fatbin -> AMDGPU metadata notes via llvm-readelf).  No GPU needed.

    python tools/kernel_occupancy.py [lib.so] [substring ...]

gfx950: 512 registers (VGPR + AGPR, unified) per SIMD lane; waves per SIMD = floor(512 / roundup(vgpr + agpr, 8)), at most 8."""
import os
import re
import subprocess
import sys
import tempfile

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def code_objects(path, arch="gfx950"):
    data = open(path, "rb").read()
    pos = 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            return
        p = pos + len(MAGIC)
        n = int.from_bytes(data[p:p + 8], "little")
        p += 8
        for _ in range(n):
            off, size, tl = (int.from_bytes(data[p + 8 * i:p + 8 * i + 8], "little") for i in range(3))
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if arch in triple and size:
                yield data[pos + off:pos + off + size]
        pos = p


def kernels(path):
    out = {}
    for blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
        for block in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            block = ".agpr_count:" + block
            get = lambda k: (re.search(r"\." + k + r":\s*(\S+)", block) or [None, None])[1]  # noqa: E731
            sym = get("name")
            if not sym:
                continue
            out[sym] = dict(vgpr=int(get("vgpr_count") or 0), agpr=int(get("agpr_count") or 0), sgpr=int(get("sgpr_count") or 0),
                            scratch=int(get("private_segment_fixed_size") or 0), lds=int(get("group_segment_fixed_size") or 0),
                            max_flat_wg=int(get("max_flat_workgroup_size") or 0))
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for sym, d in zip(names, dem):
        k = out[sym]
        regs = (k["vgpr"] + 7) // 8 * 8   # .vgpr_count is the unified total (arch VGPRs + AGPRs)
        k["waves_per_simd"] = min(8, 512 // regs) if regs else 8
        res[re.sub(r"g16::", "", re.sub(r"\(.*", "", d)).replace("void ", "")] = k
    return res


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(here, "..", "groth16_amd", "libg16_mi355x.so")
    subs = [a for a in sys.argv[1:] if not a.endswith(".so")]
    for name, k in sorted(kernels(lib).items()):
        if subs and not any(s in name for s in subs):
            continue
        print(f"{name[:100]:100s} vgpr={k['vgpr']:3d} agpr={k['agpr']:3d} waves/SIMD={k['waves_per_simd']} scratch={k['scratch']:5d} lds={k['lds']}")
