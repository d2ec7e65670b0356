#!/usr/bin/env python3
"""Instruction mix per kernel of a gfx950 code object / .o / .so:  python tools/isa/count.py file [kernel-substring ...]"""
import collections, re, subprocess, sys, os, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import kernel_occupancy as ko
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
def disasm(path):
    blobs = list(ko.code_objects(path))
    if not blobs:
        blobs = [open(path, "rb").read()]
    out = []
    for b in blobs:
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(b); f.flush()
            out.append(subprocess.run([OBJDUMP, "-d", "-C", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout)
    return "\n".join(out)
def kernels(txt):
    cur, res = None, collections.OrderedDict()
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            cur = m.group(1); res[cur] = []; continue
        if cur is None: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", line)
        if m: res[cur].append((m.group(1), m.group(2)))
    return res
if __name__ == "__main__":
    ks = kernels(disasm(sys.argv[1]))
    base = None
    for name, ins in ks.items():
        if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]): continue
        c = collections.Counter(op for op, _ in ins)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        mad = c.get("v_mad_u64_u32", 0) + c.get("v_mad_i64_i32", 0)
        print(f"== {name[:110]}: {len(ins)} instr, valu {valu}, mad64 {mad}, other valu {valu - mad}")
        print("   " + ", ".join(f"{k} {v}" for k, v in c.most_common(28)))
