#!/usr/bin/env python3
"""Basic blocks of one kernel with their instruction mix:  python tools/isa/hotloop.py lib.so kernel-substring [min-instr]
The hot path of the bucket pass (one mixed addition) is the run of large blocks holding ~3 150 v_mad_u64_u32."""
import collections, re, sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import count
def blocks(ins_lines):
    # ins_lines: (addr, op, args); split at branch targets and after branches
    targets = set()
    for a, op, args in ins_lines:
        if op.startswith("s_cbranch") or op == "s_branch":
            m = re.search(r"(\d+)\s*$", args)
    return None
if __name__ == "__main__":
    txt = count.disasm(sys.argv[1])
    want = sys.argv[2]
    minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    cur = None
    body = []
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            name = m.group(1)
            if name.startswith("L") or name.startswith(".L") or re.match(r"^\$|^BB", name):
                if cur is not None: body.append(("label", name))
                continue
            cur = name if want in name else None
            if cur: body = [("kernel", name)]
            continue
        if cur is None: continue
        m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m: body.append((m.group(1), m.group(2), int(m.group(3), 16)))
    # branch targets from the disassembler's "<label+off>" annotations are absent in --no-show-raw-insn output; compute from simm16
    ins = [b for b in body if len(b) == 3]
    addr_index = {a: i for i, (_, _, a) in enumerate(ins)}
    cuts = set([0])
    for i, (op, args, a) in enumerate(ins):
        if op.startswith("s_cbranch") or op == "s_branch" or op in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
            cuts.add(i + 1)
            m = re.search(r"(-?\d+)\s*$", args)
            if m and (op.startswith("s_cbranch") or op == "s_branch"):
                off = int(m.group(1))
                if off >= 32768: off -= 65536
                t = a + 4 + 4 * off
                if t in addr_index: cuts.add(addr_index[t])
    cuts = sorted(c for c in cuts if c <= len(ins))
    tot = collections.Counter()
    print(body[0][1][:150])
    for s, e in zip(cuts, cuts[1:] + [len(ins)]):
        blk = ins[s:e]
        if len(blk) < minn: continue
        c = collections.Counter(op for op, _, _ in blk)
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        mad = c.get("v_mad_u64_u32", 0) + c.get("v_mad_i64_i32", 0)
        lds = sum(v for k, v in c.items() if k.startswith("ds_"))
        print(f"block @{blk[0][2]:#x} n={len(blk)} valu={valu} mad={mad} other_valu={valu-mad} lds={lds} s_nop={c.get('s_nop',0)} salu={sum(v for k,v in c.items() if k.startswith('s_'))-c.get('s_nop',0)} vmem={sum(v for k,v in c.items() if k.startswith('global_') or k.startswith('buffer_') or k.startswith('scratch_'))}")
        print("      " + ", ".join(f"{k} {v}" for k, v in c.most_common(16) if k != "v_mad_u64_u32"))
