#!/usr/bin/env python3
"""Generates fips_asm_gen.hpp: the product-scanning Montgomery products of fp30.hpp (Fp30::fips_n) as hand-scheduled gfx950
assembly, one inline-asm block per product, for the four 30-bit-limb fields of the prover (BLS12-381 / BN254, Fq and Fr).

Why assembly: column-major, ONE accumulator carries a column's limb products, the reduction's products and -- as the seed of
the next column's multiply-add chain -- the carry, so the 64-bit add per column of the operand-scanning form disappears
(fp30.hpp, "product scanning").  LLVM's reassociation undoes exactly that order (it moves the late-arriving carry to the
end of a column's sum), and every way of pinning the order from C++ costs more than it saves on gfx950: an empty asm that
DEFINES the running sum draws a wait state (s_nop 0) before its reader from the hazard recognizer (2 200 per mixed
addition: measured 3.6 % slower than the operand-scanning forms), one that only READS it lets the scheduler stretch the
live ranges (G1 pass 137 -> 248 registers, the lane-pair kernel spills).  Written out, the chain is exactly
2 NL^2 (+ 2 per extra accumulator) v_mad_u64_u32 and 5 NL - 1 other vector instructions per product, ~45 live registers.

Register use inside a block: operands are compiler-allocated (%n); the accumulators are the PHYSICAL pairs v[2:3], v[4:5],
v[6:7] (declared as clobbers) because their halves must be named (v_mul_lo_u32 reads the low word, the accumulator merge
reads low and high words) and inline asm has no sub-register modifier.  m_i lives in the register of output limb r_i (m_i's
last use is column i + NL - 1, r_i is written in column NL + i).  The carry-out of every v_mad_u64_u32 goes to vcc (unused).

The column plan (which part of which column starts a new accumulator) is computed here exactly as Fp30::fips_plan computes
it at compile time; the header carries it and fp30.hpp static_asserts that the two agree, so the C++ form -- which the host
self-test runs with 128-bit shadow accumulators and whose constexpr bound propagation is the overflow proof -- vouches for
the assembly's plan.  Run by the Makefile; output is committed so the GPU box needs no step.
"""
import os

from gen_params import CURVES

MASK = (1 << 30) - 1
W32 = (1 << 32) - 1
LIM = 1 << 64
ACC = ["v[2:3]", "v[4:5]", "v[6:7]", "v[8:9]", "v[10:11]"]
ACC_LO = ["v2", "v4", "v6", "v8", "v10"]
ACC_HI = ["v3", "v5", "v7", "v9", "v11"]


def nl30(p):
    nl = (p.bit_length() + 29) // 30
    if nl * 30 - p.bit_length() < 9:
        nl += 1
    return nl


def plan(pl, NL, NS):
    """Fp30::fips_plan: accumulator index per (column, part); parts = NS sweeps then the reduction."""
    seg, nseg = [], []
    carry = 0
    for c in range(2 * NL - 1):
        lo = 0 if c < NL else c - NL + 1
        hi = c if c < NL else NL - 1
        psum = sum(MASK * pl[c - i] for i in range(lo, hi + 1) if i != c)
        B = [0] * (NS + 1)
        B[0] = carry + (MASK * pl[0] if c < NL else 0) + (NS + 2) * W32   # + 2: a fused subtraction's K p_j - s_j (and slack)
        cur = 0
        row = []
        for k in range(NS + 1):
            part = (hi - lo + 1) * MASK * MASK if k < NS else psum
            if B[cur] + part >= LIM:
                cur += 1
                assert cur <= NS and part < LIM
            B[cur] += part
            row.append(cur)
        seg.append(row)
        nseg.append(cur + 1)
        carry = (B[0] >> 30) + sum(4 * (B[s] >> 32) for s in range(1, cur + 1))
    return seg, nseg


class Block:
    def __init__(self):
        self.lines = []
        self.fresh = set()   # accumulators not yet written in this column (first multiply-add takes the literal 0)

    def mad(self, acc, x, y):
        src2 = "0" if acc in self.fresh else ACC[acc]
        self.fresh.discard(acc)
        self.lines.append("v_mad_u64_u32 %s, vcc, %s, %s, %s" % (ACC[acc], x, y, src2))

    def emit(self, s):
        self.lines.append(s)


SGPR0 = 36   # first of the NL + 1 physical SGPRs that hold p and -p^-1 when they cannot be operands (four sweeps: 8 NL + NL operands are
             # all the compiler takes -- past ~128 operands it does not terminate)


TMP = "v12"   # scratch register of the fused subtraction (physical, clobbered)


def product(pl, NL, NS, sqr, pinv_val=None, const_sgpr=False, sub=0):
    """returns (asm lines, ...).  Operand numbering: r[NL] | (sqr: t[NL]) | x_k, y_k ... | (sub: s[NL] | sub == 2: u[NL], v[NL]) | p[NL] | pinv | (sub: kp[NL])
    sub = 1: r = product + K p - s (s normalised, K p in the redundant form whose limbs cover any normalised limb);
    sub = 2: r = product + 6 p - (u + 2 v) (the X3 of the mixed addition; K p with limbs >= 3 (2^30 - 1)).
    The difference K p_j - s_j (>= 0, < 2^32) joins column NL + j before its limb is taken: one 32-bit subtraction and one
    multiply-add by 1 per limb instead of the subtraction's own carry-propagating pass (13 + 12 * 3 instructions)."""
    seg, nseg = plan(pl, NL, NS)
    n = 0
    r = ["%%%d" % (n + i) for i in range(NL)]
    n += NL
    t = None
    if sqr:
        t = ["%%%d" % (n + i) for i in range(NL)]   # doubled limbs (t[NL-1] unused: the top limb is never the smaller index)
        n += NL
    xs, ys = [], []
    for k in range(NS):
        xs.append(["%%%d" % (n + i) for i in range(NL)])
        n += NL
        if sqr:
            ys.append(xs[-1])
        else:
            ys.append(["%%%d" % (n + i) for i in range(NL)])
            n += NL
    sub_ops = []
    for _ in range(sub):
        sub_ops.append(["%%%d" % (n + i) for i in range(NL)])
        n += NL
    b = Block()
    if const_sgpr:
        p = ["s%d" % (SGPR0 + i) for i in range(NL)]
        pinv = "s%d" % (SGPR0 + NL)
        for i in range(NL):
            b.emit("s_mov_b32 %s, 0x%08x" % (p[i], pl[i]))
        b.emit("s_mov_b32 %s, 0x%08x" % (pinv, pinv_val))
    else:
        p = ["%%%d" % (n + i) for i in range(NL)]
        n += NL
        pinv = "%%%d" % n
        n += 1
    kp = None
    if sub:
        kp = ["%%%d" % (n + i) for i in range(NL)]
        n += NL

    def sub_term(j):   # TMP = K p_j - subtrahend_j
        if sub == 1:
            b.emit("v_sub_u32 %s, %s, %s" % (TMP, kp[j], sub_ops[0][j]))
        else:
            b.emit("v_lshl_add_u32 %s, %s, 1, %s" % (TMP, sub_ops[1][j], sub_ops[0][j]))
            b.emit("v_sub_u32 %s, %s, %s" % (TMP, kp[j], TMP))
    if sqr:
        for i in range(NL - 1):
            b.emit("v_lshlrev_b32 %s, 1, %s" % (t[i], xs[0][i]))
    b.fresh = set(range(len(ACC)))
    for c in range(2 * NL - 1):
        lo = 0 if c < NL else c - NL + 1
        hi = c if c < NL else NL - 1
        if c > 0:
            b.fresh = set(range(1, len(ACC)))
        for k in range(NS):
            s = seg[c][k]
            if sqr:
                i = lo
                while 2 * i < c:
                    b.mad(s, t[i], xs[0][c - i])
                    i += 1
                if c % 2 == 0:
                    b.mad(s, xs[0][c // 2], xs[0][c // 2])
            else:
                for i in range(lo, hi + 1):
                    b.mad(s, xs[k][i], ys[k][c - i])
        s = seg[c][NS]
        for i in range(lo, hi + 1):
            if i != c:
                b.mad(s, r[i], p[c - i])          # m_i sits in r_i's register
        for s in range(1, nseg[c]):
            assert s not in b.fresh
            b.emit("v_mad_u64_u32 %s, vcc, %s, 1, %s" % (ACC[0], ACC_LO[s], "0" if 0 in b.fresh else ACC[0]))
            b.fresh.discard(0)
        assert 0 not in b.fresh
        if c < NL:
            b.emit("v_mul_lo_u32 %s, %s, %s" % (r[c], ACC_LO[0], pinv))
            b.emit("v_and_b32 %s, 0x3fffffff, %s" % (r[c], r[c]))
            b.mad(0, r[c], p[0])
            b.emit("v_lshrrev_b64 %s, 30, %s" % (ACC[0], ACC[0]))
        elif c < 2 * NL - 2:
            if sub:
                sub_term(c - NL)
                b.mad(0, TMP, "1")
            b.emit("v_and_b32 %s, 0x3fffffff, %s" % (r[c - NL], ACC_LO[0]))
            b.emit("v_lshrrev_b64 %s, 30, %s" % (ACC[0], ACC[0]))
        else:
            if sub:
                sub_term(c - NL)
                b.mad(0, TMP, "1")
            b.emit("v_and_b32 %s, 0x3fffffff, %s" % (r[c - NL], ACC_LO[0]))
            if nseg[c] > 1:
                b.emit("v_lshrrev_b64 %s, 30, %s" % (ACC[0], ACC[0]))
            else:
                b.emit("v_alignbit_b32 %s, %s, %s, 30" % (r[NL - 1], ACC_HI[0], ACC_LO[0]))
        for s in range(1, nseg[c]):
            b.emit("v_mad_u64_u32 %s, vcc, %s, 4, %s" % (ACC[0], ACC_HI[s], ACC[0]))
        if c == 2 * NL - 2 and nseg[c] > 1:
            b.emit("v_mov_b32 %s, %s" % (r[NL - 1], ACC_LO[0]))
    if sub:   # the top limb takes its difference with a plain 32-bit add (no carry leaves it)
        sub_term(NL - 1)
        b.emit("v_add_u32 %s, %s, %s" % (r[NL - 1], r[NL - 1], TMP))
    return b.lines, seg, nseg, max(nseg)


def limbs30(p, nl):
    return [(p >> (30 * i)) & MASK for i in range(nl)]


def redundant(p, NL, k, floor_mult):
    """k p with every limb but the top >= floor_mult * (2^30 - 1) (and < 2^32): limb-wise K p_j - s_j never borrows for s_j up to that"""
    l = [(k * p >> (30 * i)) & MASK for i in range(NL - 1)] + [k * p >> (30 * (NL - 1))]
    r = [l[0] + floor_mult * (1 << 30)] + [x + floor_mult * (1 << 30) - floor_mult for x in l[1:-1]] + [l[-1] - floor_mult]
    assert sum(x << (30 * i) for i, x in enumerate(r)) == k * p and all(0 <= x < (1 << 32) for x in r)
    assert all(x >= floor_mult * MASK for x in r[:-1])
    return r


def emit_fn(out, name, pl, NL, NS, sqr, pinv, sub=0, kp=None):
    const_sgpr = NS > 2
    lines, seg, nseg, maxseg = product(pl, NL, NS, sqr, pinv, const_sgpr, sub)
    args = ["uint32_t* __restrict__ r"]
    for k in range(NS):
        args.append("const uint32_t* x%d" % k)
        if not sqr:
            args.append("const uint32_t* y%d" % k)
    for i in range(sub):
        args.append("const uint32_t* s%d" % i)
    out.append("    static __device__ __forceinline__ void %s(%s) {" % (name, ", ".join(args)))
    if sqr:
        out.append("        uint32_t t[%d];" % NL)
    out.append("        asm(")
    for ln in lines:
        out.append('            "%s\\n"' % ln)
    outs = ['"=&v"(r[%d])' % i for i in range(NL)]
    if sqr:
        outs += ['"=&v"(t[%d])' % i for i in range(NL)]
    ins = []
    for k in range(NS):
        ins += ['"v"(x%d[%d])' % (k, i) for i in range(NL)]
        if not sqr:
            ins += ['"v"(y%d[%d])' % (k, i) for i in range(NL)]
    for i in range(sub):
        ins += ['"v"(s%d[%d])' % (i, j) for j in range(NL)]
    clob = ['"vcc"'] + ['"v%d"' % i for i in range(2, 2 + 2 * maxseg)]
    if sub:
        clob.append('"%s"' % TMP)
    if const_sgpr:
        clob += ['"s%d"' % (SGPR0 + i) for i in range(NL + 1)]
    else:
        ins += ['"s"(0x%08xu)' % v for v in pl]
        ins.append('"s"(0x%08xu)' % pinv)
    if sub:
        ins += ['"s"(0x%08xu)' % v for v in kp]
    out.append("            : %s" % ", ".join(outs))
    out.append("            : %s" % ", ".join(ins))
    out.append("            : %s);" % ", ".join(clob))
    out.append("    }")
    return seg, nseg, len(lines), sum(1 for ln in lines if ln.startswith("v_mad_u64_u32"))


def main():
    out = []
    out.append("// GENERATED by gen_fips_asm.py -- do not edit.  Product-scanning Montgomery products as gfx950 assembly (see the generator's header).")
    out.append("#pragma once")
    out.append("#include <cstdint>")
    out.append('#include "params_gen.hpp"')
    out.append("namespace g16 {")
    out.append("// forms: mul = x0 y0, sqr = x0^2, mul2 = x0 y0 + x1 y1, mul4 = x0 y0 + ... + x3 y3 (one reduction each); operands are arrays of NL normalised 30-bit limbs")
    out.append("template <class P> struct FipsAsm { static constexpr bool available = false; static constexpr bool has_sub = false; };")
    stats = []
    for cname, c in CURVES.items():
        for fname, p in (("Fq", c["q"]), ("Fr", c["r"])):
            NL = nl30(p)
            pl = limbs30(p, NL)
            pinv = (-pow(p, -1, 1 << 30)) % (1 << 30)
            sname = "%s%sP" % (cname, fname)
            out.append("template <> struct FipsAsm<%s> {" % sname)
            out.append("    static constexpr bool available = true;")
            out.append("    static constexpr int NL = %d;" % NL)
            out.append("    static constexpr bool has_sub = %s;   // the forms with a fused subtraction (base fields only)" % ("true" if fname == "Fq" else "false"))
            plans = {}
            out.append("#if defined(__HIP_DEVICE_COMPILE__)")
            for name, NS, sqr in (("mul", 1, False), ("sqr", 1, True), ("mul2", 2, False), ("mul4", 4, False)):
                seg, nseg, n_ins, n_mad = emit_fn(out, name, pl, NL, NS, sqr, pinv)
                plans[NS] = (seg, nseg)
                stats.append((sname, name, n_ins, n_mad))
            if fname == "Fq":
                # products with the group formulas' subtractions riding in the high columns: K = 2, 4, 8 (the lazy-bound classes of
                # Acc30 / AccParked: KM, KY, KX) and the X3 form R^2 + 6 p - (PPP + 2 Q)
                for K in (2, 4, 8):
                    kpl = redundant(p, NL, K, 1)
                    for base, NS, sqr in (("mul", 1, False), ("mul2", 2, False)):
                        _, _, n_ins, n_mad = emit_fn(out, "%s_s%d" % (base, K), pl, NL, NS, sqr, pinv, 1, kpl)
                        stats.append((sname, "%s_s%d" % (base, K), n_ins, n_mad))
                kpw = redundant(p, NL, 6, 3)
                for base, NS, sqr in (("mul", 1, False), ("sqr", 1, True)):
                    _, _, n_ins, n_mad = emit_fn(out, "%s_x3" % base, pl, NL, NS, sqr, pinv, 2, kpw)
                    stats.append((sname, "%s_x3" % base, n_ins, n_mad))
            out.append("#endif")
            for NS, (seg, nseg) in sorted(plans.items()):
                flat = ", ".join(str(v) for row in seg for v in row)
                out.append("    // the generator's column plan for %d sweep(s): accumulator of part k of column c at [c * %d + k]; fp30.hpp checks it" % (NS, NS + 1))
                out.append("    static constexpr int plan%d(int i) { constexpr unsigned char t[%d] = {%s}; return t[i]; }" % (NS, len(seg) * (NS + 1), flat))
            out.append("};")
    out.append("// instruction counts of the blocks (all vector ALU; multiply-adds in brackets):")
    for sname, name, n_ins, n_mad in stats:
        out.append("//   %-14s %-5s %4d [%4d]" % (sname, name, n_ins, n_mad))
    out.append("}  // namespace g16")
    import sys
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "fips_asm_gen.hpp")
    with open(path, "w") as f:
        f.write("\n".join(out) + "\n")
    print("wrote", path)


if __name__ == "__main__":
    main()
