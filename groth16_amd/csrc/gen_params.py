#!/usr/bin/env python3
"""Generates params_gen.hpp: Montgomery constants (32-bit limbs, R = 2^(32N) which equals
arkworks' R = 2^(64*N/2)) for the two curves the prover supports.  Run by
__graft_entry__.build() / the Makefile; output is committed so the GPU box needs no step.

Constants: BLS12-381 and BN254 field moduli, Fr multiplicative generator and 2-adicity
(ark-bls12-381 / ark-bn254 0.5.0 Fr configs: GENERATOR = 7 / 5, TWO_ADICITY = 32 / 28),
curve coefficients and standard generators.
"""
import os

CURVES = {
    "Bls12_381": dict(
        q=0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB,
        r=0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
        gen=7, s=32, b1=4, b2=(4, 4),
        g1=(0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
            0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1),
        g2=(0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE),
    ),
    "Bn254": dict(
        q=21888242871839275222246405745257275088696311157297823662689037894645226208583,
        r=21888242871839275222246405745257275088548364400416034343698204186575808495617,
        gen=5, s=28, b1=3, b2=None,  # 3/(9+u), computed below
        g1=(1, 2),
        g2=(10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531),
    ),
}


def limbs32(v, n):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(vals):
    return "{" + ", ".join("0x%08xu" % v for v in vals) + "}"


def field_struct(name, p):
    n = (p.bit_length() + 31) // 32
    if n % 2:
        n += 1  # keep 64-bit limb compatibility
    R = 1 << (32 * n)
    inv = (-pow(p, -1, 1 << 32)) % (1 << 32)
    out = []
    out.append("struct %s {" % name)
    out.append("    static constexpr int N = %d;" % n)
    out.append("    static constexpr int BITS = %d;" % p.bit_length())
    out.append("    static constexpr uint32_t INV = 0x%08xu;  // -p^-1 mod 2^32" % inv)
    for fn, val in (("mod", p), ("r1", R % p), ("r2", R * R % p), ("pm2", p - 2)):
        out.append("    G16_HD static constexpr uint32_t %s(int i) {" % fn)
        out.append("        constexpr uint32_t t[N] = %s;" % arr(limbs32(val, n)))
        out.append("        return t[i];")
        out.append("    }")
    # ---- 30-bit reduced-radix view (fp30.hpp): NL limbs, R' = 2^(30 NL)
    nl = (p.bit_length() + 29) // 30
    R30 = 1 << (30 * nl)
    M30 = (1 << 30) - 1

    def limbs30(v, cnt=nl):
        return [(v >> (30 * i)) & M30 for i in range(cnt - 1)] + [v >> (30 * (cnt - 1))]

    def redundant(k):  # k*p with every limb but the top >= 2^30 - 1, so limb-wise a + KP - b never borrows
        l = limbs30(k * p)
        r = [l[0] + (1 << 30)] + [x + (1 << 30) - 1 for x in l[1:-1]] + [l[-1] - 1]
        assert sum(x << (30 * i) for i, x in enumerate(r)) == k * p
        return r

    out.append("    static constexpr int NL30 = %d;" % nl)
    out.append("    static constexpr uint32_t PINV30 = 0x%08xu;   // -p^-1 mod 2^30" % ((-pow(p, -1, 1 << 30)) % (1 << 30)))
    out.append("    static constexpr uint32_t PPINV30 = 0x%08xu;  //  p^-1 mod 2^30" % (pow(p, -1, 1 << 30)))
    for fn, vals in (("p30", limbs30(p)), ("one30", limbs30(R30 % p)), ("rstd30", limbs30(R % p)), ("r3_30", limbs30(pow(R30, 3, p))), ("kp2", redundant(2)),
                     ("kp4", redundant(4)), ("kp6", redundant(6)), ("kp8", redundant(8)), ("kp16", redundant(16)), ("np2", limbs30(2 * p)),
                     ("np4", limbs30(4 * p)), ("np8", limbs30(8 * p)), ("np16", limbs30(16 * p))):
        out.append("    G16_HD static constexpr uint32_t %s(int i) {" % fn)
        out.append("        constexpr uint32_t t[NL30] = %s;" % arr(vals))
        out.append("        return t[i];")
        out.append("    }")
    # redundant 2^(k+1) * p for k = 0..11 (lazy DIF butterflies subtract operands that double every stage)
    if name.endswith("FrP"):   # scalar fields only (the NTT); their top limb has room for 4096 p
        out.append("    G16_HD static constexpr uint32_t kp_pow2(int k, int i) {")
        out.append("        constexpr uint32_t t[12][NL30] = {%s};" % ", ".join(arr(redundant(2 << k)) for k in range(12)))
        out.append("        return t[k][i];")
        out.append("    }")
    # 2^(30 NL) mod p as a plain integer in 32-bit limbs: std-Montgomery multiplying x*R by it gives x*R' mod p
    out.append("    G16_HD static constexpr uint32_t r30_plain(int i) {")
    out.append("        constexpr uint32_t t[N] = %s;" % arr(limbs32(R30 % p, n)))
    out.append("        return t[i];")
    out.append("    }")
    out.append("};")
    return "\n".join(out), n, R


def mont(v, p, R, n):
    return arr(limbs32(v % p * R % p, n))


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    o = []
    o.append("// GENERATED by gen_params.py -- do not edit.")
    o.append("#pragma once")
    o.append("#include <cstdint>")
    o.append('#include "hd.hpp"')
    o.append("namespace g16 {")
    for cname, c in CURVES.items():
        q, r = c["q"], c["r"]
        fq_s, nq, Rq = field_struct(cname + "FqP", q)
        fr_s, nr, Rr = field_struct(cname + "FrP", r)
        o.append(fq_s)
        o.append(fr_s)
        root = pow(c["gen"], (r - 1) >> c["s"], r)
        b2 = c["b2"]
        if b2 is None:
            # 3 / (9 + u) in Fq[u]/(u^2+1):  3 * (9 - u) / 82
            inv82 = pow(82, q - 2, q)
            b2 = (27 * inv82 % q, (-3) * inv82 % q)
        o.append("struct %sConsts {" % cname)
        o.append("    static constexpr int TWO_ADICITY = %d;" % c["s"])
        for nm, val, p, R, n in (
            ("fr_generator", c["gen"], r, Rr, nr), ("fr_generator_inv", pow(c["gen"], r - 2, r), r, Rr, nr),
            ("two_adic_root", root, r, Rr, nr),
            ("b1", c["b1"], q, Rq, nq), ("b2_c0", b2[0], q, Rq, nq), ("b2_c1", b2[1], q, Rq, nq),
            ("g1_x", c["g1"][0], q, Rq, nq), ("g1_y", c["g1"][1], q, Rq, nq),
            ("g2_x0", c["g2"][0], q, Rq, nq), ("g2_x1", c["g2"][1], q, Rq, nq),
            ("g2_y0", c["g2"][2], q, Rq, nq), ("g2_y1", c["g2"][3], q, Rq, nq),
        ):
            o.append("    G16_HD static constexpr uint32_t %s(int i) {" % nm)
            o.append("        constexpr uint32_t t[%d] = %s;" % (n, mont(val, p, R, n)))
            o.append("        return t[i];")
            o.append("    }")
        o.append("};")
    o.append("}  // namespace g16")
    with open(os.path.join(here, "params_gen.hpp"), "w") as f:
        f.write("\n".join(o) + "\n")


if __name__ == "__main__":
    main()
