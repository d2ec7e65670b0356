// Host-side scalar multiplication helpers for the prover.rs:76-131 glue (the O(1) part of a proof that stays on the CPU).
//   FixedBaseTable   multiples of a point that is fixed per proving key (delta_g1, delta_g2): T[w][d-1] = d * 2^(8w) * P, so
//                    k*P is 32 table additions instead of 255 doublings + ~128 additions (r*delta, s*delta, rs*delta: prover.rs:76,
//                    :90, :100, :112).  Built once per g16_pk_load (32 * 256 additions).
//   mul_window4      variable-base k*P (s*g_a, r*g1_b: prover.rs:94, :114) with a 4-bit window: 256 doublings + 64 + 14 additions.
#pragma once
#include "curve.hpp"
#include <vector>

namespace g16 {

template <class X>
struct FixedBaseTable {
    std::vector<X> t;   // [32][255]
    bool ready() const { return !t.empty(); }
    void build(const X& p) {
        t.assign((size_t)32 * 255, X::identity());
        X base = p;
        for (int w = 0; w < 32; ++w) {
            X acc = X::identity();
            for (int d = 1; d <= 255; ++d) {
                acc.add(base);
                t[(size_t)w * 255 + (size_t)(d - 1)] = acc;
            }
            acc.add(base);   // 256 * base
            base = acc;
        }
    }
    // k: canonical little-endian integer of 8 32-bit words
    X mul(const uint32_t* k) const {
        X acc = X::identity();
        for (int w = 0; w < 32; ++w) {
            const uint32_t d = (k[w >> 2] >> (8 * (w & 3))) & 0xffu;
            if (d) acc.add(t[(size_t)w * 255 + (size_t)(d - 1)]);
        }
        return acc;
    }
};

template <class X>
X mul_window4(const X& p, const uint32_t* k) {
    X tab[16];
    tab[0] = X::identity();
    tab[1] = p;
    for (int d = 2; d < 16; ++d) { tab[d] = tab[d - 1]; tab[d].add(p); }
    X acc = X::identity();
    for (int w = 63; w >= 0; --w) {
        for (int j = 0; j < 4; ++j) acc = acc.dbl();
        const uint32_t d = (k[w >> 3] >> (4 * (w & 7))) & 0xfu;
        if (d) acc.add(tab[d]);
    }
    return acc;
}

}  // namespace g16
