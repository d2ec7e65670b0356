// Batched-affine bucket accumulation: the arithmetic half (host + device).
//
// An affine chord addition (x3 = L^2 - x1 - x2, y3 = L (x1 - x3) - y1, L = (y2 - y1) / (x2 - x1)) costs 2M + 1S + one
// inversion; with Montgomery's trick over a batch of independent additions the inversion becomes 3M per addition plus ONE
// field inversion per batch: 5M + 1S (~5.75 product-equivalents with the dedicated squaring) against 8M + 2S (~9.5) for the
// inversion-free XYZZ mixed addition of Acc30 -- the bucket pass is 86 % of a proof and is bound by exactly these products.
//
//   ModInv30<P>      x^-1 mod p by Bernstein-Yang division steps ("safegcd"), 30 steps per outer iteration on the low limbs
//                    of (f, g), signed 30-bit limbs throughout -- the same radix as Fp30, so nothing is converted.  One
//                    inversion is ~25-30 outer iterations of ~0.9 k instructions: ~25 product-equivalents where Fermat's
//                    x^(p-2) needs ~390.  That ratio is what lets every LANE invert its own batch total: no cross-lane
//                    product tree, no shuffles, a batch of 32-64 additions per lane amortises it to < 15 %.
//   AffineBatch<F>   one lane's forward step (classify the pair, multiply its denominator into the running prefix product)
//                    and backward step (peel the pair's inverse off the running inverse, finish the addition), F = Fp30<P>
//                    (G1) or Fp2p30<P> (G2, lane pair).  Exceptional pairs -- either operand the identity, P = Q (tangent),
//                    P = -Q -- are resolved exactly; they put the neutral factor 1 into the product.
#pragma once
#include "curve.hpp"
#include "fp30.hpp"

namespace g16 {

template <class P>
struct ModInv30 {
    static constexpr int NL = P::NL30;
    static constexpr int32_t M30 = (int32_t)((1u << 30) - 1u);
    struct S30 { int32_t v[NL]; };      // limbs 0 .. NL-2 in [0, 2^30), the top limb signed
    struct T2x2 { int32_t u, v, q, r; };  // 2^30 * [f', g'] = [[u, v], [q, r]] * [f, g]

    // 30 division steps on the low limbs.  zeta = -(delta + 1/2); f0 odd.  Branch-free: every lane of a wave runs the same
    // instruction stream whatever its operands.
    G16_HD static int32_t divsteps30(int32_t zeta, uint32_t f0, uint32_t g0, T2x2& t) {
        uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
        for (int i = 0; i < 30; ++i) {
            uint32_t m1 = (uint32_t)(zeta >> 31);          // zeta < 0
            const uint32_t m2 = 0u - (g & 1u);              // g odd
            const uint32_t x = (f ^ m1) - m1, y = (u ^ m1) - m1, z = (v ^ m1) - m1;   // (+-f, +-u, +-v)
            g += x & m2; q += y & m2; r += z & m2;
            m1 &= m2;                                       // swap case: zeta < 0 and g odd
            zeta = (zeta ^ (int32_t)m1) - 1;                // -zeta - 2  or  zeta - 1
            f += g & m1; u += q & m1; v += r & m1;
            g >>= 1; u <<= 1; v <<= 1;
        }
        t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
        return zeta;
    }
    // [f, g] <- t [f, g] / 2^30 (exact)
    G16_HD static void update_fg(S30& f, S30& g, const T2x2& t) {
        const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
        int64_t cf = u * f.v[0] + v * g.v[0], cg = q * f.v[0] + r * g.v[0];
        cf >>= 30; cg >>= 30;
        G16_UNROLL for (int i = 1; i < NL; ++i) {
            cf += u * f.v[i] + v * g.v[i];
            cg += q * f.v[i] + r * g.v[i];
            f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
            g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
        }
        f.v[NL - 1] = (int32_t)cf;
        g.v[NL - 1] = (int32_t)cg;
    }
    // [d, e] <- t [d, e] / 2^30 mod p, both kept in (-2p, p): a multiple of p is added that clears the low 30 bits
    G16_HD static void update_de(S30& d, S30& e, const T2x2& t) {
        const int64_t u = t.u, v = t.v, q = t.q, r = t.r;
        const int32_t sd = d.v[NL - 1] >> 31, se = e.v[NL - 1] >> 31;
        int32_t md = (t.u & sd) + (t.v & se), me = (t.q & sd) + (t.r & se);
        int64_t cd = u * d.v[0] + v * e.v[0], ce = q * d.v[0] + r * e.v[0];
        md -= (int32_t)((P::PPINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
        me -= (int32_t)((P::PPINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
        cd += (int64_t)(int32_t)P::p30(0) * md;
        ce += (int64_t)(int32_t)P::p30(0) * me;
        cd >>= 30; ce >>= 30;
        G16_UNROLL for (int i = 1; i < NL; ++i) {
            cd += u * d.v[i] + v * e.v[i] + (int64_t)(int32_t)P::p30(i) * md;
            ce += q * d.v[i] + r * e.v[i] + (int64_t)(int32_t)P::p30(i) * me;
            d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
            e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
        }
        d.v[NL - 1] = (int32_t)cd;
        e.v[NL - 1] = (int32_t)ce;
    }
    G16_HD static void carry(S30& r) {
        G16_UNROLL for (int i = 0; i + 1 < NL; ++i) { r.v[i + 1] += r.v[i] >> 30; r.v[i] &= M30; }
    }
    // r in (-2p, p), negated when sign < 0, brought to [0, p)
    G16_HD static void normalize(S30& r, int32_t sign) {
        const int32_t add1 = r.v[NL - 1] >> 31, neg = sign >> 31;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            int32_t x = r.v[i] + ((int32_t)P::p30(i) & add1);
            r.v[i] = (x ^ neg) - neg;
        }
        carry(r);
        const int32_t add2 = r.v[NL - 1] >> 31;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.v[i] += (int32_t)P::p30(i) & add2;
        carry(r);
    }
    // x^-1 mod p as a plain integer in [0, p); x: normalised limbs of any value below 2^(30 NL - 3) that p does not divide
    // (x = 0 mod p returns 0).  The loop ends when g = 0 (then f = +-1 and d = +-x^-1): ~2.3 steps per bit of p at worst.
    G16_HD static Fp30<P> inverse_plain(const Fp30<P>& x) {
        S30 d, e, f, g;
        G16_UNROLL for (int i = 0; i < NL; ++i) { d.v[i] = 0; e.v[i] = 0; f.v[i] = (int32_t)P::p30(i); g.v[i] = (int32_t)x.l[i]; }
        e.v[0] = 1;
        int32_t zeta = -1;
        for (int it = 0; it < 4 * NL; ++it) {   // (bound never reached: 2.3 * 30 NL / 30 < 2.5 NL outer iterations)
            int32_t nz = 0;
            G16_UNROLL for (int i = 0; i < NL; ++i) nz |= g.v[i];
            if (nz == 0) break;
            T2x2 t;
            zeta = divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
            update_de(d, e, t);
            update_fg(f, g, t);
        }
        normalize(d, f.v[NL - 1]);
        Fp30<P> r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = (uint32_t)d.v[i];
        return r;
    }
    // Montgomery form in, Montgomery form out: (a R')^-1 = a^-1 R'^-1, times R'^3 / R' = a^-1 R'   (output < 1.5p)
    G16_HD static Fp30<P> inverse(const Fp30<P>& x) {
        Fp30<P> c;
        G16_UNROLL for (int i = 0; i < NL; ++i) c.l[i] = P::r3_30(i);
        return inverse_plain(x).mul(c);
    }
};

// field inversion as each bucket-kernel field sees it (Montgomery form in and out, output < 1.5p per component)
template <class P>
G16_HD Fp30<P> batch_inverse(const Fp30<P>& a) { return ModInv30<P>::inverse(a); }
// Fq2 lane pair: (a0 + a1 u)^-1 = (a0 - a1 u) / (a0^2 + a1^2).  `m` is this lane's component, `o` the partner's; both lanes
// invert the same norm (a pure function of (hi, m, o): host-testable without a wave)
template <class P>
G16_HD Fp30<P> pair_inverse(bool hi, const Fp30<P>& m, const Fp30<P>& o) {
    const Fp30<P> n = m.sqr().add(o.sqr());                       // < 3p
    const Fp30<P> t = m.mul(ModInv30<P>::inverse(n));             // lane 0: a0 / n     lane 1: a1 / n
    return hi ? Fp30<P>::zero().template sub<2>(t) : t;           // lane 1: -a1 / n  (< 2p)
}
template <class P>
G16_HD Fp2p30<P> batch_inverse(const Fp2p30<P>& a) { return {pair_inverse<P>(Fp2p30<P>::lane_hi(), a.c, Fp2p30<P>::swap(a.c))}; }

// One pair of the batch.  Operands are affine points with CANONICAL coordinates (the window tables and every level's output
// list hold them so; a negated y is p - y), `pz` / `qz` flag the identity.
template <class F>
struct AffineBatch {
    enum { ADD = 0, DBL = 1, TAKE_P = 2, TAKE_Q = 3, ZERO = 4 };
    // what the pair needs and the denominator it contributes to the batch (never 0 mod p)
    G16_HD static int classify(bool pz, bool qz, const F& px, const F& py, const F& qx, const F& qy, F& d) {
        d = F::one();
        if (pz || qz) return pz ? (qz ? (int)ZERO : (int)TAKE_Q) : (int)TAKE_P;
        if (qx.equals_canonical(px)) {
            // same x: Q = -P (or a 2-torsion point, y = 0) gives the identity, otherwise Q = P and the tangent 3x^2 / 2y
            if (py.add(qy).is_zero_exact()) return ZERO;
            d = py.dbl();
            return DBL;
        }
        d = qx.template sub<2>(px);          // < 3p
        return ADD;
    }
    // inv_d = 1/d (any representative below 16p); result canonical
    G16_HD static void finish(int kind, const F& px, const F& py, const F& qx, const F& qy, const F& inv_d, F& rx, F& ry, bool& rz) {
        rz = false;
        if (kind >= TAKE_P) {
            if (kind == TAKE_P) { rx = px; ry = py; }
            else if (kind == TAKE_Q) { rx = qx; ry = qy; }
            else { rx = F::zero(); ry = F::zero(); rz = true; }
            return;
        }
        F num;
        if (kind == DBL) {
            const F x2 = px.sqr();
            num = x2.dbl().add(x2);              // 3 x^2 < 4.5p
        } else {
            num = qy.template sub<2>(py);        // < 3p
        }
        const F lam = num.mul(inv_d);                                                  // < 1.5p
        rx = lam.sqr().template sub<2>(px).template sub<2>(qx).canonical_lt8p();       // L^2 - x1 - x2 < 5.5p
        ry = lam.mul(px.template sub<2>(rx)).template sub<2>(py).canonical_lt8p();     // L (x1 - x3) - y1 < 3.5p
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// One level of the pairwise tree over the bucket-sorted, bucket-ALIGNED entry list (msm.hip pads every bucket to a multiple of
// 2^R slots, holes = the identity): the level-r list has S >> r slots, slot q of level r + 1 = slot 2q + slot 2q + 1 of level
// r, and no pair ever straddles a bucket.  Level 0 reads the sorted entry words and gathers from the window table, later
// levels read the previous level's list; all write canonical affine points (identity = zero words).
//
// Work split: a wave owns K * T consecutive output slots (T = tasks per wave: 64 lanes, or 32 lane pairs for G2), task t of
// the wave takes slots t, t + T, t + 2T, ...: at every step the wave touches T consecutive slots (2T consecutive inputs:
// whole cache lines).  Forward: per slot classify the pair and extend the running prefix product of the denominators,
// spilling each prefix to HBM (lane-coalesced: [wave][step][limb quad][lane]).  Then ONE inversion per task, then backward:
// 1/d_j = inv * prefix_(j-1), inv *= d_j, finish the addition, store the point.
struct alignas(16) Word4 { uint32_t w[4]; };
static constexpr uint32_t SORT_HOLE = 0xffffffffu;   // padding slot of the sorted entry list

template <class F>
struct AffineLevelArgs {
    const Affine<typename F::Std>* in;   // level 0: window table / bases, later: the previous level's list
    Affine<typename F::Std>* out;
    const uint32_t* sorted;              // level 0 only
    const uint32_t* total;               // device: S = padded entry count (offsets[M])
    Word4* prefix;                       // scratch: waves * K * QUADS * 64 records
    int64_t shift;                       // level 0 decode, as in the bucket kernel
    uint64_t base_count;
    uint32_t merged;
    uint32_t level;                      // inputs are level `level` slots (S >> level of them)
    uint32_t K;                          // steps per task
};

template <class F, bool LEVEL0>
struct AffineLevel {
    typedef Affine<typename F::Std> A;
    static constexpr int LPT = F::LANES_PER_TASK;
    static constexpr uint32_t T = 64 / LPT;                        // tasks per wave
    static constexpr int QUADS = (F::PREFIX_LIMBS + 3) / 4;

    G16_HD static bool load_operand(const AffineLevelArgs<F>& a, uint64_t slot, F& x, F& y) {   // false: the identity
        if constexpr (LEVEL0) {
            const uint32_t v = a.sorted[slot];
            if (v == SORT_HOLE) return false;
            const uint32_t pt = a.merged ? (v & 0x3ffffffu) : (v & 0x7fffffffu);
            const uint64_t row = a.merged ? (uint64_t)((v >> 26) & 31u) * a.base_count : 0;
            const int64_t idx = (int64_t)pt + a.shift;
            if (idx < 0 || (uint64_t)idx >= a.base_count) return false;
            if (!F::load_point(a.in, (int64_t)row + idx, x, y)) return false;
            if (v >> 31) y = y.neg_canonical();
            return true;
        } else {
            return F::load_point(a.in, (int64_t)slot, x, y);
        }
    }
    G16_HD static void store_prefix(const AffineLevelArgs<F>& a, uint32_t wave, uint32_t j, uint32_t lane, const F& v) {
        uint32_t w[4 * QUADS];
        G16_UNROLL for (int i = 0; i < 4 * QUADS; ++i) w[i] = 0;
        v.get_limbs(w);
        Word4* dst = a.prefix + ((uint64_t)wave * a.K + j) * (QUADS * 64) + lane;
        G16_UNROLL for (int c = 0; c < QUADS; ++c) {
            Word4 q;
            G16_UNROLL for (int i = 0; i < 4; ++i) q.w[i] = w[4 * c + i];
            dst[c * 64] = q;
        }
    }
    G16_HD static F load_prefix(const AffineLevelArgs<F>& a, uint32_t wave, uint32_t j, uint32_t lane) {
        uint32_t w[4 * QUADS];
        const Word4* src = a.prefix + ((uint64_t)wave * a.K + j) * (QUADS * 64) + lane;
        G16_UNROLL for (int c = 0; c < QUADS; ++c) {
            const Word4 q = src[c * 64];
            G16_UNROLL for (int i = 0; i < 4; ++i) w[4 * c + i] = q.w[i];
        }
        return F::from_limbs(w);
    }
    // `wave`: index of the wave in the grid; `lane`: 0..63.  Control flow is uniform across the lanes of a task.
    G16_HD static void run(const AffineLevelArgs<F>& a, uint32_t wave, uint32_t lane) {
        const uint64_t n_out = (uint64_t)(*a.total) >> (a.level + 1);
        const uint64_t first = (uint64_t)wave * a.K * T;
        if (first >= n_out) return;
        const uint64_t left = (n_out - first + T - 1) / T;             // steps with at least one live slot in this wave
        const uint32_t steps = left < a.K ? (uint32_t)left : a.K;
        const uint32_t task = lane / LPT;
        F pre = F::one();
        for (uint32_t j = 0; j < steps; ++j) {
            const uint64_t q = first + (uint64_t)j * T + task;
            F px = F::zero(), py = F::zero(), qx = F::zero(), qy = F::zero(), d;
            bool pz = true, qz = true;
            if (q < n_out) {
                pz = !load_operand(a, 2 * q, px, py);
                qz = !load_operand(a, 2 * q + 1, qx, qy);
            }
            (void)AffineBatch<F>::classify(pz, qz, px, py, qx, qy, d);
            pre = pre.mul(d);
            store_prefix(a, wave, j, lane, pre);
        }
        F inv = batch_inverse(pre);
        for (uint32_t j = steps; j-- > 0;) {
            const uint64_t q = first + (uint64_t)j * T + task;
            F px = F::zero(), py = F::zero(), qx = F::zero(), qy = F::zero(), d;
            bool pz = true, qz = true;
            if (q < n_out) {
                pz = !load_operand(a, 2 * q, px, py);
                qz = !load_operand(a, 2 * q + 1, qx, qy);
            }
            const F prev = j ? load_prefix(a, wave, j - 1, lane) : F::one();
            const int kind = AffineBatch<F>::classify(pz, qz, px, py, qx, qy, d);
            const F inv_d = inv.mul(prev);
            inv = inv.mul(d);
            F rx, ry;
            bool rz;
            AffineBatch<F>::finish(kind, px, py, qx, qy, inv_d, rx, ry, rz);
            if (q < n_out) F::store_point(a.out, (int64_t)q, rx, ry, rz);
        }
    }
};

}  // namespace g16
