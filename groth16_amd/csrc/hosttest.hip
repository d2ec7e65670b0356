// CPU self-test of the reduced-radix lazy arithmetic (fp30.hpp) against the standard field / group code.
// Host-only translation unit (kept apart from api.hip so that the two build in parallel); exported as g16_host_selftest.
#include "internal.hpp"
#include "fp30.hpp"
#include "batch_affine.hpp"
#include "window_tables.hpp"
#include "fixed_base.hpp"
#include <cstdlib>
#include <vector>

using namespace g16;

namespace g16 {
// Host emulation of the Fp2p30 lane pair: both lanes' components in one object, every product through the pair's own pure
// per-lane routines (pair_mul / pair_sqr / pair_inverse with hi = false and hi = true) -- what the two lanes of the G2 window-table
// task compute, without a wave.
template <class P>
struct Fp2e30 {
    typedef Fp30<P> B;
    typedef Fp2p30<P> PP;
    B c0, c1;
    static Fp2e30 one() { return {B::one(), B::zero()}; }
    Fp2e30 add(const Fp2e30& o) const { return {c0.add(o.c0), c1.add(o.c1)}; }
    Fp2e30 dbl() const { return {c0.dbl(), c1.dbl()}; }
    template <int K> Fp2e30 sub(const Fp2e30& o) const { return {c0.template sub<K>(o.c0), c1.template sub<K>(o.c1)}; }
    Fp2e30 mul(const Fp2e30& o) const { return {PP::pair_mul(false, c0, c1, o.c0, o.c1), PP::pair_mul(true, c1, c0, o.c1, o.c0)}; }
    Fp2e30 sqr() const { return {PP::pair_sqr(false, c0, c1), PP::pair_sqr(true, c1, c0)}; }
};
template <class P>
Fp2e30<P> batch_inverse(const Fp2e30<P>& a) { return {pair_inverse<P>(false, a.c0, a.c1), pair_inverse<P>(true, a.c1, a.c0)}; }

// Bound-tracking stand-in for the lazy field: a "value" is only an upper bound b (the value is < b p, per base-field component),
// every operation propagates the bound by the rule its real counterpart obeys and CHECKS the real counterpart's precondition:
//   sub<K>(o)   needs o < K p (the redundant K p never borrows)                            -> a + K
//   mul / sqr   need both operands below R' (they must fit NL limbs); with T = the sum of the sweeps' products the Montgomery
//               output (T + m p) / R' is below p (1 + T / (R' p))                        -> 1 + T / (R' p)
//               (fp30.hpp's "A B p / R' <= 0.5" is the sufficient condition for outputs < 1.5 p; the lane-pair squaring of the
//               9.5 p difference P exceeds it -- 484 of R'/p = 630 -- and yields < 1.77 p, which every later K still covers:
//               exactly what this propagation verifies)
//   exact zero tests need < 16 p; canonical_lt2p needs < 2 p
// PAIR: the lane-pair Fq2 -- a product is a0 b0 + a1 (16p - b1) (resp. a0 b1 + a1 b0): T <= A B + 16 A; a squaring multiplies
// (a0 + a1) by (a0 - a1 + 16p).  Plugged into Acc30 and jac30_double, the REAL formula code runs on bounds, so the invariants
// the kernels rely on ("x < 7.5p, y < 3.5p, ...") are derived and every K of every subtraction is checked, not argued.
struct BoundCtx {
    double ratio = 0;      // R' / p
    bool pair = false;
    const char* fail = nullptr;
    double worst_mul = 0, worst_operand = 0;
};
template <int TAG>
struct BoundF {
    typedef BoundF Std;
    typedef BoundF Raw;
    static BoundCtx& ctx() { static BoundCtx c; return c; }
    double b;
    static void check(bool ok, const char* what) { if (!ok && !ctx().fail) ctx().fail = what; }
    static BoundF zero() { return {0.0}; }
    static BoundF one() { return {1.0}; }
    BoundF add(const BoundF& o) const { return {b + o.b}; }
    BoundF dbl() const { return {2.0 * b}; }
    BoundF add_dbl(const BoundF& o) const { return {b + 2.0 * o.b}; }
    template <int K> BoundF sub(const BoundF& o) const { check(o.b <= K - 1e-3, "sub<K>: subtrahend not below K p"); return {b + K}; }
    BoundF neg2() const { check(b <= 2.0 - 1e-3 || b == 1.0, "neg2: operand not below 2p"); return {2.0}; }
    static BoundF cond_neg2(const BoundF& a, bool) { check(a.b <= 1.0, "cond_neg2: operand not canonical"); return {2.0}; }   // the worse of a and 2p - a
    static BoundF product(double A, double B, double T) {
        BoundCtx& c = ctx();
        check(A < c.ratio && B < c.ratio, "product operand does not fit NL limbs");
        if (T / c.ratio > c.worst_mul) c.worst_mul = T / c.ratio;
        if (A > c.worst_operand) c.worst_operand = A;
        if (B > c.worst_operand) c.worst_operand = B;
        return {1.0 + T / c.ratio};
    }
    BoundF mul(const BoundF& o) const { return ctx().pair ? product(b, o.b > 16.0 ? o.b : 16.0, b * o.b + b * 16.0) : product(b, o.b, b * o.b); }
    BoundF sqr() const { return ctx().pair ? product(2.0 * b, b + 16.0, 2.0 * b * (b + 16.0)) : product(b, b, b * b); }
    template <int K> BoundF mul_sub_k(const BoundF& o, const BoundF& s) const { return mul(o).template sub<K>(s); }
    BoundF sqr_sub_x3(const BoundF& u, const BoundF& v) const { return sqr().template sub<6>(u.add_dbl(v)); }
    static BoundF mul_sub(const BoundF& a, const BoundF& bb, const BoundF& c, const BoundF& d) { return a.mul(bb).template sub<2>(c.mul(d)); }
    // the fused a b - c d (one reduction over all sweeps): d enters as 2p - d (< 2p needed); one lane: a b + c (2p - d); lane pair:
    // a0 b0 + a1 (16p - b1) + c0 (2p - d0) + c1 d1  resp.  a0 b1 + a1 b0 + c0 (2p - d1) + c1 (2p - d0)
    static BoundF mul_sub_fused(const BoundF& a, const BoundF& bb, const BoundF& c, const BoundF& d) {
        check(d.b <= 2.0 - 1e-3, "mul_sub_fused: d not below 2p");
        const double A = a.b > c.b ? a.b : c.b;
#ifdef G16_PAIR_Y3_SPLIT
        if (ctx().pair) {   // two two-sweep products, summed lazily (fp30.hpp pair_mul_sub)
            const double Bb = bb.b > 16.0 ? bb.b : 16.0;
            return {product(a.b, Bb, a.b * bb.b + a.b * Bb).b + product(c.b, 2.0, 4.0 * c.b).b};
        }
#endif
        if (ctx().pair) return product(A, bb.b > 16.0 ? bb.b : 16.0, a.b * bb.b + a.b * (bb.b > 16.0 ? bb.b : 16.0) + 4.0 * c.b);
        return product(A, bb.b > 2.0 ? bb.b : 2.0, a.b * bb.b + 2.0 * c.b);
    }
    // a b + c d under one reduction (d < 2p); lane pair: a0 b0 + a1 (16p - b1) + c0 d0 + c1 (2p - d1)  resp.  a0 b1 + a1 b0 + c0 d1 + c1 d0
    static BoundF mul_add_fused(const BoundF& a, const BoundF& bb, const BoundF& c, const BoundF& d) {
        check(d.b <= 2.0 - 1e-3, "mul_add_fused: d not below 2p");
        const double A = a.b > c.b ? a.b : c.b;
        if (ctx().pair) return product(A, bb.b > 16.0 ? bb.b : 16.0, a.b * bb.b + a.b * (bb.b > 16.0 ? bb.b : 16.0) + c.b * d.b + 2.0 * c.b);
        return product(A, bb.b > d.b ? bb.b : d.b, a.b * bb.b + c.b * d.b);
    }
    // operand views (fp30.hpp): the lane pair's prepared second operand is its own component and the partner's, negated from 16p
    typedef BoundF Lhs;
    typedef BoundF Rhs;
    static BoundF lhs(const BoundF& a) { return a; }
    static BoundF rhs(const BoundF& a) { check(!ctx().pair || a.b <= 16.0 - 1e-3, "rhs: operand not below 16p"); return a; }
    static BoundF mul_v(const BoundF& a, const BoundF& b) { return a.mul(b); }
    static BoundF sqr_v(const BoundF& a) { return a.sqr(); }
    static BoundF sqr_sub_x3_v(const BoundF& a, const BoundF& u, const BoundF& v) { return a.sqr_sub_x3(u, v); }
    static BoundF mul_add_fused_v(const BoundF& a, const BoundF& bb, const BoundF& c, const BoundF& d) {
        const double A = a.b > c.b ? a.b : c.b;
        if (ctx().pair) {   // a0 b0 + a1 (16p - b1) + c0 d0 + c1 (16p - d1)
            const double Bb = bb.b > 16.0 ? bb.b : 16.0, Db = d.b > 16.0 ? d.b : 16.0;
            return product(A, Bb > Db ? Bb : Db, a.b * bb.b + a.b * Bb + c.b * d.b + c.b * Db);
        }
        return product(A, bb.b > d.b ? bb.b : d.b, a.b * bb.b + c.b * d.b);
    }
    BoundF settle() const { return *this; }
    bool maybe_zero() const { check(b < 16.0, "zero test on a value not below 16p"); return false; }
    bool is_zero_exact() const { check(b < 16.0, "zero test on a value not below 16p"); return false; }
    static constexpr int KM = 2, K2M = 4, KX = 8, KY = 4;
    static constexpr int LANES_PER_TASK = 1;
};
}  // namespace g16

namespace {

template <class C>
struct SelfTest {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    typedef typename C::G1A G1A;
    typedef typename C::G2A G2A;
    typedef typename C::G1X G1X;
    typedef typename C::G2X G2X;

    // ---------------------------------------------------------------------------------------
    // randomized CPU self-test of the 30-bit lazy arithmetic (fp30.hpp) against the standard field
    // and group code; returns 0 or the number of the first failing check
    static uint64_t sm_next(uint64_t& s) {
        s += 0x9E3779B97F4A7C15ULL;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    static Fq rand_fq(uint64_t& st) {
        Fq acc = Fq::zero();
        const Fq two64 = Fq::from_u64(1ULL << 32).sqr();
        for (int k = 0; k < 8; ++k) acc = acc * two64 + Fq::from_u64(sm_next(st));
        return acc;
    }
    typedef Fp30<typename Fq::Params> F30;
    static F30 to30(const Fq& x) { const Fq t = F30::std_to_r30(x); return F30::unpack(t.v); }
    // scalar field in the 30-bit form (NTT butterflies): products against w*R' tables, lazy DIT / DIF butterflies
    static Fr rand_fr(uint64_t& st) {
        Fr acc = Fr::zero();
        const Fr two64 = Fr::from_u64(1ULL << 32).sqr();
        for (int k = 0; k < 8; ++k) acc = acc * two64 + Fr::from_u64(sm_next(st));
        return acc;
    }
    static int selftest_fr30(uint64_t seed, int iters) {
        typedef Fp30<typename Fr::Params> R30;
        uint64_t st = seed ^ 0xF7;
        {   // canonical_quick at the edges: k p -> 0, k p - 1 -> p - 1, k p + 1 -> 1 for every k a pass can reach (and beyond)
            R30 pl, kp = R30::zero();
            for (int i = 0; i < R30::NL; ++i) pl.l[i] = Fr::Params::p30(i);
            R30 pm1 = pl;
            pm1.l[0] -= 1;   // p is odd: no borrow
            R30 one_raw = R30::zero();
            one_raw.l[0] = 1;
            for (int k = 0; k <= 5000; ++k) {
                if (k % 7 == 0 || k < 40 || (k & (k - 1)) == 0 || ((k + 1) & k) == 0) {
                    if (!kp.canonical_quick().same_limbs(R30::zero())) return 520;
                    if (!kp.add(pm1).canonical_quick().same_limbs(pm1)) return 521;
                    if (!kp.add(one_raw).canonical_quick().same_limbs(one_raw)) return 522;
                }
                kp = kp.add(pl);
            }
        }
        for (int it = 0; it < iters; ++it) {
            Fr x = rand_fr(st), y = rand_fr(st), w = rand_fr(st);
            if (it == 0) { x = Fr::zero(); }
            if (it == 1) { x = Fr::zero() - Fr::one(); y = x; w = x; }
            // data stays in the standard form x*R; tables are w*R'
            const R30 a = R30::unpack(x.v), b = R30::unpack(y.v);
            const Fr wt = R30::std_to_r30(w);
            const R30 w30 = R30::unpack(wt.v);
            auto canon = [](const R30& v) { Fr o; v.mul_impl(R30::one()).canonical_lt2p().pack(o.v); return o; };
            // canonical_quick (the NTT passes' write-back) against the product with R' mod p, on every lazy value below
            auto quick = [](const R30& v) { Fr o; v.canonical_quick().pack(o.v); return o; };
            if (!(canon(a.mul_impl(w30)) == x * w)) return 501;
            if (!(quick(a.mul_impl(w30)) == x * w) || !(quick(a) == x) || !(quick(R30::zero()) == Fr::zero())) return 505;
            {   // just below and at multiples of p, and the largest value a pass can hold (2^11 doublings of p - 1)
                R30 m = R30::unpack((Fr::zero() - Fr::one()).v);   // p - 1
                Fr want = Fr::zero() - Fr::one();
                for (int k = 0; k < 12; ++k) {
                    if (!(quick(m) == want) || !(quick(m.add(R30::unpack(Fr::one().v))) == want + Fr::one())) return 506;
                    m = m.dbl(); want = want + want;
                }
            }
            // DIT butterfly chain: 11 stages without reduction
            R30 u = a, v = b;
            Fr ru = x, rv = y;
            for (int k = 0; k < 11; ++k) {
                const R30 t = v.mul_impl(w30);
                const R30 nu = u.add(t), nv = u.template sub<2>(t);
                const Fr rt = rv * w;
                const Fr rnu = ru + rt, rnv = ru - rt;
                u = nu; v = nv; ru = rnu; rv = rnv;
            }
            if (!(canon(u) == ru) || !(canon(v) == rv)) return 502;
            if (!(quick(u) == ru) || !(quick(v) == rv)) return 507;
            // DIF butterfly chain: sums double every stage, subtraction adds 2^(k+1) p
            u = a; v = b; ru = x; rv = y;
            R30 s2 = a, d2 = b;
            for (int k = 0; k < 11; ++k) {
                const R30 nu = u.add(v);
                const R30 nv = u.sub_pow2(v, k).mul_impl(w30);
                const Fr rnu = ru + rv, rnv = (ru - rv) * w;
                // keep BOTH operands on the doubling lineage for the next stage (worst case for the bounds)
                s2 = nu; d2 = nv;
                u = nu; v = nu.add(nv).sub_pow2(nv, k + 1 < 12 ? k + 1 : 11);   // = nu, but with a grown bound
                ru = rnu; rv = rnu;
                if (!(canon(d2) == rnv)) return 503;
                if (!(quick(d2) == rnv) || !(quick(nu) == rnu) || !(quick(v) == rnu)) return 508;
            }
            if (!(canon(s2) == ru)) return 504;
        }
        return 0;
    }

    // fixed_base.hpp against plain double-and-add: table multiples of a random point and the 4-bit window, scalars incl. 0, 1,
    // r - 1 and all-ones bytes
    template <class X, class A>
    static int selftest_fixed_base(const A& gen, uint64_t seed, int first_code) {
        uint64_t st = seed ^ 0xFB;
        uint32_t k0[8];
        for (int i = 0; i < 8; ++i) k0[i] = (uint32_t)sm_next(st);
        k0[7] &= 0x0fffffffu;
        const X p = X::from_affine(gen).mul_bits(k0, 256);
        FixedBaseTable<X> tab;
        tab.build(p);
        for (int it = 0; it < 6; ++it) {
            uint32_t k[8];
            for (int i = 0; i < 8; ++i) k[i] = it == 0 ? 0u : it == 1 ? (i == 0 ? 1u : 0u) : it == 2 ? 0xffffffffu : (uint32_t)sm_next(st);
            if (it == 3) for (int i = 0; i < Fr::N; ++i) k[i] = Fr::Params::mod(i) - (i == 0 ? 1u : 0u);
            const auto want = p.mul_bits(k, 256).to_affine();
            if (!(tab.mul(k).to_affine() == want)) return first_code;
            if (!(mul_window4(p, k).to_affine() == want)) return first_code + 1;
        }
        return 0;
    }

    // Column-overflow check of the carry-free product routines (fp30.hpp: wide_mul / wide_relax / wide_redc): every routine is
    // run twice, with 64-bit and with 128-bit columns, on the worst limbs the representation allows (all 2^30 - 1) and on
    // random limbs.  The 128-bit run cannot overflow; identical limbs out mean the 64-bit run did not either.
    // WORST CASE over all inputs of every 64-bit column of the lazy product forms, by bound propagation through the very plan the
    // kernels are generated from (col_count, G16_RELAX_LIMIT): operand limbs at most 2^30 - 1, every reduction multiplier m_i at most
    // 2^30 - 1, the field's REAL modulus limbs.  (The all-ones-limbs shadow runs below exercise one m sequence; this covers them all.)
    // sweeps = operand sweeps in front of the reduction: 1 (mul, sqr), 2 (lane-pair product, a b - c d), 4 (lane-pair a b - c d).
    template <class R30>
    static int column_headroom() {
        typedef unsigned __int128 W;
        constexpr int NL = R30::NL;
        const W M = R30::MASK, LIM = (W)1 << 64;
        const int forms[3] = {1, 2, 4};
        for (int f = 0; f < 3; ++f) {
            const int sweeps = forms[f];
            W T[2 * NL];
            for (int c = 0; c < 2 * NL; ++c) T[c] = 0;
            for (int s = 1; s <= sweeps; ++s) {
                for (int c = 0; c < 2 * NL - 1; ++c) {
                    T[c] += (W)R30::col_count(c) * M * M;
                    if (T[c] >= LIM) return 605;
                }
                for (int c = 0; c + 1 < 2 * NL; ++c)     // wide_relax<s>
                    if ((s + 1) * R30::col_count(c) > G16_RELAX_LIMIT) {
                        T[c + 1] += (T[c] >> 32) << 2;
                        if (T[c + 1] >= LIM) return 606;
                        if (T[c] > 0xffffffffu) T[c] = 0xffffffffu;
                    }
            }
            W carry = 0;
            for (int i = 0; i < NL; ++i) {                // wide_redc
                T[i] += carry;
                if (T[i] >= LIM) return 607;
                for (int j = 0; j < NL; ++j) {
                    T[i + j] += M * (W)R30::Params_t::p30(j);
                    if (T[i + j] >= LIM) return 608;
                }
                carry = T[i] >> 30;
            }
            for (int j = 0; j < NL; ++j) {
                const W v = T[NL + j] + carry;
                if (v >= LIM) return 609;
                carry = v >> 30;
            }
        }
        return 0;
    }

    template <class R30>
    static int selftest_columns(uint64_t seed, int iters) {
        typedef unsigned __int128 U128;
        {
            const int rc = column_headroom<R30>();
            if (rc) return rc;
        }
        uint64_t st = seed ^ 0xC0;
        for (int it = 0; it < iters + 4; ++it) {
            R30 a, b, c, d;
            for (int i = 0; i < R30::NL; ++i) {
                const uint32_t worst = R30::MASK;
                a.l[i] = it < 2 ? worst : (uint32_t)sm_next(st) & R30::MASK;
                b.l[i] = (it == 0 || it == 2) ? worst : (uint32_t)sm_next(st) & R30::MASK;
                c.l[i] = it < 3 ? worst : (uint32_t)sm_next(st) & R30::MASK;
                d.l[i] = it < 4 ? worst : (uint32_t)sm_next(st) & R30::MASK;
            }
            if (!a.template mul_cols<uint64_t>(b).same_limbs(a.template mul_cols<U128>(b))) return 601;
            if (!a.template sqr_cols<uint64_t>().same_limbs(a.template sqr_cols<U128>())) return 602;
            // two- and four-sweep forms (Fq2 products, fused differences); d is fed as-is where the routine negates it
            {
                uint64_t T[2 * R30::NL];
                U128 W[2 * R30::NL];
                R30::wide_mul(T, a, b); R30::wide_mul(W, a, b);
                R30::template wide_relax<1>(T); R30::template wide_relax<1>(W);
                R30::wide_mul_add(T, c, d); R30::wide_mul_add(W, c, d);
                R30::template wide_relax<2>(T); R30::template wide_relax<2>(W);
                uint64_t T2[2 * R30::NL];
                U128 W2[2 * R30::NL];
                for (int k = 0; k < 2 * R30::NL; ++k) { T2[k] = T[k]; W2[k] = W[k]; }
                if (!R30::wide_redc(T2).same_limbs(R30::wide_redc(W2))) return 603;
                R30::wide_mul_add(T, b, c); R30::wide_mul_add(W, b, c);
                R30::template wide_relax<3>(T); R30::template wide_relax<3>(W);
                R30::wide_mul_add(T, a, d); R30::wide_mul_add(W, a, d);
                R30::template wide_relax<4>(T); R30::template wide_relax<4>(W);
                if (!R30::wide_redc(T).same_limbs(R30::wide_redc(W))) return 604;
            }
        }
        return 0;
    }

    // batched-affine arithmetic (batch_affine.hpp): the division-step inverse against Fermat's, the lane-pair Fq2 inverse, and
    // the tree levels themselves -- AffineLevel::run executed lane by lane on the CPU over a padded, bucket-sorted entry list
    // with every exceptional pair in it (identity bases, P + P at level 0 and at level 1, P + (-P), holes), compared bucket
    // by bucket with XYZZ accumulation in the standard field
    static int selftest_affine(uint64_t seed, int iters) {
        typedef ModInv30<typename Fq::Params> MI;
        uint64_t st = seed ^ 0xA1;
        for (int it = 0; it < iters; ++it) {
            Fq x = rand_fq(st);
            if (it == 0) x = Fq::one();
            if (it == 1) x = Fq::zero() - Fq::one();
            if (x.is_zero()) continue;
            F30 a = to30(x);
            if (it % 3 == 1) { const F30 z = F30::zero(); a = a.add(z.template sub<8>(z)); }   // a lazy representative, + 8p
            if (!(MI::inverse(a).to_std() == x.inverse())) return 6001;
            const Fq y = rand_fq(st);
            const Fq2 r = Fq2{x, y}.inverse();
            if (!(pair_inverse<typename Fq::Params>(false, to30(x), to30(y)).to_std() == r.c0)) return 6002;
            if (!(pair_inverse<typename Fq::Params>(true, to30(y), to30(x)).to_std() == r.c1)) return 6003;
        }
        const int NP = 150, NB = 90;
        std::vector<G1A> pts, pts30;
        const G1A gen = C::g1_generator();
        G1X runp = G1X::from_affine(gen);
        for (int i = 0; i < NP; ++i) {
            const uint32_t k[1] = {(uint32_t)(sm_next(st) | 1)};
            pts.push_back(runp.mul_bits(k, 32).to_affine());
            runp.add_affine(gen);
        }
        pts[7] = G1A::identity();
        pts[8] = G1A::identity();
        for (const auto& p : pts) pts30.push_back(G1A{F30::std_to_r30(p.x), F30::std_to_r30(p.y)});
        for (int R = 1; R <= 4; ++R) {
            for (uint32_t K : {1u, 8u}) {
                const uint32_t pad = 1u << R;
                std::vector<uint32_t> sorted, off(NB + 1);
                std::vector<G1X> ref(NB, G1X::identity());
                for (int b = 0; b < NB; ++b) {
                    off[b] = (uint32_t)sorted.size();
                    int m = (int)(sm_next(st) % 40);
                    if (b % 17 == 0) m = 0;
                    if (b == 5) m = 1;
                    std::vector<uint32_t> e;
                    for (int i = 0; i < m; ++i) e.push_back((uint32_t)(sm_next(st) % NP) | ((uint32_t)(sm_next(st) & 1) << 31));
                    if (m >= 4 && b % 3 == 0) e[1] = e[0];                                          // P + P
                    if (m >= 4 && b % 3 == 1) e[3] = e[2] ^ 0x80000000u;                            // P + (-P)
                    if (m >= 8 && b % 5 == 0) { e[4] = e[0]; e[5] = e[1]; e[6] = e[2]; e[7] = e[3]; }   // equal sums one level up
                    if (m >= 2 && b % 7 == 0) { e[0] = 7; e[1] = 8; }                               // identity bases
                    for (uint32_t v : e) {
                        sorted.push_back(v);
                        G1A q = pts[v & 0x7fffffffu];
                        if (v >> 31) q = q.neg();
                        ref[b].add_affine(q);
                    }
                    while (sorted.size() % pad) sorted.push_back(SORT_HOLE);
                }
                off[NB] = (uint32_t)sorted.size();
                const uint32_t S = (uint32_t)sorted.size();
                std::vector<G1A> la(S / 2 + 1), lb(S / 4 + 1);
                const uint32_t maxw = (S / 2 + K * 64 - 1) / (K * 64) + 1;
                std::vector<Word4> prefix((size_t)maxw * K * AffineLevel<F30, true>::QUADS * 64);
                AffineLevelArgs<F30> a;
                a.sorted = sorted.data(); a.total = &S; a.prefix = prefix.data(); a.shift = 0; a.base_count = NP; a.merged = 0; a.K = K;
                const G1A* cur = pts30.data();
                G1A* outs[2] = {la.data(), lb.data()};
                for (int lvl = 0; lvl < R; ++lvl) {
                    a.in = cur; a.out = outs[lvl & 1]; a.level = (uint32_t)lvl;
                    for (uint32_t w = 0; w < maxw; ++w)
                        for (uint32_t l = 0; l < 64; ++l) {
                            if (lvl == 0) AffineLevel<F30, true>::run(a, w, l);
                            else AffineLevel<F30, false>::run(a, w, l);
                        }
                    cur = a.out;
                }
                for (int b = 0; b < NB; ++b) {
                    G1X acc = G1X::identity();
                    for (uint32_t q = off[b] >> R; q < (off[b + 1] >> R); ++q) {
                        const G1A p = cur[q];
                        if (p.is_identity()) continue;
                        acc.add_affine(G1A{F30::unpack(p.x.v).to_std(), F30::unpack(p.y.v).to_std()});
                    }
                    if (!(acc.to_affine() == ref[b].to_affine())) return 6100 + R * 10 + (int)K;
                }
            }
        }
        return 0;
    }

    // ---------------------------------------------------------------------------------------
    // window tables (window_tables.hpp): the task's arithmetic -- Jacobian doublings in the lazy field, parked rows, ONE
    // inversion, canonical affine rows -- against c * j plain XYZZ doublings + to_affine of the standard code.  G1 on Fp30,
    // G2 on the two-component emulation of the lane pair.
    template <class E>
    struct TableHostIO {
        typedef E F;
        bool ident;
        F x0, y0;
        std::vector<F> park;
        std::vector<F> rx, ry;
        std::vector<int> rid;    // 1 = identity row, 0 = finite, -1 = never written
        bool load(F& x, F& y) const { if (ident) return false; x = x0; y = y0; return true; }
        void store(int row, const F& x, const F& y) { rx[row] = x; ry[row] = y; rid[row] = 0; }
        void store_identity(int row) { rid[row] = 1; }
        void put(int slot, const F& v) { park[slot] = v; }
        F get(int slot) const { return park[slot]; }
        static bool zero1(const F30& v) { return v.maybe_zero() && v.is_zero_exact(); }
        static bool is_zero_impl(const F30& v) { return zero1(v); }
        static bool is_zero_impl(const Fp2e30<typename Fq::Params>& v) { return zero1(v.c0) && zero1(v.c1); }
        bool is_zero(const F& v) const { return is_zero_impl(v); }
        static F30 canon_impl(const F30& v) { return v.canonical_lt2p(); }
        static Fp2e30<typename Fq::Params> canon_impl(const Fp2e30<typename Fq::Params>& v) { return {v.c0.canonical_lt2p(), v.c1.canonical_lt2p()}; }
        F canonical(const F& v) const { return canon_impl(v); }
    };
    static bool same30(const F30& a, const Fq& std_val) {   // lazy canonical limbs == canonical x R' of the standard value
        Fq got;
        a.pack(got.v);
        return got == F30::std_to_r30(std_val);
    }
    static int selftest_window_table(uint64_t seed) {
        typedef Fp2e30<typename Fq::Params> E2;
        uint64_t st = seed ^ 0x7AB1E;
        const int cases[4][2] = {{3, 1}, {5, 4}, {20, 13}, {19, 14}};   // (c, W)
        for (int cs = 0; cs < 4; ++cs) {
            const int c = cases[cs][0], W = cases[cs][1];
            for (int it = 0; it < 3; ++it) {
                uint32_t k[2] = {(uint32_t)sm_next(st) | 1u, (uint32_t)sm_next(st)};
                {   // G1
                    const G1A p = it == 2 ? G1A::identity() : G1X::from_affine(C::g1_generator()).mul_bits(k, 64).to_affine();
                    TableHostIO<F30> io;
                    io.ident = p.is_identity();
                    io.x0 = to30(p.x); io.y0 = to30(p.y);
                    io.park.assign((size_t)4 * W, F30::zero());
                    io.rx.assign(W, F30::zero()); io.ry.assign(W, F30::zero()); io.rid.assign(W, -1);
                    window_table_task<F30>(io, c, W);
                    G1X ref = G1X::from_affine(p);
                    for (int j = 0; j < W; ++j) {
                        if (j) for (int d = 0; d < c; ++d) ref = ref.dbl();
                        const G1A a = ref.to_affine();
                        if (io.rid[j] < 0) return 700;
                        if (a.is_identity() != (io.rid[j] == 1)) return 701;
                        if (!a.is_identity() && (!same30(io.rx[j], a.x) || !same30(io.ry[j], a.y))) return 702;
                    }
                }
                {   // G2, lane-pair emulation
                    const G2A p = it == 2 ? G2A::identity() : G2X::from_affine(C::g2_generator()).mul_bits(k, 64).to_affine();
                    TableHostIO<E2> io;
                    io.ident = p.is_identity();
                    io.x0 = E2{to30(p.x.c0), to30(p.x.c1)}; io.y0 = E2{to30(p.y.c0), to30(p.y.c1)};
                    const E2 z2 = {F30::zero(), F30::zero()};
                    io.park.assign((size_t)4 * W, z2);
                    io.rx.assign(W, z2); io.ry.assign(W, z2); io.rid.assign(W, -1);
                    window_table_task<E2>(io, c, W);
                    G2X ref = G2X::from_affine(p);
                    for (int j = 0; j < W; ++j) {
                        if (j) for (int d = 0; d < c; ++d) ref = ref.dbl();
                        const G2A a = ref.to_affine();
                        if (io.rid[j] < 0) return 710;
                        if (a.is_identity() != (io.rid[j] == 1)) return 711;
                        if (!a.is_identity() && (!same30(io.rx[j].c0, a.x.c0) || !same30(io.rx[j].c1, a.x.c1) || !same30(io.ry[j].c0, a.y.c0) ||
                                                 !same30(io.ry[j].c1, a.y.c1)))
                            return 712;
                    }
                }
            }
        }
        return 0;
    }

    // ---------------------------------------------------------------------------------------
    // lazy-arithmetic invariants by bound propagation (BoundF above): the mixed addition / full addition / doubling of the bucket and
    // reduction kernels iterated to a fixed point from the worst admissible inputs, and the Jacobian doubling of the table builder;
    // G1 rules and lane-pair rules, this curve's R' / p.  Returns 0 or a code; *report gets the failing precondition.
    template <int TAG>
    static int bounds_case(bool pair, const char** report) {
        typedef BoundF<TAG> B;
        BoundCtx& c = B::ctx();
        c = BoundCtx();
        double ratio = 1.0;   // R' / p = 2^(30 NL) / p from the limbs of p
        {
            double pv = 0.0;
            for (int i = F30::NL - 1; i >= 0; --i) pv = pv * 1073741824.0 + (double)Fq::Params::p30(i);
            double rp = 1.0;
            for (int i = 0; i < F30::NL; ++i) rp *= 1073741824.0;
            ratio = rp / pv;
        }
        c.ratio = ratio;
        c.pair = pair;
        typedef Acc30<B> A;
        // mixed additions: px canonical (< p), py canonical or 2p - y (< 2p)
        A acc = A::identity();
        double mx[4] = {0, 0, 0, 0};
        for (int it = 0; it < 200; ++it) {
            acc.add_affine(B{1.0}, B{2.0});
            const double v[4] = {acc.x.b, acc.y.b, acc.zz.b, acc.zzz.b};
            for (int k = 0; k < 4; ++k) if (v[k] > mx[k]) mx[k] = v[k];
        }
        if (c.fail) { *report = c.fail; return 1; }
        {   // the parked form (bucket pass): same formulas in another order -- same preconditions, same fixed point
            AccParked<B, ParkedArrayStore<B>> pk;
            pk.set_identity();
            for (int it = 0; it < 200; ++it) {
                pk.add_affine_signed(B{1.0}, B{1.0}, (it & 2) != 0);   // canonical y in; cond_neg2 propagates 2p whatever the signs
                const A g = pk.gather_as_parked();
                const double v[4] = {g.x.b, g.y.b, g.zz.b, g.zzz.b};
                for (int k = 0; k < 4; ++k) if (v[k] > mx[k]) mx[k] = v[k];
                (void)pk.gather();   // the flush's 4p - y: its precondition (y < 4p) is checked here
            }
            AccParked<B, ParkedArrayStore<B>> pd;
            pd.set_identity();
            pd.set_double(B{1.0}, B{2.0});
            if (c.fail) { *report = c.fail; return 7; }
        }
        // the invariants fp30.hpp documents for the accumulator
        if (!(mx[0] < 7.5 && mx[1] < 3.5 && mx[2] < 1.8 && mx[3] < 1.8)) { *report = "accumulator bounds above the documented ones"; return 2; }
        // full additions, doublings, mixed doubling on accumulators at those bounds (reduction kernels, heavy combine, P + P branch)
        // (a partial sum leaves the pass with y or 4p - y: the reductions' operands are taken at max(y bound, 4p))
        const double y_out = mx[1] > 4.0 ? mx[1] : 4.0;
        A a1; a1.x = B{mx[0]}; a1.y = B{y_out}; a1.zz = B{mx[2]}; a1.zzz = B{mx[3]}; a1.inf = false;
        A a2 = a1;
        for (int it = 0; it < 50; ++it) { a1.add(a2); a2 = a1; a2.dbl(); }
        A a3 = A::identity();
        a3.set_double(B{1.0}, B{2.0});
        if (c.fail) { *report = c.fail; return 3; }
        {   // the reductions' streamed addition / doubling on operands at those bounds, iterated like the chain above
            ParkedArrayStore<B> ds, ss;
            ds.v[0] = B{mx[0]}; ds.v[1] = B{y_out}; ds.v[2] = B{mx[2]}; ds.v[3] = B{mx[3]};
            ss = ds;
            bool dinf = false;
            for (int it = 0; it < 50; ++it) {
                acc_add_streamed<B>(ds, dinf, ss, false);
                ss = ds;
                bool sinf = false;
                acc_dbl_streamed<B>(ss, sinf);
            }
            if (c.fail) { *report = c.fail; return 8; }
        }
        // window-table builder: Jacobian doublings from an affine point
        B X{1.0}, Y{1.0}, Z{1.0};
        double jm[3] = {0, 0, 0};
        for (int it = 0; it < 400; ++it) {
            jac30_double(X, Y, Z);
            if (X.b > jm[0]) jm[0] = X.b;
            if (Y.b > jm[1]) jm[1] = Y.b;
            if (Z.b > jm[2]) jm[2] = Z.b;
        }
        if (c.fail) { *report = c.fail; return 4; }
        if (!(jm[0] < 5.8 && jm[1] < 5.5 && jm[2] < 3.0)) { *report = "Jacobian doubling bounds above the documented ones"; return 5; }
        // rows leave the builder through canonical_lt2p of a product whose operands are X (resp. Y) and an inverse power (< 2p)
        if (getenv("G16_SELFTEST_VERBOSE"))
            fprintf(stderr, "g16 self-test bounds (%s, R'/p = %.0f): accumulator x < %.2fp y < %.2fp zz < %.2fp zzz < %.2fp; Jacobian X < %.2fp Y < %.2fp "
                            "Z < %.2fp; largest product T / (R' p) = %.3f, largest operand %.1fp\n",
                    pair ? "lane pair" : "one lane", ratio, mx[0], mx[1], mx[2], mx[3], jm[0], jm[1], jm[2], c.worst_mul, c.worst_operand);
        const B ax = X.mul(B{2.0}), ay = Y.mul(B{2.0});
        if (c.fail || !(ax.b < 2.0 && ay.b < 2.0)) { *report = c.fail ? c.fail : "table row not below 2p before canonical_lt2p"; return 6; }
        return 0;
    }
    static int selftest_bounds() {
        const char* why = nullptr;
        int rc = bounds_case<0>(false, &why);
        if (rc) { fprintf(stderr, "g16 self-test: G1 lazy bounds: %s\n", why); return 800 + rc; }
        rc = bounds_case<1>(true, &why);
        if (rc) { fprintf(stderr, "g16 self-test: lane-pair lazy bounds: %s\n", why); return 810 + rc; }
        return 0;
    }

    static int selftest30(uint64_t seed, int iters) {
        uint64_t st = seed;
        {
            int rc = selftest_columns<Fp30<typename Fr::Params>>(seed, iters);
            if (rc) return rc;
            rc = selftest_fixed_base<G1X>(C::g1_generator(), seed, 611);
            if (rc) return rc;
            rc = selftest_fixed_base<G2X>(C::g2_generator(), seed, 613);
            if (rc) return rc;
            rc = selftest_columns<F30>(seed, iters);
            if (rc) return rc;
            rc = selftest_fr30(seed, iters);
            if (rc) return rc;
        }
        {
            const int rc = selftest_affine(seed, iters);
            if (rc) return rc;
        }
        {
            const int rc = selftest_window_table(seed);
            if (rc) return rc;
        }
        {
            const int rc = selftest_bounds();
            if (rc) return rc;
        }
        for (int it = 0; it < iters; ++it) {
            Fq x = rand_fq(st), y = rand_fq(st);
            if (it == 0) { x = Fq::zero(); }
            if (it == 1) { x = Fq::zero() - Fq::one(); y = x; }
            if (it == 2) { x = Fq::one(); }
            const F30 a = to30(x), b = to30(y);
            {   // pack/unpack round trip
                Fq t = F30::std_to_r30(x), u;
                F30::unpack(t.v).pack(u.v);
                if (!(t == u)) return 1;
            }
            if (!(a.mul(b).to_std() == x * y)) return 2;
            if (!(a.add(b).to_std() == x + y)) return 3;
            if (!(a.template sub<2>(b).to_std() == x - y)) return 4;
            if (!(a.template sub<8>(b).to_std() == x - y)) return 5;
            if (!(a.neg2().to_std() == x.neg())) return 6;
            // lazy chains: products of loosely reduced operands (< 16p)
            const F30 big1 = a.add(b).add(a).template sub<8>(b);        // 2a, bound < 12p
            const F30 big2 = b.template sub<8>(a).add(b);               // 2b - a, bound < 11p
            if (!(big1.mul(big2).to_std() == (x + x) * (y + y - x))) return 7;
            if (!(big1.sqr().to_std() == (x + x).sqr())) return 8;
            // zero tests
            const F30 z0 = a.template sub<8>(a);                          // == 8p
            if (!z0.maybe_zero() || !z0.is_zero_exact()) return 9;
            const F30 nz = a.template sub<8>(a).add(F30::one());
            if (nz.is_zero_exact()) return 10;
            if (a.is_zero_exact() != x.is_zero()) return 11;
        }
        // accumulator against the generic XYZZ code, including doubling and cancellation
        const G1A gen = C::g1_generator();
        std::vector<G1A> pts;
        G1X run = G1X::from_affine(gen);
        for (int i = 0; i < 24; ++i) {
            uint32_t k[2] = {(uint32_t)sm_next(st) | 1u, 0};
            pts.push_back(run.mul_bits(k, 32).to_affine());
            run.add_affine(gen);
        }
        for (int round = 0; round < 4; ++round) {
            std::vector<G1A> seq;
            for (int i = 0; i < 40; ++i) seq.push_back(pts[sm_next(st) % pts.size()]);
            if (round == 1) { seq[1] = seq[0]; }                          // P + P -> doubling branch
            if (round == 2) { seq[1] = seq[0].neg(); }                    // P - P -> identity, then keep adding
            if (round == 3) { seq[3] = seq[0]; seq[2] = seq[1]; seq[5] = seq[4].neg(); }
            Acc30<F30> acc = Acc30<F30>::identity();
            AccParked<F30, ParkedArrayStore<F30>> park;   // the bucket pass's form: coordinates outside the registers, re-ordered products
            park.set_identity();
            G1X ref = G1X::identity();
            for (size_t i = 0; i < seq.size(); ++i) {
                {   // the packed conditional negation against the limb form's
                    const Fq yw = F30::std_to_r30(seq[i].y);
                    if (!F30::unpack_cond_neg(yw, true).same_limbs(F30::unpack(yw.v).neg2()) || !F30::unpack_cond_neg(yw, false).same_limbs(F30::unpack(yw.v)))
                        return 5090;
                }
                acc.add_affine(to30(seq[i].x), to30(seq[i].y));
                if (i & 1) park.add_affine_packed(to30(seq[i].x), F30::std_to_r30(seq[i].y), false);   // both entries of the parked form
                else park.add_affine(to30(seq[i].x), to30(seq[i].y));
                ref.add_affine(seq[i]);
                const G1A got = acc.to_std().to_affine(), want = ref.to_affine();
                if (!(got == want)) return 100 + round * 100 + (int)i;
                if (!(park.gather().to_std().to_affine() == want)) return 5000 + round * 100 + (int)i;
            }
        }
        // ---- full additions / doublings / small multiples / packed storage of the lazy accumulator (G1)
        {
            typedef Acc30<F30> A30;
            auto lift = [&](const G1X& p) { return A30::from_packed(p.is_identity() ? G1X::identity()
                                                : G1X{F30::std_to_r30(p.x), F30::std_to_r30(p.y), F30::std_to_r30(p.zz), F30::std_to_r30(p.zzz)}); };
            for (int it = 0; it < 12; ++it) {
                G1X a = G1X::identity(), b = G1X::identity();
                for (int i = 0; i < 3; ++i) { a.add_affine(pts[sm_next(st) % pts.size()]); b.add_affine(pts[sm_next(st) % pts.size()]); }
                if (it == 1) b = a;
                if (it == 2) b = a.neg();
                if (it == 3) b = G1X::identity();
                if (it == 4) a = G1X::identity();
                A30 la = lift(a), lb = lift(b);
                A30 sum = la; sum.add(lb);
                G1X rs = a; rs.add(b);
                if (!(sum.to_std().to_affine() == rs.to_affine())) return 40 + it;
                {   // the reductions' streamed form of the same addition (both operands behind stores), and of the doubling
                    ParkedArrayStore<F30> ds, ss;
                    ds.v[0] = la.x; ds.v[1] = la.y; ds.v[2] = la.zz; ds.v[3] = la.zzz;
                    ss.v[0] = lb.x; ss.v[1] = lb.y; ss.v[2] = lb.zz; ss.v[3] = lb.zzz;
                    bool dinf = la.inf;
                    acc_add_streamed<F30>(ds, dinf, ss, lb.inf);
                    A30 got; got.inf = dinf; got.x = ds.v[0]; got.y = ds.v[1]; got.zz = ds.v[2]; got.zzz = ds.v[3];
                    if (!(got.to_std().to_affine() == rs.to_affine())) return 5800 + it;
                    ParkedArrayStore<F30> d2;
                    d2.v[0] = la.x; d2.v[1] = la.y; d2.v[2] = la.zz; d2.v[3] = la.zzz;
                    bool d2inf = la.inf;
                    acc_dbl_streamed<F30>(d2, d2inf);
                    A30 g2; g2.inf = d2inf; g2.x = d2.v[0]; g2.y = d2.v[1]; g2.zz = d2.v[2]; g2.zzz = d2.v[3];
                    if (!(g2.to_std().to_affine() == a.dbl().to_affine())) return 5850 + it;
                }
                // storage round trip keeps the group element
                if (!(A30::from_packed(sum.to_packed()).to_std().to_affine() == rs.to_affine())) return 60 + it;
                A30 d = la; d.dbl();
                if (!(d.to_std().to_affine() == a.dbl().to_affine())) return 80 + it;
                const uint32_t k = (uint32_t)(sm_next(st) % 40000u);
                uint32_t kw[1] = {k};
                if (!(la.mul_small(k).to_std().to_affine() == a.mul_bits(kw, 32).to_affine())) return 90;
                // chained lazy adds (bounds must hold across many operations)
                A30 chain = la;
                G1X rchain = a;
                for (int j = 0; j < 6; ++j) { chain.add(lb); rchain.add(b); chain.add(chain); rchain.add(rchain); }
                if (!(chain.to_std().to_affine() == rchain.to_affine())) return 95;
            }
        }
        // ---- Fq2 over the 30-bit field, and the G2 accumulator
        typedef Fp2x30<typename Fq::Params> F230;
        for (int it = 0; it < iters / 4 + 4; ++it) {
            Fq2 x = {rand_fq(st), rand_fq(st)}, y = {rand_fq(st), rand_fq(st)};
            if (it == 0) x = Fq2::zero();
            if (it == 1) { x = {Fq::zero() - Fq::one(), Fq::zero() - Fq::one()}; y = x; }
            const F230 a = {to30(x.c0), to30(x.c1)}, b = {to30(y.c0), to30(y.c1)};
            if (!(a.mul_impl(b).to_std() == x.mul_inlined(y))) return 20;
            if (!(a.sqr_impl().to_std() == x.sqr_inlined())) return 21;
            if (!(a.mul(b).to_std() == x * y)) return 22;
            const F230 big1 = a.add(b).add(a).template sub<8>(b), big2 = b.template sub<8>(a).add(b);  // 2a (<12p), 2b - a (<11p)
            if (!(big1.mul_impl(big2).to_std() == (x + x) * (y + y - x))) return 23;
            if (!(big1.sqr_impl().to_std() == (x + x).sqr())) return 24;
            if (a.is_zero_exact() != x.is_zero()) return 25;
            if (!a.template sub<8>(a).is_zero_exact()) return 26;
        }
        // ---- lane-pair Fq2: the pure per-lane kernels against the one-lane Fq2 product
        typedef Fp2p30<typename Fq::Params> FP;
        for (int it = 0; it < iters / 4 + 4; ++it) {
            Fq2 x = {rand_fq(st), rand_fq(st)}, y = {rand_fq(st), rand_fq(st)};
            if (it == 0) x = Fq2::zero();
            if (it == 1) { x = {Fq::zero() - Fq::one(), Fq::zero() - Fq::one()}; y = x; }
            F30 a0 = to30(x.c0), a1 = to30(x.c1), b0 = to30(y.c0), b1 = to30(y.c1);
            if (it & 1) {   // loosely reduced operands (< 12p)
                a0 = a0.add(b0).add(a0).template sub<8>(b0); a1 = a1.add(b1).add(a1).template sub<8>(b1);
                x = x + x;
            }
            const Fq2 want_m = x * y, want_s = x.sqr();
            if (!(FP::pair_mul(false, a0, a1, b0, b1).to_std() == want_m.c0)) return 36;
            if (!(FP::pair_mul(true, a1, a0, b1, b0).to_std() == want_m.c1)) return 37;
            if (!(FP::pair_sqr(false, a0, a1).to_std() == want_s.c0)) return 38;
            if (!(FP::pair_sqr(true, a1, a0).to_std() == want_s.c1)) return 39;
            // the four-sweep a b - c d of the parked bucket accumulator (d canonical: < 2p as the routine requires)
            const Fq2 z = {rand_fq(st), rand_fq(st)}, w = {rand_fq(st), rand_fq(st)};
            const F30 c0 = to30(z.c0).add(b0), c1 = to30(z.c1).add(b1), d0 = to30(w.c0), d1 = to30(w.c1);   // c lazy: z + y
            const Fq2 want_ms = x * y - (z + y) * w;
            if (!(FP::pair_mul_sub(false, a0, a1, b0, b1, c0, c1, d0, d1).to_std() == want_ms.c0)) return 3601;
            if (!(FP::pair_mul_sub(true, a1, a0, b1, b0, c1, c0, d1, d0).to_std() == want_ms.c1)) return 3602;
            if (!(F30::template mul_sub_cols<uint64_t>(a0, b0, c0, d0).to_std() == x.c0 * y.c0 - (z.c0 + y.c0) * w.c0)) return 3603;
        }
        // ---- the bucket kernel's Karatsuba Fq2 (register-passed products, settled accumulator)
        typedef Fp2k30<typename Fq::Params> FK;
        for (int it = 0; it < iters / 4 + 4; ++it) {
            Fq2 x = {rand_fq(st), rand_fq(st)}, y = {rand_fq(st), rand_fq(st)};
            if (it == 0) x = Fq2::zero();
            if (it == 1) { x = {Fq::zero() - Fq::one(), Fq::zero() - Fq::one()}; y = x; }
            const FK a = {to30(x.c0), to30(x.c1)}, b = {to30(y.c0), to30(y.c1)};
            if (!(a.mul(b).to_std() == x * y)) return 30;
            if (!(a.sqr().to_std() == x.sqr())) return 31;
            const FK big1 = a.add(b).add(a).template sub<8>(b), big2 = b.template sub<8>(a).add(b);  // 2a (<12p), 2b - a (<11p)
            if (!(big1.mul(big2).to_std() == (x + x) * (y + y - x))) return 32;
            if (!(big1.sqr().to_std() == (x + x).sqr())) return 33;
            const FK wide = big1.add(big1).template sub<16>(b);      // 4a - b, bound < 30p
            if (!(wide.settle().to_std() == (x + x + x + x - y))) return 34;
            if (!(wide.settle().mul(b).to_std() == (x + x + x + x - y) * y)) return 35;
        }
        {
            const G2A gen2 = C::g2_generator();
            std::vector<G2A> pts2;
            G2X run2 = G2X::from_affine(gen2);
            for (int i = 0; i < 12; ++i) {
                uint32_t k[2] = {(uint32_t)sm_next(st) | 1u, 0};
                pts2.push_back(run2.mul_bits(k, 32).to_affine());
                run2.add_affine(gen2);
            }
            for (int round = 0; round < 3; ++round) {
                std::vector<G2A> seq;
                for (int i = 0; i < 24; ++i) seq.push_back(pts2[sm_next(st) % pts2.size()]);
                if (round == 1) { seq[1] = seq[0]; }
                if (round == 2) { seq[1] = seq[0].neg(); seq[4] = seq[3]; }
                Acc30<F230> acc = Acc30<F230>::identity();
                AccParked<F230, ParkedArrayStore<F230>> park2;
                park2.set_identity();
                G2X ref = G2X::identity();
                for (size_t i = 0; i < seq.size(); ++i) {
                    const F230 px = {to30(seq[i].x.c0), to30(seq[i].x.c1)}, py = {to30(seq[i].y.c0), to30(seq[i].y.c1)};
                    acc.add_affine(px, py);
                    park2.add_affine(px, py);
                    ref.add_affine(seq[i]);
                    if (!(acc.to_std().to_affine() == ref.to_affine())) return 1000 + round * 100 + (int)i;
                    if (!(park2.gather().to_std().to_affine() == ref.to_affine())) return 5500 + round * 100 + (int)i;
                }
                {   // the same sequence through the Karatsuba accumulator used by the G2 bucket kernel
                    Acc30<FK> ak = Acc30<FK>::identity();
                    G2X rk = G2X::identity();
                    for (size_t i = 0; i < seq.size(); ++i) {
                        const FK px = {to30(seq[i].x.c0), to30(seq[i].x.c1)};
                        FK py = {to30(seq[i].y.c0), to30(seq[i].y.c1)};
                        G2A q = seq[i];
                        if (i & 1) { py = py.neg2(); q = q.neg(); }
                        ak.add_affine(px, py);
                        rk.add_affine(q);
                        if (!(ak.to_std().to_affine() == rk.to_affine())) return 3000 + round * 100 + (int)i;
                        if (!(Acc30<FK>::from_packed(ak.to_packed()).to_std().to_affine() == rk.to_affine())) return 3500 + round * 100 + (int)i;
                    }
                }
                // full add / dbl / small multiple on G2
                Acc30<F230> other = Acc30<F230>::from_packed(acc.to_packed());
                G2X ro = ref;
                other.dbl(); ro = ro.dbl();
                {   // streamed forms on G2: (2 ref) + ref, ref + ref (the doubling branch), ref + (-ref)
                    auto to_store = [](const Acc30<F230>& a) { ParkedArrayStore<F230> st; st.v[0] = a.x; st.v[1] = a.y; st.v[2] = a.zz; st.v[3] = a.zzz; return st; };
                    auto from_store = [](const ParkedArrayStore<F230>& st, bool inf) { Acc30<F230> a; a.inf = inf; a.x = st.v[0]; a.y = st.v[1]; a.zz = st.v[2]; a.zzz = st.v[3]; return a; };
                    ParkedArrayStore<F230> ds = to_store(other), ss = to_store(acc);
                    bool dinf = other.inf;
                    acc_add_streamed<F230>(ds, dinf, ss, acc.inf);
                    G2X want3 = ro; want3.add(ref);
                    if (!(from_store(ds, dinf).to_std().to_affine() == want3.to_affine())) return 5900 + round;
                    ParkedArrayStore<F230> same = to_store(acc);
                    bool sinf = acc.inf;
                    acc_add_streamed<F230>(same, sinf, ss, acc.inf);
                    if (!(from_store(same, sinf).to_std().to_affine() == ro.to_affine())) return 5910 + round;
                    Acc30<F230> neg = acc;
                    neg.y = neg.y.neg2();   // y < 3.5p here: neg2 needs < 2p -> go through the canonical form
                    neg = Acc30<F230>::from_packed(acc.to_packed());
                    neg.y = neg.y.neg2();
                    ParkedArrayStore<F230> ns = to_store(neg), as = to_store(Acc30<F230>::from_packed(acc.to_packed()));
                    bool ninf = false;
                    acc_add_streamed<F230>(as, ninf, ns, false);
                    if (!acc.inf && !ninf) return 5920 + round;
                }
                other.add(acc); ro.add(ref);
                if (!(other.to_std().to_affine() == ro.to_affine())) return 2000 + round;
                uint32_t kw[1] = {12345u + (uint32_t)round};
                if (!(acc.mul_small(kw[0]).to_std().to_affine() == ref.mul_bits(kw, 32).to_affine())) return 2100 + round;
                {   // the reductions' use of the Karatsuba accumulator: raw-limb hand-over from the bucket pass, full
                    // additions, doublings and small multiples chained on lazy values
                    typedef AccRaw<F230> Raw;
                    Raw slot;
                    acc.store_raw(&slot);
                    Acc30<FK> k1 = Acc30<FK>::load_raw(slot), k2 = Acc30<FK>::load_raw(slot);
                    G2X r1 = ref, r2 = ref;
                    for (int j = 0; j < 6; ++j) {
                        k1.dbl(); r1 = r1.dbl();
                        k1.add(k2); r1.add(r2);
                        k2.add(k1); r2.add(r1);
                        Raw tmp;
                        k2.store_raw(&tmp);
                        k2 = Acc30<FK>::load_raw(tmp);
                        if (!(k1.to_std().to_affine() == r1.to_affine())) return 4000 + round * 10 + j;
                        if (!(k2.to_std().to_affine() == r2.to_affine())) return 4100 + round * 10 + j;
                    }
                    Acc30<FK> same = k1;
                    same.add(k1); r2 = r1; r2.add(r1);          // equal operands -> doubling branch
                    if (!(same.to_std().to_affine() == r2.to_affine())) return 4200 + round;
                    if (!(k1.mul_small(kw[0]).to_std().to_affine() == r1.mul_bits(kw, 32).to_affine())) return 4300 + round;
                    if (!(k1.mul_small(32760u).to_std().to_affine() == r1.mul_bits((const uint32_t[]){32760u}, 16).to_affine())) return 4400 + round;
                }
            }
        }
        return 0;
    }
};

}  // namespace

extern "C" int g16_host_selftest(int curve, uint64_t seed, int iters) {
    if (curve == G16_BLS12_381) return SelfTest<Bls12_381>::selftest30(seed, iters);
    if (curve == G16_BN254) return SelfTest<Bn254>::selftest30(seed, iters);
    return -1;
}
