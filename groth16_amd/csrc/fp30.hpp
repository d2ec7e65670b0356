// Reduced-radix (30-bit limb) Montgomery arithmetic for the bucket-accumulation kernel.
//
// Why: measured on MI355X (profiles/r01_ubench.txt) v_mad_u64_u32 issues in ~4-5 cycles per
// wave, and so does every carry-handling instruction (v_add_co/v_addc, v_lshl_add_u64).  With
// saturated 32-bit limbs each limb product drags 1-2 carry instructions plus register moves,
// and the 288 multiply-adds of a BLS12-381 Fq product end up as <20 % of its issue cycles.
// With 30-bit limbs a column of up to 13 limb products (< 13 * 2^60) fits a 64-bit accumulator,
// so a product is exactly NL^2 + NL^2 v_mad_u64_u32 with NO carry instruction inside the two
// accumulation phases, and carries are resolved once per column (shift/mask).
//
// Representation: NL limbs of 30 bits (NL = 13 for the 381-bit BLS12-381 Fq, 9 for 254/255-bit
// fields), Montgomery radix R' = 2^(30 NL).  R'/p >= 2^9 gives headroom for LAZY arithmetic:
// values are kept only "loosely" reduced (bounds noted per operation, as multiples of p); a
// product of inputs < 16p comes out < 1.5p with no final subtraction.  Exact reduction happens
// only when a value leaves the kernel (canonical()), or in the (rare) exceptional-case tests.
//
// Memory format of bases handed to the kernel: canonical x*R' mod p packed into the usual
// 32-bit words (same size as the arkworks form; converted once at g16_pk_load time).
#pragma once
#include "field.hpp"
#include "fips_asm_gen.hpp"

namespace g16 {

template <class P>
struct Fp30 {
    typedef P Params_t;
    static constexpr int NL = P::NL30;
    static constexpr int NW = P::N;  // 32-bit words of the packed form
    static constexpr uint32_t MASK = (1u << 30) - 1u;
    uint32_t l[NL];

    G16_HD static Fp30 zero() {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = 0;
        return r;
    }
    G16_HD static Fp30 one() {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = P::one30(i);
        return r;
    }
    // packed little-endian words -> limbs (value unchanged)
    G16_HD static Fp30 unpack(const uint32_t* w) {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const int bit = 30 * i, wd = bit >> 5, sh = bit & 31;
            uint32_t v = w[wd] >> sh;
            if (sh > 2 && wd + 1 < NW) v |= w[wd + 1] << (32 - sh);
            r.l[i] = (i == NL - 1) ? v : (v & MASK);
        }
        return r;
    }
    // limbs (normalised, value < 2^(32 NW)) -> packed words
    G16_HD void pack(uint32_t* w) const {
        G16_UNROLL for (int k = 0; k < NW; ++k) w[k] = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const int bit = 30 * i, wd = bit >> 5, sh = bit & 31;
            w[wd] |= l[i] << sh;
            if (sh > 2 && wd + 1 < NW) w[wd + 1] |= l[i] >> (32 - sh);
        }
    }
    // carry-propagate so that every limb but the top is < 2^30 (limbs may be up to 2^32 - 1 before)
    G16_HD void normalize() {
        G16_UNROLL for (int i = 0; i + 1 < NL; ++i) {
            l[i + 1] += l[i] >> 30;
            l[i] &= MASK;
        }
    }
    // a + b          bound: A + B
    G16_HD Fp30 add(const Fp30& b) const {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = l[i] + b.l[i];
        r.normalize();
        return r;
    }
    G16_HD Fp30 dbl() const { return add(*this); }
    // a + K p - b    requires b < K p (roughly: b's top limb <= top(K p) - 1); bound: A + K
    template <int K>
    G16_HD Fp30 sub(const Fp30& b) const {
        static_assert(K == 2 || K == 4 || K == 6 || K == 8 || K == 16, "redundant K p tables exist for K = 2, 4, 6, 8, 16 only");
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const uint32_t kp = K == 2 ? P::kp2(i) : K == 4 ? P::kp4(i) : K == 6 ? P::kp6(i) : K == 8 ? P::kp8(i) : P::kp16(i);
            r.l[i] = l[i] + kp - b.l[i];
        }
        r.normalize();
        return r;
    }
    // a + 2 b with ONE normalisation (limbs < 3 * 2^30 before it)      bound: A + 2 B
    G16_HD Fp30 add_dbl(const Fp30& b) const {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = l[i] + (b.l[i] << 1);
        r.normalize();
        return r;
    }
    // a + 2^(k+1) p - b  with the redundant constant selected at run time (scalar fields only; NTT butterflies)
    G16_HD Fp30 sub_pow2(const Fp30& b, int k) const {
        Fp30 r;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = l[i] + P::kp_pow2(k, i) - b.l[i];
        r.normalize();
        return r;
    }
    // 2p - a, for a <= 2p (used on canonical inputs)
    G16_HD Fp30 neg2() const { return zero().template sub<2>(*this); }
    // the same on the PACKED form (saturated 32-bit words, as the window tables hold a coordinate), then unpacked: 2p - y is one
    // subtract-with-borrow per word (12) where the 30-bit limbs need a carry-propagating pass (13 + 36) -- the bucket pass keeps the
    // gathered y packed until the point is added
    typedef Fp<P> PackedC;
    G16_HD static constexpr uint32_t two_p32(int i) { return (P::mod(i) << 1) | (i ? P::mod(i - 1) >> 31 : 0u); }
    G16_HD static Fp30 unpack_cond_neg(const PackedC& y, bool flip) {
        uint32_t w[NW];
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned borrow = 0;
        G16_UNROLL for (int i = 0; i < NW; ++i) {
            const uint32_t d = __builtin_subc(two_p32(i), y.v[i], borrow, &borrow);
            w[i] = flip ? d : y.v[i];
        }
#else
        uint32_t borrow = 0;
        G16_UNROLL for (int i = 0; i < NW; ++i) {
            const uint64_t t = (uint64_t)two_p32(i) - y.v[i] - borrow;
            borrow = (uint32_t)(t >> 63);
            w[i] = flip ? (uint32_t)t : y.v[i];
        }
#endif
        return unpack(w);
    }
    // canonical a -> a or 2p - a (a signed digit's +-P): a trait so that the bound-propagating stand-in can return the worse of the two
    G16_HD static Fp30 cond_neg2(const Fp30& a, bool flip) { return flip ? a.neg2() : a; }
    // 16p - a, for a < 16p
    G16_HD Fp30 neg16() const { return zero().template sub<16>(*this); }
    typedef Fp<P> Std;
    G16_HD static Fp30 from_packed(const Std& x) { return unpack(x.v); }

    // Montgomery product a*b/R' (mod p).  Inputs: normalised limbs, a < A p, b < B p with
    // A*B*p/R' <= 0.5  (A = B = 16 is fine for every supported field); output < 1.5 p, normalised.
    G16_HD Fp30 mul(const Fp30& b) const {
#ifdef G16_FP30_OUTLINE
        return mul_outlined(*this, b);
#else
        return mul_impl(b);
#endif
    }
    // ---- double-width column primitives (T has 2*NL 64-bit columns) -----------------------------
    // One sweep of limb products adds CNT(c) = min(c + 1, 2 NL - 1 - c) products (each < 2^60) to column c.  A 64-bit column
    // holds 15 of them plus the small carries that travel with it, so sweeps can pile up in a column WITHOUT any carry work
    // until (sweeps so far + 1) * CNT(c) would pass G16_RELAX_LIMIT; only then is the column "relaxed" (wide_relax), and only that
    // column: for NL = 13 that is 9 of 25 columns between a product and its reduction, none at all for NL <= 8.
    // The column type is a template parameter so that the host self-test can run every routine with 128-bit columns beside the
    // 64-bit ones on all-ones limbs and demand identical results (no overflow anywhere).
#ifndef G16_RELAX_LIMIT
#define G16_RELAX_LIMIT 16   // products (< 2^60 each) a 64-bit column may hold before it is relaxed.  16 (2^30 - 1)^2 = 2^64 - 2^35 + 16 leaves
                             // 2^35 - 16 for the carries that travel with a column (a relaxation carry < 2^34, the reduction's carry < 2^34);
                             // g16_host_selftest PROVES the plan for the real moduli by worst-case bound propagation (hosttest.hip
                             // column_headroom: codes 605-609; 16 and 17 give the same plan, 18 fails with 605) beside the 128-bit shadow
                             // runs.  Rounds 2-3 used 15: 11 relaxed columns per 13-limb product instead of 9.
#endif
    static constexpr int col_count(int c) { return c + 1 < 2 * NL - 1 - c ? c + 1 : 2 * NL - 1 - c; }
    // T = a*b as NL^2 limb products
    template <class U>
    G16_HD static void wide_mul(U* T, const Fp30& a, const Fp30& b) {
        G16_UNROLL for (int c = 0; c < 2 * NL; ++c) T[c] = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            G16_UNROLL for (int j = 0; j < NL; ++j) T[i + j] += (uint64_t)a.l[i] * b.l[j];
        }
    }
    // T = a*a with the symmetric products taken once against the doubled operand: NL(NL+1)/2 multiply-adds.
    // A column holds <= CNT/2 doubled products (< 2^61) and one square: the same magnitude as CNT plain products.
    template <class U>
    G16_HD static void wide_sqr(U* T, const Fp30& a) {
        G16_UNROLL for (int c = 0; c < 2 * NL; ++c) T[c] = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            T[2 * i] += (uint64_t)a.l[i] * a.l[i];
            const uint32_t a2 = a.l[i] << 1;
            G16_UNROLL for (int j = i + 1; j < NL; ++j) T[i + j] += (uint64_t)a2 * a.l[j];
        }
    }
    // T += a*b (one more sweep; the caller relaxes in between)
    template <class U>
    G16_HD static void wide_mul_add(U* T, const Fp30& a, const Fp30& b) {
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            G16_UNROLL for (int j = 0; j < NL; ++j) T[i + j] += (uint64_t)a.l[i] * b.l[j];
        }
    }
    // SWEEPS sweeps have been accumulated and one more (a product sweep or the reduction) follows: every column that could
    // not absorb it hands its high word to the next column -- T[c] = lo32 + hi32 * 2^32 and 2^32 = 4 * 2^30, so
    // T[c + 1] += 4 * hi32: one multiply-add and one register clear instead of the shift / mask / 64-bit add of a full carry
    // step.  The column keeps < 2^32 (not < 2^30: nothing needs that before the reduction's own carry chain).
    template <int SWEEPS, class U>
    G16_HD static void wide_relax(U* T) {
#ifdef G16_FP30_FULL_NORMALIZE
        wide_normalize(T);
#else
#if defined(__HIP_DEVICE_COMPILE__)
        // the factor 4 is made opaque so that the compiler keeps ONE v_mad_u64_u32 (hi32 * 4 + T[c + 1]); written as a shift it
        // becomes a 64-bit shift, two masks and a 64-bit add -- as expensive as the full carry step this replaces
        uint32_t four = 4u;
        asm("" : "+s"(four));
        G16_UNROLL for (int c = 0; c + 1 < 2 * NL; ++c) {
            if ((SWEEPS + 1) * col_count(c) > G16_RELAX_LIMIT) {
                T[c + 1] += (uint64_t)(uint32_t)(T[c] >> 32) * four;
                T[c] = (U)(uint32_t)T[c];
            }
        }
#else
        G16_UNROLL for (int c = 0; c + 1 < 2 * NL; ++c) {
            if ((SWEEPS + 1) * col_count(c) > G16_RELAX_LIMIT) {
                T[c + 1] += (T[c] >> 32) << 2;
                T[c] &= (U)0xffffffffu;
            }
        }
#endif
#endif
    }
    // one full carry sweep -> every column < 2^30 (the top column takes the rest)
    template <class U>
    G16_HD static void wide_normalize(U* T) {
        G16_UNROLL for (int c = 0; c + 1 < 2 * NL; ++c) {
            T[c + 1] += T[c] >> 30;
            T[c] &= MASK;
        }
    }
    // Montgomery reduction of T (value < ~400 p^2; columns relaxed for one more sweep): returns T / R' mod p, < p (1 + T/(R' p))
    template <class U>
    G16_HD static Fp30 wide_redc(U* T) {
        U carry = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            T[i] += carry;
            const uint32_t m = ((uint32_t)T[i] * P::PINV30) & MASK;
            G16_UNROLL for (int j = 0; j < NL; ++j) T[i + j] += (uint64_t)m * P::p30(j);  // column c again gets <= CNT(c) products
            carry = T[i] >> 30;  // low 30 bits are zero now
        }
        Fp30 r;
        G16_UNROLL for (int j = 0; j < NL; ++j) {
            const U v = T[NL + j] + carry;
            r.l[j] = (j == NL - 1) ? (uint32_t)v : ((uint32_t)v & MASK);
            carry = v >> 30;
        }
        return r;
    }
#ifndef G16_NO_FIPS
#define G16_FIPS 1
#endif
    // ---- product scanning (round 6): the same sums COLUMN BY COLUMN --------------------------------------------------------
    // The operand-scanning forms above add the reduction's carry into a column that already holds its limb products: one 64-bit add
    // per column on top of the shift that produced the carry (26 per product; the compiler also splits column chains and adds
    // the halves, 33 v_lshl_add_u64 per product in the round-5 ISA).  Column-major, ONE running accumulator takes column c's
    // products of every sweep and of the reduction, and its carry-out `acc >> 30` is the SEED of column c + 1's multiply-add chain:
    // the add disappears (v_mad_u64_u32 has a free 64-bit addend), the 2 NL columns of T are never materialised (fewer live
    // registers), and with the field's REAL modulus limbs in the bound only 5 of 25 columns of a 13-limb product exceed what a
    // 64-bit accumulator holds (9 "relaxed" columns before).  Such a column continues in a second accumulator seeded with zero; at
    // the end its low word joins the first (one multiply-add by 1) and its high word, worth 4 units of the next column, joins the
    // carry (one multiply-add by 4).  Which part of which column starts a new accumulator is decided at compile time by
    // fips_plan: worst-case bound propagation (operand limbs <= 2^30 - 1, every m_i <= 2^30 - 1, the real p limbs) -- the plan IS
    // the overflow proof (static_assert on .ok), and the host self-test still runs every form with 128-bit accumulators beside it.
    // LLVM's reassociation would undo the order (it sorts the late-arriving carry to the END of a column's sum: back to the
    // separate add), so each accumulation step is pinned with an empty asm on the device.
    template <int NS>
    struct FipsPlan {
        unsigned char seg[2 * NL][NS + 1];   // accumulator index of part k of column c (parts: the NS sweeps, then the reduction)
        unsigned char nseg[2 * NL];
        int extra;                           // accumulators beyond the first, summed over the columns (2 multiply-adds each)
        bool ok;
    };
    template <int NS>
    static constexpr FipsPlan<NS> fips_plan() {
        typedef unsigned __int128 W;
        FipsPlan<NS> pl{};
        const W LIM = (W)1 << 64, M = MASK, W32 = 0xffffffffu;
        W carry = 0;
        pl.ok = true;
        pl.extra = 0;
        for (int c = 0; c < 2 * NL - 1; ++c) {
            const int lo = c < NL ? 0 : c - NL + 1, hi = c < NL ? c : NL - 1;
            W psum = 0;                       // reduction products of this column without m_c p_0 (added after the merge)
            for (int i = lo; i <= hi; ++i)
                if (i != c) psum += M * (W)P::p30(c - i);
            W B[NS + 1] = {};
            // accumulator 0: the carry-in, m_c p_0, the low words merged in at the end and (the assembly's fused forms) one 32-bit
            // difference K p_j - s_j are reserved up front
            B[0] = carry + (c < NL ? M * (W)P::p30(0) : (W)0) + (W)(NS + 2) * W32;   // + 2: a fused subtraction's K p_j - s_j, slack
            int cur = 0;
            for (int k = 0; k <= NS; ++k) {
                const W part = k < NS ? (W)(hi - lo + 1) * M * M : psum;
                if (B[cur] + part >= LIM) {
                    ++cur;
                    if (cur > NS || part >= LIM) { pl.ok = false; return pl; }
                }
                B[cur] += part;
                pl.seg[c][k] = (unsigned char)cur;
            }
            pl.nseg[c] = (unsigned char)(cur + 1);
            pl.extra += cur;
            carry = B[0] >> 30;
            for (int s = 1; s <= cur; ++s) carry += 4 * (B[s] >> 32);
        }
        // (the last carry is the top limb; it fits 32 bits because the VALUE is bounded -- callers' A*B*p/R' <= 0.5 -- not because
        //  of this all-ones worst case, whose sum would be several R')
        return pl;
    }
    // the generated assembly's plan (fips_asm_gen.hpp) against this one
    template <int NS>
    static constexpr bool fips_asm_plan_agrees() {
        if constexpr (!FipsAsm<P>::available) return true;
        else {
            static_assert(NS == 1 || NS == 2 || NS == 4, "assembly exists for one, two and four sweeps");
            const FipsPlan<NS> pl = fips_plan<NS>();
            if (!pl.ok || FipsAsm<P>::NL != NL) return false;
            for (int c = 0; c < 2 * NL - 1; ++c)
                for (int k = 0; k <= NS; ++k) {
                    const int want = NS == 1 ? FipsAsm<P>::plan1(c * 2 + k) : NS == 2 ? FipsAsm<P>::plan2(c * 3 + k) : FipsAsm<P>::plan4(c * 5 + k);
                    if (pl.seg[c][k] != want) return false;
                }
            return true;
        }
    }
    // The C++ form's pin: an empty asm the running sum passes through.  It keeps the order, but gfx950's hazard recognizer puts a wait
    // state (s_nop 0) between an inline asm that defines a register and the next instruction that reads it: ~2 200 per mixed addition,
    // measured 3.6 % SLOWER than the operand-scanning forms (profiles/r06_ab_fips_pins.txt; an asm that only READS the sum draws no
    // wait state but lets the scheduler stretch the live ranges: G1 pass 137 -> 248 registers, the lane-pair kernel spills).  The
    // device therefore runs the chain as hand-scheduled assembly (fips_asm_gen.hpp, generated by gen_fips_asm.py); this form is
    // what the host runs (plain C++, 64- and 128-bit accumulators) and the device's fallback under -DG16_NO_FIPS_ASM.
    template <class U>
    G16_HD static void fips_pin(U& v) {
#if defined(__HIP_DEVICE_COMPILE__)
        if constexpr (sizeof(U) == 8) asm("" : "+v"(v));
#endif
    }
    // sum_k x_k y_k / R' mod p under one reduction; SQR: the single sweep is x_0^2 (symmetric products once, doubled operand)
    template <class U, int NS, bool SQR>
    G16_HD static Fp30 fips(const Fp30* const* xs, const Fp30* const* ys) {
        constexpr FipsPlan<NS> pl = fips_plan<NS>();
        static_assert(pl.ok, "product-scanning plan: a column does not fit its accumulators");
        static_assert(!SQR || NS == 1, "the squaring form has one sweep");
#if defined(__HIP_DEVICE_COMPILE__)
        uint32_t one = 1u, four = 4u;   // opaque: keeps `acc + lo32 * 1` / `carry + hi32 * 4` ONE v_mad_u64_u32 each
        asm("" : "+s"(one));
        asm("" : "+s"(four));
#else
        const uint32_t one = 1u, four = 4u;
#endif
        uint32_t m[NL], x2[NL];
        if (SQR) { G16_UNROLL for (int i = 0; i < NL; ++i) x2[i] = xs[0]->l[i] << 1; }
        Fp30 r;
        U carry = 0;
        G16_UNROLL for (int c = 0; c < 2 * NL - 1; ++c) {
            const int lo = c < NL ? 0 : c - NL + 1, hi = c < NL ? c : NL - 1;
            U acc[NS + 1];
            acc[0] = carry;
            G16_UNROLL for (int s = 1; s <= NS; ++s) acc[s] = 0;
            G16_UNROLL for (int k = 0; k < NS; ++k) {
                const int s = pl.seg[c][k];
                if (SQR) {
                    G16_UNROLL for (int i = lo; 2 * i < c; ++i) { acc[s] += (U)((uint64_t)x2[i] * xs[0]->l[c - i]); fips_pin(acc[s]); }
                    if ((c & 1) == 0) { acc[s] += (U)((uint64_t)xs[0]->l[c / 2] * xs[0]->l[c / 2]); fips_pin(acc[s]); }
                } else {
                    G16_UNROLL for (int i = lo; i <= hi; ++i) { acc[s] += (U)((uint64_t)xs[k]->l[i] * ys[k]->l[c - i]); fips_pin(acc[s]); }
                }
            }
            {
                const int s = pl.seg[c][NS];
                G16_UNROLL for (int i = lo; i <= hi; ++i) {
                    if (i == c) continue;   // m_c follows below
                    acc[s] += (U)((uint64_t)m[i] * P::p30(c - i));
                    fips_pin(acc[s]);
                }
            }
            G16_UNROLL for (int s = 1; s <= NS; ++s)
                if (s < pl.nseg[c]) { acc[0] += (U)((uint64_t)(uint32_t)acc[s] * one); fips_pin(acc[0]); }
            if (c < NL) {
                m[c] = ((uint32_t)acc[0] * P::PINV30) & MASK;
                acc[0] += (U)((uint64_t)m[c] * P::p30(0));   // low 30 bits vanish
                fips_pin(acc[0]);
            } else {
                r.l[c - NL] = (uint32_t)acc[0] & MASK;
            }
            carry = acc[0] >> 30;
            G16_UNROLL for (int s = 1; s <= NS; ++s)
                if (s < pl.nseg[c]) {
                    if constexpr (sizeof(U) == 8) carry += (U)((uint64_t)(uint32_t)(acc[s] >> 32) * four);
                    else carry += (acc[s] >> 32) * 4;   // the 128-bit shadow run keeps every bit
                    fips_pin(carry);
                }
        }
        r.l[NL - 1] = (uint32_t)carry;
        return r;
    }
    template <class U>
    G16_HD Fp30 mul_cols(const Fp30& b) const {
#ifdef G16_FIPS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(G16_NO_FIPS_ASM)
        if constexpr (sizeof(U) == 8 && FipsAsm<P>::available) {   // the same chain as hand-scheduled assembly (gen_fips_asm.py)
            static_assert(fips_asm_plan_agrees<1>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            FipsAsm<P>::mul(r.l, l, b.l);
            return r;
        }
#endif
        const Fp30* xs[1] = {this};
        const Fp30* ys[1] = {&b};
        return fips<U, 1, false>(xs, ys);
#else
        U T[2 * NL];
        wide_mul(T, *this, b);
        wide_relax<1>(T);
        return wide_redc(T);
#endif
    }
    template <class U>
    G16_HD Fp30 sqr_cols() const {
#ifdef G16_FIPS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(G16_NO_FIPS_ASM)
        if constexpr (sizeof(U) == 8 && FipsAsm<P>::available) {
            static_assert(fips_asm_plan_agrees<1>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            FipsAsm<P>::sqr(r.l, l);
            return r;
        }
#endif
        const Fp30* xs[1] = {this};
        return fips<U, 1, true>(xs, xs);
#else
        U T[2 * NL];
        wide_sqr(T, *this);
        wide_relax<1>(T);
        return wide_redc(T);
#endif
    }
    // x1 y1 + x2 y2 under one reduction
    template <class U>
    G16_HD static Fp30 mul2_cols(const Fp30& x1, const Fp30& y1, const Fp30& x2, const Fp30& y2) {
#ifdef G16_FIPS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(G16_NO_FIPS_ASM)
        if constexpr (sizeof(U) == 8 && FipsAsm<P>::available) {
            static_assert(fips_asm_plan_agrees<2>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            FipsAsm<P>::mul2(r.l, x1.l, y1.l, x2.l, y2.l);
            return r;
        }
#endif
        const Fp30* xs[2] = {&x1, &x2};
        const Fp30* ys[2] = {&y1, &y2};
        return fips<U, 2, false>(xs, ys);
#else
        U T[2 * NL];
        wide_mul(T, x1, y1);
        wide_relax<1>(T);
        wide_mul_add(T, x2, y2);
        wide_relax<2>(T);
        return wide_redc(T);
#endif
    }
    // x1 y1 + x2 y2 + x3 y3 + x4 y4 under one reduction
    template <class U>
    G16_HD static Fp30 mul4_cols(const Fp30& x1, const Fp30& y1, const Fp30& x2, const Fp30& y2, const Fp30& x3, const Fp30& y3,
                                 const Fp30& x4, const Fp30& y4) {
#ifdef G16_FIPS
#if defined(__HIP_DEVICE_COMPILE__) && !defined(G16_NO_FIPS_ASM)
        if constexpr (sizeof(U) == 8 && FipsAsm<P>::available) {
            static_assert(fips_asm_plan_agrees<4>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            FipsAsm<P>::mul4(r.l, x1.l, y1.l, x2.l, y2.l, x3.l, y3.l, x4.l, y4.l);
            return r;
        }
#endif
        const Fp30* xs[4] = {&x1, &x2, &x3, &x4};
        const Fp30* ys[4] = {&y1, &y2, &y3, &y4};
        return fips<U, 4, false>(xs, ys);
#else
        U T[2 * NL];
        wide_mul(T, x1, y1);
        wide_relax<1>(T);
        wide_mul_add(T, x2, y2);
        wide_relax<2>(T);
        wide_mul_add(T, x3, y3);
        wide_relax<3>(T);
        wide_mul_add(T, x4, y4);
        wide_relax<4>(T);
        return wide_redc(T);
#endif
    }
    template <class U>
    G16_HD static Fp30 mul_sub_cols(const Fp30& a, const Fp30& b, const Fp30& c, const Fp30& d) {
        return mul2_cols<U>(a, b, c, d.neg2());
    }
    // ---- products with the group formulas' lazy subtractions folded in (round 6) --------------------------------------------
    // a b / R' + K p - s and a^2 / R' + 6 p - (u + 2 v): on the device the difference K p_j - s_j rides in column NL + j of the
    // product's assembly block (gen_fips_asm.py: one 32-bit subtraction + one multiply-add by 1 per limb; the stand-alone
    // subtraction costs 13 + 12 * 3 instructions for its carry pass).  Same integer, hence the same normalised limbs, as the
    // two-step forms the host runs -- preconditions and bounds are those of sub<K> (BoundF checks them through the same names).
    template <int K>
    G16_HD Fp30 mul_sub_k(const Fp30& b, const Fp30& s) const {
        static_assert(K == 2 || K == 4 || K == 8, "fused subtraction: K = 2, 4, 8");
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_FIPS) && !defined(G16_NO_FIPS_ASM) && !defined(G16_NO_FUSED_SUB)
        if constexpr (FipsAsm<P>::has_sub) {
            static_assert(fips_asm_plan_agrees<1>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            if constexpr (K == 2) FipsAsm<P>::mul_s2(r.l, l, b.l, s.l);
            else if constexpr (K == 4) FipsAsm<P>::mul_s4(r.l, l, b.l, s.l);
            else FipsAsm<P>::mul_s8(r.l, l, b.l, s.l);
            return r;
        }
#endif
        return mul(b).template sub<K>(s);
    }
    template <int K>
    G16_HD static Fp30 mul2_sub_k(const Fp30& x1, const Fp30& y1, const Fp30& x2, const Fp30& y2, const Fp30& s) {
        static_assert(K == 2 || K == 4 || K == 8, "fused subtraction: K = 2, 4, 8");
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_FIPS) && !defined(G16_NO_FIPS_ASM) && !defined(G16_NO_FUSED_SUB)
        if constexpr (FipsAsm<P>::has_sub) {
            static_assert(fips_asm_plan_agrees<2>(), "gen_fips_asm.py and Fp30::fips_plan disagree on the column plan");
            Fp30 r;
            if constexpr (K == 2) FipsAsm<P>::mul2_s2(r.l, x1.l, y1.l, x2.l, y2.l, s.l);
            else if constexpr (K == 4) FipsAsm<P>::mul2_s4(r.l, x1.l, y1.l, x2.l, y2.l, s.l);
            else FipsAsm<P>::mul2_s8(r.l, x1.l, y1.l, x2.l, y2.l, s.l);
            return r;
        }
#endif
        return mul2_cols<uint64_t>(x1, y1, x2, y2).template sub<K>(s);
    }
    // this^2 + 6 p - (u + 2 v)   (X3 = R^2 - PPP - 2 Q of the XYZZ additions; u, v < ~1.5 p normalised)
    G16_HD Fp30 sqr_sub_x3(const Fp30& u, const Fp30& v) const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_FIPS) && !defined(G16_NO_FIPS_ASM) && !defined(G16_NO_FUSED_SUB)
        if constexpr (FipsAsm<P>::has_sub) {
            Fp30 r;
            FipsAsm<P>::sqr_x3(r.l, l, u.l, v.l);
            return r;
        }
#endif
        return sqr().template sub<6>(u.add_dbl(v));
    }
    // this * b + 6 p - (u + 2 v)   (the lane pair's X3: its squaring is a product of prepared operands)
    G16_HD Fp30 mul_sub_x3(const Fp30& b, const Fp30& u, const Fp30& v) const {
#if defined(__HIP_DEVICE_COMPILE__) && defined(G16_FIPS) && !defined(G16_NO_FIPS_ASM) && !defined(G16_NO_FUSED_SUB)
        if constexpr (FipsAsm<P>::has_sub) {
            Fp30 r;
            FipsAsm<P>::mul_x3(r.l, l, b.l, u.l, v.l);
            return r;
        }
#endif
        return mul_impl(b).template sub<6>(u.add_dbl(v));
    }
    G16_HD Fp30 mul_impl(const Fp30& b) const { return mul_cols<uint64_t>(b); }
    G16_HD_NOINLINE static Fp30 mul_outlined(Fp30 a, Fp30 b) { return a.mul_impl(b); }
    // a*b - c*d with ONE Montgomery reduction (the double-width sums share it): 3 NL^2 multiply-adds instead of 4 NL^2.
    // Requires d < 2p and a*b + 2p*c < ~400 p^2; output < 1.2p for the bounds the group formulas feed it.
    // In the REGISTER-RESIDENT accumulator (Acc30: reductions, table builder) this stays opt-in (G16_FP30_MUL_SUB): round 2 measured
    // -4.7 % at equal occupancy, but the fused form needs ~15 more live registers than the 256 that keep two waves per SIMD; held
    // to 256 the allocator spilled 18 dwords and the pass was 1.8 % SLOWER than two plain products (profiles/r02_ab_g1_occupancy.txt).
    // The bucket pass itself no longer has that problem: its accumulator is parked in LDS (AccParked, round 4) and it calls
    // mul_sub_fused below -- G1 pass 9.14 -> 8.90 ms (profiles/r04_ab_parked_accumulator.txt).
    G16_HD static Fp30 mul_sub(const Fp30& a, const Fp30& b, const Fp30& c, const Fp30& d) {
#if defined(G16_FP30_OUTLINE) || !defined(G16_FP30_MUL_SUB)
        return a.mul(b).template sub<2>(c.mul(d));
#else
        return mul_sub_cols<uint64_t>(a, b, c, d);
#endif
    }
    // ---- operand views (round 6).  A lane-pair Fq2 product needs its FIRST operand as both components in both lanes (two DPP
    // broadcasts per limb) and its SECOND operand's partner component, negated for the even lane (a carry-propagating 16p - b and a
    // select).  An operand that enters several products of a group formula (PP: three, PPP and Pd and R: two) is prepared ONCE
    // through these views; for one-lane fields a view is the value itself.
    typedef Fp30 Lhs;
    typedef Fp30 Rhs;
    G16_HD static const Fp30& lhs(const Fp30& a) { return a; }
    G16_HD static const Fp30& rhs(const Fp30& a) { return a; }
    G16_HD static Fp30 mul_v(const Fp30& a, const Fp30& b) { return a.mul(b); }
    G16_HD static Fp30 sqr_v(const Fp30& a) { return a.sqr(); }
    G16_HD static Fp30 sqr_sub_x3_v(const Fp30& a, const Fp30& u, const Fp30& v) { return a.sqr_sub_x3(u, v); }
    G16_HD static Fp30 mul_add_fused_v(const Fp30& a, const Fp30& b, const Fp30& c, const Fp30& d) { return mul2_cols<uint64_t>(a, b, c, d); }
    // a*b + c*d under one reduction (the sign-tracking mixed addition's Y: no operand is negated)
    G16_HD static Fp30 mul_add_fused(const Fp30& a, const Fp30& b, const Fp30& c, const Fp30& d) { return mul2_cols<uint64_t>(a, b, c, d); }
    // the fused form unconditionally: AccParked (bucket pass) has the registers for it -- its accumulator is not in them
    G16_HD static Fp30 mul_sub_fused(const Fp30& a, const Fp30& b, const Fp30& c, const Fp30& d) {
#ifdef G16_NO_MUL_SUB_FUSED
        return a.mul(b).template sub<2>(c.mul(d));
#else
        return mul_sub_cols<uint64_t>(a, b, c, d);
#endif
    }
    G16_HD Fp30 sqr() const {
#if defined(G16_FP30_OUTLINE) || defined(G16_FP30_NO_SQR)
        return mul(*this);
#else
        return sqr_cols<uint64_t>();
#endif
    }

    // exact: is the value (any bound < 16p, normalised) congruent to 0 mod p?
    // Fast filter on the low limb: v = k p with k < 16 forces (v0 * p^-1 mod 2^30) = k < 16.
    G16_HD bool maybe_zero() const { return ((l[0] * P::PPINV30) & MASK) < 16u; }
    G16_HD bool is_zero_exact() const {
        const uint32_t k = (l[0] * P::PPINV30) & MASK;
        if (k >= 16u) return false;
        Fp30 kp;
        uint64_t c = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const uint64_t v = (uint64_t)k * P::p30(i) + c;
            kp.l[i] = (i == NL - 1) ? (uint32_t)v : ((uint32_t)v & MASK);
            c = v >> 30;
        }
        uint32_t diff = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) diff |= kp.l[i] ^ l[i];
        return diff == 0;
    }
    // v >= c ? v - c : v   for normalised v and a normalised constant c (limbs30 of k*p)
    template <int K>
    G16_HD Fp30 cond_sub() const {
        static_assert(K == 2 || K == 4 || K == 8 || K == 16, "k p constants exist for k = 2, 4, 8, 16 only");
        Fp30 d;
        int32_t br = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const uint32_t c = K == 2 ? P::np2(i) : K == 4 ? P::np4(i) : K == 8 ? P::np8(i) : P::np16(i);
            const int32_t v = (int32_t)l[i] - (int32_t)c + br;
            if (i == NL - 1) { d.l[i] = (uint32_t)v; br = v >> 31; }
            else { d.l[i] = (uint32_t)v & MASK; br = v >> 30; }
        }
        Fp30 r;
        const bool lt = br < 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = lt ? l[i] : d.l[i];
        return r;
    }
    // value < 32p  ->  value < 2p  (same residue)
    G16_HD Fp30 weak_reduce32() const { return cond_sub<16>().template cond_sub<8>().template cond_sub<4>().template cond_sub<2>(); }
    G16_HD bool raw_zero() const {
        uint32_t a = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) a |= l[i];
        return a == 0;
    }
    typedef Fp30 Raw;   // the one-lane field whose limb layout partial sums are stored in
    // accumulator trait constants (see Acc30): bounds of this field's product outputs are < 1.5p
    static constexpr int KM = 2, K2M = 4, KX = 8, KY = 4;
    G16_HD Fp30 settle() const { return *this; }
// Two waves per SIMD (<= 256 registers): at 257 the kernel silently drops to ONE wave per SIMD and loses ~20 % (measured: the
// round-2 DIRECT template parameter cost 10.1 -> 12.5 ms per pass that way; tests/test_kernel_resources.py now asserts it).
#ifndef G16_ACC_MIN_WAVES
#define G16_ACC_MIN_WAVES 2
#endif
    static constexpr int ACC_MIN_WAVES = G16_ACC_MIN_WAVES;
#ifndef G16_G1_PREFETCH
#define G16_G1_PREFETCH true
#endif
    static constexpr bool ACC_PREFETCH = G16_G1_PREFETCH;
#ifndef G16_G1_PARKED
#define G16_G1_PARKED 1
#endif
    static constexpr bool ACC_PARKED = G16_G1_PARKED != 0;   // bucket pass: accumulator coordinates in LDS (AccParked)
    // ---- bucket-kernel hooks: one lane per task
    static constexpr int LANES_PER_TASK = 1;
    template <class A>
    G16_HD static bool load_point(const A* bases, int64_t idx, Fp30& px, Fp30& py) {   // false for the identity
        const A p = bases[idx];
        if (p.is_identity()) return false;
        px = from_packed(p.x);
        py = from_packed(p.y);
        return true;
    }
    // the same with y left PACKED (unpack_cond_neg takes it from there when the point is added)
    template <class A>
    G16_HD static bool load_point_py(const A* bases, int64_t idx, Fp30& px, PackedC& yw) {
        const A p = bases[idx];
        if (p.is_identity()) return false;
        px = from_packed(p.x);
        yw = p.y;
        return true;
    }


    // exact reduction to [0, p) of a normalised value < 2p (e.g. a product output)
    G16_HD Fp30 canonical_lt2p() const {
        Fp30 d;
        int32_t br = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const int32_t v = (int32_t)l[i] - (int32_t)P::p30(i) + br;  // |v| < 2^31
            if (i == NL - 1) { d.l[i] = (uint32_t)v; br = v >> 31; }
            else { d.l[i] = (uint32_t)v & MASK; br = v >> 30; }
        }
        // br < 0  <=>  value < p
        Fp30 r;
        const bool lt = br < 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = lt ? l[i] : d.l[i];
        return r;
    }

    // value < 8p (normalised) -> canonical [0, p): three conditional subtractions (~5 % of a product)
    G16_HD Fp30 canonical_lt8p() const { return cond_sub<4>().template cond_sub<2>().canonical_lt2p(); }
    // ANY lazy value (normalised limbs, value < 2^18 p) -> canonical [0, p) without a Montgomery product: an under-estimate q of
    // the quotient from the top limb alone (q <= floor(v / p), short by at most 4: see below), ONE row of NL multiply-adds for
    // v - q p, then the three conditional subtractions of canonical_lt8p.  ~1/3 of the issue slots of mul(one) + canonical_lt2p;
    // the NTT passes canonicalise every element of every sweep this way.
    //   v >= top 2^(30 (NL-1)),  p < (ptop + 1) 2^(30 (NL-1))   =>   v / p > top / (ptop + 1) >= q   (never negative)
    //   v / p - top / (ptop + 1) < (top + ptop + 1) / (ptop (ptop + 1)) + 1 < 2.5 for top < 2^26, ptop >= 2^13; the reciprocal
    //   multiplication loses at most one more                                         =>   v - q p < 4.5 p < 8 p
    G16_HD Fp30 canonical_quick() const {
        constexpr uint32_t D = P::p30(NL - 1) + 1u;
        static_assert(D > (1u << 13), "the top limb of p carries at least 14 bits for every supported field");
        constexpr uint64_t MAGIC = ((uint64_t)1 << 40) / D;
        const uint32_t q = (uint32_t)(((uint64_t)l[NL - 1] * MAGIC) >> 40);
        Fp30 r;
        int64_t carry = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const int64_t t = (int64_t)l[i] - (int64_t)((uint64_t)q * P::p30(i)) + carry;
            r.l[i] = (i == NL - 1) ? (uint32_t)t : ((uint32_t)t & MASK);
            carry = t >> 30;
        }
        return r.canonical_lt8p();
    }
    // p - a for canonical a (a = 0 gives p: not canonical, but below 2p and harmless to the lazy arithmetic)
    G16_HD Fp30 neg_canonical() const {
        Fp30 d;
        int32_t br = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) {
            const int32_t v = (int32_t)P::p30(i) - (int32_t)l[i] + br;
            if (i == NL - 1) d.l[i] = (uint32_t)v;
            else { d.l[i] = (uint32_t)v & MASK; br = v >> 30; }
        }
        return d;
    }
    G16_HD bool same_limbs(const Fp30& o) const {
        uint32_t a = 0;
        G16_UNROLL for (int i = 0; i < NL; ++i) a |= l[i] ^ o.l[i];
        return a == 0;
    }
    // ---- batched-affine hooks (batch_affine.hpp): values of this field as the one lane holds them
    static constexpr int PREFIX_LIMBS = NL;
    G16_HD void get_limbs(uint32_t* w) const { G16_UNROLL for (int i = 0; i < NL; ++i) w[i] = l[i]; }
    G16_HD static Fp30 from_limbs(const uint32_t* w) { Fp30 r; G16_UNROLL for (int i = 0; i < NL; ++i) r.l[i] = w[i]; return r; }
    G16_HD bool equals_canonical(const Fp30& o) const { return same_limbs(o); }
    template <class A>
    G16_HD static void store_point(A* dst, int64_t idx, const Fp30& x, const Fp30& y, bool identity) {   // canonical x, y
        A p;
        if (identity) { p = A::identity(); }
        else { x.pack(p.x.v); y.pack(p.y.v); }
        dst[idx] = p;
    }

    // lazy value (< 16p) -> canonical x*R' packed in words (the kernels' own storage form)
    G16_HD Std to_packed() const {
        Std r;
        mul_impl(one()).canonical_lt2p().pack(r.v);  // x R' * R' / R' = x R', now < 1.5p -> exact
        return r;
    }
    // conversions to / from the standard arkworks Montgomery form (x*R mod p, 32-bit words)
    // std -> packed R' form: one standard product by the plain integer R' mod p
    G16_HD static Fp<P> std_to_r30(const Fp<P>& x) {
        Fp<P> c;
        G16_UNROLL for (int i = 0; i < P::N; ++i) c.v[i] = P::r30_plain(i);
        return x * c;  // x R * (R' mod p) / R = x R' mod p, canonical
    }
    // lazy internal value (< 16p) -> standard form, canonical: one 30-bit product by (R mod p)
    G16_HD Fp<P> to_std() const {
        Fp30 c;
        G16_UNROLL for (int i = 0; i < NL; ++i) c.l[i] = P::rstd30(i);
        const Fp30 t = mul(c).canonical_lt2p();  // x R' * R / R' = x R mod p
        Fp<P> r;
        t.pack(r.v);
        return r;
    }
};

// Fq2 = Fq[u]/(u^2+1) over the 30-bit lazy field.  A product is 4 limb-product sweeps + 2 reductions
// (lazy reduction: the two double-width sums are reduced once each), i.e. 6 NL^2 multiply-adds -- the
// same count as Karatsuba's 3 full products, with outputs that obey the single-product bound (< 1.5p)
// so the group formulas and their K constants are shared with G1.  Inputs: components < 16p.
// mul/sqr are out-of-line by default: fully inlined, the G2 bucket kernel is ~300 KB of code and, once
// the waves of a CU drift apart, instruction fetch (64 KB I-cache) becomes the bottleneck -- measured
// 129 ms (inlined) vs 83 ms (out-of-line) for the 2^22-point G2 bucket pass (profiles/r01_*).
template <class P>
struct Fp2x30 {
    typedef Fp30<P> B;
    typedef Fp2<P> Std;
    B c0, c1;
    G16_HD static Fp2x30 zero() { return {B::zero(), B::zero()}; }
    G16_HD static Fp2x30 one() { return {B::one(), B::zero()}; }
    G16_HD static Fp2x30 from_packed(const Std& x) { return {B::unpack(x.c0.v), B::unpack(x.c1.v)}; }
    G16_HD Fp2x30 add(const Fp2x30& o) const { return {c0.add(o.c0), c1.add(o.c1)}; }
    G16_HD Fp2x30 dbl() const { return {c0.dbl(), c1.dbl()}; }
    G16_HD Fp2x30 add_dbl(const Fp2x30& o) const { return {c0.add_dbl(o.c0), c1.add_dbl(o.c1)}; }
    template <int K>
    G16_HD Fp2x30 sub(const Fp2x30& o) const { return {c0.template sub<K>(o.c0), c1.template sub<K>(o.c1)}; }
    G16_HD Fp2x30 neg2() const { return {c0.neg2(), c1.neg2()}; }
    G16_HD static Fp2x30 cond_neg2(const Fp2x30& a, bool flip) { return flip ? a.neg2() : a; }
    G16_HD Fp2x30 mul_impl(const Fp2x30& o) const {
        Fp2x30 r;
        const B nb1 = o.c1.neg16();           // 16p - b1
        r.c0 = B::template mul2_cols<uint64_t>(c0, o.c0, c1, nb1);   // a0 b0 + a1 (16p - b1)  ==  a0 b0 - a1 b1  (mod p)
        r.c1 = B::template mul2_cols<uint64_t>(c0, o.c1, c1, o.c0);  // a0 b1 + a1 b0
        return r;
    }
    G16_HD Fp2x30 sqr_impl() const {          // (a0 + a1)(a0 - a1), 2 a0 a1
        return {c0.add(c1).mul_impl(c0.template sub<16>(c1)), c0.dbl().mul_impl(c1)};
    }
    G16_HD_NOINLINE static Fp2x30 mul_outlined(const Fp2x30& a, const Fp2x30& b) { return a.mul_impl(b); }
    G16_HD_NOINLINE static Fp2x30 sqr_outlined(const Fp2x30& a) { return a.sqr_impl(); }
    G16_HD Fp2x30 mul(const Fp2x30& o) const {
#ifdef G16_FP2X30_INLINE
        return mul_impl(o);
#else
        return mul_outlined(*this, o);
#endif
    }
    G16_HD Fp2x30 sqr() const {
#ifdef G16_FP2X30_INLINE
        return sqr_impl();
#else
        return sqr_outlined(*this);
#endif
    }
    template <int K>
    G16_HD Fp2x30 mul_sub_k(const Fp2x30& o, const Fp2x30& s) const { return mul(o).template sub<K>(s); }
    G16_HD Fp2x30 sqr_sub_x3(const Fp2x30& u, const Fp2x30& v) const { return sqr().template sub<KM + K2M>(u.add_dbl(v)); }
    G16_HD bool maybe_zero() const { return c0.maybe_zero() && c1.maybe_zero(); }
    G16_HD bool is_zero_exact() const { return c0.is_zero_exact() && c1.is_zero_exact(); }
    G16_HD Std to_std() const { return {c0.to_std(), c1.to_std()}; }
    G16_HD Std to_packed() const { return {c0.to_packed(), c1.to_packed()}; }
    static constexpr int KM = 2, K2M = 4, KX = 8, KY = 4;
    G16_HD static Fp2x30 mul_sub(const Fp2x30& a, const Fp2x30& b, const Fp2x30& c, const Fp2x30& d) { return a.mul(b).template sub<KM>(c.mul(d)); }
    G16_HD static Fp2x30 mul_sub_fused(const Fp2x30& a, const Fp2x30& b, const Fp2x30& c, const Fp2x30& d) { return mul_sub(a, b, c, d); }
    G16_HD static Fp2x30 mul_add_fused(const Fp2x30& a, const Fp2x30& b, const Fp2x30& c, const Fp2x30& d) { return a.mul(b).add(c.mul(d)); }
    typedef Fp2x30 Lhs;
    typedef Fp2x30 Rhs;
    G16_HD static const Fp2x30& lhs(const Fp2x30& a) { return a; }
    G16_HD static const Fp2x30& rhs(const Fp2x30& a) { return a; }
    G16_HD static Fp2x30 mul_v(const Fp2x30& a, const Fp2x30& b) { return a.mul(b); }
    G16_HD static Fp2x30 sqr_v(const Fp2x30& a) { return a.sqr(); }
    G16_HD static Fp2x30 sqr_sub_x3_v(const Fp2x30& a, const Fp2x30& u, const Fp2x30& v) { return a.sqr_sub_x3(u, v); }
    G16_HD static Fp2x30 mul_add_fused_v(const Fp2x30& a, const Fp2x30& b, const Fp2x30& c, const Fp2x30& d) { return mul_add_fused(a, b, c, d); }
    G16_HD Fp2x30 settle() const { return *this; }
    G16_HD bool raw_zero() const { return c0.raw_zero() && c1.raw_zero(); }
    typedef Fp2x30 Raw;
    static constexpr int ACC_MIN_WAVES = 1;
    static constexpr bool ACC_PREFETCH = false;
    static constexpr bool ACC_PARKED = false;
    static constexpr int PREFIX_LIMBS = 2 * B::NL;
    // ---- bucket-kernel hooks: one lane per task
    static constexpr int LANES_PER_TASK = 1;
    template <class A>
    G16_HD static bool load_point(const A* bases, int64_t idx, Fp2x30& px, Fp2x30& py) {   // false for the identity
        const A p = bases[idx];
        if (p.is_identity()) return false;
        px = from_packed(p.x);
        py = from_packed(p.y);
        return true;
    }

};

// Fq2 for the G2 BUCKET kernel: Karatsuba over three base-field products that are passed in registers
// (Fp30::mul_outlined by value) or inlined.  Unlike Fp2x30 no operand ever lives in scratch memory: the
// out-of-line Fp2x30 product moved ~250 GB of scratch traffic per 2^22-point launch (rocprofv3 FETCH_SIZE +
// WRITE_SIZE, profiles/r01_pmc_*).  Price: product outputs are only < 6p (c0 = v0 - v1 + 2p, c1 = v2 - v0 - v1 + 4p),
// so the accumulator subtracts with larger K and "settles" X3 / Y3 with a weak reduction (4 conditional
// subtractions, ~3 % of a mixed addition).
template <class P>
struct Fp2k30 {
    typedef Fp30<P> B;
    typedef Fp2<P> Std;
    B c0, c1;
    G16_HD static Fp2k30 zero() { return {B::zero(), B::zero()}; }
    G16_HD static Fp2k30 one() { return {B::one(), B::zero()}; }
    G16_HD static Fp2k30 from_packed(const Std& x) { return {B::unpack(x.c0.v), B::unpack(x.c1.v)}; }
    G16_HD Fp2k30 add(const Fp2k30& o) const { return {c0.add(o.c0), c1.add(o.c1)}; }
    G16_HD Fp2k30 dbl() const { return {c0.dbl(), c1.dbl()}; }
    G16_HD Fp2k30 add_dbl(const Fp2k30& o) const { return {c0.add_dbl(o.c0), c1.add_dbl(o.c1)}; }
    template <int K>
    G16_HD Fp2k30 sub(const Fp2k30& o) const { return {c0.template sub<K>(o.c0), c1.template sub<K>(o.c1)}; }
    G16_HD Fp2k30 neg2() const { return {c0.neg2(), c1.neg2()}; }
    G16_HD static Fp2k30 cond_neg2(const Fp2k30& a, bool flip) { return flip ? a.neg2() : a; }
    G16_HD static B bmul(const B& a, const B& b) {
#ifdef G16_FP2K_INLINE
        return a.mul_impl(b);
#else
        return B::mul_outlined(a, b);
#endif
    }
    // inputs: components < 16p
    G16_HD Fp2k30 mul(const Fp2k30& o) const {
        const B v0 = bmul(c0, o.c0), v1 = bmul(c1, o.c1);       // < 1.5p
        const B v2 = bmul(c0.add(c1), o.c0.add(o.c1));          // operands < 32p
        return {v0.template sub<2>(v1), v2.template sub<4>(v0.add(v1))};   // < 3.5p, < 5.5p
    }
    G16_HD Fp2k30 sqr() const {                                 // both components < 2p
        return {bmul(c0.add(c1), c0.template sub<16>(c1)), bmul(c0.dbl(), c1)};
    }
    template <int K>
    G16_HD Fp2k30 mul_sub_k(const Fp2k30& o, const Fp2k30& s) const { return mul(o).template sub<K>(s); }
    G16_HD Fp2k30 sqr_sub_x3(const Fp2k30& u, const Fp2k30& v) const { return sqr().template sub<16>(u.add_dbl(v)); }
    G16_HD bool maybe_zero() const { return c0.maybe_zero() && c1.maybe_zero(); }
    G16_HD bool is_zero_exact() const { return c0.is_zero_exact() && c1.is_zero_exact(); }
    G16_HD Std to_std() const { return {c0.to_std(), c1.to_std()}; }
    G16_HD Std to_packed() const { return {c0.to_packed(), c1.to_packed()}; }
    // product outputs < 6p -> subtract them with K = 8 (16 when doubled); x, y are settled below 2p
    static constexpr int KM = 8, K2M = 16, KX = 2, KY = 2;
    G16_HD static Fp2k30 mul_sub(const Fp2k30& a, const Fp2k30& b, const Fp2k30& c, const Fp2k30& d) { return a.mul(b).template sub<KM>(c.mul(d)); }
    G16_HD static Fp2k30 mul_sub_fused(const Fp2k30& a, const Fp2k30& b, const Fp2k30& c, const Fp2k30& d) { return mul_sub(a, b, c, d); }
    G16_HD static Fp2k30 mul_add_fused(const Fp2k30& a, const Fp2k30& b, const Fp2k30& c, const Fp2k30& d) { return a.mul(b).add(c.mul(d)); }
    typedef Fp2k30 Lhs;
    typedef Fp2k30 Rhs;
    G16_HD static const Fp2k30& lhs(const Fp2k30& a) { return a; }
    G16_HD static const Fp2k30& rhs(const Fp2k30& a) { return a; }
    G16_HD static Fp2k30 mul_v(const Fp2k30& a, const Fp2k30& b) { return a.mul(b); }
    G16_HD static Fp2k30 sqr_v(const Fp2k30& a) { return a.sqr(); }
    G16_HD static Fp2k30 sqr_sub_x3_v(const Fp2k30& a, const Fp2k30& u, const Fp2k30& v) { return a.sqr_sub_x3(u, v); }
    G16_HD static Fp2k30 mul_add_fused_v(const Fp2k30& a, const Fp2k30& b, const Fp2k30& c, const Fp2k30& d) { return mul_add_fused(a, b, c, d); }
    G16_HD Fp2k30 settle() const { return {c0.weak_reduce32(), c1.weak_reduce32()}; }
    G16_HD bool raw_zero() const { return c0.raw_zero() && c1.raw_zero(); }
    typedef Fp2x30<P> Raw;
#ifndef G16_G2_MIN_WAVES
#define G16_G2_MIN_WAVES 1
#endif
    static constexpr int ACC_MIN_WAVES = G16_G2_MIN_WAVES;
    static constexpr bool ACC_PREFETCH = false;
    static constexpr bool ACC_PARKED = false;
    static constexpr int PREFIX_LIMBS = 2 * B::NL;
    // ---- bucket-kernel hooks: one lane per task
    static constexpr int LANES_PER_TASK = 1;
    template <class A>
    G16_HD static bool load_point(const A* bases, int64_t idx, Fp2k30& px, Fp2k30& py) {   // false for the identity
        const A p = bases[idx];
        if (p.is_identity()) return false;
        px = from_packed(p.x);
        py = from_packed(p.y);
        return true;
    }

};

// Fq2 for the G2 bucket kernel, LANE-PAIR form: two adjacent lanes (2k, 2k+1) own one bucket; lane parity `hi`
// selects the component it holds (c0 or c1).  Additions are per-lane; a product fetches the partner's components with one
// DPP quad-permute per limb and computes ITS output component with the 2-product lazy reduction
//     lane 0:  c0 = a0 b0 + a1 (16p - b1)        lane 1:  c1 = a0 b1 + a1 b0
// (507 multiply-adds per lane).  Per-lane state is that of a G1 accumulator, so the kernel keeps two waves per SIMD, needs
// no calls and no scratch, and its outputs obey the tight (< 1.5p) bound.  Control flow is identical in both lanes of a pair.
template <class P>
struct Fp2p30 {
    typedef Fp30<P> B;
    typedef Fp2<P> Std;
    B c;  // this lane's component

    G16_HD static bool lane_hi() {
#if defined(__HIP_DEVICE_COMPILE__)
        return (threadIdx.x & 1u) != 0;
#else
        return false;
#endif
    }
    // partner lane's value (lane ^ 1)
    G16_HD static uint32_t swap32(uint32_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
        // bound_ctrl = true: every lane of the quad is a valid source, so no "old" value exists -- with update_dpp(0, ...) the compiler
        // initialised the destination (v_mov_b32 dst, zero) before every v_mov_b32_dpp: 156 of them per lane-pair mixed addition
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
#else
        return v;
#endif
    }
    G16_HD static B swap(const B& x) {
        B r;
        G16_UNROLL for (int i = 0; i < B::NL; ++i) r.l[i] = swap32(x.l[i]);
        return r;
    }
    G16_HD static B sel(bool c, const B& a, const B& b) {
        B r;
        G16_UNROLL for (int i = 0; i < B::NL; ++i) r.l[i] = c ? a.l[i] : b.l[i];
        return r;
    }
    // the EVEN / ODD lane's value in both lanes of the pair (component 0 / component 1 of an Fq2 value): one DPP move per limb.  Round 6:
    // the first operand of a product is needed as (a0, a1) in BOTH lanes -- two broadcasts instead of a swap and two selects per limb
    G16_HD static uint32_t bcast32(uint32_t v, bool odd) {
#if defined(__HIP_DEVICE_COMPILE__)
        return odd ? (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xF5 /* quad_perm [1,1,3,3] */, 0xF, 0xF, true)
                   : (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xA0 /* quad_perm [0,0,2,2] */, 0xF, 0xF, true);
#else
        return v;
#endif
    }
    G16_HD static B comp0(const B& x) { B r; G16_UNROLL for (int i = 0; i < B::NL; ++i) r.l[i] = bcast32(x.l[i], false); return r; }
    G16_HD static B comp1(const B& x) { B r; G16_UNROLL for (int i = 0; i < B::NL; ++i) r.l[i] = bcast32(x.l[i], true); return r; }
    // ---- pure per-lane kernels (host-testable).  `_c` forms: a0, a1 = BOTH components of the first operand (the same in both lanes),
    // mb / ob = this lane's / the partner's component of the second.  The (mine, other) forms below select (a0, a1) and forward.
    G16_HD static B pair_mul_c(bool hi, const B& a0, const B& a1, const B& mb, const B& ob) {
        const B y2 = sel(hi, ob, ob.neg16());         // lane0: 16p-b1  lane1: b0
        return pair_cols<uint64_t>(a0, mb, a1, y2);   // lane0: a0 b0 + a1 (16p-b1)   lane1: a0 b1 + a1 b0
    }
    G16_HD static B pair_mul(bool hi, const B& ma, const B& oa, const B& mb, const B& ob) {
        return pair_mul_c(hi, sel(hi, oa, ma), sel(hi, ma, oa), mb, ob);
    }
    // x1 y1 + x2 y2 under one reduction
    template <class U>
    G16_HD static B pair_cols(const B& x1, const B& y1, const B& x2, const B& y2) {
        return B::template mul2_cols<U>(x1, y1, x2, y2);
    }
    // (a0 + a1)(a0 - a1), 2 a0 a1:   lane0: x = a0 + a1, y = a0 - a1 + 16p      lane1: x = 2 a0, y = a1
    G16_HD static void pair_sqr_operands(bool hi, const B& a0, const B& a1, B& x, B& y) {
        x = a0.add(sel(hi, a0, a1));                  // ONE addition serves both lanes (was: a doubling, an addition and a select)
        y = sel(hi, a1, a0.template sub<16>(a1));
    }
    G16_HD static B pair_sqr_c(bool hi, const B& a0, const B& a1) {
        B x, y;
        pair_sqr_operands(hi, a0, a1, x, y);
        return x.mul_impl(y);
    }
    G16_HD static B pair_sqr(bool hi, const B& m, const B& o) { return pair_sqr_c(hi, sel(hi, o, m), sel(hi, m, o)); }
    // this lane's component of a*b - c*d (d's components < 2p): four limb-product sweeps, ONE reduction
    //     lane 0:  a0 b0 + a1 (16p - b1) + c0 (2p - d0) + c1 d1        lane 1:  a0 b1 + a1 b0 + c0 (2p - d1) + c1 (2p - d0)
    G16_HD static B pair_mul_sub_c(bool hi, const B& a0, const B& a1, const B& mb, const B& ob, const B& c0, const B& c1, const B& md,
                                   const B& od) {
        const B y2 = sel(hi, ob, ob.neg16());         // lane0: 16p-b1  lane1: b0
        const B z1 = md.neg2();                       // lane0: 2p-d0   lane1: 2p-d1
        const B z2 = sel(hi, od.neg2(), od);          // lane0: d1      lane1: 2p-d0
#ifdef G16_PAIR_Y3_SPLIT
        // two two-sweep products and a lazy sum (one reduction more; measured no faster at three waves per SIMD: the passes are bound by
        // their vector-instruction count, profiles/r06_ab_g2_three_waves.txt); output < 2 (1 + T / (R' p)) p
        return B::template mul2_cols<uint64_t>(a0, mb, a1, y2).add(B::template mul2_cols<uint64_t>(c0, z1, c1, z2));
#else
        return B::template mul4_cols<uint64_t>(a0, mb, a1, y2, c0, z1, c1, z2);
#endif
    }
    G16_HD static B pair_mul_sub(bool hi, const B& ma, const B& oa, const B& mb, const B& ob, const B& mc, const B& oc, const B& md,
                                 const B& od) {
        return pair_mul_sub_c(hi, sel(hi, oa, ma), sel(hi, ma, oa), mb, ob, sel(hi, oc, mc), sel(hi, mc, oc), md, od);
    }
    // this lane's component of a*b + c*d (d's components < 2p): ONE negated operand instead of the difference's three
    //     lane 0:  a0 b0 + a1 (16p - b1) + c0 d0 + c1 (2p - d1)        lane 1:  a0 b1 + a1 b0 + c0 d1 + c1 d0
    G16_HD static B pair_mul_add_c(bool hi, const B& a0, const B& a1, const B& mb, const B& ob, const B& c0, const B& c1, const B& md,
                                   const B& od) {
        const B y2 = sel(hi, ob, ob.neg16());         // lane0: 16p-b1  lane1: b0
        const B z2 = sel(hi, od, od.neg2());          // lane0: 2p-d1   lane1: d0
        return B::template mul4_cols<uint64_t>(a0, mb, a1, y2, c0, md, c1, z2);
    }
    G16_HD static Fp2p30 mul_add_fused(const Fp2p30& a, const Fp2p30& b, const Fp2p30& c, const Fp2p30& d) {
        return {pair_mul_add_c(lane_hi(), comp0(a.c), comp1(a.c), b.c, swap(b.c), comp0(c.c), comp1(c.c), d.c, swap(d.c))};
    }
    // operand views (see Fp30): first operand = both components in both lanes; second operand = this lane's component and what the
    // product takes from the partner -- lane0: 16p - b1, lane1: b0 -- i.e. the partner's `hi ? 16p - own : own`, one swap away
    struct Lhs { B a0, a1; };
    struct Rhs { B m, y2; };
    G16_HD static Lhs lhs(const Fp2p30& a) { return {comp0(a.c), comp1(a.c)}; }
    G16_HD static Rhs rhs(const Fp2p30& b) { return {b.c, swap(sel(lane_hi(), b.c.neg16(), b.c))}; }
    G16_HD static Fp2p30 mul_v(const Lhs& a, const Rhs& b) { return {pair_cols<uint64_t>(a.a0, b.m, a.a1, b.y2)}; }
    G16_HD static Fp2p30 sqr_v(const Lhs& a) { return {pair_sqr_c(lane_hi(), a.a0, a.a1)}; }
    G16_HD static Fp2p30 sqr_sub_x3_v(const Lhs& a, const Fp2p30& u, const Fp2p30& v) { return {pair_sqr_sub_x3(lane_hi(), a.a0, a.a1, u.c, v.c)}; }
    // a b + c d:  lane0: a0 b0 + a1 (16p - b1) + c0 d0 + c1 (16p - d1)      lane1: a0 b1 + a1 b0 + c0 d1 + c1 d0
    G16_HD static Fp2p30 mul_add_fused_v(const Lhs& a, const Rhs& b, const Lhs& c, const Rhs& d) {
        return {B::template mul4_cols<uint64_t>(a.a0, b.m, a.a1, b.y2, c.a0, d.m, c.a1, d.y2)};
    }
    // Measured in round 2 (profiles/r02_ab_mul_sub.txt, 2^22, same box) inside the register-resident accumulator: the fused form was
    // 4.7 % FASTER for G1 (Fp30::mul_sub) but 5 % SLOWER here (27.9 -> 29.3 ms per G2 pass): eight operand sets + the column array
    // exceeded the 256 VGPRs of a two-waves-per-SIMD kernel.  Acc30 therefore keeps two products (G16_PAIR_MUL_SUB opts in); the bucket
    // pass, whose accumulator sits in LDS since round 4, uses mul_sub_fused below.
    G16_HD static Fp2p30 mul_sub(const Fp2p30& a, const Fp2p30& b, const Fp2p30& c, const Fp2p30& d) {
#ifdef G16_PAIR_MUL_SUB
        return {pair_mul_sub_c(lane_hi(), comp0(a.c), comp1(a.c), b.c, swap(b.c), comp0(c.c), comp1(c.c), d.c, swap(d.c))};
#else
        return a.mul(b).template sub<2>(c.mul(d));
#endif
    }
    // Round 4: with the accumulator parked in LDS (AccParked) the four-sweep form fits: 23.6 vs 24.0 ms per 2^22-point G2 pass on one
    // box (profiles/r04_ab_parked_accumulator.txt), where round 2 measured it 5 % slower inside the register-resident Acc30
    G16_HD static Fp2p30 mul_sub_fused(const Fp2p30& a, const Fp2p30& b, const Fp2p30& c, const Fp2p30& d) {
#ifdef G16_NO_MUL_SUB_FUSED
        return a.mul(b).template sub<2>(c.mul(d));
#else
        return {pair_mul_sub_c(lane_hi(), comp0(a.c), comp1(a.c), b.c, swap(b.c), comp0(c.c), comp1(c.c), d.c, swap(d.c))};
#endif
    }
    G16_HD static Fp2p30 zero() { return {B::zero()}; }
    G16_HD static Fp2p30 one() { return {lane_hi() ? B::zero() : B::one()}; }
    G16_HD Fp2p30 add(const Fp2p30& o) const { return {c.add(o.c)}; }
    G16_HD Fp2p30 dbl() const { return {c.dbl()}; }
    G16_HD Fp2p30 add_dbl(const Fp2p30& o) const { return {c.add_dbl(o.c)}; }
    template <int K>
    G16_HD Fp2p30 sub(const Fp2p30& o) const { return {c.template sub<K>(o.c)}; }
    G16_HD Fp2p30 neg2() const { return {c.neg2()}; }
    G16_HD static Fp2p30 cond_neg2(const Fp2p30& a, bool flip) { return flip ? a.neg2() : a; }
    G16_HD Fp2p30 mul(const Fp2p30& o) const { return {pair_mul_c(lane_hi(), comp0(c), comp1(c), o.c, swap(o.c))}; }
    // the products with a lazy subtraction folded in (Fp30::mul2_sub_k / mul_sub_x3): each lane subtracts from ITS component
    template <int K>
    G16_HD static B pair_mul_sub_k(bool hi, const B& a0, const B& a1, const B& mb, const B& ob, const B& sub) {
        const B y2 = sel(hi, ob, ob.neg16());
        return B::template mul2_sub_k<K>(a0, mb, a1, y2, sub);
    }
    template <int K>
    G16_HD Fp2p30 mul_sub_k(const Fp2p30& o, const Fp2p30& s) const { return {pair_mul_sub_k<K>(lane_hi(), comp0(c), comp1(c), o.c, swap(o.c), s.c)}; }
    G16_HD static B pair_sqr_sub_x3(bool hi, const B& a0, const B& a1, const B& u, const B& v) {
        B x, y;
        pair_sqr_operands(hi, a0, a1, x, y);
        return x.mul_sub_x3(y, u, v);
    }
    G16_HD Fp2p30 sqr_sub_x3(const Fp2p30& u, const Fp2p30& v) const { return {pair_sqr_sub_x3(lane_hi(), comp0(c), comp1(c), u.c, v.c)}; }
    G16_HD Fp2p30 sqr() const { return {pair_sqr_c(lane_hi(), comp0(c), comp1(c))}; }
    G16_HD static bool both(bool v) { return v && (swap32(v ? 1u : 0u) != 0u); }
    G16_HD bool maybe_zero() const { return both(c.maybe_zero()); }
    G16_HD bool is_zero_exact() const { return both(c.is_zero_exact()); }
    static constexpr int KM = 2, K2M = 4, KX = 8, KY = 4;
    G16_HD Fp2p30 settle() const { return *this; }
    // ---- batched-affine hooks (batch_affine.hpp): each lane of the pair handles its own component
    G16_HD Fp2p30 canonical_lt8p() const { return {c.canonical_lt8p()}; }
    G16_HD Fp2p30 neg_canonical() const { return {c.neg_canonical()}; }
    G16_HD bool equals_canonical(const Fp2p30& o) const { return both(c.same_limbs(o.c)); }
    static constexpr int PREFIX_LIMBS = B::NL;
    G16_HD void get_limbs(uint32_t* w) const { c.get_limbs(w); }
    G16_HD static Fp2p30 from_limbs(const uint32_t* w) { return {B::from_limbs(w)}; }
    template <class A>
    G16_HD static void store_point(A* dst, int64_t idx, const Fp2p30& x, const Fp2p30& y, bool identity) {
        Fp<P>* w = reinterpret_cast<Fp<P>*>(dst + idx);   // x.c0 x.c1 y.c0 y.c1
        const int k = lane_hi() ? 1 : 0;
        Fp<P> xw = Fp<P>::zero(), yw = Fp<P>::zero();
        if (!identity) { x.c.pack(xw.v); y.c.pack(yw.v); }
        w[k] = xw;
        w[2 + k] = yw;
    }
    typedef Fp2x30<P> Raw;
#ifndef G16_PAIR_MIN_WAVES
#define G16_PAIR_MIN_WAVES 2
#endif
#ifndef G16_PAIR_PREFETCH
#define G16_PAIR_PREFETCH false
#endif
    static constexpr int ACC_MIN_WAVES = G16_PAIR_MIN_WAVES;
    static constexpr bool ACC_PREFETCH = G16_PAIR_PREFETCH;
#ifndef G16_PAIR_PARKED
#define G16_PAIR_PARKED 1
#endif
    static constexpr bool ACC_PARKED = G16_PAIR_PARKED != 0;   // bucket pass: accumulator coordinates in LDS (AccParked)
    // ---- bucket-kernel hooks: two lanes per task, each touching only its half of every Fq2 value
    static constexpr int LANES_PER_TASK = 2;
    template <class A>
    G16_HD static bool load_point(const A* bases, int64_t idx, Fp2p30& px, Fp2p30& py) {
        const Fp<P>* w = reinterpret_cast<const Fp<P>*>(bases + idx);   // x.c0 x.c1 y.c0 y.c1
        const int k = lane_hi() ? 1 : 0;
        const Fp<P> xw = w[k], yw = w[2 + k];
        const bool zero_half = xw.is_zero() && yw.is_zero();
        if (both(zero_half)) return false;
        px.c = B::unpack(xw.v);
        py.c = B::unpack(yw.v);
        return true;
    }
    typedef Fp<P> PackedC;   // this lane's component of y as the table holds it
    template <class A>
    G16_HD static bool load_point_py(const A* bases, int64_t idx, Fp2p30& px, PackedC& yout) {
        const Fp<P>* w = reinterpret_cast<const Fp<P>*>(bases + idx);   // x.c0 x.c1 y.c0 y.c1
        const int k = lane_hi() ? 1 : 0;
        const Fp<P> xw = w[k], yw = w[2 + k];
        const bool zero_half = xw.is_zero() && yw.is_zero();
        if (both(zero_half)) return false;
        px.c = B::unpack(xw.v);
        yout = yw;
        return true;
    }
    G16_HD static Fp2p30 unpack_cond_neg(const PackedC& y, bool flip) { return {B::unpack_cond_neg(y, flip)}; }
};

// Partial sums between the MSM kernels: the accumulator's lazy limbs verbatim (no conversion, no product), in the layout of
// the one-lane field R (Fp30 for G1: 208 B; Fp2x30 for G2: 416 B).  Identity <=> zz is all-zero limbs (a non-identity zz is
// a product of non-zero residues, so its limbs cannot all vanish).
template <class R>
struct alignas(16) AccRaw {
    R x, y, zz, zzz;
};

// Lazy extended-Jacobian accumulator, F = Fp30<P> (G1), Fp2x30<P> or Fp2k30<P> (G2).  Invariants between
// calls (per base-field component), "tight" fields (product outputs < 1.5p): x < 7.5p, y < 3.5p, zz, zzz < 1.8p;
// Karatsuba field (product outputs < 6p): x, y < 2p (settled), zz, zzz < 6p.  The K of every subtraction is the
// field's trait constant: KM for a product output, K2M for a doubled one, KX / KY for x / y; identity kept as a flag.
template <class F>
struct Acc30 {
    typedef typename F::Std StdF;
    F x, y, zz, zzz;
    bool inf;

    G16_HD static Acc30 identity() {
        Acc30 a;
        a.x = a.y = a.zz = a.zzz = F::zero();
        a.inf = true;
        return a;
    }
    // mdbl-2008-s-1 on an affine point (px, py < 2p)
    G16_HD void set_double(const F& px, const F& py) {
        const F U = py.dbl();                     // < 4p
        if (U.is_zero_exact()) { inf = true; return; }
        const F V = U.sqr();
        const F W = U.mul(V);
        const F S = px.mul(V);
        const F X2 = px.sqr();
        const F M = X2.dbl().add(X2);             // < 3 * (square bound)
        const F X3 = M.sqr().template sub<F::K2M>(S.dbl()).settle();
        const F Y3 = F::mul_sub(M, S.template sub<F::KX>(X3), py, W).settle();
        x = X3; y = Y3; zz = V; zzz = W;
        inf = false;
    }
    // madd-2008-s: this += (px, py), affine, px,py < 2p, not the identity
    G16_HD void add_affine(const F& px, const F& py) {
        if (inf) {
            x = px; y = py; zz = F::one(); zzz = F::one();
            inf = false;
            return;
        }
        const F U2 = px.mul(zz);
        const F S2 = py.mul(zzz);
        const F Pd = U2.template sub<F::KX>(x);
        const F R = S2.template sub<F::KY>(y);
        if (Pd.maybe_zero()) {
            if (Pd.is_zero_exact()) {
                if (R.is_zero_exact()) set_double(px, py);
                else inf = true;
                return;
            }
        }
        const F PP = Pd.sqr();
        const F PPP = Pd.mul(PP);
        const F Q = x.mul(PP);
        const F X3 = R.sqr().template sub<F::KM>(PPP).template sub<F::K2M>(Q.dbl()).settle();
        const F Y3 = F::mul_sub(R, Q.template sub<F::KX>(X3), y, PPP).settle();
        x = X3;
        y = Y3;
        zz = zz.mul(PP);
        zzz = zzz.mul(PPP);
    }
    // dbl-2008-s-1
    G16_HD void dbl() {
        if (inf) return;
        const F U = y.dbl();
        if (U.is_zero_exact()) { inf = true; return; }
        const F V = U.sqr();
        const F W = U.mul(V);
        const F S = x.mul(V);
        const F X2 = x.sqr();
        const F M = X2.dbl().add(X2);
        const F X3 = M.sqr().template sub<F::K2M>(S.dbl()).settle();
        const F Y3 = F::mul_sub(M, S.template sub<F::KX>(X3), y, W).settle();
        x = X3; y = Y3;
        zz = V.mul(zz);
        zzz = W.mul(zzz);
    }
    // add-2008-s: this += o
    G16_HD void add(const Acc30& o) {
        if (o.inf) return;
        if (inf) { *this = o; return; }
        const F U1 = x.mul(o.zz);
        const F U2 = o.x.mul(zz);
        const F S1 = y.mul(o.zzz);
        const F S2 = o.y.mul(zzz);
        const F Pd = U2.template sub<F::KM>(U1);
        const F R = S2.template sub<F::KM>(S1);
        if (Pd.maybe_zero()) {
            if (Pd.is_zero_exact()) {
                if (R.is_zero_exact()) dbl();
                else inf = true;
                return;
            }
        }
        const F PP = Pd.sqr();
        const F PPP = Pd.mul(PP);
        const F Q = U1.mul(PP);
        const F X3 = R.sqr().template sub<F::KM>(PPP).template sub<F::K2M>(Q.dbl()).settle();
        const F Y3 = F::mul_sub(R, Q.template sub<F::KX>(X3), S1, PPP).settle();
        x = X3;
        y = Y3;
        zz = zz.mul(o.zz).mul(PP);
        zzz = zzz.mul(o.zzz).mul(PPP);
    }
    // k * this for a small scalar (double-and-add, MSB first)
    G16_HD Acc30 mul_small(uint32_t k) const {
        Acc30 acc = identity();
        for (int i = 31; i >= 0; --i) {
            acc.dbl();
            if ((k >> i) & 1) acc.add(*this);
        }
        return acc;
    }
    // raw limb storage (see AccRaw): every lane of the task writes the limbs it owns
    G16_HD void store_raw(AccRaw<typename F::Raw>* dst) const {
        typedef typename F::Raw R;
        if constexpr (F::LANES_PER_TASK == 1) {
            static_assert(sizeof(F) == sizeof(R), "one-lane fields share the raw layout");
            F* w = reinterpret_cast<F*>(dst);
            w[0] = x; w[1] = y; w[2] = inf ? F::zero() : zz; w[3] = zzz;
        } else {
            typedef typename F::B B30;
            B30* w = reinterpret_cast<B30*>(dst);   // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
            const int k = F::lane_hi() ? 1 : 0;
            w[0 + k] = x.c; w[2 + k] = y.c; w[4 + k] = inf ? B30::zero() : zz.c; w[6 + k] = zzz.c;
        }
    }
    G16_HD static Acc30 load_raw(const AccRaw<typename F::Raw>& p) {
        Acc30 a;
        if constexpr (F::LANES_PER_TASK == 1) {
            const F* w = reinterpret_cast<const F*>(&p);
            a.x = w[0]; a.y = w[1]; a.zz = w[2]; a.zzz = w[3];
            a.inf = a.zz.raw_zero();
        } else {   // lane pair: each lane takes the component it owns
            typedef typename F::B B30;
            const B30* w = reinterpret_cast<const B30*>(&p);   // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
            const int k = F::lane_hi() ? 1 : 0;
            a.x.c = w[0 + k]; a.y.c = w[2 + k]; a.zz.c = w[4 + k]; a.zzz.c = w[6 + k];
            a.inf = F::both(a.zz.c.raw_zero());
        }
        return a;
    }
    // leave the device (pair-aware form of *dst = to_std()): standard-form XYZZ, the identity as all-zero words
    G16_HD void store_std(XYZZ<StdF>* dst) const {
        if constexpr (F::LANES_PER_TASK == 1) {
            *dst = to_std();
        } else {
            typedef typename F::B B30;
            typedef typename B30::Std W;
            W* w = reinterpret_cast<W*>(dst);
            const int k = F::lane_hi() ? 1 : 0;
            const W z = W::zero();
            w[0 + k] = inf ? z : x.c.to_std();
            w[2 + k] = inf ? z : y.c.to_std();
            w[4 + k] = inf ? z : zz.c.to_std();
            w[6 + k] = inf ? z : zzz.c.to_std();
        }
    }
    // bucket-kernel output: every lane of the task writes the words it owns
    G16_HD void store_packed(XYZZ<StdF>* dst) const {
        if constexpr (F::LANES_PER_TASK == 1) {
            *dst = to_packed();
        } else {
            typedef typename F::B B30;
            typedef typename B30::Std W;   // Fp<P>: one packed base-field element
            W* w = reinterpret_cast<W*>(dst);  // x.c0 x.c1 y.c0 y.c1 zz.c0 zz.c1 zzz.c0 zzz.c1
            const int k = F::lane_hi() ? 1 : 0;
            const W z = W::zero();
            w[0 + k] = inf ? z : x.c.to_packed();
            w[2 + k] = inf ? z : y.c.to_packed();
            w[4 + k] = inf ? z : zz.c.to_packed();
            w[6 + k] = inf ? z : zzz.c.to_packed();
        }
    }
    // storage form between kernels: canonical coordinates in the R' domain packed in words; identity = all zero
    G16_HD XYZZ<StdF> to_packed() const {
        if (inf) return XYZZ<StdF>::identity();
        return {x.to_packed(), y.to_packed(), zz.to_packed(), zzz.to_packed()};
    }
    G16_HD static Acc30 from_packed(const XYZZ<StdF>& p) {
        Acc30 a;
        a.inf = p.zz.is_zero();
        a.x = F::from_packed(p.x); a.y = F::from_packed(p.y); a.zz = F::from_packed(p.zz); a.zzz = F::from_packed(p.zzz);
        return a;
    }
    // leave the device: standard-form XYZZ (canonical coordinates, arkworks Montgomery radix)
    G16_HD XYZZ<StdF> to_std() const {
        if (inf) return XYZZ<StdF>::identity();
        return {x.to_std(), y.to_std(), zz.to_std(), zzz.to_std()};
    }
};

// The same accumulator with its four coordinates PARKED outside the register file (the bucket kernels: LDS, one column of words per
// lane).  Why: one lazy product already needs two operands, 2 NL 64-bit columns and the reduction's scratch -- ~110 of the 256
// registers that keep two waves per SIMD -- and madd-2008-s touches every coordinate exactly twice with a long gap in between
// (zz, zzz: first and last product; x, y: the differences and one late product).  Held in registers the allocator spilled them to
// HBM-backed scratch instead (G2 lane pair: 392 B per lane, ~2.3 GB each way per 2^22-point launch, with 0 bytes of LDS in use);
// parked, a coordinate costs 13 LDS words read twice and written once per addition and the live set of the formula is
// R, PP, PPP, Q plus the operand being fetched.  The products are issued so that each coordinate's second use follows as early
// as the data flow allows (zz * PP right after PP, zzz * PPP right after PPP).
// Store: F ld(int which) / void st(int which, const F&); every ld is a fresh read (the store orders it against its own st's).
// Same formulas, same K constants and therefore the same bounds as Acc30::add_affine (BoundF runs this code too).
template <class F, class Store>
struct AccParked {
    enum { CX = 0, CY = 1, CZZ = 2, CZZZ = 3 };
    Store s;
    bool inf;
    // Sign tracking (round 6).  madd-2008-s ends with Y3 = R (Q - X3) - y PPP: a DIFFERENCE of products, i.e. one operand negated
    // with its own carry pass (PPP -> 2p - PPP, three such operands for the lane pair).  The SUM R (X3 - Q) + y PPP = -Y3 needs
    // none -- and (X3, -Y3, ZZ, ZZZ) is simply the NEGATED result.  So the parked sum is allowed to be -A (`neg`): the next point
    // then enters with its sign flipped too (free: the signed digit already selects +-P, the two flips are one exclusive-or),
    // -A + (-P) = -(A + P), the formula's own negation makes it +(A + P), and `neg` toggles with every hot-path addition.
    // The cold branches (doubling, first point) keep the sign they find; gather() hands out +A (y -> 4p - y when neg: once per flush).
    bool neg = false;

    G16_HD void set_identity() { inf = true; neg = false; }
    G16_HD Acc30<F> gather_as_parked() const {   // the coordinates as they lie (the represented point is -A when neg)
        Acc30<F> a;
        a.inf = inf;
        if (inf) { a.x = a.y = a.zz = a.zzz = F::zero(); return a; }
        a.x = s.ld(CX); a.y = s.ld(CY); a.zz = s.ld(CZZ); a.zzz = s.ld(CZZZ);
        return a;
    }
    G16_HD Acc30<F> gather() const {
        Acc30<F> a = gather_as_parked();
        if (!inf && neg) a.y = F::zero().template sub<4>(a.y);   // y < 3.5p (the doubling's two-product Y3 is the largest) -> 4p - y < 4p
        return a;
    }
    G16_HD void scatter(const Acc30<F>& a) {
        inf = a.inf;
        neg = false;
        if (inf) return;
        s.st(CX, a.x); s.st(CY, a.y); s.st(CZZ, a.zz); s.st(CZZZ, a.zzz);
    }
    // mdbl-2008-s-1 on an affine point (px, py < 2p), parked like the addition: V and W go straight to their slots
    G16_HD void set_double(const F& px, const F& py) {
        const F U = py.dbl();                     // < 4p
        if (U.is_zero_exact()) { inf = true; return; }
        s.st(CZZ, U.sqr());
        s.st(CZZZ, U.mul(s.ld(CZZ)));
        const F S = px.mul(s.ld(CZZ));
        const F X2 = px.sqr();
        const F M = X2.dbl().add(X2);             // < 3 * (square bound)
        const F X3 = M.sqr().template sub<F::K2M>(S.dbl()).settle();
        s.st(CX, X3);
        s.st(CY, F::mul_sub(M, S.template sub<F::KX>(X3), py, s.ld(CZZZ)).settle());   // (cold path: two plain products, fewer registers)
        inf = false;
    }
    // madd-2008-s: this += (px, +-py), affine canonical px, py, not the identity; `minus`: subtract the point (the signed digit's sign)
    G16_HD void add_affine(const F& px, const F& py) { add_affine_signed(px, py, false); }
    G16_HD bool flip_for(bool minus) const {
#ifdef G16_NO_SIGN_TRACK
        return minus;
#else
        return minus != (neg && !inf);   // the point takes the parked sum's sign as well
#endif
    }
    G16_HD void add_affine_signed(const F& px, const F& py_in, bool minus) { add_affine_core(px, F::cond_neg2(py_in, flip_for(minus))); }
    // y as the window table holds it (packed words): negated there -- one subtract-with-borrow per word -- and unpacked once
    template <class YP>
    G16_HD void add_affine_packed(const F& px, const YP& yw, bool minus) { add_affine_core(px, F::unpack_cond_neg(yw, flip_for(minus))); }
    G16_HD void add_affine_core(const F& px, const F& py) {
        if (inf) {
            s.st(CX, px); s.st(CY, py); s.st(CZZ, F::one()); s.st(CZZZ, F::one());
            inf = false;
            neg = false;
            return;
        }
        const F Pd = px.template mul_sub_k<F::KX>(s.ld(CZZ), s.ld(CX));    // U2 - x
        const F R = py.template mul_sub_k<F::KY>(s.ld(CZZZ), s.ld(CY));    // S2 - y
        if (Pd.maybe_zero()) {
            if (Pd.is_zero_exact()) {   // P == +-Q: doubling (the sum is 2 (px, py), whatever the accumulator's scale) or cancellation
                if (R.is_zero_exact()) set_double(px, py);
                else inf = true;
                return;
            }
        }
#if defined(G16_NO_OPERAND_VIEWS) || defined(G16_NO_SIGN_TRACK)
        const F PP = Pd.sqr();
        s.st(CZZ, s.ld(CZZ).mul(PP));
        const F PPP = Pd.mul(PP);
        s.st(CZZZ, s.ld(CZZZ).mul(PPP));
        const F Q = s.ld(CX).mul(PP);
        static_assert(F::KM + F::K2M == 6, "sqr_sub_x3 subtracts from 6 p");
        const F X3 = R.sqr_sub_x3(PPP, Q).settle();
        s.st(CX, X3);
#ifdef G16_NO_SIGN_TRACK
        s.st(CY, F::mul_sub_fused(R, Q.template sub<F::KX>(X3), s.ld(CY), PPP).settle());
#else
        s.st(CY, F::mul_add_fused(R, X3.template sub<F::KM>(Q), s.ld(CY), PPP).settle());   // = -Y3: the parked sum changes sign
        neg = !neg;
#endif
#else
        // every operand that enters more than one product is prepared once (F::lhs / F::rhs: the lane pair's broadcasts / negation)
        const typename F::Lhs PdL = F::lhs(Pd);
        const F PP = F::sqr_v(PdL);
        const typename F::Rhs PPr = F::rhs(PP);
        s.st(CZZ, F::mul_v(F::lhs(s.ld(CZZ)), PPr));
        const F PPP = F::mul_v(PdL, PPr);
        const typename F::Rhs PPPr = F::rhs(PPP);
        s.st(CZZZ, F::mul_v(F::lhs(s.ld(CZZZ)), PPPr));
        const F Q = F::mul_v(F::lhs(s.ld(CX)), PPr);
        static_assert(F::KM + F::K2M == 6, "sqr_sub_x3 subtracts from 6 p");
        const typename F::Lhs RL = F::lhs(R);
        const F X3 = F::sqr_sub_x3_v(RL, PPP, Q).settle();   // R^2 + 6 p - (PPP + 2 Q), the subtraction inside the squaring's high columns
        s.st(CX, X3);
        // R (X3 - Q) + y PPP = -Y3: a SUM of products, so the parked sum changes sign (see `neg`)
        s.st(CY, F::mul_add_fused_v(RL, F::rhs(X3.template sub<F::KM>(Q)), F::lhs(s.ld(CY)), PPPr).settle());
        neg = !neg;
#endif
    }
};

// The FULL addition (add-2008-s: XYZZ + XYZZ, 12 products + 2 squarings) with BOTH operands behind stores: d += s.  The reduction
// kernels add partial sums that already live in memory (the bucket pass's flushed records, LDS tree nodes), so neither operand needs
// to sit in registers as a whole -- a coordinate is fetched right before the product that consumes it.  Register-resident (Acc30::add)
// the two operands alone are 8 coordinates = 104 registers per lane of a lane pair, and the G2 reduction kernels needed 496 registers:
// ONE wave per SIMD, and no bucket-pass wave beside it.  Streamed, the live set is S1, R, PP, PPP, Q plus one operand.
// d's store must support st(); s is read-only.  Same formulas, K constants and bounds as Acc30::add.  The exceptional cases (equal or
// opposite points) take the register form -- cold.
// dbl-2008-s-1 on a stored point, coordinates fetched where they are consumed (the cold branch of the streamed addition: kept as lean
// as the hot path so that the kernels' register budget is not set by a branch that almost never runs)
template <class F, class D>
G16_HD void acc_dbl_streamed(D& d, bool& d_inf) {
    enum { CX = 0, CY = 1, CZZ = 2, CZZZ = 3 };
    if (d_inf) return;
    const F U = d.ld(CY).dbl();
    if (U.is_zero_exact()) { d_inf = true; return; }
    const F V = U.sqr();
    const F W = U.mul(V);
    d.st(CZZ, V.mul(d.ld(CZZ)));
    d.st(CZZZ, W.mul(d.ld(CZZZ)));
    const F S = d.ld(CX).mul(V);
    const F X2 = d.ld(CX).sqr();
    const F M = X2.dbl().add(X2);
    const F X3 = M.sqr().template sub<F::K2M>(S.dbl()).settle();
    d.st(CX, X3);
    d.st(CY, F::mul_sub(M, S.template sub<F::KX>(X3), d.ld(CY), W).settle());
}

template <class F, class D, class S>
G16_HD void acc_add_streamed(D& d, bool& d_inf, const S& s, bool s_inf) {
    enum { CX = 0, CY = 1, CZZ = 2, CZZZ = 3 };
    if (s_inf) return;
    if (d_inf) {
        d.st(CX, s.ld(CX)); d.st(CY, s.ld(CY)); d.st(CZZ, s.ld(CZZ)); d.st(CZZZ, s.ld(CZZZ));
        d_inf = false;
        return;
    }
    const F U1 = d.ld(CX).mul(s.ld(CZZ));
    const F S1 = d.ld(CY).mul(s.ld(CZZZ));
    const F Pd = s.ld(CX).template mul_sub_k<F::KM>(d.ld(CZZ), U1);
    const F R = s.ld(CY).template mul_sub_k<F::KM>(d.ld(CZZZ), S1);
    if (Pd.maybe_zero()) {
        if (Pd.is_zero_exact()) {
            if (R.is_zero_exact()) acc_dbl_streamed<F>(d, d_inf);   // the same point: double it
            else d_inf = true;
            return;
        }
    }
    const F PP = Pd.sqr();
    d.st(CZZ, d.ld(CZZ).mul(s.ld(CZZ)).mul(PP));
    const F PPP = Pd.mul(PP);
    d.st(CZZZ, d.ld(CZZZ).mul(s.ld(CZZZ)).mul(PPP));
    const F Q = U1.mul(PP);
    static_assert(F::KM + F::K2M == 6, "sqr_sub_x3 subtracts from 6 p");
    const F X3 = R.sqr_sub_x3(PPP, Q).settle();
    d.st(CX, X3);
    d.st(CY, F::mul_sub(R, Q.template sub<F::KX>(X3), S1, PPP).settle());
}

// host-side Store for the self-tests: plain memory
template <class F>
struct ParkedArrayStore {
    F v[4];
    G16_HD F ld(int k) const { return v[k]; }
    G16_HD void st(int k, const F& a) { v[k] = a; }
};

}  // namespace g16
