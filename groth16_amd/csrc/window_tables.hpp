// Window tables of a proving-key query: table[j * n + i] = 2^(c j) * P_i, affine, in the bucket kernel's radix (msm.hip,
// "merged windows").  Built once per key by g16_pk_load; 13x the key at c = 20.
//
// Round 3 rewrite.  The first version kept W XYZZ points and W prefix products in per-lane arrays (17.6 KB of scratch per
// lane for G2), ran the c (W - 1) doublings per point in the standard 32-bit-limb arithmetic, and took 1.39 s of a 2.86 s key
// load for the G2 table alone.  This one
//   * works in the bucket kernels' own 30-bit lazy arithmetic (fp30.hpp): one lane per G1 point, a lane PAIR per G2 point
//     (Fp2p30: one Fq2 component per lane), two waves per SIMD, no scratch;
//   * doubles in Jacobian coordinates with a = 0 (3 products + 4 squarings per doubling against 6 + 3 in XYZZ);
//   * parks what the backward sweep of Montgomery's trick needs (X, Y, Z and the running product per row) in an explicit
//     limb-planar HBM buffer -- every store / load instruction of a wave touches 64 consecutive words -- instead of a
//     dynamically indexed private array;
//   * still spends ONE field inversion per point (Bernstein-Yang division steps, batch_affine.hpp) for all W - 1 rows.
// Per (point, row): 20 doublings (~122 product-equivalents) + 7 products + 1/12 inversion.
#pragma once
#include "fp30.hpp"
#include "batch_affine.hpp"

namespace g16 {

// (X, Y, Z) <- 2 (X, Y, Z) on y^2 = x^3 + b, lazy bounds per base-field component: in X < 5.8p, Y < 5.5p, Z < 3p (an affine
// start: < p, < p, one); out the same.
//   A = X^2, B = Y^2, D = X * 4B (= 4 X Y^2), E = 3A, X3 = E^2 - 2D, Y3 = E (D - X3) - 8 B^2, Z3 = 2 Y Z
template <class F>
G16_HD void jac30_double(F& X, F& Y, F& Z) {
    const F A = X.sqr();                                        // < 1.8p
    const F B2 = Y.sqr().dbl();                                 // 2B < 3.6p
    const F D = X.mul(B2.dbl());                                // X * 4B, < 1.5p
    const F E = A.dbl().add(A);                                 // 3A < 5.4p
    const F Z3 = Y.mul(Z).dbl();                                // < 3p
    const F X3 = E.sqr().template sub<4>(D.dbl());              // E^2 + 4p - 2D < 5.8p   (2D < 3p)
    const F C8 = B2.sqr().dbl();                                // 2 (2B)^2 = 8 B^2 < 3.6p
    Y = E.mul(D.template sub<8>(X3)).template sub<4>(C8);       // < 5.5p               (X3 < 8p, C8 < 4p)
    X = X3;
    Z = Z3;
}

// The arithmetic of one task (= one point): rows 1 .. W - 1 of its window table.  F is the lazy field as the task's lane(s)
// hold it (Fp30 for G1, the lane-pair Fp2p30 for G2; the host self-test runs a two-component emulation of the pair); IO is
// where the task's data lives:
//   bool load(F& x, F& y)            the point in the R' radix, canonical; false for the identity
//   void store(int row, x, y)        canonical affine coordinates of a row (row 0 included)
//   void store_identity(int row)
//   void put(int slot, v) / F get(int slot)     4 (W - 1) parking slots
//   bool is_zero(v)                  exact test of a lazy value (< 16p per component), uniform over the task's lanes
//   F canonical(v)                   lazy (< 2p per component) -> [0, p)
template <class F, class IO>
G16_HD void window_table_task(IO& io, int c, int W) {
    F X, Y;
    if (!io.load(X, Y)) {
        for (int j = 0; j < W; ++j) io.store_identity(j);
        return;
    }
    io.store(0, X, Y);
    F Z = F::one(), run = F::one();
    // forward: rows 1 .. live - 1 are finite; a point of 2-power order (never a subgroup point of these curves, but the caller's
    // bases are not checked) reaches the identity at some row, from which on every row is the identity
    int live = W;
    for (int j = 1; j < W; ++j) {
        for (int d = 0; d < c; ++d) jac30_double(X, Y, Z);
        if (io.is_zero(Z)) { live = j; break; }
        io.put(4 * (j - 1) + 0, X);
        io.put(4 * (j - 1) + 1, Y);
        io.put(4 * (j - 1) + 2, Z);
        io.put(4 * (j - 1) + 3, run);   // product of the Z's of rows 1 .. j - 1
        run = run.mul(Z);
    }
    F inv = batch_inverse(run);          // run != 0: a product of non-zero Z's (or one)
    for (int j = W - 1; j >= live; --j) io.store_identity(j);
    for (int j = live - 1; j >= 1; --j) {
        const F Zj = io.get(4 * (j - 1) + 2);
        const F zi = inv.mul(io.get(4 * (j - 1) + 3));   // 1 / Z_j
        inv = inv.mul(Zj);
        const F zi2 = zi.sqr();
        const F ax = io.get(4 * (j - 1) + 0).mul(zi2);              // X / Z^2
        const F ay = io.get(4 * (j - 1) + 1).mul(zi2.mul(zi));      // Y / Z^3
        io.store(j, io.canonical(ax), io.canonical(ay));
    }
}

// ---- device side: where a lane of the task finds its share of the data ----------------------------------------------------
template <class F30> struct TableLane;   // what one lane of a task holds of a field value
template <class P>
struct TableLane<Fp30<P>> {
    typedef Fp30<P> B;
    G16_HD static const B& comp(const Fp30<P>& v) { return v; }
    G16_HD static Fp30<P> wrap(const B& b) { return b; }
    G16_HD static bool all(bool v) { return v; }
    G16_HD static int part() { return 0; }                      // which Fq of an affine coordinate this lane owns
    static constexpr int PARTS = 1;                             // base-field elements per coordinate
};
template <class P>
struct TableLane<Fp2p30<P>> {
    typedef Fp30<P> B;
    G16_HD static const B& comp(const Fp2p30<P>& v) { return v.c; }
    G16_HD static Fp2p30<P> wrap(const B& b) { return Fp2p30<P>{b}; }
    G16_HD static bool all(bool v) { return Fp2p30<P>::both(v); }
    G16_HD static int part() { return Fp2p30<P>::lane_hi() ? 1 : 0; }
    static constexpr int PARTS = 2;
};

// src / table: arrays of packed base-field elements, a point = 2 * PARTS of them (x parts, then y parts); this lane reads / writes
// element `part` of each coordinate.  park: slot s, limb l of this lane at park[(s * NL + l) * stride].
template <class F30>
struct TableDeviceIO {
    typedef TableLane<F30> TL;
    typedef typename TL::B B;
    typedef typename B::Std W32;           // one packed base-field element
    static constexpr int NL = B::NL;
    const W32* src_pt;
    W32* table_pt;
    uint64_t row_stride;                  // packed elements between two rows of the table
    uint32_t* park;
    uint64_t stride;
    G16_HD bool load(F30& x, F30& y) const {
        const int k = TL::part();
        const W32 xs = src_pt[k], ys = src_pt[TL::PARTS + k];
        if (TL::all(xs.is_zero() && ys.is_zero())) return false;
        const W32 x0 = B::std_to_r30(xs), y0 = B::std_to_r30(ys);   // canonical x R', y R'
        x = TL::wrap(B::unpack(x0.v));
        y = TL::wrap(B::unpack(y0.v));
        return true;
    }
    G16_HD void store(int row, const F30& x, const F30& y) const {
        const int k = TL::part();
        W32 ox, oy;
        TL::comp(x).pack(ox.v);
        TL::comp(y).pack(oy.v);
        table_pt[(uint64_t)row * row_stride + k] = ox;
        table_pt[(uint64_t)row * row_stride + TL::PARTS + k] = oy;
    }
    G16_HD void store_identity(int row) const {
        const int k = TL::part();
        table_pt[(uint64_t)row * row_stride + k] = W32::zero();
        table_pt[(uint64_t)row * row_stride + TL::PARTS + k] = W32::zero();
    }
    G16_HD void put(int slot, const F30& v) const {
        const B& b = TL::comp(v);
        G16_UNROLL for (int l = 0; l < NL; ++l) park[((uint64_t)slot * NL + l) * stride] = b.l[l];
    }
    G16_HD F30 get(int slot) const {
        B b;
        G16_UNROLL for (int l = 0; l < NL; ++l) b.l[l] = park[((uint64_t)slot * NL + l) * stride];
        return TL::wrap(b);
    }
    G16_HD bool is_zero(const F30& v) const { return TL::all(TL::comp(v).maybe_zero()) && TL::all(TL::comp(v).is_zero_exact()); }
    G16_HD F30 canonical(const F30& v) const { return TL::wrap(TL::comp(v).canonical_lt2p()); }
};

}  // namespace g16
