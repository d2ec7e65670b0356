// Short-Weierstrass (a = 0) group arithmetic in extended Jacobian "XYZZ" coordinates
// (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2), generic over the coordinate field (Fp for G1, Fp2
// for G2).  XYZZ mixed addition is 8M+2S against 7M+4S for Jacobian and needs no field
// inversion, which is what the bucket-accumulation kernel spends its time in.
//
// Replaces, for the prover path, ark-ec's Projective/Affine group law used at
// /root/reference/src/prover.rs:76,90,94,100,112,114,128-130,265-267.
// Result points are canonical group elements, so any correct formula set is bit-exact
// with the reference after into_affine().
#pragma once
#include "field.hpp"

namespace g16 {

// Affine point; the identity is encoded as (0, 0), which is on neither curve (b != 0).
template <class F>
struct alignas(16) Affine {
    F x, y;
    G16_HD static Affine identity() { return {F::zero(), F::zero()}; }
    G16_HD bool is_identity() const { return x.is_zero() && y.is_zero(); }
    G16_HD Affine neg() const { return {x, y.neg()}; }  // neg(0) = 0 keeps the identity
    G16_HD bool operator==(const Affine& o) const { return x == o.x && y == o.y; }
};

template <class F>
struct alignas(16) XYZZ {
    F x, y, zz, zzz;

    G16_HD static XYZZ identity() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    G16_HD bool is_identity() const { return zz.is_zero(); }
    G16_HD static XYZZ from_affine(const Affine<F>& p) {
        if (p.is_identity()) return identity();
        return {p.x, p.y, F::one(), F::one()};
    }
    G16_HD XYZZ neg() const { return {x, y.neg(), zz, zzz}; }

    // dbl-2008-s-1 (a = 0)
    G16_HD XYZZ dbl() const {
        if (is_identity()) return *this;
        F U = y.dbl();
        if (U.is_zero()) return identity();
        F V = U.sqr();
        F W = U * V;
        F S = x * V;
        F X2 = x.sqr();
        F M = X2.dbl() + X2;
        F X3 = M.sqr() - S.dbl();
        F Y3 = M * (S - X3) - W * y;
        return {X3, Y3, V * zz, W * zzz};
    }
    // mdbl-2008-s-1: double an affine point
    G16_HD static XYZZ dbl_affine(const Affine<F>& p) {
        if (p.is_identity()) return identity();
        F U = p.y.dbl();
        if (U.is_zero()) return identity();
        F V = U.sqr();
        F W = U * V;
        F S = p.x * V;
        F X2 = p.x.sqr();
        F M = X2.dbl() + X2;
        F X3 = M.sqr() - S.dbl();
        F Y3 = M * (S - X3) - W * p.y;
        return {X3, Y3, V, W};
    }
    // madd-2008-s: this += affine p   (handles identity, doubling and inverse cases)
    G16_HD void add_affine(const Affine<F>& p) {
        if (p.is_identity()) return;
        if (is_identity()) {
            x = p.x; y = p.y; zz = F::one(); zzz = F::one();
            return;
        }
        F U2 = p.x * zz;
        F S2 = p.y * zzz;
        F Pd = U2 - x;
        F R = S2 - y;
        if (Pd.is_zero()) {
            if (R.is_zero()) *this = dbl_affine(p);
            else *this = identity();
            return;
        }
        F PP = Pd.sqr();
        F PPP = Pd * PP;
        F Q = x * PP;
        F X3 = R.sqr() - PPP - Q.dbl();
        F Y3 = R * (Q - X3) - y * PPP;
        x = X3;
        y = Y3;
        zz = zz * PP;
        zzz = zzz * PPP;
    }
    // add-2008-s: this += o
    G16_HD void add(const XYZZ& o) {
        if (o.is_identity()) return;
        if (is_identity()) { *this = o; return; }
        F U1 = x * o.zz;
        F U2 = o.x * zz;
        F S1 = y * o.zzz;
        F S2 = o.y * zzz;
        F Pd = U2 - U1;
        F R = S2 - S1;
        if (Pd.is_zero()) {
            if (R.is_zero()) *this = dbl();
            else *this = identity();
            return;
        }
        F PP = Pd.sqr();
        F PPP = Pd * PP;
        F Q = U1 * PP;
        F X3 = R.sqr() - PPP - Q.dbl();
        F Y3 = R * (Q - X3) - S1 * PPP;
        x = X3;
        y = Y3;
        zz = zz * o.zz * PP;
        zzz = zzz * o.zzz * PPP;
    }
    // scalar multiple by a canonical little-endian integer (nbits bits), MSB-first
    G16_HD_NOINLINE XYZZ mul_bits(const uint32_t* k, int nbits) const {
        XYZZ acc = identity();
        for (int i = nbits - 1; i >= 0; --i) {
            acc = acc.dbl();
            if ((k[i >> 5] >> (i & 31)) & 1) acc.add(*this);
        }
        return acc;
    }
    // 1/ZZZ -> y; (ZZ/ZZZ)^2 = 1/ZZ -> x
    G16_HD_NOINLINE Affine<F> to_affine() const {
        if (is_identity()) return Affine<F>::identity();
        F i3 = zzz.inverse();
        F iz = i3 * zz;  // = 1/Z
        return {x * iz.sqr(), y * i3};
    }
};

// curve bundles ------------------------------------------------------------------------
template <class FRP, class FQP, class CONSTS, int ID>
struct CurveT {
    static constexpr int CURVE_ID = ID;
    typedef Fp<FRP> Fr;
    typedef Fp<FQP> Fq;
    typedef Fp2<FQP> Fq2;
    typedef Affine<Fq> G1A;
    typedef XYZZ<Fq> G1X;
    typedef Affine<Fq2> G2A;
    typedef XYZZ<Fq2> G2X;
    typedef CONSTS K;
    static constexpr int TWO_ADICITY = CONSTS::TWO_ADICITY;

    G16_HD static Fr fr_generator() { Fr r; G16_UNROLL for (int i = 0; i < Fr::N; ++i) r.v[i] = CONSTS::fr_generator(i); return r; }
    G16_HD static Fr fr_generator_inv() { Fr r; G16_UNROLL for (int i = 0; i < Fr::N; ++i) r.v[i] = CONSTS::fr_generator_inv(i); return r; }
    G16_HD static Fr two_adic_root() { Fr r; G16_UNROLL for (int i = 0; i < Fr::N; ++i) r.v[i] = CONSTS::two_adic_root(i); return r; }
    G16_HD static G1A g1_generator() {
        G1A p;
        G16_UNROLL for (int i = 0; i < Fq::N; ++i) { p.x.v[i] = CONSTS::g1_x(i); p.y.v[i] = CONSTS::g1_y(i); }
        return p;
    }
    G16_HD static G2A g2_generator() {
        G2A p;
        G16_UNROLL for (int i = 0; i < Fq::N; ++i) {
            p.x.c0.v[i] = CONSTS::g2_x0(i); p.x.c1.v[i] = CONSTS::g2_x1(i);
            p.y.c0.v[i] = CONSTS::g2_y0(i); p.y.c1.v[i] = CONSTS::g2_y1(i);
        }
        return p;
    }
    G16_HD static Fq b1() { Fq r; G16_UNROLL for (int i = 0; i < Fq::N; ++i) r.v[i] = CONSTS::b1(i); return r; }
    G16_HD static Fq2 b2() {
        Fq2 r;
        G16_UNROLL for (int i = 0; i < Fq::N; ++i) { r.c0.v[i] = CONSTS::b2_c0(i); r.c1.v[i] = CONSTS::b2_c1(i); }
        return r;
    }
};

typedef CurveT<Bls12_381FrP, Bls12_381FqP, Bls12_381Consts, 0> Bls12_381;
typedef CurveT<Bn254FrP, Bn254FqP, Bn254Consts, 1> Bn254;

}  // namespace g16
