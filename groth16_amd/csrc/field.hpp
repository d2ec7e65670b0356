// Montgomery prime-field arithmetic on 32-bit limbs, shared by gfx950 kernels and the host
// glue.  Element layout = arkworks' in-memory form: little-endian limbs of a*R mod p with
// R = 2^(32*N) (= 2^(64*N/2), ark-ff MontBackend), so buffers cross the C ABI untouched.
//
// gfx950 has no 64-bit multiplier; the widest integer multiply-add is v_mad_u64_u32
// (32x32 + 64 -> 64).  Every product below is written as (u64)a*b + c [+ d] with 32-bit
// a,b,c,d, which cannot overflow 64 bits and lowers to exactly one v_mad_u64_u32 plus carry
// adds.  No MFMA: this is modular integer arithmetic, not a dense contraction.
#pragma once
#include <cstdint>
#include <cstring>
#include "hd.hpp"
#include "params_gen.hpp"

#ifndef G16_NOINLINE_MUL_LIMBS
#define G16_NOINLINE_MUL_LIMBS 12
#endif

namespace g16 {

template <class P>
struct alignas(16) Fp {
    static constexpr int N = P::N;
    typedef P Params;
    uint32_t v[N];

    G16_HD static Fp zero() {
        Fp r;
        G16_UNROLL for (int i = 0; i < N; ++i) r.v[i] = 0;
        return r;
    }
    G16_HD static Fp one() {
        Fp r;
        G16_UNROLL for (int i = 0; i < N; ++i) r.v[i] = P::r1(i);
        return r;
    }
    G16_HD static Fp r2() {
        Fp r;
        G16_UNROLL for (int i = 0; i < N; ++i) r.v[i] = P::r2(i);
        return r;
    }
    G16_HD bool is_zero() const {
        uint32_t a = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) a |= v[i];
        return a == 0;
    }
    G16_HD bool operator==(const Fp& o) const {
        uint32_t a = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) a |= v[i] ^ o.v[i];
        return a == 0;
    }
    G16_HD bool operator!=(const Fp& o) const { return !(*this == o); }
    G16_HD bool is_one() const { return *this == one(); }

    // r = t - p if t >= p else t      (t < 2p < 2^(32N))
    G16_HD static void reduce_once(uint32_t* t) {
        uint32_t d[N];
        int64_t br = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            br += (int64_t)t[i] - (int64_t)P::mod(i);
            d[i] = (uint32_t)br;
            br >>= 32;
        }
        bool ge = (br == 0);
        G16_UNROLL for (int i = 0; i < N; ++i) t[i] = ge ? d[i] : t[i];
    }

    G16_HD Fp operator+(const Fp& o) const {
        Fp r;
        uint64_t c = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            c += (uint64_t)v[i] + o.v[i];
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
        reduce_once(r.v);
        return r;
    }
    G16_HD Fp operator-(const Fp& o) const {
        Fp r;
        int64_t br = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            br += (int64_t)v[i] - (int64_t)o.v[i];
            r.v[i] = (uint32_t)br;
            br >>= 32;
        }
        uint32_t mask = (uint32_t)br;  // 0 or 0xffffffff
        uint64_t c = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            c += (uint64_t)r.v[i] + (P::mod(i) & mask);
            r.v[i] = (uint32_t)c;
            c >>= 32;
        }
        return r;
    }
    G16_HD Fp neg() const {
        Fp z = zero();
        return is_zero() ? z : (z - *this);
    }
    G16_HD Fp dbl() const { return *this + *this; }

    // Montgomery product, coarsely-integrated operand scanning with the reduction row merged
    // into the multiplication row (2N^2 + N v_mad_u64_u32 in total).
    // Wide fields keep the ~900-instruction product out of line: an XYZZ addition inlines
    // 10 (G1) or 30 (G2) products and would otherwise overflow the 64 KB instruction cache.
    G16_HD Fp operator*(const Fp& o) const {
        if constexpr (N >= G16_NOINLINE_MUL_LIMBS) return mul_outlined(*this, o);
        else return mul_inlined(o);
    }
#ifdef G16_MUL_BYVAL
    G16_HD_NOINLINE static Fp mul_outlined(Fp a, Fp b) { return a.mul_inlined(b); }
#else
    G16_HD_NOINLINE static Fp mul_outlined(const Fp& a, const Fp& b) { return a.mul_inlined(b); }
#endif
    G16_HD Fp mul_inlined(const Fp& o) const {
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__SIZEOF_INT128__)
        // host: the same Montgomery product on 64-bit limbs (R = 2^(32N) = 2^(64 N/2) is the same radix),
        // ~3x faster on x86 than the 32-bit formulation that suits the GPU
        {
            constexpr int M = N / 2;
            typedef unsigned __int128 u128;
            uint64_t a[M], b[M], p[M], t[M + 2];
            for (int i = 0; i < M; ++i) {
                a[i] = (uint64_t)v[2 * i] | ((uint64_t)v[2 * i + 1] << 32);
                b[i] = (uint64_t)o.v[2 * i] | ((uint64_t)o.v[2 * i + 1] << 32);
                p[i] = (uint64_t)P::mod(2 * i) | ((uint64_t)P::mod(2 * i + 1) << 32);
            }
            // -p^-1 mod 2^64 from the 32-bit constant by one Newton step
            const uint64_t inv32 = P::INV;               // -p^-1 mod 2^32
            uint64_t x = (uint64_t)0 - inv32;            // p^-1 mod 2^32 (as a 64-bit value, correct in the low 32 bits)
            x *= 2 - p[0] * x;                           // now correct mod 2^64
            const uint64_t inv64 = (uint64_t)0 - x;
            for (int i = 0; i < M + 2; ++i) t[i] = 0;
            for (int i = 0; i < M; ++i) {
                uint64_t carry = 0;
                for (int j = 0; j < M; ++j) {
                    const u128 y = (u128)a[j] * b[i] + t[j] + carry;
                    t[j] = (uint64_t)y;
                    carry = (uint64_t)(y >> 64);
                }
                u128 sacc = (u128)t[M] + carry;
                t[M] = (uint64_t)sacc;
                t[M + 1] = (uint64_t)(sacc >> 64);
                const uint64_t mq = t[0] * inv64;
                u128 y = (u128)mq * p[0] + t[0];
                carry = (uint64_t)(y >> 64);
                for (int j = 1; j < M; ++j) {
                    y = (u128)mq * p[j] + t[j] + carry;
                    t[j - 1] = (uint64_t)y;
                    carry = (uint64_t)(y >> 64);
                }
                sacc = (u128)t[M] + carry;
                t[M - 1] = (uint64_t)sacc;
                t[M] = t[M + 1] + (uint64_t)(sacc >> 64);
            }
            Fp r;
            for (int i = 0; i < M; ++i) { r.v[2 * i] = (uint32_t)t[i]; r.v[2 * i + 1] = (uint32_t)(t[i] >> 32); }
            reduce_once(r.v);
            return r;
        }
#endif
        uint32_t t[N];
        G16_UNROLL for (int i = 0; i < N; ++i) t[i] = 0;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            const uint32_t bi = o.v[i];
            uint64_t x = (uint64_t)v[0] * bi + t[0];
            const uint32_t m = (uint32_t)x * P::INV;
            uint64_t y = (uint64_t)m * P::mod(0) + (uint32_t)x;
            uint32_t c1 = (uint32_t)(x >> 32), c2 = (uint32_t)(y >> 32);
            G16_UNROLL for (int j = 1; j < N; ++j) {
                x = (uint64_t)v[j] * bi + t[j] + c1;
                c1 = (uint32_t)(x >> 32);
                y = (uint64_t)m * P::mod(j) + (uint32_t)x + c2;
                c2 = (uint32_t)(y >> 32);
                t[j - 1] = (uint32_t)y;
            }
            t[N - 1] = c1 + c2;  // < 2^32 because the running value stays < 2p < 2^(32N)
        }
        reduce_once(t);
        Fp r;
        G16_UNROLL for (int i = 0; i < N; ++i) r.v[i] = t[i];
        return r;
    }
    G16_HD Fp sqr() const { return *this * *this; }

    // Montgomery form -> canonical integer (ark-ff into_bigint): one reduction by R
    G16_HD void to_canonical(uint32_t* out) const {
        uint32_t t[N];
        G16_UNROLL for (int i = 0; i < N; ++i) t[i] = v[i];
        G16_UNROLL for (int i = 0; i < N; ++i) {
            const uint32_t m = t[0] * P::INV;
            uint64_t y = (uint64_t)m * P::mod(0) + t[0];
            uint32_t c = (uint32_t)(y >> 32);
            G16_UNROLL for (int j = 1; j < N; ++j) {
                y = (uint64_t)m * P::mod(j) + t[j] + c;
                c = (uint32_t)(y >> 32);
                t[j - 1] = (uint32_t)y;
            }
            t[N - 1] = c;
        }
        reduce_once(t);
        G16_UNROLL for (int i = 0; i < N; ++i) out[i] = t[i];
    }
    G16_HD static Fp from_canonical(const uint32_t* in) {
        Fp a;
        G16_UNROLL for (int i = 0; i < N; ++i) a.v[i] = in[i];
        return a * r2();
    }
    G16_HD static Fp from_u64(uint64_t k) {
        uint32_t t[N];
        G16_UNROLL for (int i = 0; i < N; ++i) t[i] = 0;
        t[0] = (uint32_t)k;
        t[1] = (uint32_t)(k >> 32);
        return from_canonical(t);
    }

    // generic square-and-multiply; exponent as 32-bit limbs
    G16_HD_NOINLINE Fp pow(const uint32_t* e, int nlimbs) const {
        Fp acc = one();
        bool started = false;
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            if (started) acc = acc.sqr();
            if ((e[i >> 5] >> (i & 31)) & 1) {
                acc = acc * *this;
                started = true;
            }
        }
        return acc;
    }
    G16_HD Fp pow_u64(uint64_t e) const {
        uint32_t t[2] = {(uint32_t)e, (uint32_t)(e >> 32)};
        return pow(t, 2);
    }
    // Fermat inverse; the caller guarantees a != 0
    G16_HD_NOINLINE Fp inverse() const {
        uint32_t e[N];
        for (int i = 0; i < N; ++i) e[i] = P::pm2(i);
        return pow(e, N);
    }
};

// Fq2 = Fq[u]/(u^2 + 1): the quadratic extension of both supported curves
template <class P>
struct alignas(16) Fp2 {
    typedef Fp<P> Base;
    Base c0, c1;

    G16_HD static Fp2 zero() { return {Base::zero(), Base::zero()}; }
    G16_HD static Fp2 one() { return {Base::one(), Base::zero()}; }
    G16_HD bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    G16_HD bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    G16_HD bool operator!=(const Fp2& o) const { return !(*this == o); }
    G16_HD Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    G16_HD Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    G16_HD Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    G16_HD Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    G16_HD Fp2 operator*(const Fp2& o) const { return mul_outlined(*this, o); }
    G16_HD Fp2 sqr() const { return sqr_outlined(*this); }
    G16_HD_NOINLINE static Fp2 mul_outlined(const Fp2& a, const Fp2& b) { return a.mul_inlined(b); }
    G16_HD_NOINLINE static Fp2 sqr_outlined(const Fp2& a) { return a.sqr_inlined(); }
    G16_HD Fp2 mul_inlined(const Fp2& o) const {  // Karatsuba, 3 base multiplications
        Base a = c0 * o.c0, b = c1 * o.c1;
        Base c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    G16_HD Fp2 sqr_inlined() const {  // complex squaring, 2 base multiplications
        Base a = (c0 + c1) * (c0 - c1);
        Base b = c0 * c1;
        return {a, b.dbl()};
    }
    G16_HD_NOINLINE Fp2 inverse() const {
        Base n = (c0.sqr() + c1.sqr()).inverse();
        return {c0 * n, (c1 * n).neg()};
    }
};

}  // namespace g16
