// Canonical (de)serialisation of curve points: SURVEY.md row f1, the data format either side of the prover path
// (Proof / VerifyingKey / ProvingKey derive CanonicalSerialize, /root/reference/src/data_structures.rs:8,31,125).
// Host code only.  The byte formats live in un-vendored crates and are restated from their published definitions
// ("[EXT-MEM]": cannot be checked against the reference in this environment; the only external known answer held is the
// IETF/zcash compressed BLS12-381 G1 generator):
//   * BLS12-381 (ark-bls12-381 overrides the default with the zcash / IETF format): big-endian coordinates, for Fq2 c1 before
//     c0; top three bits of the first byte: 0x80 compressed, 0x40 infinity, 0x20 (compressed only) y is the larger of {y, -y}.
//   * BN254 (ark-ec short-Weierstrass default, ark-serialize flags): little-endian coordinates, for Fq2 c0 before c1; top two
//     bits of the LAST byte of the last coordinate written: 0x80 y is the larger of {y, -y}, 0x40 infinity -- compressed
//     writes x with the flags, uncompressed writes x then y with the flags.
// "Larger" compares canonical integers; Fq2 compares c1 first, then c0.  The identity is (0, 0) with the infinity flag.
#include "internal.hpp"
#include <atomic>
#include <thread>

namespace g16 {

namespace {

// f(i) for i < n on the host's threads (a 2^22-constraint key holds 2 * 10^7 points; decompression is a 381-bit
// exponentiation per point, the subgroup check a 255-bit scalar multiplication)
template <class Fn>
int parallel_points(uint64_t n, Fn f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 64) nt = 64;
    if (n < 256 || nt == 1) {
        for (uint64_t i = 0; i < n; ++i) G16_TRY(f(i));
        return G16_OK;
    }
    std::atomic<uint64_t> next{0};
    std::atomic<int> status{G16_OK};
    auto work = [&]() {
        for (;;) {
            const uint64_t lo = next.fetch_add(64);
            if (lo >= n || status.load() != G16_OK) return;
            const uint64_t hi = lo + 64 < n ? lo + 64 : n;
            for (uint64_t i = lo; i < hi; ++i) {
                const int rc = f(i);
                if (rc != G16_OK) { status.store(rc); return; }
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto& th : pool) th.join();
    return status.load();
}

template <class F> struct FieldIo;

template <class P>
struct FieldIo<Fp<P>> {
    typedef Fp<P> F;
    static constexpr int BYTES = (P::BITS + 7) / 8;
    // canonical little-endian words -> bytes
    static void to_bytes(const F& x, bool big_endian, uint8_t* out) {
        uint32_t w[F::N];
        x.to_canonical(w);
        for (int i = 0; i < BYTES; ++i) {
            const uint8_t b = (uint8_t)(w[i >> 2] >> (8 * (i & 3)));
            out[big_endian ? BYTES - 1 - i : i] = b;
        }
    }
    // false if the integer is >= p
    static bool from_bytes(const uint8_t* in, bool big_endian, F* out) {
        uint32_t w[F::N] = {0};
        for (int i = 0; i < BYTES; ++i) w[i >> 2] |= (uint32_t)in[big_endian ? BYTES - 1 - i : i] << (8 * (i & 3));
        for (int i = F::N - 1; i >= 0; --i) {
            if (w[i] < P::mod(i)) break;
            if (w[i] > P::mod(i) || i == 0) return false;
        }
        *out = F::from_canonical(w);
        return true;
    }
    // -1, 0, 1 comparing canonical integers
    static int cmp(const F& a, const F& b) {
        uint32_t x[F::N], y[F::N];
        a.to_canonical(x);
        b.to_canonical(y);
        for (int i = F::N - 1; i >= 0; --i)
            if (x[i] != y[i]) return x[i] < y[i] ? -1 : 1;
        return 0;
    }
    static bool is_larger_than_neg(const F& y) { return cmp(y, y.neg()) > 0; }
    // p = 3 mod 4 for both base fields: sqrt(a) = a^((p+1)/4) when it exists
    static bool sqrt(const F& a, F* out) {
        uint32_t e[F::N];
        uint64_t carry = 1;   // (p + 1) >> 2
        for (int i = 0; i < F::N; ++i) { carry += P::mod(i); e[i] = (uint32_t)carry; carry >>= 32; }
        for (int i = 0; i < F::N; ++i) e[i] = (e[i] >> 2) | (i + 1 < F::N ? e[i + 1] << 30 : (uint32_t)carry << 30);
        const F r = a.pow(e, F::N);
        if (!(r.sqr() == a)) return false;
        *out = r;
        return true;
    }
};

template <class P>
struct FieldIo<Fp2<P>> {
    typedef Fp2<P> F;
    typedef Fp<P> B;
    typedef FieldIo<B> Bio;
    static constexpr int BYTES = 2 * Bio::BYTES;
    static void to_bytes(const F& x, bool big_endian, uint8_t* out) {   // big-endian order is c1 | c0, little-endian c0 | c1
        Bio::to_bytes(big_endian ? x.c1 : x.c0, big_endian, out);
        Bio::to_bytes(big_endian ? x.c0 : x.c1, big_endian, out + Bio::BYTES);
    }
    static bool from_bytes(const uint8_t* in, bool big_endian, F* out) {
        return Bio::from_bytes(in, big_endian, big_endian ? &out->c1 : &out->c0) &&
               Bio::from_bytes(in + Bio::BYTES, big_endian, big_endian ? &out->c0 : &out->c1);
    }
    static bool is_larger_than_neg(const F& y) {
        const F n = y.neg();
        const int c = Bio::cmp(y.c1, n.c1);
        return c ? c > 0 : Bio::cmp(y.c0, n.c0) > 0;
    }
    // u^2 = -1:  (x0 + x1 u)^2 = a0 + a1 u  with  x0^2 = (a0 +- sqrt(a0^2 + a1^2)) / 2,  x1 = a1 / (2 x0)
    static bool sqrt(const F& a, F* out) {
        if (a.c1.is_zero()) {
            B r;
            if (Bio::sqrt(a.c0, &r)) { *out = {r, B::zero()}; return true; }
            if (Bio::sqrt(a.c0.neg(), &r)) { *out = {B::zero(), r}; return true; }
            return false;
        }
        B s;
        if (!Bio::sqrt(a.c0.sqr() + a.c1.sqr(), &s)) return false;
        const B half = B::from_u64(2).inverse();
        B x0;
        if (!Bio::sqrt((a.c0 + s) * half, &x0) && !Bio::sqrt((a.c0 - s) * half, &x0)) return false;
        if (x0.is_zero()) return false;
        const B x1 = a.c1 * x0.dbl().inverse();
        const F r = {x0, x1};
        if (!(r.sqr() == a)) return false;
        *out = r;
        return true;
    }
};

template <class C, class F>
struct PointIo {
    typedef FieldIo<F> Fio;
    typedef Affine<F> A;
    static constexpr bool ZCASH = C::CURVE_ID == G16_BLS12_381;   // big-endian, flags in the first byte
    static F coeff_b() {
        if constexpr (sizeof(F) == sizeof(typename C::Fq)) return C::b1();
        else return C::b2();
    }
    static size_t size(bool compressed) { return compressed ? Fio::BYTES : 2 * Fio::BYTES; }

    static void write(const A& p, bool compressed, uint8_t* out) {
        const bool inf = p.is_identity();
        const bool larger = !inf && Fio::is_larger_than_neg(p.y);
        Fio::to_bytes(p.x, ZCASH, out);
        if (!compressed) Fio::to_bytes(p.y, ZCASH, out + Fio::BYTES);
        if (ZCASH) {
            out[0] |= (compressed ? 0x80 : 0) | (inf ? 0x40 : 0) | (compressed && larger ? 0x20 : 0);
        } else {
            out[size(compressed) - 1] |= (larger ? 0x80 : 0) | (inf ? 0x40 : 0);
        }
    }

    static bool on_curve(const A& p) { return p.y.sqr() == p.x.sqr() * p.x + coeff_b(); }
    static bool in_subgroup(const A& p) {
        uint32_t r[C::Fr::N];
        for (int i = 0; i < C::Fr::N; ++i) r[i] = C::Fr::Params::mod(i);
        return XYZZ<F>::from_affine(p).mul_bits(r, C::Fr::Params::BITS).is_identity();
    }

    // validate: 0 = Validate::No, 1 = on-curve, 2 = on-curve and in the prime-order subgroup (Validate::Yes)
    static int read(const uint8_t* in, bool compressed, int validate, A* out) {
        const size_t sz = size(compressed);
        uint8_t buf[4 * 48];
        memcpy(buf, in, sz);
        bool inf, larger;
        if (ZCASH) {
            const uint8_t fl = buf[0];
            if (((fl & 0x80) != 0) != compressed) return G16_ERR_INVALID_DATA;
            inf = (fl & 0x40) != 0;
            larger = (fl & 0x20) != 0;
            if (!compressed && larger) return G16_ERR_INVALID_DATA;
            buf[0] &= 0x1f;
        } else {
            const uint8_t fl = buf[sz - 1];
            inf = (fl & 0x40) != 0;
            larger = (fl & 0x80) != 0;
            if (inf && larger) return G16_ERR_INVALID_DATA;
            buf[sz - 1] &= 0x3f;
        }
        A p;
        if (!Fio::from_bytes(buf, ZCASH, &p.x)) return G16_ERR_INVALID_DATA;
        if (inf) {
            bool zero = p.x.is_zero();
            if (!compressed) {
                if (!Fio::from_bytes(buf + Fio::BYTES, ZCASH, &p.y)) return G16_ERR_INVALID_DATA;
                zero = zero && p.y.is_zero();
            }
            if (!zero || (ZCASH && larger)) return G16_ERR_INVALID_DATA;
            *out = A::identity();
            return G16_OK;
        }
        if (compressed) {
            F y;
            if (!Fio::sqrt(p.x.sqr() * p.x + coeff_b(), &y)) return G16_ERR_INVALID_DATA;   // x is not on the curve
            if (Fio::is_larger_than_neg(y) != larger) y = y.neg();
            p.y = y;
        } else {
            if (!Fio::from_bytes(buf + Fio::BYTES, ZCASH, &p.y)) return G16_ERR_INVALID_DATA;   // (a BN254 sign flag is not re-checked)
            if (validate >= 1 && !on_curve(p)) return G16_ERR_INVALID_DATA;
        }
        if (validate >= 2 && !in_subgroup(p)) return G16_ERR_INVALID_DATA;
        *out = p;
        return G16_OK;
    }
};

template <class C, class F>
int serialize_t(int compressed, const uint64_t* points, uint64_t n, uint8_t* out) {
    typedef PointIo<C, F> Io;
    const Affine<F>* p = reinterpret_cast<const Affine<F>*>(points);
    const size_t sz = Io::size(compressed != 0);
    return parallel_points(n, [&](uint64_t i) -> int { Io::write(p[i], compressed != 0, out + i * sz); return G16_OK; });
}

template <class C, class F>
int deserialize_t(int compressed, const uint8_t* in, uint64_t n, int validate, uint64_t* points_out) {
    typedef PointIo<C, F> Io;
    Affine<F>* p = reinterpret_cast<Affine<F>*>(points_out);
    const size_t sz = Io::size(compressed != 0);
    return parallel_points(n, [&](uint64_t i) -> int { return Io::read(in + i * sz, compressed != 0, validate, &p[i]); });
}

}  // namespace

int serialize_points(int curve, int g2, int compressed, const uint64_t* points, uint64_t n, uint8_t* out) {
    if (curve == G16_BLS12_381) return g2 ? serialize_t<Bls12_381, Bls12_381::Fq2>(compressed, points, n, out)
                                          : serialize_t<Bls12_381, Bls12_381::Fq>(compressed, points, n, out);
    if (curve == G16_BN254) return g2 ? serialize_t<Bn254, Bn254::Fq2>(compressed, points, n, out)
                                      : serialize_t<Bn254, Bn254::Fq>(compressed, points, n, out);
    return G16_ERR_BAD_ARG;
}

int deserialize_points(int curve, int g2, int compressed, const uint8_t* in, uint64_t n, int validate, uint64_t* points_out) {
    if (curve == G16_BLS12_381) return g2 ? deserialize_t<Bls12_381, Bls12_381::Fq2>(compressed, in, n, validate, points_out)
                                          : deserialize_t<Bls12_381, Bls12_381::Fq>(compressed, in, n, validate, points_out);
    if (curve == G16_BN254) return g2 ? deserialize_t<Bn254, Bn254::Fq2>(compressed, in, n, validate, points_out)
                                      : deserialize_t<Bn254, Bn254::Fq>(compressed, in, n, validate, points_out);
    return G16_ERR_BAD_ARG;
}

uint64_t serialized_point_size(int curve, int g2, int compressed) {
    const uint64_t fq = curve == G16_BLS12_381 ? 48 : 32;
    return fq * (g2 ? 2 : 1) * (compressed ? 1 : 2);
}

}  // namespace g16
