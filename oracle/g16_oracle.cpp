// oracle/g16_oracle.cpp -- CPU restatement of the ark-groth16 prover hot path.
//
// TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load this library; the product (groth16_amd/) never links it.
//
// PARITY STATUS: "parity unpinned".  The reference (/root/reference, ark-groth16 0.5.0)
// holds no golden vectors / KATs for this path and cannot be built here (no Rust, the
// arithmetic lives in un-vendored crates ark-ff/ark-ec/ark-poly/ark-relations 0.5.0,
// Cargo.toml:18-25).  This restatement is pinned against oracle/pymodel.py (independent
// big-int model), the known-trapdoor check and structural KATs -- see tests/.
//
// What is restated, and from where (file:line relative to /root/reference):
//   evaluate_constraint                 src/r1cs_to_qap.rs:28-67
//   witness_map_from_matrices           src/r1cs_to_qap.rs:172-235
//   instance_map_with_evaluation        src/r1cs_to_qap.rs:128-170
//   h_query_scalars                     src/r1cs_to_qap.rs:237-247
//   create_proof_with_assignment        src/prover.rs:54-132
//   calculate_coeff                     src/prover.rs:252-270
//   create_proof_with_reduction_and_matrices  src/prover.rs:26-51
//   generate_parameters_with_qap        src/generator.rs:47-208
// External algorithms restated from their published definitions (crate @ 0.5.0):
//   ark-ff  MontBackend (R = 2^(64N), LE u64 limbs)
//   ark-ec  short-Weierstrass Jacobian add/double/mixed-add; VariableBaseMSM::msm_bigint
//           (signed-digit windows, c = 3 if n < 32 else ln_without_floats(n) + 2,
//           windows in parallel, running-sum bucket reduction)
//   ark-poly Radix2EvaluationDomain (natural order in/out, ifft scales by 1/n, coset
//           fft multiplies coefficient k by g^k first, coset ifft by g^-k last)
//
// Implementation is deliberately independent of the product code: 64-bit limbs with
// unsigned __int128, Jacobian coordinates, OpenMP threads.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <omp.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

// ------------------------------------------------------------------------------------
// small multi-limb helpers (non-Montgomery, used once at init)
// ------------------------------------------------------------------------------------
template <int N>
static bool geq(const u64* a, const u64* b) {
    for (int i = N - 1; i >= 0; --i) {
        if (a[i] != b[i]) return a[i] > b[i];
    }
    return true;
}
template <int N>
static u64 sub_n(u64* r, const u64* a, const u64* b) {
    u64 borrow = 0;
    for (int i = 0; i < N; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}
template <int N>
static u64 add_n(u64* r, const u64* a, const u64* b) {
    u64 carry = 0;
    for (int i = 0; i < N; ++i) {
        u128 s = (u128)a[i] + b[i] + carry;
        r[i] = (u64)s;
        carry = (u64)(s >> 64);
    }
    return carry;
}

template <int N>
struct FieldCtx {
    u64 mod[N];
    u64 inv;    // -mod^-1 mod 2^64
    u64 r1[N];  // R mod p
    u64 r2[N];  // R^2 mod p
    int bits;
    void init(const u64* m) {
        memcpy(mod, m, sizeof(mod));
        u64 x = 1;  // Newton: x = m^-1 mod 2^64
        for (int i = 0; i < 6; ++i) x *= 2 - m[0] * x;
        inv = (u64)0 - x;
        bits = 0;
        for (int i = N - 1; i >= 0 && !bits; --i)
            if (m[i]) bits = 64 * i + (64 - __builtin_clzll(m[i]));
        // r1 = 2^(64N) mod p by doubling 1, 64N times; r2 = doubling r1 another 64N times
        u64 t[N] = {1};
        for (int rep = 0; rep < 2; ++rep) {
            for (int k = 0; k < 64 * N; ++k) {
                u64 c = add_n<N>(t, t, t);
                if (c || geq<N>(t, mod)) sub_n<N>(t, t, mod);
            }
            memcpy(rep == 0 ? r1 : r2, t, sizeof(t));
        }
    }
};

// ------------------------------------------------------------------------------------
// Montgomery prime field, 64-bit limbs (ark-ff MontBackend semantics)
// ------------------------------------------------------------------------------------
template <int N_, int ID>
struct Fp {
    static constexpr int N = N_;
    static FieldCtx<N_> C;
    u64 v[N_];

    static Fp zero() { Fp r; memset(r.v, 0, sizeof(r.v)); return r; }
    static Fp one() { Fp r; memcpy(r.v, C.r1, sizeof(r.v)); return r; }
    bool is_zero() const { u64 a = 0; for (int i = 0; i < N; ++i) a |= v[i]; return a == 0; }
    bool operator==(const Fp& o) const { return memcmp(v, o.v, sizeof(v)) == 0; }
    bool operator!=(const Fp& o) const { return !(*this == o); }
    bool is_one() const { return memcmp(v, C.r1, sizeof(v)) == 0; }

    Fp operator+(const Fp& o) const {
        Fp r;
        u64 c = add_n<N>(r.v, v, o.v);
        if (c || geq<N>(r.v, C.mod)) sub_n<N>(r.v, r.v, C.mod);
        return r;
    }
    Fp operator-(const Fp& o) const {
        Fp r;
        if (sub_n<N>(r.v, v, o.v)) add_n<N>(r.v, r.v, C.mod);
        return r;
    }
    Fp neg() const { return is_zero() ? *this : (zero() - *this); }
    Fp dbl() const { return *this + *this; }

    Fp operator*(const Fp& o) const {
        // CIOS Montgomery multiplication
        u64 t[N + 2];
        memset(t, 0, sizeof(t));
        for (int i = 0; i < N; ++i) {
            u64 carry = 0;
            for (int j = 0; j < N; ++j) {
                u128 x = (u128)v[j] * o.v[i] + t[j] + carry;
                t[j] = (u64)x;
                carry = (u64)(x >> 64);
            }
            u128 s = (u128)t[N] + carry;
            t[N] = (u64)s;
            t[N + 1] = (u64)(s >> 64);
            u64 m = t[0] * C.inv;
            u128 x = (u128)m * C.mod[0] + t[0];
            carry = (u64)(x >> 64);
            for (int j = 1; j < N; ++j) {
                x = (u128)m * C.mod[j] + t[j] + carry;
                t[j - 1] = (u64)x;
                carry = (u64)(x >> 64);
            }
            s = (u128)t[N] + carry;
            t[N - 1] = (u64)s;
            t[N] = t[N + 1] + (u64)(s >> 64);
        }
        Fp r;
        if (t[N] || geq<N>(t, C.mod)) sub_n<N>(r.v, t, C.mod);
        else memcpy(r.v, t, sizeof(r.v));
        return r;
    }
    Fp sqr() const { return *this * *this; }

    // canonical integer (into_bigint): one Montgomery reduction
    void to_bigint(u64* out) const {
        Fp o; memset(o.v, 0, sizeof(o.v)); o.v[0] = 1;
        Fp r = *this * o;
        memcpy(out, r.v, sizeof(r.v));
    }
    static Fp from_bigint(const u64* in) {
        Fp a, r2; memcpy(a.v, in, sizeof(a.v)); memcpy(r2.v, C.r2, sizeof(r2.v));
        return a * r2;
    }
    static Fp from_u64(u64 k) { u64 t[N] = {k}; return from_bigint(t); }

    Fp pow(const u64* e, int nlimbs) const {
        Fp acc = one();
        bool started = false;
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            if (started) acc = acc.sqr();
            if ((e[i / 64] >> (i % 64)) & 1) { acc = acc * *this; started = true; }
        }
        return acc;
    }
    Fp pow_u64(u64 e) const { return pow(&e, 1); }
    Fp inverse() const {  // Fermat; caller guarantees non-zero
        u64 e[N]; u64 two[N] = {2};
        sub_n<N>(e, C.mod, two);
        return pow(e, N);
    }
};
template <int N_, int ID> FieldCtx<N_> Fp<N_, ID>::C;

// ------------------------------------------------------------------------------------
// Fq2 = Fq[u]/(u^2 + 1)   (both BLS12-381 and BN254 use non-residue -1)
// ------------------------------------------------------------------------------------
template <class F>
struct Fp2 {
    F c0, c1;
    static Fp2 zero() { return {F::zero(), F::zero()}; }
    static Fp2 one() { return {F::one(), F::zero()}; }
    bool is_zero() const { return c0.is_zero() && c1.is_zero(); }
    bool operator==(const Fp2& o) const { return c0 == o.c0 && c1 == o.c1; }
    bool operator!=(const Fp2& o) const { return !(*this == o); }
    Fp2 operator+(const Fp2& o) const { return {c0 + o.c0, c1 + o.c1}; }
    Fp2 operator-(const Fp2& o) const { return {c0 - o.c0, c1 - o.c1}; }
    Fp2 neg() const { return {c0.neg(), c1.neg()}; }
    Fp2 dbl() const { return {c0.dbl(), c1.dbl()}; }
    Fp2 operator*(const Fp2& o) const {
        F a = c0 * o.c0, b = c1 * o.c1;
        F c = (c0 + c1) * (o.c0 + o.c1);
        return {a - b, c - a - b};
    }
    Fp2 sqr() const {
        F a = (c0 + c1) * (c0 - c1);
        F b = (c0 * c1).dbl();
        return {a, b};
    }
    Fp2 inverse() const {
        F n = (c0.sqr() + c1.sqr()).inverse();
        return {c0 * n, (c1 * n).neg()};
    }
};

// ------------------------------------------------------------------------------------
// short Weierstrass a = 0: affine + Jacobian (ark-ec models::short_weierstrass)
// ------------------------------------------------------------------------------------
template <class F>
struct Aff {
    F x, y;
    bool inf;
    static Aff identity() { return {F::zero(), F::zero(), true}; }
    Aff neg() const { return inf ? *this : Aff{x, y.neg(), false}; }
};

template <class F>
struct Jac {
    F X, Y, Z;
    static Jac identity() { return {F::one(), F::one(), F::zero()}; }
    static Jac from_affine(const Aff<F>& p) { return p.inf ? identity() : Jac{p.x, p.y, F::one()}; }
    bool is_identity() const { return Z.is_zero(); }

    Jac dbl() const {  // dbl-2009-l
        if (is_identity() || Y.is_zero()) return identity();
        F A = X.sqr(), B = Y.sqr(), Cc = B.sqr();
        F D = ((X + B).sqr() - A - Cc).dbl();
        F E = A.dbl() + A;
        F Fv = E.sqr();
        F X3 = Fv - D.dbl();
        F Y3 = E * (D - X3) - Cc.dbl().dbl().dbl();
        F Z3 = (Y * Z).dbl();
        return {X3, Y3, Z3};
    }
    Jac add(const Jac& o) const {  // add-2007-bl
        if (is_identity()) return o;
        if (o.is_identity()) return *this;
        F Z1Z1 = Z.sqr(), Z2Z2 = o.Z.sqr();
        F U1 = X * Z2Z2, U2 = o.X * Z1Z1;
        F S1 = Y * o.Z * Z2Z2, S2 = o.Y * Z * Z1Z1;
        if (U1 == U2) return (S1 == S2) ? dbl() : identity();
        F H = U2 - U1, R = S2 - S1;
        F HH = H.sqr(), HHH = H * HH, V = U1 * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - S1 * HHH;
        F Z3 = Z * o.Z * H;
        return {X3, Y3, Z3};
    }
    Jac add_affine(const Aff<F>& p) const {  // madd-2007-bl
        if (p.inf) return *this;
        if (is_identity()) return from_affine(p);
        F Z1Z1 = Z.sqr();
        F U2 = p.x * Z1Z1, S2 = p.y * Z * Z1Z1;
        if (X == U2) return (Y == S2) ? dbl() : identity();
        F H = U2 - X, R = S2 - Y;
        F HH = H.sqr(), HHH = H * HH, V = X * HH;
        F X3 = R.sqr() - HHH - V.dbl();
        F Y3 = R * (V - X3) - Y * HHH;
        F Z3 = Z * H;
        return {X3, Y3, Z3};
    }
    Jac neg() const { return {X, Y.neg(), Z}; }
    Jac mul_bigint(const u64* k, int nlimbs) const {  // double-and-add, MSB first
        Jac acc = identity();
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            acc = acc.dbl();
            if ((k[i / 64] >> (i % 64)) & 1) acc = acc.add(*this);
        }
        return acc;
    }
    Aff<F> to_affine() const {
        if (is_identity()) return Aff<F>::identity();
        F zi = Z.inverse(), zi2 = zi.sqr();
        return {X * zi2, Y * zi2 * zi, false};
    }
};

// ------------------------------------------------------------------------------------
// VariableBaseMSM::msm_bigint (ark-ec 0.5.0), called at src/prover.rs:66,74,262
// ------------------------------------------------------------------------------------
static inline int ln_without_floats(size_t a) {
    // ark-ec: log2(a) * 69 / 100, log2 = ceil(log2)
    int lg = 0;
    while (((size_t)1 << lg) < a) ++lg;
    return lg * 69 / 100;
}

// signed radix-2^c digits of a canonical scalar (ark-ec make_digits)
static void make_digits(const u64* k, int nlimbs, int c, int num_bits, std::vector<int64_t>& out, size_t off) {
    const u64 radix = (u64)1 << c, window_mask = radix - 1;
    int digits_count = (num_bits + c - 1) / c;
    u64 carry = 0;
    for (int i = 0; i < digits_count; ++i) {
        int bit_offset = i * c, u64_idx = bit_offset / 64, bit_idx = bit_offset % 64;
        u64 bit_buf;
        if (bit_idx < 64 - c || u64_idx == nlimbs - 1) bit_buf = k[u64_idx] >> bit_idx;
        else bit_buf = (k[u64_idx] >> bit_idx) | (k[u64_idx + 1] << (64 - bit_idx));
        u64 coef = carry + (bit_buf & window_mask);
        carry = (coef + radix / 2) >> c;
        int64_t d = (int64_t)coef - (int64_t)(carry << c);
        out[off + i] = d;
    }
    out[off + digits_count - 1] += (int64_t)(carry << c);
}

template <class F, int FRN>
static Jac<F> msm_bigint(const Aff<F>* bases, const u64* bigints /* FRN limbs each */, size_t size, int num_bits) {
    if (size == 0) return Jac<F>::identity();
    int c = size < 32 ? 3 : ln_without_floats(size) + 2;
    int digits_count = (num_bits + c - 1) / c;
    std::vector<int64_t> digits(size * (size_t)digits_count);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < size; ++i) make_digits(bigints + i * FRN, FRN, c, num_bits, digits, i * digits_count);
    std::vector<Jac<F>> window_sums(digits_count);
#pragma omp parallel for schedule(dynamic, 1)
    for (int w = 0; w < digits_count; ++w) {
        std::vector<Jac<F>> buckets((size_t)1 << c, Jac<F>::identity());
        for (size_t i = 0; i < size; ++i) {
            int64_t s = digits[i * digits_count + w];
            if (s > 0) buckets[s - 1] = buckets[s - 1].add_affine(bases[i]);
            else if (s < 0) buckets[-s - 1] = buckets[-s - 1].add_affine(bases[i].neg());
        }
        Jac<F> running = Jac<F>::identity(), res = Jac<F>::identity();
        for (size_t b = buckets.size(); b-- > 0;) {
            running = running.add(buckets[b]);
            res = res.add(running);
        }
        window_sums[w] = res;
    }
    Jac<F> total = Jac<F>::identity();
    for (int w = digits_count - 1; w >= 1; --w) {
        total = total.add(window_sums[w]);
        for (int k = 0; k < c; ++k) total = total.dbl();
    }
    return window_sums[0].add(total);
}

// ------------------------------------------------------------------------------------
// Radix-2 evaluation domain (ark-poly 0.5.0), used at src/r1cs_to_qap.rs:178-232
// ------------------------------------------------------------------------------------
template <class Fr>
struct Domain {
    size_t n;
    int log_n;
    Fr omega, omega_inv, n_inv;
    bool ok;
    Domain(size_t num_coeffs, const Fr& two_adic_root, int two_adicity) {
        n = 1; log_n = 0;
        while (n < num_coeffs) { n <<= 1; ++log_n; }
        ok = log_n <= two_adicity;
        if (!ok) return;
        omega = two_adic_root;
        for (int i = log_n; i < two_adicity; ++i) omega = omega.sqr();
        omega_inv = omega.inverse();
        n_inv = Fr::from_u64((u64)n).inverse();
    }
    // in-place, natural order in and out: bit-reverse then DIT butterflies
    void ntt(Fr* a, const Fr& w) const {
        for (size_t i = 0; i < n; ++i) {
            size_t j = 0;
            for (int b = 0; b < log_n; ++b) j |= ((i >> b) & 1) << (log_n - 1 - b);
            if (i < j) std::swap(a[i], a[j]);
        }
        std::vector<Fr> tw(n / 2 ? n / 2 : 1);
        tw[0] = Fr::one();
        for (size_t i = 1; i < n / 2; ++i) tw[i] = tw[i - 1] * w;
        for (int s = 0; s < log_n; ++s) {
            size_t half = (size_t)1 << s, step = n >> (s + 1);
#pragma omp parallel for schedule(static) if (n >= 4096)
            for (size_t k = 0; k < n / 2; ++k) {
                size_t grp = k >> s, j = k & (half - 1);
                size_t i0 = (grp << (s + 1)) + j, i1 = i0 + half;
                Fr t = a[i1] * tw[j * step];
                Fr u = a[i0];
                a[i0] = u + t;
                a[i1] = u - t;
            }
        }
    }
    void fft(Fr* a) const { ntt(a, omega); }
    void ifft(Fr* a) const {
        ntt(a, omega_inv);
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t i = 0; i < n; ++i) a[i] = a[i] * n_inv;
    }
    static void distribute_powers(Fr* a, size_t n, const Fr& g) {
        // a[k] *= g^k ; chunked so that it parallelises
        const size_t CH = 1024;
        size_t nch = (n + CH - 1) / CH;
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t ch = 0; ch < nch; ++ch) {
            u64 e = ch * CH;
            Fr p = g.pow_u64(e);
            for (size_t i = ch * CH; i < std::min(n, (ch + 1) * CH); ++i) { a[i] = a[i] * p; p = p * g; }
        }
    }
    void coset_fft(Fr* a, const Fr& g) const { distribute_powers(a, n, g); fft(a); }
    void coset_ifft(Fr* a, const Fr& g) const { ifft(a); distribute_powers(a, n, g.inverse()); }
    Fr vanishing(const Fr& tau) const { return tau.pow_u64((u64)n) - Fr::one(); }
};

// ------------------------------------------------------------------------------------
// curve bundles
// ------------------------------------------------------------------------------------
struct SplitMix64 {
    u64 s;
    explicit SplitMix64(u64 seed) : s(seed) {}
    u64 next() {
        s += 0x9E3779B97F4A7C15ULL;
        u64 z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
};

template <int FQN, int ID>
struct CurveT {
    typedef Fp<4, 2 * ID> Fr;
    typedef Fp<FQN, 2 * ID + 1> Fq;
    typedef Fp2<Fq> Fq2;
    typedef Aff<Fq> G1A;
    typedef Jac<Fq> G1J;
    typedef Aff<Fq2> G2A;
    typedef Jac<Fq2> G2J;
    static constexpr int FQ_LIMBS = FQN;
    static Fr fr_generator, two_adic_root;
    static int two_adicity;
    static G1A g1;
    static G2A g2;
    static Fq b1;
    static Fq2 b2;

    // 512 random bits reduced mod r, identical to pymodel.SplitMix64.field
    static Fr rand_fr(SplitMix64& rng) {
        // v = sum limb_k * 2^(64*(7-k)); reduce by Horner with 2^64 multiplications in Fr
        Fr acc = Fr::zero();
        Fr two64 = Fr::from_u64(1ULL << 32).sqr();
        for (int k = 0; k < 8; ++k) {
            u64 limb = rng.next();
            acc = acc * two64 + Fr::from_u64(limb);
        }
        return acc;
    }
};
template <int FQN, int ID> typename CurveT<FQN, ID>::Fr CurveT<FQN, ID>::fr_generator;
template <int FQN, int ID> typename CurveT<FQN, ID>::Fr CurveT<FQN, ID>::two_adic_root;
template <int FQN, int ID> int CurveT<FQN, ID>::two_adicity;
template <int FQN, int ID> typename CurveT<FQN, ID>::G1A CurveT<FQN, ID>::g1;
template <int FQN, int ID> typename CurveT<FQN, ID>::G2A CurveT<FQN, ID>::g2;
template <int FQN, int ID> typename CurveT<FQN, ID>::Fq CurveT<FQN, ID>::b1;
template <int FQN, int ID> typename CurveT<FQN, ID>::Fq2 CurveT<FQN, ID>::b2;

typedef CurveT<6, 0> Bls;
typedef CurveT<4, 1> Bn;

static void hex_to_limbs(const char* hex, u64* out, int n) {
    memset(out, 0, sizeof(u64) * n);
    size_t len = strlen(hex);
    for (size_t i = 0; i < len; ++i) {
        char ch = hex[len - 1 - i];
        u64 d = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
        out[i / 16] |= d << (4 * (i % 16));
    }
}
template <class F>
static F fp_from_hex(const char* hex) {
    u64 t[F::N];
    hex_to_limbs(hex, t, F::N);
    return F::from_bigint(t);
}

static bool g_inited = false;
static void init_all() {
    if (g_inited) return;
    u64 t[6];
    // ---- BLS12-381
    hex_to_limbs("73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001", t, 4);
    Bls::Fr::C.init(t);
    hex_to_limbs("1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab", t, 6);
    Bls::Fq::C.init(t);
    Bls::fr_generator = Bls::Fr::from_u64(7);
    Bls::two_adicity = 32;
    {
        u64 e[4], one[4] = {1};
        sub_n<4>(e, Bls::Fr::C.mod, one);
        // (r-1) >> 32
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
        Bls::two_adic_root = Bls::fr_generator.pow(e, 4);
    }
    Bls::b1 = Bls::Fq::from_u64(4);
    Bls::b2 = {Bls::Fq::from_u64(4), Bls::Fq::from_u64(4)};
    Bls::g1 = {fp_from_hex<Bls::Fq>("17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"),
               fp_from_hex<Bls::Fq>("08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1"), false};
    Bls::g2 = {{fp_from_hex<Bls::Fq>("024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"),
                fp_from_hex<Bls::Fq>("13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e")},
               {fp_from_hex<Bls::Fq>("0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801"),
                fp_from_hex<Bls::Fq>("0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be")},
               false};
    // ---- BN254
    hex_to_limbs("30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001", t, 4);
    Bn::Fr::C.init(t);
    hex_to_limbs("30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47", t, 4);
    Bn::Fq::C.init(t);
    Bn::fr_generator = Bn::Fr::from_u64(5);
    Bn::two_adicity = 28;
    {
        u64 e[4], one[4] = {1};
        sub_n<4>(e, Bn::Fr::C.mod, one);
        for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 28) | (i < 3 ? e[i + 1] << 36 : 0);
        Bn::two_adic_root = Bn::fr_generator.pow(e, 4);
    }
    Bn::b1 = Bn::Fq::from_u64(3);
    {   // b2 = 3 / (9 + u)
        Bn::Fq2 xi = {Bn::Fq::from_u64(9), Bn::Fq::from_u64(1)};
        Bn::Fq2 three = {Bn::Fq::from_u64(3), Bn::Fq::zero()};
        Bn::b2 = three * xi.inverse();
    }
    Bn::g1 = {Bn::Fq::from_u64(1), Bn::Fq::from_u64(2), false};
    Bn::g2 = {{fp_from_hex<Bn::Fq>("1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed"),
               fp_from_hex<Bn::Fq>("198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2")},
              {fp_from_hex<Bn::Fq>("12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa"),
               fp_from_hex<Bn::Fq>("090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b")},
              false};
    g_inited = true;
}

// ------------------------------------------------------------------------------------
// flat views used across the C boundary
//   Fr / Fq element  : N u64 limbs, Montgomery form (arkworks in-memory form)
//   G1 affine        : x | y              (2*FQN limbs); identity = flag byte (or x=y=0)
//   G2 affine        : x.c0 x.c1 y.c0 y.c1 (4*FQN limbs)
//   CSR matrix       : row_ptr u64[nc+1], col u32[nnz], val Fr[nnz]
// ------------------------------------------------------------------------------------
struct CsrView {
    const u64* row_ptr;
    const uint32_t* col;
    const u64* val;
};

template <class C>
static typename C::G1A load_g1(const u64* p) {
    typename C::G1A a;
    memcpy(a.x.v, p, sizeof(a.x.v));
    memcpy(a.y.v, p + C::FQ_LIMBS, sizeof(a.y.v));
    a.inf = a.x.is_zero() && a.y.is_zero();
    return a;
}
template <class C>
static void store_g1(u64* p, const typename C::G1A& a) {
    if (a.inf) { memset(p, 0, sizeof(u64) * 2 * C::FQ_LIMBS); return; }
    memcpy(p, a.x.v, sizeof(a.x.v));
    memcpy(p + C::FQ_LIMBS, a.y.v, sizeof(a.y.v));
}
template <class C>
static typename C::G2A load_g2(const u64* p) {
    typename C::G2A a;
    const int L = C::FQ_LIMBS;
    memcpy(a.x.c0.v, p, 8 * L); memcpy(a.x.c1.v, p + L, 8 * L);
    memcpy(a.y.c0.v, p + 2 * L, 8 * L); memcpy(a.y.c1.v, p + 3 * L, 8 * L);
    a.inf = a.x.is_zero() && a.y.is_zero();
    return a;
}
template <class C>
static void store_g2(u64* p, const typename C::G2A& a) {
    const int L = C::FQ_LIMBS;
    if (a.inf) { memset(p, 0, 8 * 4 * L); return; }
    memcpy(p, a.x.c0.v, 8 * L); memcpy(p + L, a.x.c1.v, 8 * L);
    memcpy(p + 2 * L, a.y.c0.v, 8 * L); memcpy(p + 3 * L, a.y.c1.v, 8 * L);
}

// ------------------------------------------------------------------------------------
// evaluate_constraint  (src/r1cs_to_qap.rs:28-67)
// ------------------------------------------------------------------------------------
template <class Fr>
static Fr evaluate_constraint(const CsrView& m, size_t row, const Fr* z) {
    Fr sum = Fr::zero();
    for (u64 k = m.row_ptr[row]; k < m.row_ptr[row + 1]; ++k) {
        Fr coeff; memcpy(coeff.v, m.val + 4 * k, sizeof(coeff.v));
        const Fr& val = z[m.col[k]];
        sum = sum + (coeff.is_one() ? val : val * coeff);
    }
    return sum;
}

// ------------------------------------------------------------------------------------
// witness_map_from_matrices  (src/r1cs_to_qap.rs:172-235)
// returns 0 ok, 1 = PolynomialDegreeTooLarge
// ------------------------------------------------------------------------------------
template <class C>
static int witness_map(const CsrView abc[3], size_t num_inputs, size_t num_constraints, const typename C::Fr* z,
                       std::vector<typename C::Fr>& h, typename C::Fr* abc_out /* 3n or null */) {
    typedef typename C::Fr Fr;
    Domain<Fr> dom(num_constraints + num_inputs, C::two_adic_root, C::two_adicity);  // :178-179
    if (!dom.ok) return 1;
    size_t n = dom.n;
    std::vector<Fr> a(n, Fr::zero()), b(n, Fr::zero()), c(n, Fr::zero());
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < num_constraints; ++i) {  // :186-193, :213-218
        a[i] = evaluate_constraint<Fr>(abc[0], i, z);
        b[i] = evaluate_constraint<Fr>(abc[1], i, z);
        c[i] = evaluate_constraint<Fr>(abc[2], i, z);
    }
    for (size_t j = 0; j < num_inputs; ++j) a[num_constraints + j] = z[j];  // :195-199
    if (abc_out) {
        memcpy(abc_out, a.data(), n * sizeof(Fr));
        memcpy(abc_out + n, b.data(), n * sizeof(Fr));
        memcpy(abc_out + 2 * n, c.data(), n * sizeof(Fr));
    }
    const Fr g = C::fr_generator;
    dom.ifft(a.data()); dom.ifft(b.data());               // :201-202
    dom.coset_fft(a.data(), g); dom.coset_fft(b.data(), g);  // :206-207
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) a[i] = a[i] * b[i];    // :209
    dom.ifft(c.data()); dom.coset_fft(c.data(), g);       // :220-221
    Fr zinv = dom.vanishing(g).inverse();                 // :223-226
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) a[i] = (a[i] - c[i]) * zinv;  // :227-230
    dom.coset_ifft(a.data(), g);                          // :232
    h.swap(a);
    return 0;
}

// ------------------------------------------------------------------------------------
// proving key view (src/data_structures.rs:125-143 + the vk fields the prover reads)
// ------------------------------------------------------------------------------------
struct PkView {
    const u64 *alpha_g1, *beta_g1, *delta_g1;  // G1 affine
    const u64 *beta_g2, *delta_g2;             // G2 affine
    const u64* a_query;    u64 a_len;          // m+1
    const u64* b_g1_query; u64 b_g1_len;       // m+1
    const u64* b_g2_query; u64 b_g2_len;       // m+1
    const u64* h_query;    u64 h_len;          // n-1
    const u64* l_query;    u64 l_len;          // w
};

template <class C>
static std::vector<typename C::G1A> load_g1_vec(const u64* p, size_t n) {
    std::vector<typename C::G1A> v(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) v[i] = load_g1<C>(p + i * 2 * C::FQ_LIMBS);
    return v;
}
template <class C>
static std::vector<typename C::G2A> load_g2_vec(const u64* p, size_t n) {
    std::vector<typename C::G2A> v(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) v[i] = load_g2<C>(p + i * 4 * C::FQ_LIMBS);
    return v;
}
template <class Fr>
static std::vector<u64> to_bigints(const Fr* s, size_t n) {  // src/prover.rs:63-65
    std::vector<u64> out(n * 4);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) s[i].to_bigint(&out[4 * i]);
    return out;
}

template <class C>
struct ProofOut {
    typename C::G1A a;
    typename C::G2A b;
    typename C::G1A c;
};

struct PhaseTimes {  // seconds, named after the reference's timers (src/prover.rs:36,62,89,99,111,119)
    double witness_map, compute_c, compute_a, compute_b_g1, compute_b_g2, finish_c;
};

// calculate_coeff  (src/prover.rs:252-270)
template <class F, class A>
static Jac<F> calculate_coeff(const Jac<F>& initial, const std::vector<A>& query, const A& vk_param,
                              const std::vector<u64>& assignment, int num_bits) {
    size_t n = std::min(query.size() - 1, assignment.size() / 4);
    Jac<F> acc = msm_bigint<F, 4>(query.data() + 1, assignment.data(), n, num_bits);
    Jac<F> res = initial;
    res = res.add_affine(query[0]);
    res = res.add(acc);
    res = res.add_affine(vk_param);
    return res;
}

// create_proof_with_assignment  (src/prover.rs:54-132)
template <class C>
static void create_proof_with_assignment(const PkView& pk, const typename C::Fr& r, const typename C::Fr& s,
                                         const typename C::Fr* h, size_t h_len, const typename C::Fr* input_assignment,
                                         size_t n_input, const typename C::Fr* aux_assignment, size_t n_aux,
                                         ProofOut<C>& out, PhaseTimes* pt, u64* msm_parts /* optional */) {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    const int nb = Fr::C.bits;
    double t0 = omp_get_wtime();
    auto h_query = load_g1_vec<C>(pk.h_query, pk.h_len);
    auto l_query = load_g1_vec<C>(pk.l_query, pk.l_len);
    auto a_query = load_g1_vec<C>(pk.a_query, pk.a_len);
    auto b_g1_query = load_g1_vec<C>(pk.b_g1_query, pk.b_g1_len);
    auto b_g2_query = load_g2_vec<C>(pk.b_g2_query, pk.b_g2_len);
    auto alpha_g1 = load_g1<C>(pk.alpha_g1), beta_g1 = load_g1<C>(pk.beta_g1), delta_g1 = load_g1<C>(pk.delta_g1);
    auto beta_g2 = load_g2<C>(pk.beta_g2), delta_g2 = load_g2<C>(pk.delta_g2);
    double t_load = omp_get_wtime() - t0;
    (void)t_load;

    t0 = omp_get_wtime();
    std::vector<u64> h_assignment = to_bigints(h, h_len);                                        // :63-65
    Jac<Fq> h_acc = msm_bigint<Fq, 4>(h_query.data(), h_assignment.data(), std::min(h_query.size(), h_len), nb);  // :66
    std::vector<u64> aux_big = to_bigints(aux_assignment, n_aux);                               // :70-72
    Jac<Fq> l_aux_acc = msm_bigint<Fq, 4>(l_query.data(), aux_big.data(), std::min(l_query.size(), n_aux), nb);  // :74
    u64 rs_big[4]; (r * s).to_bigint(rs_big);
    Jac<Fq> r_s_delta_g1 = Jac<Fq>::from_affine(delta_g1).mul_bigint(rs_big, 4);                 // :76
    if (pt) pt->compute_c = omp_get_wtime() - t0;

    std::vector<u64> assignment = to_bigints(input_assignment, n_input);                        // :80-85
    assignment.insert(assignment.end(), aux_big.begin(), aux_big.end());
    u64 r_big[4], s_big[4];
    r.to_bigint(r_big); s.to_bigint(s_big);

    t0 = omp_get_wtime();
    Jac<Fq> r_g1 = Jac<Fq>::from_affine(delta_g1).mul_bigint(r_big, 4);                          // :90
    Jac<Fq> g_a = calculate_coeff<Fq>(r_g1, a_query, alpha_g1, assignment, nb);                  // :92
    Jac<Fq> s_g_a = g_a.mul_bigint(s_big, 4);                                                    // :94
    if (pt) pt->compute_a = omp_get_wtime() - t0;

    t0 = omp_get_wtime();
    Jac<Fq> g1_b = Jac<Fq>::identity();
    if (!r.is_zero()) {                                                                          // :98-108
        Jac<Fq> s_g1 = Jac<Fq>::from_affine(delta_g1).mul_bigint(s_big, 4);
        g1_b = calculate_coeff<Fq>(s_g1, b_g1_query, beta_g1, assignment, nb);
    }
    if (pt) pt->compute_b_g1 = omp_get_wtime() - t0;

    t0 = omp_get_wtime();
    Jac<Fq2> s_g2 = Jac<Fq2>::from_affine(delta_g2).mul_bigint(s_big, 4);                        // :112
    Jac<Fq2> g2_b = calculate_coeff<Fq2>(s_g2, b_g2_query, beta_g2, assignment, nb);             // :113
    Jac<Fq> r_g1_b = g1_b.mul_bigint(r_big, 4);                                                  // :114
    if (pt) pt->compute_b_g2 = omp_get_wtime() - t0;

    t0 = omp_get_wtime();
    Jac<Fq> g_c = s_g_a;                                                                         // :119-124
    g_c = g_c.add(r_g1_b);
    g_c = g_c.add(r_s_delta_g1.neg());
    g_c = g_c.add(l_aux_acc);
    g_c = g_c.add(h_acc);
    out.a = g_a.to_affine();                                                                     // :127-131
    out.b = g2_b.to_affine();
    out.c = g_c.to_affine();
    if (pt) pt->finish_c = omp_get_wtime() - t0;

    if (msm_parts) {  // the five raw MSM results, affine, for unit-level parity tests
        const int L = C::FQ_LIMBS;
        size_t na = std::min(a_query.size() - 1, assignment.size() / 4);
        store_g1<C>(msm_parts, h_acc.to_affine());
        store_g1<C>(msm_parts + 2 * L, l_aux_acc.to_affine());
        store_g1<C>(msm_parts + 4 * L, msm_bigint<Fq, 4>(a_query.data() + 1, assignment.data(), na, nb).to_affine());
        store_g1<C>(msm_parts + 6 * L, msm_bigint<Fq, 4>(b_g1_query.data() + 1, assignment.data(), na, nb).to_affine());
        store_g2<C>(msm_parts + 8 * L, msm_bigint<Fq2, 4>(b_g2_query.data() + 1, assignment.data(), na, nb).to_affine());
    }
}

// create_proof_with_reduction_and_matrices  (src/prover.rs:26-51)
template <class C>
static int prove(const PkView& pk, const CsrView abc[3], size_t num_inputs, size_t num_constraints,
                 const u64* full_assignment, size_t n_assign, const u64* r_, const u64* s_, u64* proof_out,
                 u64* h_out, u64* msm_parts, PhaseTimes* pt) {
    typedef typename C::Fr Fr;
    const Fr* z = reinterpret_cast<const Fr*>(full_assignment);
    Fr r, s;
    memcpy(r.v, r_, 32); memcpy(s.v, s_, 32);
    std::vector<Fr> h;
    double t0 = omp_get_wtime();
    int rc = witness_map<C>(abc, num_inputs, num_constraints, z, h, nullptr);  // :37-42
    if (pt) pt->witness_map = omp_get_wtime() - t0;
    if (rc) return rc;
    if (h_out) memcpy(h_out, h.data(), h.size() * sizeof(Fr));
    ProofOut<C> out;
    create_proof_with_assignment<C>(pk, r, s, h.data(), h.size(), z + 1, num_inputs - 1, z + num_inputs,
                                    n_assign - num_inputs, out, pt, msm_parts);  // :44-47
    const int L = C::FQ_LIMBS;
    store_g1<C>(proof_out, out.a);
    store_g2<C>(proof_out + 2 * L, out.b);
    store_g1<C>(proof_out + 6 * L, out.c);
    return 0;
}

// ------------------------------------------------------------------------------------
// fixed-base batch multiplication (restates the role of ark-ec BatchMulPreprocessing,
// src/generator.rs:129-183): 8-bit windows, table of 32 x 255 multiples
// ------------------------------------------------------------------------------------
template <class F>
struct FixedBase {
    static constexpr int W = 8, NW = 32;
    std::vector<Aff<F>> table;  // [NW][255]
    explicit FixedBase(const Jac<F>& g) {
        table.resize((size_t)NW * 255);
        Jac<F> base = g;
        for (int w = 0; w < NW; ++w) {
            Jac<F> acc = base;
            std::vector<Jac<F>> row(255);
            for (int d = 1; d <= 255; ++d) { row[d - 1] = acc; acc = acc.add(base); }
            for (int d = 0; d < 255; ++d) table[(size_t)w * 255 + d] = row[d].to_affine();
            base = acc;  // 256 * base
        }
    }
    Jac<F> mul(const u64* k) const {
        Jac<F> acc = Jac<F>::identity();
        for (int w = 0; w < NW; ++w) {
            unsigned d = (k[w / 8] >> (8 * (w % 8))) & 0xFF;
            if (d) acc = acc.add_affine(table[(size_t)w * 255 + d - 1]);
        }
        return acc;
    }
};

// ------------------------------------------------------------------------------------
// generate_parameters_with_qap with a known trapdoor  (src/generator.rs:47-208)
// ------------------------------------------------------------------------------------
template <class C>
static int setup(const CsrView abc[3], size_t num_inputs, size_t num_constraints, size_t num_vars, u64 seed,
                 u64* g1_out /* alpha,beta,delta,g1gen : 4 pts */, u64* g2_out /* beta,delta,gamma,g2gen : 4 pts */,
                 u64* a_query, u64* b_g1_query, u64* b_g2_query, u64* h_query, u64* l_query, u64* gamma_abc,
                 u64* trapdoor_out /* alpha beta gamma delta t zt : 6 Fr */, u64* abc_t_out /* 3*(m+1) Fr or null */) {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;
    SplitMix64 rng(seed ^ 0x5E7ULL);
    auto nz = [&]() { Fr x; do { x = C::rand_fr(rng); } while (x.is_zero()); return x; };
    Fr alpha = nz(), beta = nz(), gamma = nz(), delta = nz();
    Fr k1 = nz(), k2 = nz();
    u64 kb[4];
    k1.to_bigint(kb);
    Jac<Fq> g1 = Jac<Fq>::from_affine(C::g1).mul_bigint(kb, 4);   // random generators, generator.rs:31-32
    k2.to_bigint(kb);
    Jac<Fq2> g2 = Jac<Fq2>::from_affine(C::g2).mul_bigint(kb, 4);

    Domain<Fr> dom(num_constraints + num_inputs, C::two_adic_root, C::two_adicity);   // :88-89
    if (!dom.ok) return 1;
    size_t n = dom.n;
    Fr t, zt;
    do { t = C::rand_fr(rng); zt = dom.vanishing(t); } while (zt.is_zero());          // :90

    // instance_map_with_evaluation  (src/r1cs_to_qap.rs:128-170)
    std::vector<Fr> u(n);  // evaluate_all_lagrange_coefficients(t): u_i = zt/n * w^i / (t - w^i)
    {
        std::vector<Fr> wp(n), den(n);
        wp[0] = Fr::one();
        for (size_t i = 1; i < n; ++i) wp[i] = wp[i - 1] * dom.omega;
        // batch inversion of (t - w^i)
        std::vector<Fr> pre(n);
        Fr run = Fr::one();
        for (size_t i = 0; i < n; ++i) { den[i] = t - wp[i]; pre[i] = run; run = run * den[i]; }
        Fr inv = run.inverse();
        Fr zn = zt * dom.n_inv;
        for (size_t i = n; i-- > 0;) {
            Fr di = inv * pre[i];
            inv = inv * den[i];
            u[i] = zn * wp[i] * di;
        }
    }
    size_t m = num_vars - 1;  // qap_num_variables = (num_inputs - 1) + num_witness
    std::vector<Fr> a(m + 1, Fr::zero()), b(m + 1, Fr::zero()), c(m + 1, Fr::zero());
    for (size_t j = 0; j < num_inputs; ++j) a[j] = u[num_constraints + j];            // :150-155
    for (size_t i = 0; i < num_constraints; ++i) {                                    // :157-167
        for (int which = 0; which < 3; ++which) {
            std::vector<Fr>& dst = which == 0 ? a : which == 1 ? b : c;
            const CsrView& mtx = abc[which];
            for (u64 k = mtx.row_ptr[i]; k < mtx.row_ptr[i + 1]; ++k) {
                Fr coeff; memcpy(coeff.v, mtx.val + 4 * k, 32);
                dst[mtx.col[k]] = dst[mtx.col[k]] + u[i] * coeff;
            }
        }
    }
    Fr gamma_inv = gamma.inverse(), delta_inv = delta.inverse();
    std::vector<Fr> gabc(num_inputs), l(m + 1 - num_inputs);
    for (size_t i = 0; i < num_inputs; ++i) gabc[i] = (beta * a[i] + alpha * b[i] + c[i]) * gamma_inv;  // :113-117
    for (size_t i = num_inputs; i <= m; ++i) l[i - num_inputs] = (beta * a[i] + alpha * b[i] + c[i]) * delta_inv;  // :119-123
    std::vector<Fr> hs(n - 1);                                                          // r1cs_to_qap.rs:243-245
    {
        Fr base = zt * delta_inv, p = Fr::one();
        for (size_t i = 0; i + 1 < n; ++i) { hs[i] = base * p; p = p * t; }
    }
    FixedBase<Fq> t1(g1);
    FixedBase<Fq2> t2(g2);
    const int L = C::FQ_LIMBS;
    auto batch_g1 = [&](const std::vector<Fr>& sc, u64* out) {
#pragma omp parallel for schedule(dynamic, 64)
        for (size_t i = 0; i < sc.size(); ++i) {
            u64 kk[4]; sc[i].to_bigint(kk);
            store_g1<C>(out + i * 2 * L, t1.mul(kk).to_affine());
        }
    };
    auto batch_g2 = [&](const std::vector<Fr>& sc, u64* out) {
#pragma omp parallel for schedule(dynamic, 64)
        for (size_t i = 0; i < sc.size(); ++i) {
            u64 kk[4]; sc[i].to_bigint(kk);
            store_g2<C>(out + i * 4 * L, t2.mul(kk).to_affine());
        }
    };
    batch_g2(b, b_g2_query);   // :129-135
    batch_g1(a, a_query);      // :155
    batch_g1(b, b_g1_query);   // :161
    batch_g1(hs, h_query);     // :167-169
    batch_g1(l, l_query);      // :174
    batch_g1(gabc, gamma_abc); // :183
    std::vector<Fr> sc1 = {alpha, beta, delta, Fr::one()};
    batch_g1(sc1, g1_out);
    std::vector<Fr> sc2 = {beta, delta, gamma, Fr::one()};
    batch_g2(sc2, g2_out);
    Fr td[6] = {alpha, beta, gamma, delta, t, zt};
    memcpy(trapdoor_out, td, sizeof(td));
    if (abc_t_out) {
        memcpy(abc_t_out, a.data(), (m + 1) * 32);
        memcpy(abc_t_out + 4 * (m + 1), b.data(), (m + 1) * 32);
        memcpy(abc_t_out + 8 * (m + 1), c.data(), (m + 1) * 32);
    }
    return 0;
}

// Expected proof from the trapdoor, as scalar * generator: shares no MSM / NTT code
// with the prover (SURVEY.md 8(c) pin 2).  h enters through h(t) only.
template <class C>
static void trapdoor_proof(const u64* trapdoor, const u64* abc_t, size_t m, size_t num_inputs, size_t n,
                           const u64* g1gen, const u64* g2gen, const u64* z_, const u64* h_, const u64* r_,
                           const u64* s_, u64* proof_out) {
    typedef typename C::Fr Fr;
    const Fr* td = reinterpret_cast<const Fr*>(trapdoor);
    Fr alpha = td[0], beta = td[1], delta = td[3], t = td[4], zt = td[5];
    const Fr* a = reinterpret_cast<const Fr*>(abc_t);
    const Fr* b = a + (m + 1);
    const Fr* c = b + (m + 1);
    const Fr* z = reinterpret_cast<const Fr*>(z_);
    const Fr* h = reinterpret_cast<const Fr*>(h_);
    Fr r, s; memcpy(r.v, r_, 32); memcpy(s.v, s_, 32);
    Fr A = alpha + r * delta, B = beta + s * delta, lsum = Fr::zero();
    for (size_t i = 0; i <= m; ++i) {
        A = A + z[i] * a[i];
        B = B + z[i] * b[i];
        if (i >= num_inputs) lsum = lsum + z[i] * (beta * a[i] + alpha * b[i] + c[i]);
    }
    Fr ht = Fr::zero(), p = Fr::one();
    for (size_t i = 0; i + 1 < n; ++i) { ht = ht + h[i] * p; p = p * t; }
    Fr di = delta.inverse();
    Fr Cc = lsum * di + ht * zt * di + s * A + r * B - r * s * delta;
    u64 k[4];
    const int L = C::FQ_LIMBS;
    A.to_bigint(k);
    store_g1<C>(proof_out, Jac<typename C::Fq>::from_affine(load_g1<C>(g1gen)).mul_bigint(k, 4).to_affine());
    B.to_bigint(k);
    store_g2<C>(proof_out + 2 * L, Jac<typename C::Fq2>::from_affine(load_g2<C>(g2gen)).mul_bigint(k, 4).to_affine());
    Cc.to_bigint(k);
    store_g1<C>(proof_out + 6 * L, Jac<typename C::Fq>::from_affine(load_g1<C>(g1gen)).mul_bigint(k, 4).to_affine());
}

// ------------------------------------------------------------------------------------
// synthetic inputs
// ------------------------------------------------------------------------------------
// SYN(k, seed): Fibonacci product chain.  n_c = 2^k - 2, l = 2, w = n_c + 1.
// z = [1, x, u_0..u_{n_c}], outputs CSR for A, B, C with one unit entry per row.
template <class C>
static void syn_circuit(int k, u64 seed, u64* z_out, u64* row_ptr /* nc+1, shared */, uint32_t* colA, uint32_t* colB,
                        uint32_t* colC, u64* val /* nc Fr, all one, shared */) {
    typedef typename C::Fr Fr;
    SplitMix64 rng(seed);
    size_t nc = ((size_t)1 << k) - 2;
    std::vector<Fr> u(nc + 2);
    u[0] = C::rand_fr(rng); u[1] = C::rand_fr(rng);
    for (size_t i = 0; i < nc; ++i) u[i + 2] = u[i] * u[i + 1];
    Fr* z = reinterpret_cast<Fr*>(z_out);
    z[0] = Fr::one(); z[1] = u[nc + 1];
    for (size_t j = 0; j <= nc; ++j) z[2 + j] = u[j];
    auto col = [&](size_t j) -> uint32_t { return j == nc + 1 ? 1u : (uint32_t)(2 + j); };
    Fr one = Fr::one();
    for (size_t i = 0; i < nc; ++i) {
        row_ptr[i] = i;
        colA[i] = col(i); colB[i] = col(i + 1); colC[i] = col(i + 2);
        memcpy(val + 4 * i, one.v, 32);
    }
    row_ptr[nc] = nc;
}

// distinct non-identity bases P_i = (s0 + i) * G : chunked chains + batch normalisation
template <class F, class A>
static void synth_bases(const A& gen, u64 seed, size_t n, std::vector<A>& out) {
    out.resize(n);
    const size_t CH = 4096;
    size_t nch = (n + CH - 1) / CH;
    SplitMix64 rng(seed ^ 0xBA5E5ULL);
    u64 s0 = (rng.next() >> 8) | 1;
    Jac<F> G = Jac<F>::from_affine(gen);
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t ch = 0; ch < nch; ++ch) {
        size_t lo = ch * CH, hi = std::min(n, lo + CH);
        u64 k[1] = {s0 + lo};
        Jac<F> p = G.mul_bigint(k, 1);
        std::vector<Jac<F>> pts(hi - lo);
        for (size_t i = lo; i < hi; ++i) { pts[i - lo] = p; p = p.add_affine(gen); }
        // batch inversion of Z
        std::vector<F> pre(hi - lo);
        F run = F::one();
        for (size_t i = 0; i < pts.size(); ++i) { pre[i] = run; run = run * pts[i].Z; }
        F inv = run.inverse();
        for (size_t i = pts.size(); i-- > 0;) {
            F zi = inv * pre[i];
            inv = inv * pts[i].Z;
            F zi2 = zi.sqr();
            out[lo + i] = A{pts[i].X * zi2, pts[i].Y * zi2 * zi, false};
        }
    }
}

// ------------------------------------------------------------------------------------
// C interface (ctypes).  curve: 0 = BLS12-381, 1 = BN254.
// ------------------------------------------------------------------------------------
#define DISPATCH(curve, ...)                                \
    do {                                                    \
        init_all();                                         \
        if ((curve) == 0) { typedef Bls C; __VA_ARGS__; }   \
        else { typedef Bn C; __VA_ARGS__; }                 \
    } while (0)

extern "C" {

int orc_num_threads() { return omp_get_max_threads(); }
void orc_set_num_threads(int n) { omp_set_num_threads(n); }

// field ops for unit tests: which = 0 Fr, 1 Fq; op: 0 add 1 sub 2 mul 3 inverse(a) 4 to_bigint(a) 5 from_bigint(a)
int orc_field_op(int curve, int which, int op, const u64* a, const u64* b, u64* out) {
    auto run = [&](auto tag) {
        typedef decltype(tag) F;
        F x, y, r;
        memcpy(x.v, a, sizeof(x.v));
        if (b) memcpy(y.v, b, sizeof(y.v));
        switch (op) {
            case 0: r = x + y; break;
            case 1: r = x - y; break;
            case 2: r = x * y; break;
            case 3: r = x.inverse(); break;
            case 4: x.to_bigint(r.v); break;
            case 5: r = F::from_bigint(x.v); break;
            default: return 1;
        }
        memcpy(out, r.v, sizeof(r.v));
        return 0;
    };
    int rc = 0;
    DISPATCH(curve, { rc = which == 0 ? run(typename C::Fr()) : run(typename C::Fq()); });
    return rc;
}

// constants for tests: which 0 = fr two-adic root, 1 = fr generator, 2 = G1 generator (2*L), 3 = G2 generator (4*L)
int orc_constant(int curve, int which, u64* out) {
    DISPATCH(curve, {
        switch (which) {
            case 0: memcpy(out, C::two_adic_root.v, 32); break;
            case 1: memcpy(out, C::fr_generator.v, 32); break;
            case 2: store_g1<C>(out, C::g1); break;
            case 3: store_g2<C>(out, C::g2); break;
            default: return 1;
        }
    });
    return 0;
}

// group ops for unit tests.  op: 0 add(p,q) 1 mul(p, k_bigint[4]);  g2 = 0/1
int orc_group_op(int curve, int g2, int op, const u64* p, const u64* q_or_k, u64* out) {
    DISPATCH(curve, {
        if (!g2) {
            typedef Jac<typename C::Fq> J;
            J a = J::from_affine(load_g1<C>(p));
            J r = op == 0 ? a.add_affine(load_g1<C>(q_or_k)) : a.mul_bigint(q_or_k, 4);
            store_g1<C>(out, r.to_affine());
        } else {
            typedef Jac<typename C::Fq2> J;
            J a = J::from_affine(load_g2<C>(p));
            J r = op == 0 ? a.add_affine(load_g2<C>(q_or_k)) : a.mul_bigint(q_or_k, 4);
            store_g2<C>(out, r.to_affine());
        }
    });
    return 0;
}

// data: n = 2^log_n Fr in natural order; inverse/coset select ifft / coset variants
int orc_ntt(int curve, u64* data, int log_n, int inverse, int coset) {
    int rc = 0;
    DISPATCH(curve, {
        typedef typename C::Fr Fr;
        Domain<Fr> dom((size_t)1 << log_n, C::two_adic_root, C::two_adicity);
        if (!dom.ok) rc = 1;
        else {
            Fr* a = reinterpret_cast<Fr*>(data);
            if (!inverse) { if (coset) dom.coset_fft(a, C::fr_generator); else dom.fft(a); }
            else { if (coset) dom.coset_ifft(a, C::fr_generator); else dom.ifft(a); }
        }
    });
    return rc;
}

// h_out: n Fr ; abc_out: 3n Fr or null.  returns 0 / 1 (PolynomialDegreeTooLarge)
int orc_witness_map(int curve, const CsrView* abc, u64 num_inputs, u64 num_constraints, const u64* z, u64* h_out,
                    u64* abc_out) {
    int rc = 0;
    DISPATCH(curve, {
        std::vector<typename C::Fr> h;
        rc = witness_map<C>(abc, num_inputs, num_constraints, reinterpret_cast<const typename C::Fr*>(z), h,
                            reinterpret_cast<typename C::Fr*>(abc_out));
        if (!rc) memcpy(h_out, h.data(), h.size() * 32);
    });
    return rc;
}

u64 orc_domain_size(u64 num_coeffs) { u64 n = 1; while (n < num_coeffs) n <<= 1; return n; }

// scalars in Montgomery form (as the prover holds them); into_bigint applied inside
int orc_msm_g1(int curve, const u64* bases, const u64* scalars, u64 n, u64* out_affine) {
    DISPATCH(curve, {
        auto b = load_g1_vec<C>(bases, n);
        auto k = to_bigints(reinterpret_cast<const typename C::Fr*>(scalars), n);
        store_g1<C>(out_affine, msm_bigint<typename C::Fq, 4>(b.data(), k.data(), n, C::Fr::C.bits).to_affine());
    });
    return 0;
}
int orc_msm_g2(int curve, const u64* bases, const u64* scalars, u64 n, u64* out_affine) {
    DISPATCH(curve, {
        auto b = load_g2_vec<C>(bases, n);
        auto k = to_bigints(reinterpret_cast<const typename C::Fr*>(scalars), n);
        store_g2<C>(out_affine, msm_bigint<typename C::Fq2, 4>(b.data(), k.data(), n, C::Fr::C.bits).to_affine());
    });
    return 0;
}

// proof_out: a(2L) b(4L) c(2L); h_out (n Fr) / msm_parts (h,l,a,b1: 2L each, b2: 4L) / times (6 doubles) optional
int orc_prove(int curve, const PkView* pk, const CsrView* abc, u64 num_inputs, u64 num_constraints, const u64* z,
              u64 n_assign, const u64* r, const u64* s, u64* proof_out, u64* h_out, u64* msm_parts, double* times) {
    int rc = 0;
    PhaseTimes pt = {};
    DISPATCH(curve, { rc = prove<C>(*pk, abc, num_inputs, num_constraints, z, n_assign, r, s, proof_out, h_out, msm_parts, &pt); });
    if (times) memcpy(times, &pt, sizeof(pt));
    return rc;
}

int orc_setup(int curve, const CsrView* abc, u64 num_inputs, u64 num_constraints, u64 num_vars, u64 seed, u64* g1_out,
              u64* g2_out, u64* a_query, u64* b_g1_query, u64* b_g2_query, u64* h_query, u64* l_query, u64* gamma_abc,
              u64* trapdoor_out, u64* abc_t_out) {
    int rc = 0;
    DISPATCH(curve, {
        rc = setup<C>(abc, num_inputs, num_constraints, num_vars, seed, g1_out, g2_out, a_query, b_g1_query, b_g2_query,
                      h_query, l_query, gamma_abc, trapdoor_out, abc_t_out);
    });
    return rc;
}

int orc_trapdoor_proof(int curve, const u64* trapdoor, const u64* abc_t, u64 m, u64 num_inputs, u64 n, const u64* g1gen,
                       const u64* g2gen, const u64* z, const u64* h, const u64* r, const u64* s, u64* proof_out) {
    DISPATCH(curve, { trapdoor_proof<C>(trapdoor, abc_t, m, num_inputs, n, g1gen, g2gen, z, h, r, s, proof_out); });
    return 0;
}

int orc_syn_circuit(int curve, int k, u64 seed, u64* z_out, u64* row_ptr, uint32_t* colA, uint32_t* colB,
                    uint32_t* colC, u64* val) {
    DISPATCH(curve, { syn_circuit<C>(k, seed, z_out, row_ptr, colA, colB, colC, val); });
    return 0;
}

int orc_synth_bases(int curve, int g2, u64 seed, u64 n, u64* out) {
    DISPATCH(curve, {
        if (!g2) {
            std::vector<typename C::G1A> v;
            synth_bases<typename C::Fq>(C::g1, seed, n, v);
            for (size_t i = 0; i < n; ++i) store_g1<C>(out + i * 2 * C::FQ_LIMBS, v[i]);
        } else {
            std::vector<typename C::G2A> v;
            synth_bases<typename C::Fq2>(C::g2, seed, n, v);
            for (size_t i = 0; i < n; ++i) store_g2<C>(out + i * 4 * C::FQ_LIMBS, v[i]);
        }
    });
    return 0;
}

// on-curve check (0 = on curve / identity, 1 = not)
int orc_on_curve(int curve, int g2, const u64* p) {
    int rc = 0;
    DISPATCH(curve, {
        if (!g2) {
            auto a = load_g1<C>(p);
            if (!a.inf) rc = (a.y.sqr() == a.x.sqr() * a.x + C::b1) ? 0 : 1;
        } else {
            auto a = load_g2<C>(p);
            if (!a.inf) rc = (a.y.sqr() == a.x.sqr() * a.x + C::b2) ? 0 : 1;
        }
    });
    return rc;
}

// random Fr (Montgomery) stream identical to pymodel.SplitMix64.field
int orc_rand_fr(int curve, u64 seed, u64 n, u64* out) {
    DISPATCH(curve, {
        SplitMix64 rng(seed);
        for (u64 i = 0; i < n; ++i) { auto x = C::rand_fr(rng); memcpy(out + 4 * i, x.v, 32); }
    });
    return 0;
}

}  // extern "C"
