// CRS generation for a circuit given the toxic waste: SURVEY.md row f3, the caller in front of the prover path.
//
// Replaces Groth16::generate_parameters_with_qap (/root/reference/src/generator.rs:47-208) with the same split the
// reference has:
//   host   instance_map_with_evaluation (src/r1cs_to_qap.rs:120-170): Lagrange coefficients at t by one batch
//          inversion, the a/b/c(t) accumulation over the constraint matrices in the reference's row order, gamma_abc / l
//          (generator.rs:113-123) and the h-query scalars zt/delta * t^i (r1cs_to_qap.rs:236-246) -- O(nnz + n) field
//          products, seconds at 2^22;
//   GPU    the ~5n fixed-base scalar multiplications (BatchMulPreprocessing::batch_mul, generator.rs:129-183):
//          8-bit windows over a 32 x 255 table of multiples of the generator (L2-resident, 0.8 / 1.6 MB), one lane per
//          scalar, 32 mixed additions and one field inversion each.
// The reference samples t itself (domain.sample_element_outside_domain, generator.rs:90); here the caller passes it with the
// rest of the toxic waste, so that a run can be reproduced (the parity tests replay the oracle's trapdoor).
#include "internal.hpp"
#include <algorithm>

namespace g16 {

static constexpr int FB_WINDOWS = 32, FB_DIGITS = 255;   // 8-bit windows cover 256 >= 255 scalar bits

// table[w * 255 + d - 1] = d * 2^(8w) * g
template <class F>
__global__ __launch_bounds__(64) void fixed_base_table_kernel(Affine<F> g, Affine<F>* __restrict__ table) {
    const uint32_t id = blockIdx.x * 64 + threadIdx.x;
    if (id >= FB_WINDOWS * FB_DIGITS) return;
    const uint32_t w = id / FB_DIGITS, d = id % FB_DIGITS + 1;
    XYZZ<F> base = XYZZ<F>::from_affine(g);
    for (uint32_t k = 0; k < 8 * w; ++k) base = base.dbl();
    uint32_t kw[1] = {d};
    table[id] = base.mul_bits(kw, 8).to_affine();
}

template <class F, class Fr>
__global__ __launch_bounds__(64) void fixed_base_mul_kernel(const Affine<F>* __restrict__ table, const Fr* __restrict__ scalars, uint64_t n,
                                                            Affine<F>* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    uint32_t s[Fr::N];
    scalars[i].to_canonical(s);   // into_bigint
    XYZZ<F> acc = XYZZ<F>::identity();
    for (int w = 0; w < FB_WINDOWS; ++w) {
        const uint32_t d = (s[w >> 2] >> (8 * (w & 3))) & 0xffu;
        if (d) acc.add_affine(table[w * FB_DIGITS + d - 1]);
    }
    out[i] = acc.to_affine();   // identity for a zero scalar: (0, 0)
}

template <class C>
struct Setup {
    typedef typename C::Fr Fr;
    typedef typename C::Fq Fq;
    typedef typename C::Fq2 Fq2;

    static Fr load_fr(const uint64_t* p) { Fr x; memcpy(&x, p, sizeof(Fr)); return x; }

    // instance_map_with_evaluation; a, b, c get num_variables entries
    static int qap_evaluations(const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv, const Fr& t, std::vector<Fr>& a,
                               std::vector<Fr>& b, std::vector<Fr>& c, Fr* zt_out, uint64_t* n_out) {
        if (ni == 0 || nv < ni) return G16_ERR_BAD_LENGTH;
        const uint64_t need = nc + ni;   // D::new(num_constraints + num_instance_variables), r1cs_to_qap.rs:131-132
        int log_n = 0;
        while (((uint64_t)1 << log_n) < need) {
            ++log_n;
            if (log_n > 40) return G16_ERR_DEGREE_TOO_LARGE;
        }
        if (log_n > C::TWO_ADICITY || log_n > 30) return G16_ERR_DEGREE_TOO_LARGE;
        const uint64_t n = (uint64_t)1 << log_n;
        Fr omega = C::two_adic_root();
        for (int k = log_n; k < C::TWO_ADICITY; ++k) omega = omega.sqr();
        Fr tn = t;
        for (int k = 0; k < log_n; ++k) tn = tn.sqr();
        const Fr zt = tn - Fr::one();    // evaluate_vanishing_polynomial
        if (zt.is_zero()) return G16_ERR_BAD_ARG;   // t inside the domain: the reference never samples such a t
        // evaluate_all_lagrange_coefficients(t): u_i = zt / n * w^i / (t - w^i), one inversion for all i
        std::vector<Fr> u(n), wp(n), pre(n);
        Fr run = Fr::one(), wpow = Fr::one();
        for (uint64_t i = 0; i < n; ++i) {
            wp[i] = wpow;
            u[i] = t - wpow;
            pre[i] = run;
            run = run * u[i];
            wpow = wpow * omega;
        }
        Fr inv = run.inverse();
        Fr n_fr = Fr::from_u64(n);
        const Fr zn = zt * n_fr.inverse();
        for (uint64_t i = n; i-- > 0;) {
            const Fr di = inv * pre[i];
            inv = inv * u[i];
            u[i] = zn * wp[i] * di;
        }
        a.assign(nv, Fr::zero());
        b.assign(nv, Fr::zero());
        c.assign(nv, Fr::zero());
        for (uint64_t j = 0; j < ni; ++j) a[j] = u[nc + j];   // r1cs_to_qap.rs:150-155
        for (int which = 0; which < 3; ++which) {             // :157-167
            std::vector<Fr>& dst = which == 0 ? a : which == 1 ? b : c;
            const g16_csr_view& m = abc[which];
            if (!m.row_ptr) return G16_ERR_BAD_ARG;
            const Fr* val = reinterpret_cast<const Fr*>(m.val);
            for (uint64_t i = 0; i < nc; ++i)
                for (uint64_t k = m.row_ptr[i]; k < m.row_ptr[i + 1]; ++k) {
                    if (m.col[k] >= nv) return G16_ERR_BAD_LENGTH;
                    dst[m.col[k]] = dst[m.col[k]] + u[i] * val[k];
                }
        }
        *zt_out = zt;
        *n_out = n;
        return G16_OK;
    }

    template <class F>
    static int batch_mul(const Affine<F>* d_table, const std::vector<Fr>& sc, Arena& arena, hipStream_t st, uint64_t* out, bool out_on_device) {
        typedef Affine<F> A;
        const uint64_t n = sc.size();
        if (n == 0) return G16_OK;
        if (!out) return G16_ERR_BAD_ARG;
        Fr* d_s = nullptr;
        G16_TRY(arena.alloc_n(n, &d_s));
        G16_HIP_TRY(hipMemcpyAsync(d_s, sc.data(), n * sizeof(Fr), hipMemcpyHostToDevice, st));
        A* d_out = reinterpret_cast<A*>(out);
        if (!out_on_device) G16_TRY(arena.alloc_n(n, &d_out));
        hipLaunchKernelGGL((fixed_base_mul_kernel<F, Fr>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, d_table, d_s, n, d_out);
        G16_LAUNCH_CHECK();
        if (!out_on_device) G16_HIP_TRY(hipMemcpyAsync(out, d_out, n * sizeof(A), hipMemcpyDeviceToHost, st));
        // the scalars live in a std::vector the caller may drop: finish the upload before returning
        G16_HIP_TRY(hipStreamSynchronize(st));
        return G16_OK;
    }

    static int generate(hipStream_t st, Arena& arena, const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv,
                        const g16_toxic_waste* tw, const uint64_t* g1_gen, const uint64_t* g2_gen, const g16_params_view* out) {
        const Fr alpha = load_fr(tw->alpha), beta = load_fr(tw->beta), gamma = load_fr(tw->gamma), delta = load_fr(tw->delta),
                 t = load_fr(tw->t);
        if (gamma.is_zero() || delta.is_zero()) return G16_ERR_UNEXPECTED_IDENTITY;   // generator.rs:110-111
        std::vector<Fr> a, b, c;
        Fr zt;
        uint64_t n = 0;
        G16_TRY(qap_evaluations(abc, ni, nc, nv, t, a, b, c, &zt, &n));
        const Fr gamma_inv = gamma.inverse(), delta_inv = delta.inverse();
        std::vector<Fr> gabc(ni), l(nv - ni), hs(n - 1);
        for (uint64_t i = 0; i < ni; ++i) gabc[i] = (beta * a[i] + alpha * b[i] + c[i]) * gamma_inv;          // generator.rs:113-117
        for (uint64_t i = ni; i < nv; ++i) l[i - ni] = (beta * a[i] + alpha * b[i] + c[i]) * delta_inv;      // :119-123
        {
            std::vector<Fr>().swap(c);
            const Fr base = zt * delta_inv;                                                                   // r1cs_to_qap.rs:243-245
            Fr p = Fr::one();
            for (uint64_t i = 0; i + 1 < n; ++i) { hs[i] = base * p; p = p * t; }
        }
        arena.reset();
        Affine<Fq>* t1 = nullptr;
        Affine<Fq2>* t2 = nullptr;
        G16_TRY(arena.alloc_n((size_t)FB_WINDOWS * FB_DIGITS, &t1));
        G16_TRY(arena.alloc_n((size_t)FB_WINDOWS * FB_DIGITS, &t2));
        Affine<Fq> g1;
        Affine<Fq2> g2;
        memcpy(&g1, g1_gen, sizeof(g1));
        memcpy(&g2, g2_gen, sizeof(g2));
        const unsigned tb = (FB_WINDOWS * FB_DIGITS + 63) / 64;
        hipLaunchKernelGGL((fixed_base_table_kernel<Fq>), dim3(tb), dim3(64), 0, st, g1, t1);
        G16_LAUNCH_CHECK();
        hipLaunchKernelGGL((fixed_base_table_kernel<Fq2>), dim3(tb), dim3(64), 0, st, g2, t2);
        G16_LAUNCH_CHECK();
        const bool dev = (out->flags & G16_PARAMS_DEVICE_PTRS) != 0;
        G16_TRY(batch_mul<Fq2>(t2, b, arena, st, out->b_g2_query, dev));        // generator.rs:129-135
        G16_TRY(batch_mul<Fq>(t1, a, arena, st, out->a_query, dev));            // :155
        G16_TRY(batch_mul<Fq>(t1, b, arena, st, out->b_g1_query, dev));         // :161
        G16_TRY(batch_mul<Fq>(t1, hs, arena, st, out->h_query, dev));           // :165-169
        G16_TRY(batch_mul<Fq>(t1, l, arena, st, out->l_query, dev));            // :174
        G16_TRY(batch_mul<Fq>(t1, gabc, arena, st, out->gamma_abc_g1, false));  // :183
        std::vector<Fr> s1 = {alpha, beta, delta}, s2 = {beta, delta, gamma};   // :143-147, :182
        std::vector<Affine<Fq>> o1(3);
        std::vector<Affine<Fq2>> o2(3);
        G16_TRY(batch_mul<Fq>(t1, s1, arena, st, reinterpret_cast<uint64_t*>(o1.data()), false));
        G16_TRY(batch_mul<Fq2>(t2, s2, arena, st, reinterpret_cast<uint64_t*>(o2.data()), false));
        memcpy(out->alpha_g1, &o1[0], sizeof(o1[0]));
        memcpy(out->beta_g1, &o1[1], sizeof(o1[1]));
        memcpy(out->delta_g1, &o1[2], sizeof(o1[2]));
        memcpy(out->beta_g2, &o2[0], sizeof(o2[0]));
        memcpy(out->delta_g2, &o2[1], sizeof(o2[1]));
        memcpy(out->gamma_g2, &o2[2], sizeof(o2[2]));
        return G16_OK;
    }
};

template <class C>
int generate_parameters_device(hipStream_t st, Arena& arena, const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv,
                               const g16_toxic_waste* tw, const uint64_t* g1_gen, const uint64_t* g2_gen, const g16_params_view* out) {
    return Setup<C>::generate(st, arena, abc, ni, nc, nv, tw, g1_gen, g2_gen, out);
}

template <class C>
int qap_evaluations_host(const g16_csr_view abc[3], uint64_t ni, uint64_t nc, uint64_t nv, const uint64_t* t, uint64_t* a_out, uint64_t* b_out,
                         uint64_t* c_out, uint64_t* zt_out) {
    typedef typename C::Fr Fr;
    std::vector<Fr> a, b, c;
    Fr zt;
    uint64_t n = 0;
    G16_TRY(Setup<C>::qap_evaluations(abc, ni, nc, nv, Setup<C>::load_fr(t), a, b, c, &zt, &n));
    memcpy(a_out, a.data(), nv * sizeof(Fr));
    memcpy(b_out, b.data(), nv * sizeof(Fr));
    memcpy(c_out, c.data(), nv * sizeof(Fr));
    memcpy(zt_out, &zt, sizeof(Fr));
    return G16_OK;
}

#define G16_INSTANTIATE_SETUP(C)                                                                                                        \
    template int generate_parameters_device<C>(hipStream_t, Arena&, const g16_csr_view*, uint64_t, uint64_t, uint64_t,                  \
                                               const g16_toxic_waste*, const uint64_t*, const uint64_t*, const g16_params_view*);       \
    template int qap_evaluations_host<C>(const g16_csr_view*, uint64_t, uint64_t, uint64_t, const uint64_t*, uint64_t*, uint64_t*,      \
                                         uint64_t*, uint64_t*);
G16_INSTANTIATE_SETUP(Bls12_381)
G16_INSTANTIATE_SETUP(Bn254)

}  // namespace g16
