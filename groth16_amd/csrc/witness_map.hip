// R1CS -> QAP witness map on gfx950: h = (A z * B z - C z) / Z over the coset.
//
// Restates LibsnarkReduction::witness_map_from_matrices, /root/reference/src/r1cs_to_qap.rs:172-235
// (row evaluation: evaluate_constraint, :28-67), with the seven size-n transforms of :201-232
// arranged so that no standalone bit-reversal or scaling sweep is needed:
//   a,b,c  --DIF(w^-1)-->  bit-reversed coefficients          (ifft :201,202,220 without the 1/n)
//          --DIT(w), load pre-scaled by n^-1 g^bitrev(i)-->   coset evaluations, natural order (:206,207,221)
//   q = (a*b - c) / Z(g)                                      (:209, :223-230)
//   q      --DIF(w^-1)-->  bit-reversed                        (:232)
//   h[k] = q[bitrev(k)] * n^-1 g^-k                           (coset ifft's tail, fused with the un-permute)
// All kernels are HBM-streaming (32 B per element per sweep); the CSR mat-vec is gather-bound.
#include "internal.hpp"
#include <algorithm>
#include <new>

namespace g16 {

// rows [0, nc): <M_row, z>; matrix 0 additionally copies z[0..num_inputs) to rows nc.. (r1cs_to_qap.rs:195-199);
// every other row up to n is zero.  blockIdx.y selects A/B/C.
template <class Fr>
struct SpmvArgs {
    const uint64_t* row_ptr[3];
    const uint32_t* col[3];
    const Fr* val[3];
    Fr* out[3];
};

// The distributed witness map evaluates only the rows row0 + stride * i, i < rows, and stores row i's value at the
// bit-reversed position of i (out_rev_bits > 0), ready for a decimation-in-time transform; the single-GPU map passes
// (0, 1, n, 0).
template <class Fr>
__global__ void spmv3_kernel(SpmvArgs<Fr> args, const Fr* __restrict__ z, uint64_t num_inputs, uint64_t nc, uint64_t row0, uint64_t stride,
                             uint64_t rows, int out_rev_bits, uint64_t out0) {
    const int m = blockIdx.y;
    const uint64_t li = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= rows) return;
    const uint64_t row = row0 + stride * li;
    Fr acc = Fr::zero();
    if (row < nc) {
        const uint64_t b = args.row_ptr[m][row], e = args.row_ptr[m][row + 1];
        const Fr one = Fr::one();
        for (uint64_t k = b; k < e; ++k) {
            const uint32_t c = args.col[m][k];
            if (c >> 31) {   // marked at load time (mark_unit_coefficients): the coefficient is one -- its 32 bytes are not even read
                acc = acc + z[c & 0x7fffffffu];                     // coeff.is_one() fast path, r1cs_to_qap.rs:37,57
            } else {
                const Fr coeff = args.val[m][k];
                const Fr v = z[c];
                acc = acc + ((coeff == one) ? v : v * coeff);
            }
        }
    } else if (m == 0 && row - nc < num_inputs) {
        acc = z[row - nc];
    }
    args.out[m][out0 + (out_rev_bits ? (uint64_t)(__brev((uint32_t)li) >> (32 - out_rev_bits)) : li)] = acc;
}

// a <- (a*b - c) * zinv
template <class Fr>
__global__ void quotient_kernel(Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, Fr zinv, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = (a[i] * b[i] - c[i]) * zinv;
}

template <class C>
int witness_map_device(const DeviceCircuit<C>* ck, const typename C::Fr* d_z, typename C::Fr* d_h, Arena& arena, hipStream_t st,
                       EventTimer* ntt_timers, ZUpload* up) {
    typedef typename C::Fr Fr;
    const Domain<C>* dom = ck->dom;
    const size_t n = dom->n;
    Fr *a = nullptr, *b = nullptr, *c = nullptr;
    G16_TRY(arena.alloc_n(n, &a));
    G16_TRY(arena.alloc_n(n, &b));
    G16_TRY(arena.alloc_n(n, &c));
    SpmvArgs<Fr> args;
    Fr* outs[3] = {a, b, c};
    for (int m = 0; m < 3; ++m) {
        args.row_ptr[m] = ck->row_ptr[m];
        args.col[m] = ck->col[m];
        args.val[m] = ck->val[m];
        args.out[m] = outs[m];
    }
    if (!up) {
        hipLaunchKernelGGL((spmv3_kernel<Fr>), dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, st, args, d_z, ck->num_inputs,
                           ck->num_constraints, (uint64_t)0, (uint64_t)1, (uint64_t)n, 0, (uint64_t)0);
        G16_LAUNCH_CHECK();
    } else {
        // The assignment is still on the host (full_assignment: &[F], prover.rs:33).  Everything after the mat-vec needs all of a, b, c,
        // but the mat-vec itself does not need all of z: piece by piece on the copy stream, and after each piece the row blocks it
        // unlocks (need_col, computed at load time: circuits allocate their variables as they emit constraints, so row block k
        // mostly reads piece <= k -- exactly so for the benchmark's product chain; a circuit whose first row reads the last variable
        // simply gets no overlap).  Hides most of the mat-vec (0.44 ms at 2^22) behind the 2.4 ms a 128 MiB upload takes at 55 GB/s; the
        // rest of the proof -- every transform needs all of a, b, c, every sort all of z -- cannot start before the last byte has landed.
        constexpr int ZC = DeviceCircuit<C>::Z_CHUNKS;
        const uint64_t nz = ck->num_variables;
        const int pieces = (int)std::min<uint64_t>(4, std::max<uint64_t>(1, nz >> 16));   // >= 2 MiB per piece; four of them
        up->pieces = pieces;
        const uint64_t per = (nz + pieces - 1) / pieces;
        // all the copies first, back to back (pinned memory: each call returns at once, the pieces follow each other on the copy
        // engine with ~10 us between them; enqueued in turn with the launches they unlock the host's ~45 us per round of calls sat
        // BETWEEN the copies and cost more than the overlap gave -- kernel + copy trace of round 5, profiles/r05_upload_timeline.txt)
        for (int k = 0; k < pieces; ++k) {
            const uint64_t lo = per * k, hi = std::min(nz, per * (k + 1));
            if (hi > lo)
                G16_HIP_TRY(hipMemcpyAsync(const_cast<Fr*>(d_z) + lo, static_cast<const Fr*>(up->host) + lo, (hi - lo) * sizeof(Fr),
                                           hipMemcpyHostToDevice, up->copy_stream));
            G16_HIP_TRY(hipEventRecord(up->landed[k], up->copy_stream));
        }
        int next_block = 0;
        for (int k = 0; k < pieces; ++k) {
            const uint64_t hi = std::min(nz, per * (k + 1));
            int last_block = next_block;
            while (last_block < ZC && (k == pieces - 1 || ck->need_col[last_block] < hi)) ++last_block;
            if (last_block > next_block) {
                const uint64_t r_lo = n * (uint64_t)next_block / ZC, r_hi = n * (uint64_t)last_block / ZC;
                G16_HIP_TRY(hipStreamWaitEvent(st, up->landed[k], 0));
                if (r_hi > r_lo) {
                    hipLaunchKernelGGL((spmv3_kernel<Fr>), dim3((unsigned)((r_hi - r_lo + 255) / 256), 3), dim3(256), 0, st, args, d_z, ck->num_inputs,
                                       ck->num_constraints, r_lo, (uint64_t)1, r_hi - r_lo, 0, r_lo);
                    G16_LAUNCH_CHECK();
                }
                next_block = last_block;
            }
        }
    }
    // ntt_timers (optional, two of them): the six transforms of :201-207,220-221, then the seventh (:232) with its un-permute
    if (ntt_timers) G16_TRY(ntt_timers[0].start(st));
    // ifft then coset fft of a, b, c (r1cs_to_qap.rs:201-207, 220-221): inverse DIF, n^-1 g^bitrev(i), forward DIT -- the three
    // chains in one launch per sweep
    G16_TRY((ntt_dif_dit_batch<C>(dom, outs, 3, /*dif_inverse=*/true, dom->s1_br, st)));
    if (ntt_timers) G16_TRY(ntt_timers[0].stop(st));
    if (ntt_timers) G16_TRY(ntt_timers[1].start(st));
#ifdef G16_NO_FUSED_QUOTIENT
    hipLaunchKernelGGL((quotient_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, c, dom->zinv, n);
    G16_LAUNCH_CHECK();
    G16_TRY((ntt_dif<C>(dom, a, /*inverse=*/true, st)));
#else
    // (a b - c) / Z(g) (r1cs_to_qap.rs:223-230) is computed by the first sweep of the last transform as it loads its tile
    if (dom->log_n == 0) {
        hipLaunchKernelGGL((quotient_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, c, dom->zinv, n);
        G16_LAUNCH_CHECK();
    } else {
        G16_TRY((ntt_dif_quotient<C>(dom, a, b, c, dom->zinv, /*inverse=*/true, st)));
    }
#endif
    G16_TRY((bitrev_scale<C>(dom, d_h, a, dom->s2, nullptr, st)));
    if (ntt_timers) G16_TRY(ntt_timers[1].stop(st));
    return G16_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Distributed witness map (see DistWm in internal.hpp; index maps as oracle/pymodel.py::distributed_witness_map)
// ---------------------------------------------------------------------------------------------------------------------
template <class Fr>
struct SmallRoots { Fr w[8]; };   // root^k, k < N / 2

// N-point transform of v (N = 2^LOGN <= 16) in registers: decimation in frequency, then the output un-permuted
template <class Fr, int LOGN>
__device__ __forceinline__ void dft_small(Fr* v, const SmallRoots<Fr>& roots) {
    constexpr int N = 1 << LOGN;
    G16_UNROLL for (int s = LOGN - 1; s >= 0; --s) {
        const int half = 1 << s;
        G16_UNROLL for (int i = 0; i < N; ++i) {
            if (i & half) continue;
            const int k = (i & (half - 1)) << (LOGN - 1 - s);
            const Fr u = v[i], t = v[i + half];
            v[i] = u + t;
            v[i + half] = k ? (u - t) * roots.w[k] : (u - t);
        }
    }
    G16_UNROLL for (int i = 0; i < N; ++i) {
        int j = 0;
        G16_UNROLL for (int b = 0; b < LOGN; ++b) j |= ((i >> b) & 1) << (LOGN - 1 - b);
        if (j > i) { const Fr t = v[i]; v[i] = v[j]; v[j] = t; }
    }
}

// one lane per column j < blk of an [N][blk] array (blockIdx.y: which of the chains):
//   v[i] = in[i blk + j];  v = DFT_N(v, roots_a) .* sc_a[. blk + j];  TWO: v = DFT_N(v, roots_b) .* sc_b[. blk + j];  out[. blk + j] = v
// (launch bounds: 128 lanes per workgroup, or the compiler assumes 1024, caps the kernel at 128 registers and spills the
// 16-element column -- 1.3 KB of scratch per lane)
template <class Fr, int LOGN, bool TWO>
__global__ __launch_bounds__(128) void dwm_column_kernel(const Fr* __restrict__ in0, const Fr* __restrict__ in1, const Fr* __restrict__ in2, Fr* __restrict__ out0,
                                  Fr* __restrict__ out1, Fr* __restrict__ out2, size_t blk, SmallRoots<Fr> roots_a, const Fr* __restrict__ sc_a,
                                  SmallRoots<Fr> roots_b, const Fr* __restrict__ sc_b) {
    constexpr int N = 1 << LOGN;
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= blk) return;
    const Fr* in = blockIdx.y == 0 ? in0 : blockIdx.y == 1 ? in1 : in2;
    Fr* out = blockIdx.y == 0 ? out0 : blockIdx.y == 1 ? out1 : out2;
    Fr v[N];
    G16_UNROLL for (int i = 0; i < N; ++i) v[i] = in[(size_t)i * blk + j];
    dft_small<Fr, LOGN>(v, roots_a);
    G16_UNROLL for (int i = 0; i < N; ++i) v[i] = v[i] * sc_a[(size_t)i * blk + j];
    if (TWO) {
        dft_small<Fr, LOGN>(v, roots_b);
        G16_UNROLL for (int i = 0; i < N; ++i) v[i] = v[i] * sc_b[(size_t)i * blk + j];
    }
    G16_UNROLL for (int i = 0; i < N; ++i) out[(size_t)i * blk + j] = v[i];
}

template <class C>
void dwm_destroy(DistWm<C>* d) {
    if (!d) return;
    (void)hipFree(d->tw1); (void)hipFree(d->sc_mid); (void)hipFree(d->tw2); (void)hipFree(d->sc_out);
    domain_destroy<C>(d->dom_m);
    delete d;
}

template <class C>
int dwm_create(const DeviceCircuit<C>* ck, int rank, int world, hipStream_t st, DistWm<C>** out) {
    typedef typename C::Fr Fr;
    const int log_n = ck->dom->log_n;
    int lw = 0;
    while ((1 << lw) < world) ++lw;
    if (world < 1 || world > 16 || (1 << lw) != world || rank < 0 || rank >= world) return G16_ERR_BAD_ARG;
    if (2 * lw > log_n) return G16_ERR_BAD_LENGTH;   // needs world^2 | n
    DistWm<C>* d = new (std::nothrow) DistWm<C>();
    if (!d) return G16_ERR_OOM;
    d->rank = rank; d->world = world; d->log_world = lw;
    d->M = (size_t)1 << (log_n - lw);
    d->blk = d->M >> lw;
    auto fail = [&](int code) { dwm_destroy<C>(d); return code; };
    int rc = domain_create<C>(log_n - lw, st, &d->dom_m);
    if (rc) return fail(rc);
    const size_t M = d->M, blk = d->blk, N = (size_t)world;
    Fr omega = C::two_adic_root();
    for (int i = log_n; i < C::TWO_ADICITY; ++i) omega = omega.sqr();
    const Fr omega_inv = omega.inverse();
    const Fr n_inv = Fr::from_u64((uint64_t)1 << log_n).inverse();
    const Fr g = C::fr_generator(), g_inv = C::fr_generator_inv();
    if (hipMalloc((void**)&d->tw1, M * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&d->sc_mid, M * sizeof(Fr)) != hipSuccess ||
        hipMalloc((void**)&d->tw2, M * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&d->sc_out, M * sizeof(Fr)) != hipSuccess)
        return fail(G16_ERR_OOM);
    const uint64_t r = (uint64_t)rank;
    if ((rc = gen_power_table<C>(d->tw1, M, omega_inv.pow_u64(r), Fr::one(), true, st))) return fail(rc);
    for (uint64_t k = 0; k < N; ++k) {
        const uint64_t e0 = r * blk + (uint64_t)M * k;      // index of column j = 0 in row k1 = k
        if ((rc = gen_power_table<C>(d->sc_mid + k * blk, blk, g, n_inv * g.pow_u64(e0), false, st))) return fail(rc);
        if ((rc = gen_power_table<C>(d->sc_out + k * blk, blk, g_inv, n_inv * g_inv.pow_u64(e0), false, st))) return fail(rc);
        const Fr wk = omega.pow_u64(k);                     // w_n^ka, exponent r blk + j
        if ((rc = gen_power_table<C>(d->tw2 + k * blk, blk, wk, wk.pow_u64(r * blk), false, st))) return fail(rc);
    }
    const Fr wN = omega.pow_u64((uint64_t)M), wN_inv = omega_inv.pow_u64((uint64_t)M);
    for (int k = 0; k < 8; ++k) { d->wn_fwd[k] = wN.pow_u64((uint64_t)k); d->wn_inv[k] = wN_inv.pow_u64((uint64_t)k); }
    d->zinv = ck->dom->zinv;
    if (hipStreamSynchronize(st) != hipSuccess) return fail(G16_ERR_HIP);
    *out = d;
    return G16_OK;
}

template <class C, bool TWO>
static int dwm_columns(const DistWm<C>* d, int chains, typename C::Fr* const in[3], typename C::Fr* const outp[3], const typename C::Fr* sc_a,
                       const typename C::Fr* sc_b, hipStream_t st) {
    typedef typename C::Fr Fr;
    SmallRoots<Fr> ra, rb;
    for (int k = 0; k < 8; ++k) { ra.w[k] = d->wn_inv[k]; rb.w[k] = d->wn_fwd[k]; }
    const dim3 grid((unsigned)((d->blk + 127) / 128), (unsigned)chains), block(128);
#define G16_DWM_LAUNCH(LOGN)                                                                                                          \
    hipLaunchKernelGGL((dwm_column_kernel<Fr, LOGN, TWO>), grid, block, 0, st, in[0], in[1], in[2], outp[0], outp[1], outp[2], d->blk, ra, sc_a, \
                       rb, sc_b)
    switch (d->log_world) {
        case 0: G16_DWM_LAUNCH(0); break;
        case 1: G16_DWM_LAUNCH(1); break;
        case 2: G16_DWM_LAUNCH(2); break;
        case 3: G16_DWM_LAUNCH(3); break;
        case 4: G16_DWM_LAUNCH(4); break;
        default: return G16_ERR_INTERNAL;
    }
#undef G16_DWM_LAUNCH
    G16_LAUNCH_CHECK();
    return G16_OK;
}

template <class C>
int dwm_stage(const DeviceCircuit<C>* ck, const DistWm<C>* d, int stage, const typename C::Fr* d_z, typename C::Fr* const work[3],
              typename C::Fr* const recv[3], typename C::Fr* h_local, hipStream_t st) {
    typedef typename C::Fr Fr;
    const Domain<C>* dm = d->dom_m;
    const size_t M = d->M;
    if (stage == 0) {
        // residue rows r + N i2 of A z, B z, C z (r1cs_to_qap.rs:186-199), stored bit-reversed; type-1 transform with w^-1
        // (the ifft without its 1/n): M-point decimation in time, then the twiddle w_n^(-r k2); chunk p of the result goes to rank p
        SpmvArgs<Fr> args;
        for (int m = 0; m < 3; ++m) {
            args.row_ptr[m] = ck->row_ptr[m];
            args.col[m] = ck->col[m];
            args.val[m] = ck->val[m];
            args.out[m] = work[m];
        }
        hipLaunchKernelGGL((spmv3_kernel<Fr>), dim3((unsigned)((M + 255) / 256), 3), dim3(256), 0, st, args, d_z, ck->num_inputs,
                           ck->num_constraints, (uint64_t)d->rank, (uint64_t)d->world, (uint64_t)M, dm->log_n, (uint64_t)0);
        G16_LAUNCH_CHECK();
        G16_TRY((ntt_dit_batch<C>(dm, work, 3, /*inverse=*/true, nullptr, st)));
        for (int m = 0; m < 3; ++m) G16_TRY((scale_by_table<C>(work[m], d->tw1, M, st)));
        return G16_OK;
    }
    if (stage == 1) {
        // finish the ifft (N-point transform over the source rank, w_N^-1), scale by n^-1 g^index (block distribution), start the
        // coset fft (type 2: N-point transform with w_N, twiddle w_n^((r blk + j) ka)); chunk ka goes to rank ka
        return dwm_columns<C, true>(d, 3, recv, work, d->sc_mid, d->tw2, st);
    }
    if (stage == 2) {
        // the received chunks are x[ib], ib natural: M-point transform with w_M (bit-reversed out), pointwise
        // (a b - c) / Z(g) (r1cs_to_qap.rs:209, 223-230; any common order works), then the last transform's type-1 half
        G16_TRY((ntt_dif_batch<C>(dm, recv, 3, /*inverse=*/false, st)));
        hipLaunchKernelGGL((quotient_kernel<Fr>), dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, recv[0], recv[1], recv[2], d->zinv, M);
        G16_LAUNCH_CHECK();
        G16_HIP_TRY(hipMemcpyAsync(work[0], recv[0], M * sizeof(Fr), hipMemcpyDeviceToDevice, st));
        G16_TRY((ntt_dit<C>(dm, work[0], /*inverse=*/true, nullptr, st)));
        G16_TRY((scale_by_table<C>(work[0], d->tw1, M, st)));
        return G16_OK;
    }
    if (stage == 3) {
        // N-point transform over the source rank and the coset ifft's tail n^-1 g^-index: h in block distribution
        typename C::Fr* const outp[3] = {h_local, nullptr, nullptr};
        return dwm_columns<C, false>(d, 1, recv, outp, d->sc_out, nullptr, st);
    }
    return G16_ERR_BAD_ARG;
}

#define G16_INSTANTIATE_DWM(C)                                                                                            \
    template int dwm_create<C>(const DeviceCircuit<C>*, int, int, hipStream_t, DistWm<C>**);                             \
    template void dwm_destroy<C>(DistWm<C>*);                                                                             \
    template int dwm_stage<C>(const DeviceCircuit<C>*, const DistWm<C>*, int, const typename C::Fr*, typename C::Fr* const[3], \
                              typename C::Fr* const[3], typename C::Fr*, hipStream_t);
G16_INSTANTIATE_DWM(Bls12_381)
G16_INSTANTIATE_DWM(Bn254)

// Circuit load: bit 31 of a device column index says "this coefficient is 1" (R1CS matrices are mostly +-1), so the sparse mat-vec
// skips the 32-byte coefficient read for those entries.  Only when every index fits 31 bits; the caller's arrays are not touched.
template <class Fr>
__global__ void mark_unit_kernel(uint32_t* __restrict__ col, const Fr* __restrict__ val, uint64_t nnz) {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nnz && val[k] == Fr::one()) col[k] |= 0x80000000u;
}
template <class C>
int mark_unit_coefficients(DeviceCircuit<C>* ck, hipStream_t st) {
    if (ck->num_variables >= (1ull << 31)) return G16_OK;
    for (int m = 0; m < 3; ++m) {
        if (!ck->nnz[m]) continue;
        hipLaunchKernelGGL((mark_unit_kernel<typename C::Fr>), dim3((unsigned)((ck->nnz[m] + 255) / 256)), dim3(256), 0, st, ck->col[m], ck->val[m],
                           ck->nnz[m]);
        G16_LAUNCH_CHECK();
    }
    return G16_OK;
}
template int mark_unit_coefficients<Bls12_381>(DeviceCircuit<Bls12_381>*, hipStream_t);
template int mark_unit_coefficients<Bn254>(DeviceCircuit<Bn254>*, hipStream_t);

template int witness_map_device<Bls12_381>(const DeviceCircuit<Bls12_381>*, const Bls12_381::Fr*, Bls12_381::Fr*, Arena&, hipStream_t, EventTimer*, ZUpload*);
template int witness_map_device<Bn254>(const DeviceCircuit<Bn254>*, const Bn254::Fr*, Bn254::Fr*, Arena&, hipStream_t, EventTimer*, ZUpload*);

}  // namespace g16
