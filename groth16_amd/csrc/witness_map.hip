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

template <class Fr>
__global__ void spmv3_kernel(SpmvArgs<Fr> args, const Fr* __restrict__ z, uint64_t num_inputs, uint64_t nc, uint64_t n) {
    const int m = blockIdx.y;
    const uint64_t row = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n) return;
    Fr acc = Fr::zero();
    if (row < nc) {
        const uint64_t b = args.row_ptr[m][row], e = args.row_ptr[m][row + 1];
        const Fr one = Fr::one();
        for (uint64_t k = b; k < e; ++k) {
            const Fr coeff = args.val[m][k];
            const Fr v = z[args.col[m][k]];
            acc = acc + ((coeff == one) ? v : v * coeff);  // coeff.is_one() fast path, r1cs_to_qap.rs:37,57
        }
    } else if (m == 0 && row - nc < num_inputs) {
        acc = z[row - nc];
    }
    args.out[m][row] = acc;
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
                       EventTimer* ntt_timers) {
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
    hipLaunchKernelGGL((spmv3_kernel<Fr>), dim3((unsigned)((n + 255) / 256), 3), dim3(256), 0, st, args, d_z, ck->num_inputs,
                       ck->num_constraints, (uint64_t)n);
    G16_LAUNCH_CHECK();
    // ntt_timers (optional, two of them): the six transforms of :201-207,220-221, then the seventh (:232) with its un-permute
    if (ntt_timers) G16_TRY(ntt_timers[0].start(st));
    for (int m = 0; m < 3; ++m) {
        // ifft then coset fft (r1cs_to_qap.rs:201-207): inverse DIF, n^-1 g^bitrev(i), forward DIT
        G16_TRY((ntt_dif_dit<C>(dom, outs[m], /*dif_inverse=*/true, dom->s1_br, st)));
    }
    if (ntt_timers) G16_TRY(ntt_timers[0].stop(st));
    hipLaunchKernelGGL((quotient_kernel<Fr>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a, b, c, dom->zinv, n);
    G16_LAUNCH_CHECK();
    if (ntt_timers) G16_TRY(ntt_timers[1].start(st));
    G16_TRY((ntt_dif<C>(dom, a, /*inverse=*/true, st)));
    G16_TRY((bitrev_scale<C>(dom, d_h, a, dom->s2, nullptr, st)));
    if (ntt_timers) G16_TRY(ntt_timers[1].stop(st));
    return G16_OK;
}

template int witness_map_device<Bls12_381>(const DeviceCircuit<Bls12_381>*, const Bls12_381::Fr*, Bls12_381::Fr*, Arena&, hipStream_t, EventTimer*);
template int witness_map_device<Bn254>(const DeviceCircuit<Bn254>*, const Bn254::Fr*, Bn254::Fr*, Arena&, hipStream_t, EventTimer*);

}  // namespace g16
