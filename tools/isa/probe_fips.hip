// ISA probe for the product-scanning (FIPS) Montgomery product prototypes.
#include <hip/hip_runtime.h>
#include "../../groth16_amd/csrc/curve.hpp"
#include "../../groth16_amd/csrc/fp30.hpp"
#include "../../groth16_amd/csrc/params_gen.hpp"
using namespace g16;
typedef Bls12_381FqP P;
typedef Fp30<P> F;
extern "C" __global__ void k_mul(F* a, const F* b) { const int i = threadIdx.x; const F x = a[i], y = b[i]; a[i] = x.mul(y); }
extern "C" __global__ void k_sqr(F* a, const F* b) { const int i = threadIdx.x; const F x = a[i]; a[i] = x.sqr(); }
extern "C" __global__ void k_mulsub(F* a, const F* b) { const int i = threadIdx.x; const F x = a[i], y = b[i], z = a[i + 64], w = b[i + 64]; a[i] = F::mul_sub_fused(x, y, z, w); }
extern "C" __global__ void k_2mul(F* a, const F* b) { const int i = threadIdx.x; const F x = a[i], y = b[i], z = a[i + 64], w = b[i + 64]; const F u = x.mul(y), v = z.mul(w); a[i] = u; a[i + 64] = v; }
typedef Fp2p30<P> F2;
extern "C" __global__ void k_pair_mul(F2* a, const F2* b) { const int i = threadIdx.x; const F2 x = a[i], y = b[i]; a[i] = x.mul(y); }
extern "C" __global__ void k_pair_mulsub(F2* a, const F2* b) { const int i = threadIdx.x; const F2 x = a[i], y = b[i], z = a[i + 64], w = b[i + 64]; a[i] = F2::mul_sub_fused(x, y, z, w); }
extern "C" __global__ void k_negpk(F* a, const Fp<P>* b, const int* f) {
    const int i = threadIdx.x;
    a[i] = F::unpack_cond_neg(b[i], f[i] != 0);
}
extern "C" __global__ void k_negold(F* a, const Fp<P>* b, const int* f) {
    const int i = threadIdx.x;
    a[i] = F::cond_neg2(F::from_packed(b[i]), f[i] != 0);
}
