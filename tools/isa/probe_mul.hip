// ISA probe: one kernel per field primitive so that llvm-objdump shows what a single call costs (tools/isa/count.py).
#include <hip/hip_runtime.h>
#include "../../groth16_amd/csrc/curve.hpp"
#include "../../groth16_amd/csrc/fp30.hpp"
#include "../../groth16_amd/csrc/params_gen.hpp"
using namespace g16;
typedef Fp30<Bls12_381FqP> F;
extern "C" __global__ void k_mul(F* a, const F* b) { const int i = threadIdx.x; a[i] = a[i].mul(b[i]); }
extern "C" __global__ void k_sqr(F* a, const F* b) { const int i = threadIdx.x; a[i] = a[i].sqr(); }
extern "C" __global__ void k_mulsub(F* a, const F* b) { const int i = threadIdx.x; a[i] = F::mul_sub_fused(a[i], b[i], a[i + 64], b[i + 64]); }
extern "C" __global__ void k_add(F* a, const F* b) { const int i = threadIdx.x; a[i] = a[i].add(b[i]); }
extern "C" __global__ void k_sub(F* a, const F* b) { const int i = threadIdx.x; a[i] = a[i].sub<8>(b[i]); }
extern "C" __global__ void k_copy(F* a, const F* b) { const int i = threadIdx.x; a[i] = b[i]; }
