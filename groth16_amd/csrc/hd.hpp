// Host/device attribute shim.  The arithmetic headers (field.hpp, curve.hpp) are compiled
// for gfx950 device code and for the host glue that finishes a proof.
#pragma once
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define G16_HD __host__ __device__ __forceinline__
#define G16_HD_NOINLINE __host__ __device__ __noinline__
#else
#define G16_HD inline
#define G16_HD_NOINLINE
#endif
#define G16_UNROLL _Pragma("unroll")
