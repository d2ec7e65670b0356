// Known-bytes kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: the
// counters are exact only up to a per-access-width factor; "calibrate on a known byte count in your own access pattern").
// Each kernel below reproduces ONE access pattern of the prover's kernels over a footprint far beyond the 256 MiB Infinity
// Cache, and the host prints the true byte counts as JSON.  Run it inside the SAME `rocprofv3 --pmc FETCH_SIZE` /
// `--pmc WRITE_SIZE` passes as the bench (tools/gpu_session.sh pmc): factor = true bytes / (counter * 1024) per pattern,
// applied by tools/pmc_summary.py --calib.
//
//   calib_read16     coalesced streaming read, 16 B per lane (the guide's reference pattern)
//   calib_read32     coalesced read of 32-byte elements, one element per lane (Fr loads of the NTT / pointwise / SpMV kernels)
//   calib_write32    coalesced write of 32-byte elements
//   calib_gather96   one random 96-byte record per lane out of a 6 GB table (G1 bucket pass: Affine<Fq> gathers)
//   calib_gather192p one random 192-byte record per lane PAIR, each lane reading 2 x 48 B of it (G2 lane-pair bucket pass)
//   calib_write208   one 208-byte record per lane, scattered (partial-sum flushes of the bucket pass)
// Build: hipcc -O3 --offload-arch=gfx950 tools/calib.hip -o tools/bin/calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct alignas(16) V16 { uint32_t w[4]; };
struct alignas(16) R32 { uint32_t w[8]; };
struct alignas(16) R48 { uint32_t w[12]; };
struct alignas(16) R96 { uint32_t w[24]; };
struct alignas(16) R208 { uint32_t w[52]; };

__device__ __forceinline__ uint32_t fold(const uint32_t* w, int n) { uint32_t a = 0; for (int i = 0; i < n; ++i) a ^= w[i]; return a; }

__global__ __launch_bounds__(256) void calib_read16(const V16* __restrict__ in, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const V16 v = in[i]; acc ^= fold(v.w, 4); }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_read32(const R32* __restrict__ in, size_t n, uint32_t* __restrict__ sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const R32 v = in[i]; acc ^= fold(v.w, 8); }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(256) void calib_write32(R32* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        R32 v;
        for (int k = 0; k < 8; ++k) v.w[k] = (uint32_t)i + k;
        out[i] = v;
    }
}
__device__ __forceinline__ uint32_t lcg(uint32_t x) { return x * 1664525u + 1013904223u; }
// every lane: `per` random records (the bucket pass walks 64 entries per lane)
__global__ __launch_bounds__(128) void calib_gather96(const R96* __restrict__ tab, uint32_t nrec, uint32_t per, uint32_t* __restrict__ sink) {
    uint32_t x = (blockIdx.x * 128 + threadIdx.x) * 2654435761u + 7u, acc = 0;
    for (uint32_t e = 0; e < per; ++e) {
        x = lcg(x);
        const R96 v = tab[(uint64_t)(x >> 2) % nrec];
        acc ^= fold(v.w, 24);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
// lane pair (2k, 2k+1) shares one random 192-byte record = 4 x R48 (x.c0 x.c1 y.c0 y.c1); lane parity k reads parts k and 2 + k
__global__ __launch_bounds__(128) void calib_gather192p(const R48* __restrict__ tab, uint32_t nrec, uint32_t per, uint32_t* __restrict__ sink) {
    const uint32_t t = (blockIdx.x * 128 + threadIdx.x) >> 1, k = threadIdx.x & 1u;
    uint32_t x = t * 2654435761u + 7u, acc = 0;
    for (uint32_t e = 0; e < per; ++e) {
        x = lcg(x);
        const R48* rec = tab + (uint64_t)((x >> 2) % nrec) * 4;
        const R48 a = rec[k], b = rec[2 + k];
        acc ^= fold(a.w, 12) ^ fold(b.w, 12);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
__global__ __launch_bounds__(128) void calib_write208(R208* __restrict__ out, uint32_t nrec, uint32_t per) {
    uint32_t x = (blockIdx.x * 128 + threadIdx.x) * 2654435761u + 9u;
    for (uint32_t e = 0; e < per; ++e) {
        x = lcg(x);
        R208 v;
        for (int k = 0; k < 52; ++k) v.w[k] = x + k;
        out[(uint64_t)(x >> 2) % nrec] = v;
    }
}

int main() {
    const size_t STREAM_BYTES = (size_t)2 << 30;      // 2 GiB streamed (8x the Infinity Cache)
    const size_t TABLE_BYTES = (size_t)6 << 30;       // 6 GiB table for the gathers (the 2^22 G1 window table is 5 GB)
    void* buf = nullptr;
    uint32_t* sink = nullptr;
    CK(hipMalloc(&buf, TABLE_BYTES));
    CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 0x5a, TABLE_BYTES));
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto report = [&](const char* name, double rd, double wr, float ms) {
        printf("{\"kernel\": \"%s\", \"read_bytes\": %.0f, \"write_bytes\": %.0f, \"ms\": %.4f, \"GBps\": %.1f}\n", name, rd, wr, ms,
               (rd + wr) / (ms * 1e-3) / 1e9);
    };
    float ms;
    const unsigned G = 256 * 16;
#define TIMED(name, rd, wr, ...)                                   \
    for (int rep = 0; rep < 2; ++rep) {                            \
        CK(hipEventRecord(e0));                                    \
        __VA_ARGS__;                                               \
        CK(hipEventRecord(e1));                                    \
        CK(hipDeviceSynchronize());                                \
    }                                                              \
    CK(hipEventElapsedTime(&ms, e0, e1));                          \
    report(name, rd, wr, ms);
    TIMED("calib_read16", (double)STREAM_BYTES, 0.0,
          hipLaunchKernelGGL(calib_read16, dim3(G), dim3(256), 0, 0, (const V16*)buf, STREAM_BYTES / 16, sink));
    TIMED("calib_read32", (double)STREAM_BYTES, 0.0,
          hipLaunchKernelGGL(calib_read32, dim3(G), dim3(256), 0, 0, (const R32*)buf, STREAM_BYTES / 32, sink));
    TIMED("calib_write32", 0.0, (double)STREAM_BYTES, hipLaunchKernelGGL(calib_write32, dim3(G), dim3(256), 0, 0, (R32*)buf, STREAM_BYTES / 32));
    {
        const uint32_t lanes = 1u << 18, per = 64;   // 2^24 gathers of 96 B = 1.6 GB
        const uint32_t nrec = (uint32_t)(TABLE_BYTES / 96);
        TIMED("calib_gather96", (double)lanes * per * 96.0, 0.0,
              hipLaunchKernelGGL(calib_gather96, dim3(lanes / 128), dim3(128), 0, 0, (const R96*)buf, nrec, per, sink));
    }
    {
        const uint32_t lanes = 1u << 19, per = 64;   // 2^18 pairs x 64 records of 192 B = 3.2 GB
        const uint32_t nrec = (uint32_t)(TABLE_BYTES / 192);
        TIMED("calib_gather192p", (double)(lanes / 2) * per * 192.0, 0.0,
              hipLaunchKernelGGL(calib_gather192p, dim3(lanes / 128), dim3(128), 0, 0, (const R48*)buf, nrec, per, sink));
    }
    {
        const uint32_t lanes = 1u << 18, per = 8;    // 2^21 scattered 208-byte records = 436 MB
        const uint32_t nrec = (uint32_t)(TABLE_BYTES / 208);
        TIMED("calib_write208", 0.0, (double)lanes * per * 208.0,
              hipLaunchKernelGGL(calib_write208, dim3(lanes / 128), dim3(128), 0, 0, (R208*)buf, nrec, per));
    }
    (void)hipFree(buf);
    (void)hipFree(sink);
    return 0;
}
