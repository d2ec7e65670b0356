// Signed-digit window decomposition shared by the device kernels and the host model.
//
// A canonical scalar s is biased once, s' = s + K with K = sum_w 2^(c-1) 2^(c w); window w's
// digit is then d_w = ((s' >> c w) & (2^c - 1)) - 2^(c-1), in [-2^(c-1), 2^(c-1) - 1], and
// s = sum_w d_w 2^(c w) exactly -- every window's digit is independent of the others, so
// digit planes can be produced and consumed window-parallel with no carry chain.
// |d_w| - 1 indexes one of B = 2^(c-1) buckets; d_w = 0 contributes nothing.
#pragma once
#include <cstdint>
#include "hd.hpp"

namespace g16 {

static constexpr int MSM_SWORDS = 11;  // 32-bit words holding s' (<= 272 bits) + 1 guard word

// raw c-bit window of a little-endian word array (guard word makes word+1 always readable)
G16_HD uint32_t window_raw(const uint32_t* sp, int w, int c) {
    const int bit = w * c, word = bit >> 5, sh = bit & 31;
    uint64_t two = (uint64_t)sp[word] | ((uint64_t)sp[word + 1] << 32);
    return (uint32_t)(two >> sh) & ((1u << c) - 1u);
}

// out: bucket index (0-based) and sign; returns false for a zero digit
G16_HD bool digit_to_bucket(uint32_t raw, int c, uint32_t* bucket, uint32_t* neg) {
    const int32_t d = (int32_t)raw - (int32_t)(1u << (c - 1));
    if (d == 0) return false;
    *neg = d < 0 ? 1u : 0u;
    *bucket = (uint32_t)(d < 0 ? -d : d) - 1u;
    return true;
}

}  // namespace g16
