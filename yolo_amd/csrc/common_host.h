// Host-usable pieces shared by the kernels and the host-side checkers.
#pragma once
#include <stdint.h>

namespace yolo_amd {

// ---- exact unsigned division by a runtime constant -------------------------
// q = n / d for 0 <= n < 2^31, 1 <= d < 2^31 via one 32x32->64 multiply.
// Host computes (mul, shift); the device uses __umulhi.
struct FastDiv {
    uint32_t mul;
    uint32_t shift;
    uint32_t d;
};

static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.mul = 0; f.shift = 0; return f; }
    uint32_t l = 0;
    while ((1u << l) < d) ++l;                 // ceil(log2 d)
    uint64_t m = (((uint64_t)1 << (32 + l)) + d - 1) / d - ((uint64_t)1 << 32);
    f.mul = (uint32_t)m;
    f.shift = l;
    return f;
}

// host replica of the device routine (same arithmetic, 64-bit multiply)
static inline uint32_t fdiv_u32_host(uint32_t n, FastDiv f) {
    if (f.d == 1) return n;
    uint32_t hi = (uint32_t)(((uint64_t)n * f.mul) >> 32);
    return (hi + ((n - hi) >> 1)) >> (f.shift - 1);
}

}  // namespace yolo_amd
