// Shared helpers for the gfx950 kernels of libyolo_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace yolo_amd {

// ---- error reporting -------------------------------------------------------
// Every entry point returns 0 on success or a negative code and leaves a
// message for yolo_amd_last_error().
enum : int {
    YA_OK = 0,
    YA_ERR_ARG = -1,       // bad argument (shape, alignment, null pointer)
    YA_ERR_HIP = -2,       // a HIP runtime call failed
    YA_ERR_WORKSPACE = -3, // workspace too small
    YA_ERR_UNSUPPORTED = -4,
    YA_ERR_BUSY = -5,      // a shared resource is still in use by work queued on another stream
};

void set_error(const char* fmt, ...);

#define YA_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            yolo_amd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                                __FILE__, __LINE__);                                    \
            return yolo_amd::YA_ERR_HIP;                                                \
        }                                                                               \
    } while (0)

#define YA_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) {                             \
            yolo_amd::set_error(__VA_ARGS__);      \
            return yolo_amd::YA_ERR_ARG;           \
        }                                          \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static inline uint32_t next_pow2(uint32_t x) {
    uint32_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

}  // namespace yolo_amd

#include "common_host.h"

namespace yolo_amd {

__device__ __forceinline__ uint32_t fdiv_u32(uint32_t n, FastDiv f) {
    if (f.d == 1) return n;
    // Granlund-Montgomery round-up form: q = (hi + ((n - hi) >> 1)) >> (l - 1)
    uint32_t hi = __umulhi(n, f.mul);
    return (hi + ((n - hi) >> 1)) >> (f.shift - 1);
}

// One element of yolo_v3.predict (/root/reference/yolo_v3.py:205-262) for
// channel d of a box at grid cell (gx, gy): shared by the decode kernel and by
// the detection convolutions that decode in their epilogue, so both give the
// same bits.  anc = anchor width (d == 2) or height (d == 3) in grid units.
// exp and the reciprocal are the hardware's (v_exp_f32 on x * log2(e), v_rcp_f32:
// about 1 ulp each; the scaled argument adds |x| * 1e-7 relative): for |x| <= 20 the
// result is within 2e-6 relative of the correctly rounded value, inside the 1e-5 the
// fp32 decode is held to, and a detection layer's 2.8 million values per image cost a
// fifth of the instructions of expf + IEEE division; up to |x| = 87, where fp32 exp
// leaves the normal range, the scaled argument keeps the error under 6e-6.  The decode
// is therefore held to a TOLERANCE (1e-5 relative), not to the reference's bits: an
// objectness within an ulp or two of the confidence threshold can fall on the other
// side of it than in the reference.  Overflow behaves as in the reference: exp of a
// large argument is +inf, sigmoid of a large negative one is 0.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
__device__ __forceinline__ float decode_value(float x, int d, float gx, float gy, float anc, float stride) {
    if (d >= 4) return sigmoid_fast(x);
    if (d < 2) return __fmul_rn(__fadd_rn(sigmoid_fast(x), d == 0 ? gx : gy), stride);
    return __fmul_rn(__fmul_rn(exp_fast(x), anc), stride);
}

}  // namespace yolo_amd
