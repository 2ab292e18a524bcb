// Sustained rate of v_mfma_f32_16x16x32_bf16 on the whole chip with nothing else going on:
// every wave keeps ACC independent accumulator tiles in registers and issues MFMAs on
// constant operands in a loop.  Build:  make -C tools/probe   Run: python tools/mfma_peak.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

template <int ACC>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters, float seed) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(seed + (float)((threadIdx.x + i) & 7) * 0.125f);
        b[i] = (__bf16)(seed * 0.5f + (float)((threadIdx.x * 3 + i) & 7) * 0.0625f);
    }
    f32x4 acc[ACC];
    for (int t = 0; t < ACC; ++t) acc[t] = f32x4{seed * (float)t, seed + (float)t, 0.f, (float)threadIdx.x};   // distinct: nothing to merge
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < ACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int t = 1; t < ACC; ++t) s += acc[t];
    if (s[0] == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s[1] + s[2] + s[3];   // keeps the loop alive
}

extern "C" int mfma_probe_launch(float* out, int blocks, int iters, int acc, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    // Twelve accumulator tiles is what a wave of the eight-wave 3x3 kernel issues per tap.  (With 4 or
    // 8 the compiler rotates the accumulators through v_accvgpr copies in this loop and the
    // figure says nothing about the matrix pipe; check the ISA before adding a variant.)
    if (acc == 12) hipLaunchKernelGGL(mfma_loop<12>, dim3(blocks), dim3(256), 0, st, out, iters, 1.0f);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
