// What register-only MFMA loops sustain on this GPU: the practical ceiling under the 2.5 PFLOP/s that
// 16-cycle v_mfma_f32_16x16x32_bf16 instructions would give at 2.4 GHz on 1024 SIMDs.  The MFMAs are
// assembly statements with twelve (16x16x32) or three (32x32x16) independent accumulators per wave, so
// that the compiler can neither merge the accumulators nor reorder the loop.  Every workgroup (four
// waves, one per SIMD) reports the shader cycles (clock64) its loop took.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int KIND>     // 0: 16x16x32, 1: 32x32x16
__global__ void __launch_bounds__(256) mfma_loop(const __bf16* src, float* dst, int iters, long long* clocks) {
    const int lane = threadIdx.x & 63;
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(src + lane * 8);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(src + 512 + lane * 8);
    float s = 0.f;
    long long c0, c1;
    if (KIND == 0) {
        f32x4 acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{(float)i, 0.f, (float)lane, 0.f};
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 12; ++i)
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        c1 = clock64();
#pragma unroll
        for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = (float)(i + j + lane);
        c0 = clock64();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        c1 = clock64();
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) s += acc[i][j];
    }
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clocks[blockIdx.x] = c1 - c0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int max_grid = 256 * 4;
    __bf16* src; float* dst; long long* clocks;
    CK(hipMalloc(&src, 1024 * sizeof(__bf16)));
    CK(hipMalloc(&dst, (size_t)max_grid * 256 * sizeof(float)));
    CK(hipMalloc(&clocks, max_grid * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int zero = 0; zero < 2; ++zero) {
        std::vector<__bf16> h(1024);
        srand(1);
        for (auto& v : h) v = zero ? (__bf16)0.f : (__bf16)((rand() % 2001 - 1000) / 1000.f);
        CK(hipMemcpy(src, h.data(), 1024 * sizeof(__bf16), hipMemcpyHostToDevice));
        for (int kind = 0; kind < 2; ++kind)
            for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
                const int grid = 256 * per_cu, per_iter = kind ? 3 : 12, iters = kind ? 40000 : 20000;
                const double flop_per = kind ? 32768.0 : 16384.0;
                for (int rep = 0; rep < 2; ++rep) {
                    const int n = rep ? iters : 200;
                    if (rep) CK(hipEventRecord(e0));
                    if (kind) hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, src, dst, n, clocks);
                    else hipLaunchKernelGGL(mfma_loop<0>, dim3(grid), dim3(256), 0, 0, src, dst, n, clocks);
                    if (rep) CK(hipEventRecord(e1));
                    CK(hipDeviceSynchronize());
                }
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<long long> hc(grid);
                CK(hipMemcpy(hc.data(), clocks, grid * sizeof(long long), hipMemcpyDeviceToHost));
                std::sort(hc.begin(), hc.end());
                const double n_mfma = (double)iters * per_iter;
                printf("%s, %s operands, %d workgroups (one wave per SIMD each) per CU: %.2f ms, %.0f TFLOP/s; cycles per MFMA "
                       "of one wave: min %.1f median %.1f max %.1f\n", kind ? "32x32x16" : "16x16x32", zero ? "zero" : "random",
                       per_cu, ms, (double)grid * 4 * n_mfma * flop_per / (ms * 1e-3) / 1e12, hc.front() / n_mfma,
                       hc[grid / 2] / n_mfma, hc.back() / n_mfma);
            }
    }
    return 0;
}
