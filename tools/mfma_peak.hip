// What a register-only MFMA loop sustains on this GPU: the practical ceiling under the 2.5 PFLOP/s
// that 16-cycle v_mfma_f32_16x16x32_bf16 instructions would give at 2.4 GHz on 1024 SIMDs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma_peak tools/mfma_peak.hip && tools/bin/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(const __bf16* src, float* dst, int iters, long long* clocks) {
    const int lane = threadIdx.x & 63;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(src + lane * 8);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(src + 512 + lane * 8);
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same with v_mfma_f32_32x32x16_bf16 (twice the work per instruction, 32 cycles each)
template <int NACC>
__global__ void __launch_bounds__(256) mfma32_loop(const __bf16* src, float* dst, int iters, long long* clocks) {
    const int lane = threadIdx.x & 63;
    bf16x8 a = *reinterpret_cast<const bf16x8*>(src + lane * 8);
    bf16x8 b = *reinterpret_cast<const bf16x8*>(src + 512 + lane * 8);
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    dst[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = c1 - c0; clocks[1] = w1 - w0; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    const int iters = 20000, NACC = 12;
    __bf16* src; float* dst; long long* clocks;
    const int max_grid = 256 * 4;
    CK(hipMalloc(&src, 1024 * sizeof(__bf16)));
    CK(hipMalloc(&dst, (size_t)max_grid * 256 * sizeof(float)));
    CK(hipMalloc(&clocks, 2 * sizeof(long long)));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int zero = 0; zero < 2; ++zero) {
        std::vector<__bf16> h(1024);
        srand(1);
        for (auto& v : h) v = zero ? (__bf16)0.f : (__bf16)((rand() % 2001 - 1000) / 1000.f);
        CK(hipMemcpy(src, h.data(), 1024 * sizeof(__bf16), hipMemcpyHostToDevice));
        for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
            const int grid = 256 * wgs_per_cu;
            hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, src, dst, 200, clocks);   // warm-up
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma_loop<NACC>, dim3(grid), dim3(256), 0, 0, src, dst, iters, clocks);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            long long hc[2];
            CK(hipMemcpy(hc, clocks, sizeof(hc), hipMemcpyDeviceToHost));
            const double flop = (double)grid * 4 * iters * NACC * 16384.0;
            const double mfma_per_simd = (double)wgs_per_cu * iters * NACC;      // one wave of each workgroup per SIMD
            printf("%s operands, %d waves per SIMD: %.2f ms, %.0f TFLOP/s, %.2f GHz if an MFMA takes 16 cycles; "
                   "clock64 %lld, wall_clock64 %lld (100 MHz)\n", zero ? "zero" : "random", wgs_per_cu, ms,
                   flop / (ms * 1e-3) / 1e12, mfma_per_simd * 16 / (ms * 1e-3) / 1e9, hc[0], hc[1]);
        }
    }
    {
        std::vector<__bf16> h(1024);
        srand(1);
        for (auto& v : h) v = (__bf16)((rand() % 2001 - 1000) / 1000.f);
        CK(hipMemcpy(src, h.data(), 1024 * sizeof(__bf16), hipMemcpyHostToDevice));
        for (int wgs_per_cu = 1; wgs_per_cu <= 2; wgs_per_cu *= 2) {
            const int grid = 256 * wgs_per_cu, it32 = iters;
            hipLaunchKernelGGL(mfma32_loop<3>, dim3(grid), dim3(256), 0, 0, src, dst, 200, clocks);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(mfma32_loop<3>, dim3(grid), dim3(256), 0, 0, src, dst, it32, clocks);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            long long hc[2];
            CK(hipMemcpy(hc, clocks, sizeof(hc), hipMemcpyDeviceToHost));
            const double flop = (double)grid * 4 * it32 * 3 * 32768.0;
            printf("32x32x16, random operands, %d waves per SIMD: %.2f ms, %.0f TFLOP/s; clock64 %lld, wall_clock64 %lld; "
                   "%.1f cycles per MFMA per SIMD\n", wgs_per_cu, ms, flop / (ms * 1e-3) / 1e12, hc[0], hc[1],
                   (double)hc[0] / ((double)wgs_per_cu * it32 * 3));
        }
    }
    return 0;
}
