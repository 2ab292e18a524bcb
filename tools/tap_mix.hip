// What LDS fragment reads and LDS-DMA instructions cost the MFMA stream of a SIMD when four waves share it:
// the tap loop of the 3x3 kernels as a synthetic loop.  Per step and wave: NDMA LDS-DMA instructions of 1 KiB
// from an L2-resident source, NREAD ds_read_b128 whose results feed the MFMAs, NMFMA v_mfma_f32_16x16x32_bf16
// on independent accumulators (in VGPRs, or in AGPRs with ACC_A).  512 workgroups of eight waves (two per CU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/tap_mix tools/tap_mix.hip && tools/bin/tap_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* sbase, uint32_t lane_bytes, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_bytes), "s"(sbase), "s"(lds_base) : "memory", "m0");
}

template <int NREAD, int NDMA, int NMFMA, bool ACC_A>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
mix_kernel(const char* src, float* sink, int iters, long long* clocks, uint32_t seed) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * 8192];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* mine = lds + w * 8192;
    const uint32_t mybase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)mine;
    const uint32_t lane_bytes = lane * 16;
    // something that is not zero, so that the clock is the clock of real data
    for (int i = 0; i < 8; ++i) {
        u32x4 v = u32x4{0x3f803f80u ^ (uint32_t)(lane * 0x01010101u), 0x3fa0bf90u + (uint32_t)i, 0xbf853f33u ^ (uint32_t)(lane << 3), 0x3f4cbf19u};
        if (seed) {
            // random signs and mantissas, magnitudes in [0.5, 2): what activations and weights look like to the multipliers
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t h = seed ^ (blockIdx.x * 0x9E3779B1u) ^ ((uint32_t)(threadIdx.x * 8 + i) * 0x85EBCA77u) ^ ((uint32_t)e * 0xC2B2AE3Du);
                h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
                v[e] = (h & 0x80ff80ffu) | 0x3f003f00u;
            }
        }
        *reinterpret_cast<u32x4*>(mine + i * 1024 + lane * 16) = v;
    }
    __syncthreads();
    f32x4 acc[NMFMA];
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) acc[i] = f32x4{(float)i, 0.f, (float)lane, 0.f};
    bf16x8 afrag = *reinterpret_cast<const bf16x8*>(mine + lane * 16);
    bf16x8 bfix = *reinterpret_cast<const bf16x8*>(mine + 1024 + lane * 16);
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const char* sb = src + ((it & 15) * 4096 + (w & 1) * 2048);
#pragma unroll
        for (int k = 0; k < NDMA; ++k) dma16(sb + k * 1024, lane_bytes, mybase + 4096 + (it & 1) * 2048 + (k & 1) * 1024);
        if (NDMA) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_setprio(2);
        // fragments are read four at a time, each batch right before the MFMAs that use it
        constexpr int NB = NREAD > 0 ? (NREAD + 3) / 4 : 1;
#pragma unroll
        for (int bch = 0; bch < NB; ++bch) {
            bf16x8 fr[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (bch * 4 + k < NREAD)
                    fr[k] = *reinterpret_cast<const bf16x8*>(mine + (((bch * 4 + k) * 272 + lane * 16) & 4095));
            const int i0 = NREAD > 0 ? (bch * 4 * NMFMA + NREAD - 1) / NREAD : 0;
            const int i1 = NREAD > 0 ? (min(bch * 4 + 4, NREAD) * NMFMA + NREAD - 1) / NREAD : NMFMA;
#pragma unroll
            for (int i = 0; i < NMFMA; ++i) {
                if (i < i0 || i >= i1) continue;
                const bf16x8 b = NREAD > 0 ? fr[(i * NREAD) / NMFMA - bch * 4] : bfix;
                if (ACC_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(afrag), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(afrag), "v"(b));
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NMFMA; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = s;
    if (threadIdx.x == 0) clocks[blockIdx.x] = c1 - c0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static char* src; static float* sink; static long long* clocks; static hipEvent_t e0, e1;

static uint32_t g_seed = 0;

template <int NREAD, int NDMA, int NMFMA, bool ACC_A>
int run(int grid) {
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        if (rep) CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<NREAD, NDMA, NMFMA, ACC_A>), dim3(grid), dim3(512), 0, 0, src, sink, rep ? iters : 200, clocks, g_seed);
        if (rep) CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
    }
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> hc(grid);
    CK(hipMemcpy(hc.data(), clocks, grid * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end());
    const double cyc = (double)hc[grid / 2] / iters;
    const int per_simd = grid * 8 / 1024;
    // (s_memtime ticks are not shader cycles under this load -- twelve MFMAs of four waves cannot issue in fewer
    // than 768 cycles and take 615 ticks -- so the rate comes from the HIP events alone)
    const double tf = (double)grid * 8 * iters * NMFMA * 16384.0 / (ms * 1e-3) / 1e12;
    printf("%s data: %2d reads %d DMA %2d MFMA per step and wave, accumulators in %s, %d waves/SIMD: %5.0f TFLOP/s = %.2f us per step "
           "(%.0f s_memtime ticks per step and wave)\n",
           g_seed ? "random" : "patterned", NREAD, NDMA, NMFMA, ACC_A ? "AGPRs" : "VGPRs", per_simd, tf, ms * 1e3 / iters, cyc);
    return 0;
}

int main() {
    CK(hipMalloc(&src, 1 << 17));
    CK(hipMemset(src, 0x3c, 1 << 17));
    CK(hipMalloc(&sink, (size_t)512 * 512 * sizeof(float)));
    CK(hipMalloc(&clocks, 512 * sizeof(long long)));
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // the same loops on random operands first (the board's power limit sets the clock, and the multipliers'
    // power follows the bits that toggle): the MFMA stream alone, and the 3x3 kernel's mix
    g_seed = 0x1234567u;
    run<0, 0, 12, false>(512);
    run<8, 2, 12, false>(512);
    run<9, 2, 14, false>(512);
    g_seed = 0;
    for (int grid = 512; grid >= 256; grid -= 256) {
        run<0, 0, 12, false>(grid);
        run<4, 0, 12, false>(grid);
        run<8, 0, 12, false>(grid);
        run<0, 2, 12, false>(grid);
        run<8, 2, 12, false>(grid);
        run<8, 2, 12, true>(grid);
        run<8, 0, 12, true>(grid);
        run<0, 0, 12, true>(grid);
        run<13, 2, 22, false>(grid);
        run<13, 2, 22, true>(grid);
        run<10, 4, 24, false>(grid);
    }
    return 0;
}
