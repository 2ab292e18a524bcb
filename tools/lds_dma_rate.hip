// How fast one CU takes LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) from an L2-resident
// source, alone and next to ds_read_b128 traffic: the LDS-side budget of the 3x3 kernel's tap loop.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/lds_dma_rate tools/lds_dma_rate.hip && tools/bin/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(const char* sbase, uint32_t lane_bytes, uint32_t lds_base) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_bytes), "s"(sbase), "s"(lds_base) : "memory", "m0");
}

// MODE 0: DMA only; 1: ds_read_b128 only; 2: per step 2 DMA + 8 reads (the tap loop's mix)
template <int MODE>
__global__ void __launch_bounds__(512) rate_kernel(const char* src, uint32_t* sink, int iters, long long* clocks) {
    __shared__ __attribute__((aligned(1024))) char lds[8 * 4096];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t mybase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + w * 4096);
    const uint32_t lane_bytes = lane * 16;
    u32x4 acc = u32x4{0, 0, 0, 0};
    const long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
        const char* sb = src + ((it & 15) * 4096 + w * 2048 % 4096);
        if (MODE != 1) {
            dma16(sb, lane_bytes, mybase + (it & 1) * 2048);
            dma16(sb + 1024, lane_bytes, mybase + (it & 1) * 2048 + 1024);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        if (MODE != 0) {
            asm volatile("" ::: "memory");                 // LDS may have changed: read again
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(lds + w * 4096 + ((k * 384 + lane * 16) & 4095));
                acc += v;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long c1 = clock64();
    sink[(size_t)blockIdx.x * 512 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) clocks[0] = c1 - c0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    char* src; uint32_t* sink; long long* clocks;
    CK(hipMalloc(&src, 1 << 17));
    CK(hipMemset(src, 1, 1 << 17));
    CK(hipMalloc(&sink, (size_t)512 * 512 * sizeof(uint32_t)));
    CK(hipMalloc(&clocks, sizeof(long long)));
    const int iters = 20000;
    const char* names[3] = {"LDS-DMA only (2 KiB per wave and step)", "ds_read_b128 only (8 KiB per wave and step)",
                            "2 LDS-DMA + 8 ds_read_b128 per wave and step"};
    for (int mode = 0; mode < 3; ++mode)
        for (int wgs = 1; wgs <= 2; ++wgs) {
            const int grid = 256 * wgs;
            for (int rep = 0; rep < 2; ++rep) {
                const int n = rep ? iters : 100;
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(grid), dim3(512), 0, 0, src, sink, n, clocks);
                if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(grid), dim3(512), 0, 0, src, sink, n, clocks);
                if (mode == 2) hipLaunchKernelGGL(rate_kernel<2>, dim3(grid), dim3(512), 0, 0, src, sink, n, clocks);
                CK(hipDeviceSynchronize());
            }
            long long c;
            CK(hipMemcpy(&c, clocks, sizeof(c), hipMemcpyDeviceToHost));
            printf("%s, %d workgroup(s) of 8 waves per CU: %.0f cycles per step of workgroup 0\n", names[mode], wgs,
                   (double)c / iters);
        }
    return 0;
}
