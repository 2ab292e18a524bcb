// Fused convolution layers of the Darknet-53 / YOLOv3 graph for gfx950.
//
// One launch = conv (1x1 or 3x3, stride 1 or 2, padding ksize/2) + bias
// (BatchNorm folded on the host) + LeakyReLU(0.1) + residual add, reading an
// input that may be concat(upsample2x(src0), src1).  This is what
// yolo_v3.forward executes block by block through nn.Sequential(Conv2d,
// BatchNorm2d, LeakyReLU), nn.Upsample, torch.cat and the in-place shortcut
// add (/root/reference/yolo_v3.py:449-473, /root/reference/utilis.py:169-236).
//
// Formulation: implicit GEMM  D[channel][pixel] = sum_k W[channel][k] * X[k][pixel]
// with k = (kh*ksize + kw)*Cin + c, activations NHWC bf16, fp32 accumulation
// on v_mfma_f32_16x16x32_bf16.  Weights are the MFMA "A" operand and pixels
// the "B" operand, so every lane ends up owning 4 consecutive output channels
// of one pixel: the epilogue packs them into one 8-byte NHWC store.
//
// Both operand tiles are staged into LDS by LDS-DMA (global_load_lds_dwordx4):
// each lane supplies the global address of one 16-byte chunk, which is how the
// im2col gather, the zero padding (lanes outside the image read a zero page),
// the nearest-neighbour upsample and the concat are all folded into addressing
// with no extra pass over memory.  Two LDS stages: the loads of K-tile t+1 are
// in flight while the MFMAs of tile t run; one barrier per K-tile.
#include <stdlib.h>
#include <mutex>
#include <type_traits>

#include "common.h"
#include "conv_layout.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

// bn.hip
int launch_bn_train(const yolo_amd_bn_train& bn, long long M, int C, const void* residual, void* dst, int leaky,
                    hipStream_t stream);

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct ConvArgs {
    const __bf16* src0;
    const __bf16* src1;
    const __bf16* wgt;
    const float* bias;
    const __bf16* res;
    void* dst;
    const __bf16* zero;
    int C0, C1, up0;
    int H, W, Ho, Wo, M;
    int Cout, Kpad, ksize, stride;
    int leaky, dst_fp32;
    int tiles_m, tiles_n;
    FastDiv divHoWo, divWo;
    // fused head decode (dec_D > 0): dst is the (B, dec_rows, dec_D) prediction tensor
    int dec_D, dec_rows, dec_row_offset;
    float dec_stride;
    float dec_aw[YOLO_AMD_MAX_ANCHORS], dec_ah[YOLO_AMD_MAX_ANCHORS];
    FastDiv divD;
    // split K (small batches): workgroup (x, y) reduces K-steps [y*ksplit, (y+1)*ksplit)
    // and writes raw fp32 sums to slice y of dst; conv_splitk_finish_kernel adds them up
    int ksplit, kslices;
    long long* trace;      // development aid, null unless built with -DYOLO_AMD_DEV_TRACE: per workgroup
                           // eight words as in Conv3Args::trace
    // TAIL kernels: the 1x1 layer (128 -> 64) that reads this layer's output
    const __bf16* w2;      // [4][64][32] bf16 (K-tile major)
    const float* b2;
    __bf16* dst2;
    int leaky2;
};


// One LDS-DMA instruction: 16 bytes per lane from gsrc (per lane) to lds_dst (wave-uniform) +
// 16 * lane.  Issued as inline assembly rather than through the builtin: once the compiler
// sees an LDS-DMA instruction it stops trusting the order in which LDS reads return and puts a
// full `s_waitcnt lgkmcnt(0)` in front of the first use of ANY ds_read result, which serialises
// "eight fragment reads, then twelve MFMAs" where counted waits let the first MFMAs start
// after the first reads.  The loads' own completion is tracked by hand anyway (wait_vmcnt).
#ifndef YOLO_AMD_ASM_DMA
#define YOLO_AMD_ASM_DMA 1
#endif
__device__ __forceinline__ void lds_dma16(const __bf16* gsrc, char* lds_dst) {
#if YOLO_AMD_ASM_DMA
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
    // (m0 is named as clobbered on purpose -- nothing else in these kernels uses it)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off"
                 :
                 : "v"(gsrc), "s"(base)
                 : "memory", "m0");
#pragma clang diagnostic pop
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
#endif
}

// The same with the address split into a wave-uniform base (a scalar register pair) and a
// 32-bit per-lane byte offset, plus OFF constant bytes (the instruction's offset field, which the
// hardware adds to the source address AND to the LDS address): the form for a source whose per-lane part
// never changes (weights: one offset register for the whole kernel, the walk over taps and
// channel chunks done in scalar arithmetic, no vector instruction per load -- the vector
// instructions of the tap loop compete with the MFMAs of the other waves for issue slots).
template <int OFF = 0>
__device__ __forceinline__ void lds_dma16_sv(const __bf16* sbase, uint32_t lane_bytes, char* lds_dst) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed instruction offset");
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(lane_bytes), "s"(sbase), "s"(base), "n"(OFF)
                 : "memory", "m0");
#pragma clang diagnostic pop
}

// The same from a raw buffer: 16 bytes per lane from (descriptor base + soff + voff) to lds_dst + 16 * lane.
// The per-lane part is one 32-bit byte offset, the walk over a tensor goes through the scalar offset, and a
// lane whose offset lies beyond the descriptor's size reads as zero: padding needs no zero page, no select
// and no 64-bit address per lane.
__device__ __forceinline__ void lds_dma16_buf(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff, char* lds_dst) {
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds_dst;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(base)
                 : "memory", "m0");
#pragma clang diagnostic pop
}
constexpr uint32_t kRawBufferWord3 = 0x00020000u;  // untyped 32-bit raw buffer descriptor word 3
constexpr uint32_t kBufferNowhere = 0xFFFFFF00u;   // a byte offset beyond any descriptor these kernels build

// ---------------------------------------------------------------- tile epilogue through LDS
// The accumulator layout (one pixel x 4 channels per lane) gives 8-byte stores
// scattered over 16 rows, and a residual that is loaded just before it is
// needed costs one memory round trip per accumulator tile.  Instead the
// workgroup's output tile goes through LDS (free once the K loop is done):
//   1. the residual tile is copied in by LDS-DMA (no registers, every load in
//      flight at once);
//   2. each lane adds its accumulators (+bias, LeakyReLU) to the residual it
//      finds there in fp32, rounds ONCE to bf16 and writes back in place;
//   3. the tile is written out with 16-byte coalesced stores.
// Tile image: NSLOT rows (pixels) of BNT bf16, no padding (LDS-DMA writes 1 KiB
// runs); the 16-byte chunk c of row r sits at chunk position c ^ (r mod CH8),
// which spreads the per-lane 8-byte accesses of step 2 (16 rows, same channel)
// over the banks.
template <int BNT>
__device__ __forceinline__ int tile_addr(int row, int chl) {
    constexpr int CH8 = BNT / 8;
    const int c = chl >> 3;
    return row * (BNT * 2) + ((c ^ (row & (CH8 - 1))) << 4) + ((chl & 7) << 1);
}

template <int NSLOT, int BNT, int NT, typename PixFn>
__device__ __forceinline__ void tile_load_residual(char* lds, const __bf16* __restrict__ res,
                                                   const __bf16* zero, int Cout, int n0, int w,
                                                   int lane, PixFn pix) {
    constexpr int CH8 = BNT / 8, RPI = 64 / CH8;         // rows per wave-instruction
    constexpr int NINST = (NSLOT * CH8 + 63) / 64, NWAVE = NT / 64;
#pragma unroll
    for (int i = 0; i < (NINST + NWAVE - 1) / NWAVE; ++i) {
        const int j = w + i * NWAVE;
        if (j < NINST) {
            const int row = j * RPI + lane / CH8;
            const int c = (lane % CH8) ^ (row & (CH8 - 1));
            const __bf16* src = zero;
            if (row < NSLOT && n0 + c * 8 < Cout) {
                const long long m = pix(row);
                if (m >= 0) src = res + (size_t)m * Cout + n0 + c * 8;
            }
            lds_dma16(src, lds + j * 1024);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// 16 bytes of a layer's output: a plain store.  (With the non-temporal hint the conv stack
// measured 3.10 ms against 2.96: the next layer does find part of this output in L2.)
__device__ __forceinline__ void store_out16(__bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

template <int NSLOT, int BNT, int NT, typename PixFn>
__device__ __forceinline__ void tile_store(const char* lds, __bf16* __restrict__ dst, int Cout, int n0,
                                           PixFn pix) {
    constexpr int CH8 = BNT / 8, ITEMS = NSLOT * CH8, IT = (ITEMS + NT - 1) / NT;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = threadIdx.x + i * NT;
        const int slot = e / CH8, c8 = e - slot * CH8;
        if (e < ITEMS && n0 + c8 * 8 < Cout) {
            const long long m = pix(slot);
            if (m >= 0)
                store_out16(dst + (size_t)m * Cout + n0 + c8 * 8,
                            *reinterpret_cast<const bf16x8*>(lds + tile_addr<BNT>(slot, c8 * 8)));
        }
    }
}

// lane-side step: v[4] (bias and activation applied) for tile row `slot`,
// tile channel `chl`; adds the parked residual, rounds once, writes in place
template <int BNT>
__device__ __forceinline__ void tile_put(char* lds, int slot, int chl, float v[4], bool has_res) {
    bf16x4* p = reinterpret_cast<bf16x4*>(lds + tile_addr<BNT>(slot, chl));
    if (has_res) {
        const bf16x4 rr = *p;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
    }
    bf16x4 pk;
#pragma unroll
    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)v[e];
    *p = pk;
}

// max(v, slope*v) for 0 < slope <= 1: one multiply and one bare v_max_f32 (fmaxf / fmed3
// put a canonicalising v_max(v, v) in front of it)
__device__ __forceinline__ float leaky_max(float v, float slope) {
    float r;
    const float sv = slope * v;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(sv));
    return r;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// s_waitcnt vmcnt(n) for a count known only after unrolling (n <= 11; larger counts wait for 11)
__device__ __forceinline__ void wait_vmcnt_upto(int n) {
    switch (n) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 8: wait_vmcnt<8>(); break;
        case 9: wait_vmcnt<9>(); break;
        case 10: wait_vmcnt<10>(); break;
        default: wait_vmcnt<11>(); break;
    }
}

// wait until at most `stages` of the youngest stages (PER loads each) are in flight
template <int MAXS, int PER>
__device__ __forceinline__ void wait_stages(int stages) {
    if constexpr (MAXS == 0) {
        wait_vmcnt<0>();
    } else {
        if (stages >= MAXS) wait_vmcnt<MAXS * PER>();
        else wait_stages<MAXS - 1, PER>(stages);
    }
}

// AMUL = 2: the activation tile is two K-steps deep (BK*2 channels, a whole
// 128-byte line per pixel when BK = 32) while the weight tile stays BK deep, so
// every line the activation gather pulls from L2 is used completely and the
// ring still fits three workgroups per CU.
// TAIL (BN = 128 = all of the layer's channels): the 1x1 layer behind this one (128 -> 64
// channels) is computed from the output tile while it sits in LDS -- four K-steps per pixel
// group in ascending order, as this kernel would run that layer on its own (same bits) --
// and stored as well: that layer's launch, which read the 128 channels back, is saved.
template <int BM, int BN, int BK, int WAVES_M, int NSTAGE, int AMUL = 1, bool TAIL = false>
__global__ void __launch_bounds__(256)
conv_igemm_kernel(const ConvArgs a) {
    constexpr int BKA = BK * AMUL;             // depth of the activation tile
    constexpr int CH = BKA / 8;                // 16-byte chunks per activation tile row
    constexpr int RPI = 64 / CH;               // activation tile rows filled by one wave-instruction
    constexpr int CHW = BK / 8, RPIW = 64 / CHW;   // the same for the weight tile
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int WMP = BM / WAVES_M;          // pixels per wave
    constexpr int WNC = BN / WAVES_N;          // channels per wave
    constexpr int TP = WMP / 16, TC = WNC / 16;
    constexpr int TA = BM * CH / 64;           // wave-instructions per activation tile
    constexpr int TB = BN * CHW / 64;          // ... per weight tile
    constexpr int NA = (TA + 3) / 4, NB = (TB + 3) / 4;   // per wave
    constexpr int A_BYTES = BM * BKA * 2, B_BYTES = BN * BK * 2;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int KSTEPS = BK / 32;
    static_assert(AMUL == 1 || (AMUL == 2 && NSTAGE == 2 && BK == 32 && TA % 4 == 0),
                  "the two-step activation tile is built for BK = 32 and a two-deep ring");
    constexpr int PRE = NSTAGE - 1;            // stages issued ahead of the one being computed
    // counted waits need every wave to issue the same number of loads per stage
    constexpr bool UNIFORM = (TA % 4 == 0) && (TB % 4 == 0);
    constexpr int LPS = UNIFORM ? NA + NB : 0; // LDS-DMA instructions per wave per stage
    static_assert(WMP % 16 == 0 && WNC % 16 == 0, "wave tile must be a multiple of 16x16");
    static_assert(NSTAGE >= 2 && (PRE - 1) * (NA + NB) < 64, "vmcnt holds 6 bits");

    constexpr int EPI_BYTES = BM * BN * 2;
    constexpr int LDS_BYTES = NSTAGE * STAGE > EPI_BYTES ? NSTAGE * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) char lds[LDS_BYTES];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w % WAVES_M, wn = w / WAVES_M;

    // XCD-aware tile order: blocks that share blockIdx % 8 share an L2; give
    // each such group a contiguous run of tiles, channel tiles fastest, so the
    // blocks that re-read one pixel tile run on one XCD close in time.
    int tile;
    {
        const int T = a.tiles_m * a.tiles_n;
        const int q = T >> 3, r = T & 7;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile_n = tile % a.tiles_n, tile_m = tile / a.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    auto stamp = [&](int i) {
        if (a.trace && threadIdx.x == 0) a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + i] = (long long)wall_clock64();
    };
    stamp(0);
    if (a.trace && threadIdx.x == 0)
        a.trace[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 4] =
            ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4);

    const int Cin = a.C0 + a.C1;
    const int H0 = a.H >> a.up0, W0 = a.W >> a.up0;

    // ---- per-lane source bookkeeping for the LDS-DMA slots this lane fills
    const int cout_pad = a.tiles_n * BN;
    int off0[NA], off1[NA], vmask[NA], woff[NB];
    int koff;                                   // element offset of this lane's chunk in a K-tile
    {
        const int cprime = lane % CH;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int j = i * 4 + w;
            const int row = j * RPI + lane / CH;
            if (i == 0) koff = (cprime ^ chunk_swizzle<BKA>(row)) * 8;  // same for every i (see conv_layout.h)
            const int m = m0 + row;
            const bool vm = (j < TA) && (m < a.M);
            const int mm = vm ? m : 0;
            const uint32_t b = fdiv_u32((uint32_t)mm, a.divHoWo);
            const uint32_t rem = (uint32_t)mm - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t oy = fdiv_u32(rem, a.divWo);
            const uint32_t ox = rem - oy * (uint32_t)a.Wo;
            const int iyc = (int)oy * a.stride, ixc = (int)ox * a.stride;
            off0[i] = (((int)b * H0 + (iyc >> a.up0)) * W0 + (ixc >> a.up0)) * a.C0;
            off1[i] = (((int)b * a.H + iyc) * a.W + ixc) * a.C1;
            int vbits = 0;
            if (vm) {
                if (a.ksize == 1) {
                    vbits = 1;
                } else {
                    // bit kh*3 + kw: input row iyc + kh - 1 and column ixc + kw - 1 exist.  The centre
                    // always does; the column bits times the row bits spread to 0, 3, 6 gives all nine.
                    const int cm = (ixc > 0 ? 1 : 0) | 2 | (ixc + 1 < a.W ? 4 : 0);
                    const int rm = (iyc > 0 ? 1 : 0) | 8 | (iyc + 1 < a.H ? 64 : 0);
                    vbits = cm * rm;
                }
            }
            vmask[i] = vbits;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = i * 4 + w;
            const int row = j * RPIW + lane / CHW;
            const int kw = ((lane % CHW) ^ chunk_swizzle<BK>(row)) * 8;
            // weights are stored K-tile major, [K/32][cout_pad][32]: the rows of a
            // workgroup's tile are one contiguous run of whole cache lines
            woff[i] = ((kw >> 5) * cout_pad + n0 + row) * 32 + (kw & 31);
        }
    }
    uint32_t woffb[NB];                         // the same in bytes: the weight loads' per-lane part
#pragma unroll
    for (int i = 0; i < NB; ++i) woffb[i] = (uint32_t)woff[i] * 2u;

    // ---- K-steps of this workgroup: all of them, or one slice of a split reduction
    int kt0 = 0, kt1 = a.Kpad / BK;
    if (a.ksplit) {
        kt0 = (int)blockIdx.y * a.ksplit;
        kt1 = min(kt1, kt0 + a.ksplit);
    }
    // ---- uniform K-tile walker: (tap, channel) of the current tile
    int k_c = 0, k_tap = 0, k_kh = 0, k_kw = 0;
    if (kt0) {
        const int k = kt0 * BK;
        k_tap = k / Cin;
        k_c = k - k_tap * Cin;
        k_kh = k_tap / a.ksize;
        k_kw = k_tap - k_kh * a.ksize;
    }
    const int pad = a.ksize >> 1;

    // activation tile at the walker's position: uniform part, one LDS-DMA
    // instruction of this wave's NA, and the walker's advance
    const __bf16* act_base = nullptr;
    int act_delta = 0, act_tap = 0;
    bool act_second = false;
    // 1x1 layers have no border taps: a lane's byte offset into its source never changes, so the
    // walk over the channels is scalar arithmetic and the loads take the scalar-base form (rows
    // beyond M read pixel 0; their outputs are never stored)
    const bool flat = a.ksize == 1;
    uint32_t offb0[NA], offb1[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        offb0[i] = (uint32_t)(off0[i] + koff) * 2u;
        offb1[i] = (uint32_t)(off1[i] + koff) * 2u;
    }
    const __bf16* act_sbase = nullptr;
    auto act_begin = [&]() {
        act_second = k_c >= a.C0;
        act_base = act_second ? a.src1 : a.src0;
        const int Csrc = act_second ? a.C1 : a.C0;
        const int Wsrc = act_second ? a.W : W0;
        act_sbase = act_base + (act_second ? k_c - a.C0 : k_c);
        act_delta = ((k_kh - pad) * Wsrc + (k_kw - pad)) * Csrc + (act_second ? k_c - a.C0 : k_c) + koff;
        act_tap = k_tap;
        k_c += BKA;
        if (k_c >= Cin) {
            k_c = 0;
            ++k_tap;
            if (++k_kw == a.ksize) { k_kw = 0; ++k_kh; }
        }
    };
    auto act_one = [&](char* sA, int i) {
        const int j = i * 4 + w;
        if ((TA % 4 == 0) || j < TA) {
            if (flat) {
                lds_dma16_sv<0>(act_sbase, act_second ? offb1[i] : offb0[i], sA + j * 1024);
            } else {
                const bool v = (vmask[i] >> act_tap) & 1;
                const int eo = (act_second ? off1[i] : off0[i]) + act_delta;
                const __bf16* p = v ? (act_base + eo) : a.zero;
                lds_dma16(p, sA + j * 1024);
            }
        }
    };
    auto stage_act = [&](char* sA) {
        act_begin();
#pragma unroll
        for (int i = 0; i < NA; ++i) act_one(sA, i);
    };
    auto stage_wgt = [&](char* sB, int kt) {    // weight tile of K-step kt
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int j = i * 4 + w;
            if ((TB % 4 == 0) || j < TB) lds_dma16_sv<0>(a.wgt + kt * (BK * cout_pad), woffb[i], sB + j * 1024);
        }
    };
    auto stage = [&](int buf, int kt) {
        char* sA = lds + buf * STAGE;
        stage_act(sA);
        stage_wgt(sA + A_BYTES, kt);
    };

    // ---- fragment read offsets (bytes inside a tile), one per k-step
    int frag_off[KSTEPS * AMUL], frag_off_w[KSTEPS];
    {
        const int r15 = lane & 15, q = lane >> 4;
#pragma unroll
        for (int ks = 0; ks < KSTEPS * AMUL; ++ks) frag_off[ks] = slot_of<BKA>(r15, ks * 4 + q) * 16;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) frag_off_w[ks] = slot_of<BK>(r15, ks * 4 + q) * 16;
    }

    f32x4 acc[TC][TP];
#pragma unroll
    for (int tc = 0; tc < TC; ++tc)
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) acc[tc][tp] = f32x4{0.f, 0.f, 0.f, 0.f};

    // Ring of NSTAGE LDS stages.  Stage t+PRE is issued while tile t is being
    // multiplied, so PRE-1 whole stages stay in flight across each barrier.
    // The wait is counted (never vmcnt(0) in the steady state) and the barrier
    // is a bare s_barrier: a __syncthreads() here would drain the DMA queue.
    // hook(tc) runs after the MFMAs of channel group tc: the place to issue
    // loads one at a time between the matrix instructions
    auto multiply = [&](const char* sA, const char* sB, int ks_a, int ks_w, auto&& hook) {
        bf16x8 wf[TC], xf[TP];
#pragma unroll
        for (int tc = 0; tc < TC; ++tc)
            wf[tc] = *reinterpret_cast<const bf16x8*>(sB + tc * 16 * (BK * 2) + frag_off_w[ks_w]);
#pragma unroll
        for (int tp = 0; tp < TP; ++tp)
            xf[tp] = *reinterpret_cast<const bf16x8*>(sA + tp * 16 * (BKA * 2) + frag_off[ks_a]);
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
#pragma unroll
            for (int tp = 0; tp < TP; ++tp)
                acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][tp], 0, 0, 0);
            hook(tc);
        }
    };
    auto no_hook = [](int) {};
    stamp(1);
    if constexpr (AMUL == 2) {
        // LDS: two activation tiles (two K-steps each), then two weight tiles.
        // Even step: needs its activation tile and weight tile; issues the next
        // weight tile, then the next activation tile.  Odd step: needs only its
        // weight tile, so the younger activation loads stay in flight across it.
        char* const ldsW = lds + 2 * A_BYTES;
        const int a_row = (wm * WMP) * (BKA * 2), w_row = (wn * WNC) * (BK * 2);
        stage_act(lds);
        stage_wgt(ldsW, kt0);
        static_assert(NA % TC == 0 || TC % NA == 0, "activation loads spread evenly over the MFMA groups");
        for (int kt = kt0; kt < kt1; kt += 2) {
            char* const sA = lds + (((kt - kt0) >> 1) & 1) * A_BYTES;
            char* const sAn = lds + ((((kt - kt0) >> 1) & 1) ^ 1) * A_BYTES;
            const bool more = kt + 2 < kt1;
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_wgt(ldsW + B_BYTES, kt + 1);
            if (more) act_begin();
            multiply(sA + a_row, ldsW + w_row, 0, 0, [&](int tc) {
                // the next activation tile goes out a load at a time between the
                // MFMA groups: a burst right after the barrier would queue every
                // wave of the CU on the one address unit
                if (more) {
                    if constexpr (NA >= TC) {
#pragma unroll
                        for (int i = 0; i < NA / TC; ++i) act_one(sAn, tc * (NA / TC) + i);
                    } else {
                        if (tc % (TC / NA) == 0) act_one(sAn, tc / (TC / NA));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            if (more) wait_vmcnt<NA>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (more) stage_wgt(ldsW, kt + 2);
            multiply(sA + a_row, ldsW + B_BYTES + w_row, 1, 0, no_hook);
        }
    } else {
#pragma unroll
        for (int s = 0; s < PRE; ++s)
            if (kt0 + s < kt1) stage(s, kt0 + s);
        int cur = 0, nxt = PRE % NSTAGE;
        for (int kt = kt0; kt < kt1; ++kt) {
            wait_stages<PRE - 1, LPS>(min(PRE - 1, kt1 - 1 - kt));  // this wave's part of tile kt landed
            __builtin_amdgcn_s_barrier();                          // everyone's did; tile kt-1 fully read
            asm volatile("" ::: "memory");
            if (kt + PRE < kt1) stage(nxt, kt + PRE);
            const char* sA = lds + cur * STAGE + (wm * WMP) * (BK * 2);
            const char* sB = lds + cur * STAGE + A_BYTES + (wn * WNC) * (BK * 2);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) multiply(sA, sB, ks, ks, no_hook);
            cur = (cur + 1 == NSTAGE) ? 0 : cur + 1;
            nxt = (nxt + 1 == NSTAGE) ? 0 : nxt + 1;
        }
    }

    // ---- epilogue: bias, LeakyReLU, residual, store
    stamp(2);
    const int px = lane & 15, cq = (lane >> 4) * 4;
    if (a.dst_fp32 && !a.res) {
        // fp32 output (the detection convolutions, 255 channels: rows of 1020
        // bytes).  The tile goes through LDS 32 channels at a time so that the
        // stores are runs of 256 contiguous bytes per pixel, and the head decode
        // (yolo_v3.predict), when asked for, is applied on the way out.
        // 64 channels per pass where the operand stages leave that much LDS, else 32
        constexpr int PC = (BN >= 64 && LDS_BYTES >= BM * (64 + 4) * 4 + BM * 16) ? 64 : 32;
        constexpr int PITCH = PC + 4;                      // +4 floats: the 16 pixel rows of a b128 write hit distinct banks
        float* const tile = reinterpret_cast<float*>(lds);
        float* const geo = tile + BM * PITCH;              // per tile pixel: destination offset, gx, gy
        __syncthreads();                                   // operand stages are dead from here on
        if (threadIdx.x < BM) {
            const int m = m0 + threadIdx.x;
            long long off = -1;
            float gx = 0.f, gy = 0.f;
            if (m < a.M) {
                if (a.dec_D) {
                    const uint32_t b = fdiv_u32((uint32_t)m, a.divHoWo);
                    const uint32_t pix = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
                    const uint32_t oy = fdiv_u32(pix, a.divWo);
                    gx = (float)(int)(pix - oy * (uint32_t)a.Wo);
                    gy = (float)(int)oy;
                    off = ((long long)b * a.dec_rows + a.dec_row_offset) * a.dec_D + (long long)pix * a.Cout;
                } else {
                    off = (long long)m * a.Cout;
                }
            }
            reinterpret_cast<long long*>(geo)[threadIdx.x] = off;     // 8-byte slots first ...
            geo[2 * BM + threadIdx.x] = gx;                            // ... then gx, gy
            geo[3 * BM + threadIdx.x] = gy;
        }
        static_assert(BM * PITCH * 4 + BM * 16 <= LDS_BYTES, "fp32 epilogue tile and pixel table do not fit");
        const int ch_l = threadIdx.x & (PC - 1), sub = threadIdx.x / PC;
        constexpr int SUBS = 256 / PC;
#pragma unroll 1
        for (int p = 0; p < BN / PC; ++p) {
            const int c_lo = p * PC;
            // every wave whose channels fall into this pass parks them (wave-uniform tests)
            const int wbase = wn * WNC - c_lo;                 // tile column of this wave's first channel
            if (wbase + WNC > 0 && wbase < PC) {
#pragma unroll
                for (int tp = 0; tp < TP; ++tp) {
                    const int slot = wm * WMP + tp * 16 + px;
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) {
                        const int col = wbase + tc * 16;
                        if (col < 0 || col >= PC) continue;
                        f32x4 v = acc[tc][tp];
                        f32x4 bs = f32x4{0.f, 0.f, 0.f, 0.f};     // partial sums of a split reduction stay raw
                        if (!a.ksplit) bs = *reinterpret_cast<const f32x4*>(a.bias + n0 + c_lo + col + cq);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = v[r] + bs[r];
                            if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                            v[r] = x;
                        }
                        *reinterpret_cast<f32x4*>(tile + slot * PITCH + col + cq) = v;
                    }
                }
            }
            __syncthreads();
            const int ch = n0 + c_lo + ch_l;
            if (ch < a.Cout) {
                int d = 0;
                float anc = 0.f;
                if (a.dec_D) {
                    const int ai = (int)fdiv_u32((uint32_t)ch, a.divD);
                    d = ch - ai * a.dec_D;
                    anc = d == 2 ? a.dec_aw[ai] : a.dec_ah[ai];
                }
                float* const dstf = reinterpret_cast<float*>(a.dst) +
                                    (a.ksplit ? (size_t)blockIdx.y * a.M * a.Cout : (size_t)0);
#pragma unroll 4
                for (int i = 0; i < BM / SUBS; ++i) {
                    const int slot = i * SUBS + sub;
                    const long long off = reinterpret_cast<const long long*>(geo)[slot];
                    if (off >= 0) {
                        float v = tile[slot * PITCH + ch_l];
                        if (a.dec_D) v = decode_value(v, d, geo[2 * BM + slot], geo[3 * BM + slot], anc, a.dec_stride);
                        dstf[off + ch] = v;
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    if (!a.dst_fp32 && (a.Cout & 7) == 0) {
        auto pix = [&](int slot) -> long long { return m0 + slot < a.M ? (long long)(m0 + slot) : -1; };
        __syncthreads();                       // operand stages are dead from here on
        constexpr int W2_BYTES = 4 * 64 * 32 * 2;
        char* const w2l = lds + EPI_BYTES;     // TAIL: the 1x1 layer's weights, later its output tile
        if constexpr (TAIL) {
            static_assert(BM == 128 && BN == 128 && EPI_BYTES + W2_BYTES <= LDS_BYTES, "tail: tile and weights share the LDS");
#pragma unroll
            for (int i = 0; i < W2_BYTES / 1024 / 4; ++i)
                lds_dma16(a.w2 + ((w + 4 * i) * 64 + lane) * 8, w2l + (w + 4 * i) * 1024);
        }
        if (a.res) {
            tile_load_residual<BM, BN, 256>(lds, a.res, a.zero, a.Cout, n0, w, lane, pix);
            __syncthreads();
        }
        stamp(5);
#pragma unroll
        for (int tp = 0; tp < TP; ++tp) {
            const int slot = wm * WMP + tp * 16 + px;
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                const int chl = wn * WNC + tc * 16 + cq;
                const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias + n0 + chl);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[tc][tp][r] + bs[r];
                    if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[r] = x;
                }
                tile_put<BN>(lds, slot, chl, v, a.res != nullptr);
            }
        }
        if constexpr (TAIL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        stamp(6);
        if constexpr (!TAIL) {
            tile_store<BM, BN, 256>(lds, (__bf16*)a.dst, a.Cout, n0, pix);
        } else {
            // wave w: pixel groups 2w and 2w + 1, all four channel groups of the 1x1 layer
            const int q = lane >> 4;
            // (its bias is asked for ahead of this layer's stores: a load is answered in order with the
            // stores issued before it and would wait for their acknowledgements)
            f32x4 b2v[4];
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) b2v[cg] = *reinterpret_cast<const f32x4*>(a.b2 + cg * 16 + cq);
            f32x4 acc2[2][4];
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) acc2[g][cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 yf[2], wq[4];
#pragma unroll
                for (int g = 0; g < 2; ++g)
                    yf[g] = *reinterpret_cast<const bf16x8*>(lds + tile_addr<BN>((2 * w + g) * 16 + px, ks * 32 + q * 8));
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    wq[cg] = *reinterpret_cast<const bf16x8*>(w2l + ((ks * 64 + cg * 16 + px) * 32 + q * 8) * 2);
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int cg = 0; cg < 4; ++cg)
                        acc2[g][cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[cg], yf[g], acc2[g][cg], 0, 0, 0);
            }
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) asm volatile("" : "+v"(b2v[cg]));
            tile_store<BM, BN, 256>(lds, (__bf16*)a.dst, a.Cout, n0, pix);
            __syncthreads();                   // every wave has read the weights: their LDS takes the output tile
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int cg = 0; cg < 4; ++cg) {
                    const int chl = cg * 16 + cq;
                    const f32x4 bs = b2v[cg];
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc2[g][cg][r] + bs[r];
                        if (a.leaky2) x = x > 0.f ? x : 0.1f * x;
                        v[r] = x;
                    }
                    tile_put<64>(w2l, (2 * w + g) * 16 + px, chl, v, false);
                }
            __syncthreads();
            tile_store<BM, 64, 256>(w2l, a.dst2, 64, 0, pix);
        }
        stamp(3);
        return;
    }
    const bool vec = (a.Cout & 3) == 0;
#pragma unroll
    for (int tp = 0; tp < TP; ++tp) {
        const int m = m0 + wm * WMP + tp * 16 + px;
        if (m >= a.M) continue;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int ch = n0 + wn * WNC + tc * 16 + cq;
            if (ch >= a.Cout) continue;
            const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias + ch);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[tc][tp][r] + bs[r];
                if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                v[r] = x;
            }
            const size_t o = (size_t)m * a.Cout + ch;
            if (vec) {
                if (a.res) {
                    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(a.res + o);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += (float)rr[r];
                }
                if (a.dst_fp32) {
                    *reinterpret_cast<f32x4*>((float*)a.dst + o) = f32x4{v[0], v[1], v[2], v[3]};
                } else {
                    bf16x4 pk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) pk[r] = (__bf16)v[r];
                    *reinterpret_cast<bf16x4*>((__bf16*)a.dst + o) = pk;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (ch + r >= a.Cout) break;
                    float x = v[r];
                    if (a.res) x += (float)a.res[o + r];
                    if (a.dst_fp32) ((float*)a.dst)[o + r] = x;
                    else ((__bf16*)a.dst)[o + r] = (__bf16)x;
                }
            }
        }
    }
}

// ---------------------------------------------------------------- 3x3 with patch reuse
// Stride-1 3x3 layers re-read every input pixel nine times when each tap is
// staged as its own im2col tile, and the CU's vector-memory path (64 B/clk, and
// every 1 KiB LDS-DMA instruction costs its wave 60-180 cycles of issue), not
// the MFMA pipe, becomes the limit.  This kernel stages, per 32-channel chunk,
// the (13+2)x(13+2) input PATCH of a 13x13 output tile once and feeds all nine
// taps from it, so only the weights are streamed per tap:
//   LDS-DMA instructions per wave per chunk = 4 (patch) + 9 x 2 (weights) = 22
//   for 198 MFMAs, against 1 per 4 MFMAs in the im2col kernel.
//
// Workgroup = one 13x13 spatial tile x 128 output channels, 4 waves; wave cg
// owns all 11 pixel groups (176 slots for the 169 pixels) for channels
// cg*32..+32 (88 accumulator registers).
//   patch   LDS planes [q = 16-byte chunk 0..3][row 0..15][16 slots], slot of
//           patch pixel (py, px) = (px + 13*py) mod 16 in row py: for every tap
//           the 16 lanes of a pixel group then read 16 distinct bank slots.
//           Wave w fills plane w with four LDS-DMA instructions (4 rows each),
//           double buffered, issued one per tap during the previous chunk.
//   weights LDS-DMA ring of two (chunk, tap) stages = 128 channels x 32 k,
//           XOR-swizzled rows as in conv_layout.h, counted vmcnt waits.
// 48 KiB of LDS and <= 168 registers: three workgroups per CU (four waves); the
// eight-wave form takes 76 KiB, two workgroups per CU (see the LDS map in the kernel).
// 13 divides every feature-map side of a 416-pixel input (13 * 2^n).
//
// The tile is TH rows x TW columns of output pixels (TW + 2 <= 16: a patch row fits the 16
// slots of a plane row).  13 x 13 serves the 13 * 2^n maps of a 416-pixel input; 19 x 10
// serves the 19 * 2^n maps of a 608-pixel input (190 pixels = twelve groups of 16 less two
// slots, six groups per wave pair half; the maps' widths are not multiples of 10, so the last
// tile of a row hangs over the edge: its patch reads beyond the map come from the zero page
// like any padding, its outputs beyond the map are not stored -- 19 of 20 columns are real).
constexpr int P3_TS = 13;                  // spatial tile side of the default shape

struct Conv3Args {
    const __bf16* src;
    const __bf16* wgt;
    const float* bias;
    const __bf16* res;
    void* dst;
    const __bf16* zero;
    int C, H, W, Cout, K;
    int leaky, dst_fp32;
    int tiles_y, tiles_x, ntiles;      // spatial tiles per image (y, x) and in the batch
    int tiles_n;                       // channel tiles
    long long* trace;                  // development aid, always null unless the file is built with -DYOLO_AMD_DEV_TRACE
                                       // (then YOLO_AMD_CONV3_TRACE=<device address>): per workgroup eight words:
                                       // 100 MHz timestamps (start, K loop entered, K loop done, stores issued), XCC_ID << 32 | HW_ID,
                                       // then two more timestamps (residual tile landed, output tile complete in LDS)
    int prio;                          // raise the wave priority over the MFMA groups (about 1 %; YOLO_AMD_CONV3_PRIO=0 turns it off)
    // TAIL kernels: the 1x1 layer (128 -> 64) that reads this layer's output (tiles_n = 1)
    const __bf16* w2;      // [4][64][32] bf16 (K-tile major)
    const float* b2;
    __bf16* dst2;
    int leaky2;
};

// WAVES = 4: wave cg owns all 11 pixel groups (three workgroups per CU).
// WAVES = 8: waves cg and cg + 4 split them 6 + 5, 48 accumulator registers each,
// so that two workgroups per CU still give four waves per SIMD -- for the maps
// whose 13x13 tiles number no more than two per CU.
// STG = 3 (eight waves): a weight stage is a whole kernel row (three taps, 24 KiB; the ring
// holds two) and the workgroup meets at a barrier once per row instead of once per tap.
// TC = channel groups of 16 per wave: 2 (workgroup = 128 channels) or 4 (256 channels: a wave
// then issues 24 MFMAs per tap for 6 + 4 fragment reads instead of 12 for 6 + 2; the weight
// ring grows to 96 KiB, one workgroup of eight waves per CU, up to 256 registers).
// VAR (private-ring form only, YOLO_AMD_CONV3_VAR): ABLATIONS that compute garbage, for timing only -- where
// does a tap's time go: 0x100 no weight loads, 0x200 no fragment reads (the matrix instructions run on
// what the registers held before), 0x400 no patch loads, 0x800 no matrix instructions.
template <int WAVES, int STG = 1, int TC = 2, int TH = P3_TS, int TW = P3_TS, int RING = 3, int VAR = 0>
__global__ void __launch_bounds__(64 * WAVES)
    __attribute__((amdgpu_waves_per_eu((TC == 4 || (STG == 9 && RING == 4)) ? 2 : (WAVES == 4 ? 3 : 4),
                                       (TC == 4 || (STG == 9 && RING == 4)) ? 2 : (WAVES == 4 ? 3 : 4))))
conv3x3_patch_kernel(const Conv3Args a) {
    static_assert(STG == 1 || ((STG == 3 || STG == 9) && WAVES == 8), "row-deep weight stages exist for the eight-wave form");
    static_assert(TC == 2 || (TC == 4 && WAVES == 8 && (STG == 3 || STG == 9)),
                  "the 256-channel tile exists for eight waves, row-deep stages or private rings");
    // STG = 9: every wave stages the weights of ITS 32 channels itself, into a ring of its own
    // (three taps deep, 2 KiB per tap; the two waves of a channel group each hold a copy), so
    // weights never wait for another wave: counted vmcnt waits only, and the workgroup meets
    // at a barrier once per 32-channel chunk (nine taps), for the shared input patch.
    constexpr bool PRIV = STG == 9;
    constexpr int PRIV_R = RING;                   // taps held per private ring (2, 3 or 4)
    static_assert(!PRIV || (PRIV_R >= 2 && PRIV_R <= 4), "private ring depth");
    constexpr int CT = 64 * TC;                    // channels per workgroup
    constexpr int WSTAGE = CT * 64;                // bytes of one (chunk, tap) weight stage: CT channels x 32 bf16
    static_assert(TW + 2 <= 16, "a patch row must fit the 16 slots of a plane row");
    constexpr int P3_NPIX = TH * TW;               // output pixels per tile
    constexpr int P3_TP = (P3_NPIX + 15) / 16;     // pixel groups of 16
    constexpr int PROWS = (TH + 2 + 3) / 4 * 4;    // patch rows, in whole LDS-DMA instructions of four rows
    constexpr int P3_PLANE = PROWS * 256;          // bytes per patch plane (rows x 16 slots x 16 B)
    constexpr int P3_PATCH = 4 * P3_PLANE;         // one buffer: four 16-byte chunks of the 32 channels
    constexpr int NH = WAVES / 4;                  // waves per channel group
    constexpr int NTP = NH == 1 ? P3_TP : (P3_TP + 1) / 2;   // pixel groups per wave (upper bound)
    constexpr int PP = (PROWS / 4) / NH;           // patch LDS-DMA instructions per wave per chunk
    static_assert((PROWS / 4) % NH == 0 && PP <= 4 && (NH == 2 ? NTP <= 6 : P3_TP == 11),
                  "tile shape not laid out for this wave count");
    constexpr int WP = (CT / 16) / WAVES;          // weight LDS-DMA instructions per wave per stage
    // LDS: weight ring, then the two patch buffers X and Y.  The eight-wave form adds
    // 28 KiB behind Y: Y + that is where the residual tile (176 rows x 256 B = 44 KiB)
    // is parked by LDS-DMA during the LAST chunk, which is made to read buffer X
    // (chunks alternate X/Y counted from the end), so that the residual round trip to
    // memory runs under the matrix work instead of after it.
    constexpr bool RPRE = WAVES == 8 && STG == 1 && TH == 13 && TW == 13;
    constexpr int TILE_BYTES = P3_TP * 16 * CT * 2;
    constexpr int PSLOT = TC * 1024;               // bytes of one tap in a private ring: 16 TC channels x 64 B
    constexpr int RING_BYTES = PRIV ? WAVES * PRIV_R * PSLOT : 2 * STG * WSTAGE;   // two stages of STG taps
    __shared__ __attribute__((aligned(1024))) char lds_all[RING_BYTES + 2 * P3_PATCH + (RPRE ? TILE_BYTES - P3_PATCH : 0)];
    static_assert(sizeof(lds_all) >= TILE_BYTES, "the epilogue tile reuses this memory");
    char* const wring = lds_all;
    char* const lds = lds_all + RING_BYTES;                    // patch buffers
    char* const rtile = RPRE ? lds + P3_PATCH : lds_all;       // epilogue tile (and parked residual)

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto stamp = [&](int i) {
        if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + i] = (long long)wall_clock64();
    };
    stamp(0);
    if (a.trace && threadIdx.x == 0)       // where it runs: HW_ID (CU, SE, ...) and XCC_ID
        a.trace[(size_t)blockIdx.x * 8 + 4] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    const int cg = w & 3, half = w >> 2;           // channel group (= patch plane), pixel-group half
    const int tp0 = half * NTP;                    // first pixel group of this wave
    const int ntp = NH == 1 ? P3_TP : (half == 0 ? NTP : P3_TP - NTP);

    int tile;
    {
        const int T = a.ntiles * a.tiles_n;
        const int q8 = T >> 3, r8 = T & 7;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int tile_n = tile % a.tiles_n, gt = tile / a.tiles_n;
    const int n0 = tile_n * CT;
    const int per_img = a.tiles_y * a.tiles_x;
    const int cb = gt / per_img;
    const int oy0 = ((gt - cb * per_img) / a.tiles_x) * TH;
    const int ox0 = ((gt - cb * per_img) % a.tiles_x) * TW;

    const int r = lane & 15, q = lane >> 4;
    int tyb[NTP];                              // byte offset of (plane q, patch row ty) per pixel group
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) {
        const int p = min((tp0 + tp) * 16 + r, P3_NPIX - 1);
        tyb[tp] = q * P3_PLANE + (p / TW) * 256;
    }
    const int wfrag = (cg * 16 * TC) * 64 + slot_of<32>(r, q) * 16;

    const int NCH = a.C / 32;
    // ---- patch fill: 16 LDS-DMA instructions of 4 rows each; this wave's are
    // rows 4i..4i+3 of plane cg for i = half*PP .. half*PP + PP - 1
    int poff[PP];
#pragma unroll
    for (int ii = 0; ii < PP; ++ii) {
        const int py = 4 * (half * PP + ii) + q, slot = r;
        const int px = (slot - TW * py) & 15;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        poff[ii] = (py < TH + 2 && px < TW + 2 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                       ? ((cb * a.H + iy) * a.W + ix) * a.C + cg * 8
                       : -1;
    }
    auto issue_patch = [&](int chunk, int ii) {            // ii: wave-uniform
        // (selected with compares: a run-time index would move the array out of registers)
        int po;
        if constexpr (PP == 4) po = ii == 0 ? poff[0] : ii == 1 ? poff[1] : ii == 2 ? poff[2] : poff[3];
        else if constexpr (PP == 3) po = ii == 0 ? poff[0] : ii == 1 ? poff[1] : poff[2];
        else po = ii == 0 ? poff[0] : poff[1];
        const __bf16* pp = po >= 0 ? a.src + (po + chunk * 32) : a.zero;
        lds_dma16(pp, lds + ((NCH - 1 - chunk) & 1) * P3_PATCH + cg * P3_PLANE + (half * PP + ii) * 1024);
    };
    // ---- weight fill: 8 LDS-DMA instructions per (chunk, tap) stage, WP per wave
    const int cout_pad = a.tiles_n * CT;
    int woff[WP];
    {
        const int qw = (lane & 3) ^ chunk_swizzle<32>(lane >> 2);
#pragma unroll
        for (int i = 0; i < WP; ++i) woff[i] = (n0 + (w * WP + i) * 16 + (lane >> 2)) * 32 + qw * 8;
    }
    const int wstep_tap = (a.C / 32) * cout_pad * 32;     // [K/32][cout_pad][32]: one tap further
    const int wstep_chunk = cout_pad * 32;
    auto issue_w = [&](int chunk, int tap, int buf) {
        const int delta = tap * wstep_tap + chunk * wstep_chunk;
#pragma unroll
        for (int i = 0; i < WP; ++i)
            lds_dma16(a.wgt + (woff[i] + delta), wring + buf * WSTAGE + (w * WP + i) * 1024);
    };

    // the accumulators start from the bias (one add less per output in the epilogue, and the
    // bias is fetched under the first loads instead of after the last MFMA)
    f32x4 acc[TC][NTP];
#pragma unroll
    for (int tc = 0; tc < TC; ++tc) {
        const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias + n0 + cg * 16 * TC + tc * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) acc[tc][tp] = bs;
    }

#pragma unroll
    for (int ii = 0; ii < PP; ++ii) issue_patch(0, ii);
    if constexpr (!PRIV) issue_w(0, 0, 0);
    stamp(1);
    bool patch_behind = false;                 // a patch load was issued after the weights now awaited
    // the vector epilogue will want the residual tile in LDS (all 128 channels of the tile exist)
    const bool res_pre = RPRE && (a.prio & 2) && a.res && !a.dst_fp32 && (a.Cout & 7) == 0 && n0 + CT <= a.Cout;
    int buf = 0;
    // Private rings three taps deep: each wave fetches ITS part of the residual tile (<= 6 pixel
    // groups x 32 channels = six 1 KiB pieces) itself, into memory nobody else reads, so that
    // nothing waits for a barrier, and early, during the last 32-channel chunk: pieces 0-1 at
    // its first tap into this wave's 2 KiB of the idle patch buffer, pieces 2-3 and 4-5 at the
    // last two taps into the ring slots those taps no longer fetch weights for.  Layout of a
    // piece: 16 pixels x 64 B, the 16-byte chunk c of pixel p at position c ^ ((p >> 2) & 3)
    // (conflict-free 8-byte reads).
    constexpr bool OWN_RES = PRIV && TC == 2 && PRIV_R == 3 && 2 * P3_PATCH >= P3_PATCH + WAVES * 2048 && NTP <= 6;
    char* const myring = wring + w * (PRIV_R * PSLOT);
    const bool own_res = OWN_RES && a.res && !a.dst_fp32 && (a.Cout & 7) == 0;
    auto res_at = [&](int tp) -> char* {
        return tp < 2 ? lds + P3_PATCH + w * 2048 + tp * 1024 : myring + ((tp - 2) >> 1) * 2048 + (tp & 1) * 1024;
    };
    auto res_piece = [&](int tp) {
        // (the addresses are worked out where the piece is issued, three times per tile: hoisted
        // out of the K loop they would cost registers the loop does not have)
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int slot = (tp0 + tp) * 16 + (ln >> 2);
        const int ty = slot / TW, tx = slot - ty * TW;
        const bool okp = tp < ntp && slot < P3_NPIX && oy0 + ty < a.H && ox0 + tx < a.W;
        const int ch = n0 + cg * 32 + (((ln & 3) ^ ((ln >> 4) & 3)) * 8);
        const __bf16* src = (okp && ch + 8 <= a.Cout)
                                ? a.res + ((((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx) * a.Cout + ch)
                                : a.zero;
        lds_dma16(src, res_at(tp));
    };
    if constexpr (PRIV) {
        static_assert(!PRIV || PP <= 3, "private weight rings: patch pieces per wave and chunk");
        // this lane's 16 bytes inside a 16-channel piece (1 KiB, [channel][32] with swizzled chunks);
        // the piece's own address is wave-uniform
        const uint32_t wlane = (uint32_t)(((lane >> 2) * 32 + ((lane & 3) ^ chunk_swizzle<32>(lane >> 2)) * 8) * 2);
        const int wbase = (n0 + cg * 16 * TC) * 32;
        auto issue_wp = [&](int chunk, int tap, int slot) {
            const __bf16* const sb = a.wgt + (wbase + tap * wstep_tap + chunk * wstep_chunk);
            char* const dst = myring + slot * PSLOT;
            static_assert(TC == 2 || TC == 4, "pieces per tap");
            lds_dma16_sv<0>(sb, wlane, dst);               // piece i: 1 KiB further in memory and in the slot
            lds_dma16_sv<1024>(sb, wlane, dst);
            if constexpr (TC == 4) {
                lds_dma16_sv<2048>(sb, wlane, dst);
                lds_dma16_sv<3072>(sb, wlane, dst);
            }
        };
        const int wfragp = slot_of<32>(r, q) * 16;
#pragma unroll
        for (int t = 0; t < PRIV_R - 1; ++t) issue_wp(0, t, t);
        int slot0 = 0;                             // ring slot of this chunk's tap 0
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const char* pb = lds + ((NCH - 1 - chunk) & 1) * P3_PATCH;
            const bool more_chunks = chunk + 1 < NCH;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                constexpr int dummy = 0; (void)dummy;
                const int kh = tap / 3, kw = tap - 3 * (tap / 3);
                // Loads are answered in issue order.  Younger than this tap's weights (issued
                // PRIV_R - 1 taps ago, the last thing that tap issued) is what the taps since
                // then issued: two weight pieces each while there were taps left to fetch for,
                // and one patch piece during the first PP taps of a chunk that has a successor.
                // (In the last chunk the residual pieces take the place of the patch pieces:
                // two each at taps 0, 7 and 8.)
                {
                    int n_more = 0, n_last = 0, n_res = 0;
#pragma unroll
                    for (int u = tap - PRIV_R + 2; u <= tap - 1; ++u) {
                        n_more += TC + ((u >= 0 && u < PP) ? 1 : 0);
                        n_last += (u + PRIV_R - 1 <= 8) ? TC : 0;
                        n_res += ((u + PRIV_R - 1 <= 8) ? TC : 0) + ((u == 0 || u == 7) ? 2 : 0);
                    }
                    wait_vmcnt_upto(more_chunks ? n_more : (own_res ? n_res : n_last));
                }
                if (tap == 0) {
                    // every wave's pieces of this chunk's patch are older than the weights it
                    // has just waited for; behind the barrier nobody reads the other patch
                    // buffer any more, so the next chunk's pieces may overwrite it
                    __builtin_amdgcn_s_barrier();
                }
                asm volatile("" ::: "memory");
                // this tap's loads: a piece of the next chunk's patch (or of the residual tile), and the
                // weights of the tap PRIV_R - 1 ahead
                auto tap_loads = [&]() {
                    if constexpr ((VAR & 0x400) == 0) {
                        if (tap < PP && more_chunks) issue_patch(chunk + 1, tap);
                    }
                    if constexpr (OWN_RES) {
                        if (own_res && !more_chunks && (tap == 0 || tap == 7 || tap == 8)) {
                            const int first = tap == 0 ? 0 : (tap == 7 ? 2 : 4);
                            res_piece(first);
                            res_piece(first + 1);
                        }
                    }
                    // slot of tap t of this chunk = (slot0 + t) mod PRIV_R
                    int sl = slot0 + (tap + PRIV_R - 1) % PRIV_R;
                    if (sl >= PRIV_R) sl -= PRIV_R;
                    if constexpr ((VAR & 0x100) == 0) {
                        if (tap + PRIV_R - 1 < 9) issue_wp(chunk, tap + PRIV_R - 1, sl);
                        else if (more_chunks) issue_wp(chunk + 1, tap + PRIV_R - 1 - 9, sl);
                    }
                };
                tap_loads();
                const int colb = ((r + kw + TW * kh) & 15) * 16 + kh * 256;
                int cur = slot0 + tap % PRIV_R;
                if (cur >= PRIV_R) cur -= PRIV_R;
                const char* ws = myring + cur * PSLOT + wfragp;
                bf16x8 wf[TC];
                bf16x8 xf[6];
                if constexpr ((VAR & 0x200) == 0) {
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) wf[tc] = *reinterpret_cast<const bf16x8*>(ws + tc * 1024);
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) xf[tp] = *reinterpret_cast<const bf16x8*>(pb + tyb[tp] + colb);
                } else {
                    // timing ablation: operands that look like data (not constants the compiler could fold)
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) {
                        u32x4 v = u32x4{0x3f803f80u ^ (uint32_t)(lane * 0x01010101u), 0x3fa0bf90u + (uint32_t)tc,
                                        0xbf853f33u ^ (uint32_t)(lane << 3), 0x3f4cbf19u};
                        asm volatile("" : "+v"(v));
                        wf[tc] = __builtin_bit_cast(bf16x8, v);
                    }
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) {
                        u32x4 v = u32x4{0x3f913f22u ^ (uint32_t)(lane * 0x00030005u), 0xbf6a3f47u + (uint32_t)tp,
                                        0x3f0dbfa1u ^ (uint32_t)(lane << 5), 0xbf2b3f7eu};
                        asm volatile("" : "+v"(v));
                        xf[tp] = __builtin_bit_cast(bf16x8, v);
                    }
                }
                __builtin_amdgcn_s_setprio(2);
                if constexpr ((VAR & 0x800) == 0) {
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) {
                        if (tp >= ntp) break;
#pragma unroll
                        for (int tc = 0; tc < TC; ++tc)
                            acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][tp], 0, 0, 0);
                    }
                } else {
                    // keep the fragment reads alive without the matrix work
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) asm volatile("" ::"v"(wf[tc]));
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) asm volatile("" ::"v"(xf[tp]));
                }
                __builtin_amdgcn_s_setprio(0);
            }
            slot0 += 9 % PRIV_R;
            if (slot0 >= PRIV_R) slot0 -= PRIV_R;
        }
    } else if constexpr (STG == 3) {
        // Ring slots 0-2 and 3-5 hold the three taps of the current and of the next
        // kernel row.  At the row's barrier everything issued so far has landed (full
        // wait); during the row each tap issues the matching tap of the next row into
        // the other half, whose last readers passed this barrier.
        issue_w(0, 1, 1);
        issue_w(0, 2, 2);
        int base = 0;
        for (int chunk = 0; chunk < NCH; ++chunk) {
            const char* pb = lds + ((NCH - 1 - chunk) & 1) * P3_PATCH;
            const bool more_chunks = chunk + 1 < NCH;
#pragma unroll 1
            for (int kh = 0; kh < 3; ++kh) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                const bool next_row = kh < 2 || more_chunks;
                const int nchunk = kh < 2 ? chunk : chunk + 1, ntap0 = kh < 2 ? (kh + 1) * 3 : 0;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    if (next_row) issue_w(nchunk, ntap0 + kw, (base ^ 3) + kw);
                    if (kh == 0 && kw < PP && more_chunks) issue_patch(chunk + 1, kw);
                    const int colb = ((r + kw + TW * kh) & 15) * 16 + kh * 256;
                    const char* ws = wring + (base + kw) * WSTAGE + wfrag;
                    bf16x8 wf[TC], xf[6];
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc) wf[tc] = *reinterpret_cast<const bf16x8*>(ws + tc * 1024);
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) xf[tp] = *reinterpret_cast<const bf16x8*>(pb + tyb[tp] + colb);
                    if (a.prio & 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
                    for (int tp = 0; tp < 6; ++tp) {
                        if (tp >= ntp) break;                           // the second half may have a group less
#pragma unroll
                        for (int tc = 0; tc < TC; ++tc)
                            acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][tp], 0, 0, 0);
                    }
                    if (a.prio & 1) __builtin_amdgcn_s_setprio(0);
                }
                base ^= 3;
            }
        }
    } else
    for (int chunk = 0; chunk < NCH; ++chunk) {
        const char* pb = lds + ((NCH - 1 - chunk) & 1) * P3_PATCH;
        const bool more_chunks = chunk + 1 < NCH;
        int kh = 0, kw = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            if (patch_behind) wait_vmcnt<1>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // next stage's weights first (they are awaited first), then one
            // instruction of the next chunk's patch
            if (tap < 8) issue_w(chunk, tap + 1, buf ^ 1);
            else if (more_chunks) issue_w(chunk + 1, 0, buf ^ 1);
            patch_behind = tap < PP && more_chunks;
            if (patch_behind) issue_patch(chunk + 1, tap);
            if constexpr (RPRE) {
                // last chunk: no patch to fetch any more; instruction w + 8 * tap of the
                // 44 that park the residual tile goes out instead (one per wave and tap)
                if (res_pre && !more_chunks && w + 8 * tap < TILE_BYTES / 1024) {
                    const int j = w + 8 * tap;
                    const int row = j * 4 + (lane >> 4);
                    const int c = (lane & 15) ^ (row & 15);
                    const __bf16* src = a.zero;
                    if (row < P3_NPIX) {
                        const int ty = row / TW, tx = row - ty * TW;
                        src = a.res + (((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx) * a.Cout + n0 + c * 8;
                    }
                    lds_dma16(src, rtile + j * 1024);
                    patch_behind = true;
                }
            }

            const int colb = ((r + kw + TW * kh) & 15) * 16 + kh * 256;
            const char* ws = wring + buf * WSTAGE + wfrag;
            bf16x8 wf[TC];
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) wf[tc] = *reinterpret_cast<const bf16x8*>(ws + tc * 1024);
            {
                bf16x8 xf[6];
#pragma unroll
                for (int tp = 0; tp < 6; ++tp) xf[tp] = *reinterpret_cast<const bf16x8*>(pb + tyb[tp] + colb);
                if (a.prio & 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int tp = 0; tp < 6; ++tp) {
                    if (NH == 2 && tp >= ntp) break;                    // the second half may have a group less
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc)
                        acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][tp], 0, 0, 0);
                }
            }
            if constexpr (NH == 1) {
                __builtin_amdgcn_sched_barrier(0);
                bf16x8 xf[5];
#pragma unroll
                for (int tp = 0; tp < 5; ++tp) xf[tp] = *reinterpret_cast<const bf16x8*>(pb + tyb[6 + tp] + colb);
#pragma unroll
                for (int tp = 0; tp < 5; ++tp)
#pragma unroll
                    for (int tc = 0; tc < TC; ++tc)
                        acc[tc][6 + tp] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][6 + tp], 0, 0, 0);
            }
            if (a.prio & 1) __builtin_amdgcn_s_setprio(0);
            buf ^= 1;
            if (++kw == 3) { kw = 0; ++kh; }
        }
    }

    // ---- epilogue
    stamp(2);
    const int cq = q * 4;
    if (!a.dst_fp32 && (a.Cout & 7) == 0) {
        constexpr int NSLOT = P3_TP * 16;
        auto pix = [&](int slot) -> long long {
            if (slot >= P3_NPIX) return -1;
            const int ty = slot / TW, tx = slot - ty * TW;
            if (oy0 + ty >= a.H || ox0 + tx >= a.W) return -1;         // a tile that hangs over the edge
            return ((long long)cb * a.H + oy0 + ty) * a.W + ox0 + tx;
        };
        // (a wave's own part of the residual tile is already on its way: see own_res above)
        const float slope = a.leaky ? 0.1f : 1.0f;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[tc][tp][e] = leaky_max(acc[tc][tp][e], slope);
            }
        }
        if constexpr (OWN_RES) {
            // The wave's outputs take the place of its residual pieces (same 16 pixels x 64 B
            // layout, in the wave's own LDS), are read back as whole 16-byte chunks and stored
            // by the wave itself: no workgroup barrier anywhere in this epilogue, every wave
            // leaves as soon as its own pixels are out.  A 64-byte run per pixel and wave; the
            // wave of the neighbouring channel group writes the other half of the line.
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
                if (own_res) {                   // the pieces went out in pairs, in this order
                    if (tp == 0) wait_vmcnt<4>();
                    if (tp == 2) wait_vmcnt<2>();
                    if (tp == 4) wait_vmcnt<0>();
                }
                if (tp >= ntp) break;
#pragma unroll
                for (int tc = 0; tc < TC; ++tc) {
                    const int c = (tc * 2 + (q >> 1)) ^ ((r >> 2) & 3);
                    bf16x4* p = reinterpret_cast<bf16x4*>(res_at(tp) + r * 64 + c * 16 + (q & 1) * 8);
                    float v[4] = {acc[tc][tp][0], acc[tc][tp][1], acc[tc][tp][2], acc[tc][tp][3]};
                    if (own_res) {
                        const bf16x4 rr = *p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
                    }
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)v[e];
                    *p = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stamp(6);
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
                if (tp >= ntp) break;
                const long long pm = pix((tp0 + tp) * 16 + (lane >> 2));
                const int ch = n0 + cg * 32 + (((lane & 3) ^ ((lane >> 4) & 3)) * 8);
                const bf16x8 val = *reinterpret_cast<const bf16x8*>(res_at(tp) + lane * 16);
                if (pm >= 0 && ch + 8 <= a.Cout) store_out16((__bf16*)a.dst + (pm * a.Cout + ch), val);
            }
            stamp(3);
            return;
        }
        if (!own_res) wait_vmcnt<0>();
        if (own_res) {
#pragma unroll
            for (int tp = 0; tp < NTP; ++tp) {
                // the pieces went out in pairs, in this order
                if (tp == 0) wait_vmcnt<4>();
                if (tp == 2) wait_vmcnt<2>();
                if (tp == 4) wait_vmcnt<0>();
                if (tp >= ntp) break;
#pragma unroll
                for (int tc = 0; tc < TC; ++tc) {
                    const int c = (tc * 2 + (q >> 1)) ^ ((r >> 2) & 3);
                    const bf16x4 rr = *reinterpret_cast<const bf16x4*>(res_at(tp) + r * 64 + c * 16 + (q & 1) * 8);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[tc][tp][e] += (float)rr[e];
                }
            }
        }
        __syncthreads();
        if (a.res && !res_pre && !own_res) {
            tile_load_residual<NSLOT, CT, 64 * WAVES>(rtile, a.res, a.zero, a.Cout, n0, w, lane, pix);
            __syncthreads();
        }
        stamp(5);
#pragma unroll
        for (int tp = 0; tp < NTP; ++tp) {
            if (tp >= ntp) break;
            const int slot = (tp0 + tp) * 16 + r;
#pragma unroll
            for (int tc = 0; tc < TC; ++tc) {
                const int chl = cg * 16 * TC + tc * 16 + cq;
                float v[4] = {acc[tc][tp][0], acc[tc][tp][1], acc[tc][tp][2], acc[tc][tp][3]};
                tile_put<CT>(rtile, slot, chl, v, a.res != nullptr && !own_res);
            }
        }
        __syncthreads();
        stamp(6);
        tile_store<NSLOT, CT, 64 * WAVES>(rtile, (__bf16*)a.dst, a.Cout, n0, pix);
        stamp(3);
        return;
    }
#pragma unroll
    for (int tp = 0; tp < NTP; ++tp) {
        const int p = (tp0 + tp) * 16 + r;
        if (tp >= ntp || p >= P3_NPIX) continue;
        const int ty = p / TW, tx = p - ty * TW;
        if (oy0 + ty >= a.H || ox0 + tx >= a.W) continue;
        const size_t m = ((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx;
#pragma unroll
        for (int tc = 0; tc < TC; ++tc) {
            const int ch = n0 + cg * 16 * TC + tc * 16 + cq;
            if (ch >= a.Cout) continue;
            const size_t o = m * a.Cout + ch;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (ch + e >= a.Cout) break;
                float x = acc[tc][tp][e];
                if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                if (a.res) x += (float)a.res[o + e];
                if (a.dst_fp32) ((float*)a.dst)[o + e] = x;
                else ((__bf16*)a.dst)[o + e] = (__bf16)x;
            }
        }
    }
}

// ---------------------------------------------------------------- 3x3 with patch reuse, flat patch
// The private-ring form of the kernel above (eight waves, 128 channels, rings three taps deep, the
// barrier-free epilogue) with ONE change of layout that takes the address arithmetic out of the tap
// loop.  A patch plane is the (TH+2) x (TW+2) patch stored flat with pitch PW = TW + 2 slots of 16
// bytes: slot(py, px) = py * PW + px.  The slot an output pixel (ty, tx) reads for tap (kh, kw) is then
// slot(ty, tx) + (kh * PW + kw): a per-lane base that never changes plus a constant that goes into the
// ds_read's offset field, so a tap is 8 fragment reads, 2 LDS-DMA instructions and 12 MFMAs with no
// vector ALU instruction at all (the rotated 16-slot rows above need seven per tap and wave).
// Bank conflicts: the 16 lanes of a pixel group read slots whose residues mod 16 must differ.  Along
// a tile row consecutive pixels are consecutive slots; from the end of row ty to the start of row
// ty' the residue steps by 1 exactly when (ty' - ty) * PW = TW (mod 16): for the 13 x 13 tile
// (PW = 15) that is ty' = ty + 3, so the pixel groups walk the tile's rows in the order
// 0, 3, 6, 9, 12, 2, 5, ... (RSTEP = 3; the two wrap-arounds leave a two-way conflict in two of the
// eleven groups).  The 19 x 10 tile (PW = 12) has no such step; it keeps the natural row order and
// two-way conflicts at every row end, and gains 4 KiB planes instead of 6 KiB ones: rings three
// taps deep and the barrier-free epilogue fit, which the rotated layout had no room for.
// Measured and left out: reading the fragments of tap t + 1 (the weights' and three of the six pixel groups')
// into a second register set before the MFMAs of tap t, the weights running three taps ahead -- the schedule
// came out as written, no spill in the loop, and every layer was 3-6 % slower (52.1 against 50.5 us on
// 256 -> 512 @ 26x26): with four waves per SIMD the wave's own LDS latency is already hidden by the others.
// TAIL (one channel tile: all 128 output channels of the layer in this workgroup): the 1x1 layer
// behind this one (128 -> 64) is computed in the epilogue.  The staged 1 KiB piece of wave (channel
// group c, half h) for pixel group g IS the MFMA B operand of that layer's K-step c for those 16
// pixels, so after one workgroup barrier wave (c, h) reads the four pieces of each of its pixel
// groups -- its own and its three neighbours' -- against the weight fragments of output channels
// 16 c .. 16 c + 15 (loaded from memory into registers after the K loop), K-steps in ascending order
// as the im2col kernel runs that layer (same bits), and stores its 32 bytes per pixel.
template <int TH, int TW, int PW, int RSTEP, int NT0, bool TAIL = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
conv3x3_flat_kernel(const Conv3Args a) {
    constexpr int NPIX = TH * TW;                  // output pixels per tile
    constexpr int NG = (NPIX + 15) / 16;           // pixel groups of 16: NT0 for each of waves 0-3, the rest for waves 4-7
    constexpr int IDLE0 = NT0 - 4, IDLE1 = NG - NT0 - 4;   // KiB of the idle patch buffer a wave of either half parks residual pieces in
    constexpr int PLANE = 4096, PATCH = 4 * PLANE; // 256 slots per plane, four 16-byte planes of a 32-channel chunk
    constexpr int PP = 2;                          // patch LDS-DMA instructions per wave and chunk
    constexpr int PSLOT = 2048, RING = 3;          // a tap of a wave's ring: 32 channels x 64 B
    constexpr int RING_BYTES = 8 * RING * PSLOT;
    static_assert(PW >= TW + 2 && (TH + 2) * PW <= 256 && NT0 <= 8 && IDLE1 >= 0 && 4 * (IDLE0 + IDLE1) * 1024 <= PATCH,
                  "tile shape not laid out for this kernel");
    __shared__ __attribute__((aligned(1024))) char lds_all[RING_BYTES + 2 * PATCH];
    char* const lds = lds_all;                     // patch buffers 0 and 1 first: every fragment offset fits the ds_read offset field
    char* const wring = lds_all + 2 * PATCH;

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    auto stamp = [&](int i) {
        if (a.trace && threadIdx.x == 0) a.trace[(size_t)blockIdx.x * 8 + i] = (long long)wall_clock64();
    };
    stamp(0);
    if (a.trace && threadIdx.x == 0)
        a.trace[(size_t)blockIdx.x * 8 + 4] = ((long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) |
                                              (unsigned int)__builtin_amdgcn_s_getreg((31 << 11) | 4);
    const int cg = w & 3, half = w >> 2;           // channel group (= patch plane it fills), pixel-group half
    const int tp0 = half * NT0;                    // first pixel group of this wave

    int tile;
    {
        const int T = a.ntiles * a.tiles_n;
        const int q8 = T >> 3, r8 = T & 7;
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    }
    const int tile_n = tile % a.tiles_n, gt = tile / a.tiles_n;
    const int n0 = tile_n * 128;
    const int per_img = a.tiles_y * a.tiles_x;
    const int cb = gt / per_img;
    const int oy0 = ((gt - cb * per_img) / a.tiles_x) * TH;
    const int ox0 = ((gt - cb * per_img) % a.tiles_x) * TW;

    // slot s of the tile's pixel order -> (ty, tx): rows are visited RSTEP apart
    auto row_of = [](int s, int& tx) {
        const int i = s / TW;
        tx = s - i * TW;
        return (i * RSTEP) % TH;
    };
    const int r = lane & 15, q = lane >> 4;

    const int NCH = a.C / 32;
    // ---- patch fill: wave w fills slots [64 j, 64 j + 64) of plane cg for j = half * PP + ii, from a raw
    // buffer over the input tensor; a slot outside the image (padding) or the patch reads beyond the buffer
    const __amdgpu_buffer_rsrc_t src_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.src, 0, (int)((uint32_t)a.ntiles / (uint32_t)(a.tiles_y * a.tiles_x) * (uint32_t)(a.H * a.W) * (uint32_t)a.C * 2u),
        (int)kRawBufferWord3);
    uint32_t poff[PP];
#pragma unroll
    for (int ii = 0; ii < PP; ++ii) {
        const int f = 64 * (half * PP + ii) + lane;
        const int py = f / PW, px = f - py * PW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        poff[ii] = (py < TH + 2 && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W)
                       ? (uint32_t)((((cb * a.H + iy) * a.W + ix) * a.C + cg * 8) * 2)
                       : kBufferNowhere;
    }
    auto issue_patch = [&](int chunk, int ii) {            // ii: compile-time after unrolling
        lds_dma16_buf(src_rs, ii == 0 ? poff[0] : poff[1], (uint32_t)chunk * 64u,
                      lds + (chunk & 1) * PATCH + cg * PLANE + (half * PP + ii) * 1024);
    };
    const int cout_pad = a.tiles_n * 128;
    const int wstep_tap = (a.C / 32) * cout_pad * 32;     // [K/32][cout_pad][32]: one tap further
    const int wstep_chunk = cout_pad * 32;
#pragma unroll
    for (int ii = 0; ii < PP; ++ii) issue_patch(0, ii);
    stamp(1);
    char* const myring = wring + w * (RING * PSLOT);
    const bool own_res = a.res && !a.dst_fp32 && (a.Cout & 7) == 0;
    const uint32_t wlane = (uint32_t)(((lane >> 2) * 32 + ((lane & 3) ^ chunk_swizzle<32>(lane >> 2)) * 8) * 2);
    const int wbase = (n0 + cg * 32) * 32;
    auto issue_wp = [&](int chunk, int tap, int slot) {
        const __bf16* const sb = a.wgt + (wbase + tap * wstep_tap + chunk * wstep_chunk);
        char* const dst = myring + slot * PSLOT;
        lds_dma16_sv<0>(sb, wlane, dst);
        lds_dma16_sv<1024>(sb, wlane, dst);
    };
    const int wfragp = slot_of<32>(r, q) * 16;

    // Development aid (compiled in with -DYOLO_AMD_DEV_TRACE only, active when a trace buffer is handed
    // over): every wave adds up, over its K loop, the shader cycles between five points of a tap --
    // [0] end of the burst -> next tap entered, [1] the wait for the tap's weights, [5] the chunk's
    // barrier, [2] issuing the tap's LDS-DMA instructions, [3] issuing the fragment reads until they
    // are all back, [4] the MFMA burst -- and writes the sums behind the per-workgroup stamps.  Each
    // stamp drains the wave's LDS reads (about 40 cycles): a diagnostic build, not a timing one.
#ifdef YOLO_AMD_DEV_TRACE
    uint32_t seg[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    auto tick = [&](int i) {
        if (a.trace) {
            unsigned long long t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            const uint32_t tl = (uint32_t)t;
            if (i >= 0) seg[i] += tl - tprev;
            tprev = tl;
        }
    };
#else
    auto tick = [&](int) {};
#endif

    // The rest of the kernel for a wave that owns NTW pixel groups (a compile-time count per half, so the
    // tap loop has no branch): NT0 for waves 0-3, NG - NT0 for waves 4-7.
    auto body = [&](auto NTWC) {
        constexpr int NTW = decltype(NTWC)::value;
        constexpr int NI = NTW - 4;                // residual pieces parked in the idle patch buffer (the rest: two ring slots)
        static_assert(NI >= 0, "at least four pixel groups per wave");
        int xbase[NTW];                            // byte offset of this lane's pixel in plane q, per pixel group
#pragma unroll
        for (int tp = 0; tp < NTW; ++tp) {
            int tx;
            const int ty = row_of(min((tp0 + tp) * 16 + r, NPIX - 1), tx);
            xbase[tp] = q * PLANE + (ty * PW + tx) * 16;
        }
        // the accumulators start from the bias
        f32x4 acc[2][NTW];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) {
            const f32x4 bs = *reinterpret_cast<const f32x4*>(a.bias + n0 + cg * 32 + tc * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int tp = 0; tp < NTW; ++tp) acc[tc][tp] = bs;
        }
        // Residual pieces (1 KiB: 16 pixels x 64 B of this wave's 32 channels), fetched during the last chunk
        // into memory nobody else reads: the first NI at tap 0 into this wave's share of the patch buffer the
        // last chunk does not read (IDLE0 KiB for each of waves 0-3, IDLE1 KiB for each of waves 4-7), two at
        // tap 7 and two at tap 8 into the ring slots those taps no longer fetch weights for.
        auto piece_of = [&](int wv, int tp) -> char* {    // wave wv's piece tp (wv in this wave's half)
            char* const idle_w = lds + (NCH & 1) * PATCH + (half == 0 ? wv * (IDLE0 * 1024) : 4 * IDLE0 * 1024 + (wv - 4) * (IDLE1 * 1024));
            return tp < NI ? idle_w + tp * 1024 : wring + wv * (RING * PSLOT) + ((tp - NI) >> 1) * 2048 + ((tp - NI) & 1) * 1024;
        };
        auto res_at = [&](int tp) -> char* { return piece_of(w, tp); };
        auto res_piece = [&](int tp) {
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const int slot = (tp0 + tp) * 16 + (ln >> 2);
            int tx;
            const int ty = row_of(slot, tx);
            const bool okp = slot < NPIX && oy0 + ty < a.H && ox0 + tx < a.W;
            const int ch = n0 + cg * 32 + (((ln & 3) ^ ((ln >> 4) & 3)) * 8);
            const __bf16* src = (okp && ch + 8 <= a.Cout)
                                    ? a.res + ((((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx) * a.Cout + ch)
                                    : a.zero;
            lds_dma16(src, res_at(tp));
        };
        // Two chunks are unrolled (PAR = chunk & 1) so that the chunk's patch buffer, chunk & 1, is a constant
        // in every fragment address.
        auto chunk_body = [&](auto PARC, int chunk) {
            constexpr int PAR = decltype(PARC)::value;
            const char* pb = lds + PAR * PATCH;
            const bool more_chunks = chunk + 1 < NCH;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // Loads are answered in issue order.  Younger than this tap's weights (the last loads
                // issued two taps ago) is what the tap before issued: a patch piece during the first
                // PP taps of a chunk that has a successor (in the last chunk residual pieces at taps 0,
                // 7 and 8 instead), then two weight pieces while there are taps left to fetch for.
                {
                    const int u = tap - 1;
                    const int n_more = 2 + ((u >= 0 && u < PP) ? 1 : 0);
                    const int n_last = (u + 2 <= 8) ? 2 : 0;
                    const int n_res = n_last + (u == 0 ? NI : (u == 7 ? 2 : 0));
                    tick(0);
                    wait_vmcnt_upto(more_chunks ? n_more : (own_res ? n_res : n_last));
                }
                // every wave's pieces of this chunk's patch are older than the weights it has just waited
                // for; behind the barrier nobody reads the other patch buffer any more, so the next chunk's
                // pieces may overwrite it
                tick(1);
                if (tap == 0) {
                    __builtin_amdgcn_s_barrier();
                    tick(5);
                }
                asm volatile("" ::: "memory");
                if (tap < PP && more_chunks) issue_patch(chunk + 1, tap);
                if (own_res && !more_chunks) {
                    if (tap == 0) {
#pragma unroll
                        for (int i = 0; i < NI; ++i) res_piece(i);
                    }
                    if (tap == 7 || tap == 8) {
                        res_piece(NI + (tap - 7) * 2);
                        res_piece(NI + (tap - 7) * 2 + 1);
                    }
                }
                if (tap + 2 < 9) issue_wp(chunk, tap + 2, (tap + 2) % RING);
                else if (more_chunks) issue_wp(chunk + 1, tap + 2 - 9, (tap + 2) % RING);
                tick(2);
                const int kh = tap / 3, kw = tap - 3 * kh;
                const char* ws = myring + (tap % RING) * PSLOT + wfragp;
                bf16x8 wf[2], xf[NTW];
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) wf[tc] = *reinterpret_cast<const bf16x8*>(ws + tc * 1024);
#pragma unroll
                for (int tp = 0; tp < NTW; ++tp) xf[tp] = *reinterpret_cast<const bf16x8*>(pb + xbase[tp] + (kh * PW + kw) * 16);
                tick(3);
                __builtin_amdgcn_s_setprio(2);
#pragma unroll
                for (int tp = 0; tp < NTW; ++tp) {
#pragma unroll
                    for (int tc = 0; tc < 2; ++tc)
                        acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tc], xf[tp], acc[tc][tp], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                tick(4);
            }
        };
        issue_wp(0, 0, 0);
        issue_wp(0, 1, 1);
        tick(-1);
        for (int chunk = 0; chunk < NCH; chunk += 2) {
            chunk_body(std::integral_constant<int, 0>{}, chunk);
            if (chunk + 1 < NCH) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
        }
#ifdef YOLO_AMD_DEV_TRACE
        if (a.trace && lane == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) a.trace[65536 + ((size_t)blockIdx.x * 8 + w) * 8 + i] = seg[i];
        }
#endif

        // ---- epilogue (no workgroup barrier: see the kernel above)
        stamp(2);
        if (!a.dst_fp32 && (a.Cout & 7) == 0) {
            // TAIL: the 1x1 layer's weight fragments and bias, asked for before this layer's stores go out
            // (loads and stores are answered in one order: behind the stores they would wait for every store's
            // acknowledgement)
            bf16x8 w2f[TAIL ? 4 : 1];
            f32x4 b2v = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (TAIL) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    w2f[ks] = *reinterpret_cast<const bf16x8*>(a.w2 + ((ks * 64 + cg * 16 + r) * 32 + q * 8));
                b2v = *reinterpret_cast<const f32x4*>(a.b2 + cg * 16 + q * 4);
            }
            const float slope = a.leaky ? 0.1f : 1.0f;
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
#pragma unroll
                for (int tp = 0; tp < NTW; ++tp) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[tc][tp][e] = leaky_max(acc[tc][tp][e], slope);
                }
            }
            // (without a residual the pieces' places still serve as the wave's staging memory: every wave
            // is past its last read of the patch buffer the last chunk does not use)
#pragma unroll
            for (int tp = 0; tp < NTW; ++tp) {
                if (own_res) {                   // NI pieces went out first, then two pairs (then TAIL's five loads)
                    constexpr int TL = TAIL ? 5 : 0;
                    if (tp == 0) wait_vmcnt<4 + TL>();
                    if (tp == NI) wait_vmcnt<2 + TL>();
                    if (tp == NI + 2) wait_vmcnt<0 + TL>();
                }
#pragma unroll
                for (int tc = 0; tc < 2; ++tc) {
                    const int c = (tc * 2 + (q >> 1)) ^ ((r >> 2) & 3);
                    bf16x4* p = reinterpret_cast<bf16x4*>(res_at(tp) + r * 64 + c * 16 + (q & 1) * 8);
                    float v[4] = {acc[tc][tp][0], acc[tc][tp][1], acc[tc][tp][2], acc[tc][tp][3]};
                    if (own_res) {
                        const bf16x4 rr = *p;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += (float)rr[e];
                    }
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) pk[e] = (__bf16)v[e];
                    *p = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            stamp(6);
            auto store_pieces = [&]() {
#pragma unroll
                for (int tp = 0; tp < NTW; ++tp) {
                    const int slot = (tp0 + tp) * 16 + (lane >> 2);
                    int tx;
                    const int ty = row_of(slot, tx);
                    const int ch = n0 + cg * 32 + (((lane & 3) ^ ((lane >> 4) & 3)) * 8);
                    const bf16x8 val = *reinterpret_cast<const bf16x8*>(res_at(tp) + lane * 16);
                    if (slot < NPIX && oy0 + ty < a.H && ox0 + tx < a.W && ch + 8 <= a.Cout)
                        store_out16((__bf16*)a.dst + ((((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx) * a.Cout + ch), val);
                }
            };
            if constexpr (!TAIL) store_pieces();
            if constexpr (TAIL) {
                // (this layer's own stores go out behind the 1x1 layer: a load is answered in order with the
                // stores issued before it, so the weight fragments would wait for their acknowledgements)
                __builtin_amdgcn_s_barrier();          // every wave's pieces are written (its LDS writes were waited for above)
                asm volatile("" ::: "memory");
                // (the bias is first used under the store's condition: waited for there, the wait would be
                // repeated in every pixel group's branch, behind the stores of the groups before)
                asm volatile("" : "+v"(b2v));
                const float slope2 = a.leaky2 ? 0.1f : 1.0f;
                // The 16 x 16-channel results go through LDS once more so that whole 128-byte pixels leave in
                // 16-byte stores (as 8 bytes per lane straight from the accumulators the 44 MB of this output
                // cost 15 us): 2 KiB per pixel group, in the patch buffer the last chunk read (seven groups)
                // and, for waves 4-7, the third ring slot of the wave with the group's number -- everybody is
                // past the K loop behind the barrier above.
                static_assert(NT0 <= 7 && NG - NT0 <= 5, "places for the 1x1 layer's output groups");
                char* const dead = lds + ((NCH - 1) & 1) * PATCH;
                auto out2_at = [&](int tp) -> char* {
                    return half == 0 ? dead + tp * 2048
                                     : (tp < 4 ? wring + (4 + tp) * (RING * PSLOT) + 2 * PSLOT : dead + 7 * 2048);
                };
#pragma unroll
                for (int tp = 0; tp < NTW; ++tp) {
                    f32x4 acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const bf16x8 yf = *reinterpret_cast<const bf16x8*>(piece_of(half * 4 + ks, tp) + r * 64 + ((q ^ ((r >> 2) & 3)) * 16));
                        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks], yf, acc2, 0, 0, 0);
                    }
                    bf16x4 pk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = acc2[e] + b2v[e];
                        x = x > 0.f ? x : slope2 * x;
                        pk[e] = (__bf16)x;
                    }
                    *reinterpret_cast<bf16x4*>(out2_at(tp) + r * 128 + (((cg * 2 + (q >> 1)) ^ (r & 7)) * 16) + (q & 1) * 8) = pk;
                }
                store_pieces();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < (2 * NTW + 3) / 4; ++k) {
                    const int it = cg + 4 * k;               // eight pixels of group it / 2, wave-uniform
                    if (it < 2 * NTW) {
                        const int tp = it >> 1, rr = (it & 1) * 8 + (lane >> 3), c8 = lane & 7;
                        const int slot = (tp0 + tp) * 16 + rr;
                        int tx;
                        const int ty = row_of(slot, tx);
                        const bf16x8 val = *reinterpret_cast<const bf16x8*>(out2_at(tp) + rr * 128 + ((c8 ^ (rr & 7)) * 16));
                        if (slot < NPIX && oy0 + ty < a.H && ox0 + tx < a.W)
                            store_out16(a.dst2 + ((((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx) * 64 + c8 * 8), val);
                    }
                }
            }
            stamp(3);
            return;
        }
        const int cq = q * 4;
#pragma unroll
        for (int tp = 0; tp < NTW; ++tp) {
            const int p = (tp0 + tp) * 16 + r;
            if (p >= NPIX) continue;
            int tx;
            const int ty = row_of(p, tx);
            if (oy0 + ty >= a.H || ox0 + tx >= a.W) continue;
            const size_t m = ((size_t)cb * a.H + oy0 + ty) * a.W + ox0 + tx;
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const int ch = n0 + cg * 32 + tc * 16 + cq;
                if (ch >= a.Cout) continue;
                const size_t o = m * a.Cout + ch;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (ch + e >= a.Cout) break;
                    float x = acc[tc][tp][e];
                    if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                    if (a.res) x += (float)a.res[o + e];
                    if (a.dst_fp32) ((float*)a.dst)[o + e] = x;
                    else ((__bf16*)a.dst)[o + e] = (__bf16)x;
                }
            }
        }
    };
    if (half == 0) body(std::integral_constant<int, NT0>{});
    else body(std::integral_constant<int, NG - NT0>{});
}

__device__ __forceinline__ TileWalk tile_walk(int ntiles) {
    return tile_walk_of(ntiles, (int)gridDim.x, (int)blockIdx.x);
}

// ---------------------------------------------------------------- first layer
// Network input is NCHW fp32 with at most 3 channels, so K = 9*Cin <= 32 is a
// single MFMA step: D[channel][pixel] = W[channel][k] * X[k][pixel] with
// k = (kh*3 + kw)*Cin + c.  A workgroup takes 8x32-pixel tiles: it stages the
// (8+2)x(32+2) input patch of every channel in LDS with range-checked buffer
// loads (a coordinate outside the image gets an out-of-range offset and reads
// as zero, so the padding is materialised once), then each wave handles four
// strips of 16 consecutive pixels.  Lane (pixel n = lane & 15, k group
// q = lane >> 4) gathers its eight im2col values from the patch, splits each
// into a bf16 high part and a bf16 remainder (as are the weights, once, in
// registers) and hi*hi + hi*lo + lo*hi accumulate on the otherwise idle matrix
// pipe, which keeps the layer at fp32-level accuracy.  Weight rows are permuted
// so that a lane's accumulators are eight consecutive channels of its own
// pixel: one 16-byte store per lane, 1 KiB contiguous per wave.  The layer is
// bound by its 64-byte-per-pixel output stream.
constexpr uint32_t kRawBufferFlags = 0x00020000u;   // untyped 32-bit raw buffer descriptor word 3
constexpr int FIRST_TH = 8, FIRST_TW = 32, FIRST_PH = FIRST_TH + 2, FIRST_PW = FIRST_TW + 2;

// two fp32 -> one dword of two bf16 (one v_cvt_pk_bf16_f32; converting element by element
// costs a conversion and a v_perm per value)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}

// F32: the output is fp32 (32 or 64 bytes per pixel and lane quad instead of 16 or 32), for
// layers whose BatchNorm runs on batch statistics and needs the un-normalised sums.
template <int COUT, bool F32 = false>
__global__ void __launch_bounds__(256)
conv_first_kernel(const float* __restrict__ src, const float* __restrict__ wgt,
                  const float* __restrict__ bias, void* __restrict__ dst, int B, int Cin, int H, int W,
                  float slope, uint32_t src_bytes, uint32_t dst_bytes, int tiles_x, int tiles_y) {
    constexpr int NM = COUT / 16;
    static_assert(NM == 1 || NM == 2, "first layer: 16 or 32 output channels");
    constexpr int PH = FIRST_PH, PW = FIRST_PW, PLANE = PH * PW;
    __shared__ float patch[3 * PLANE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int K = 9 * Cin;

    // patch index of this lane's eight k values relative to the top-left corner
    // of its pixel's window; k >= K (zero weight) re-reads the corner
    int offl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        offl[j] = 0;
        if (k < K) {
            const int tap = k / Cin, c = k - tap * Cin;
            const int kh = tap / 3, kw = tap - 3 * kh;
            offl[j] = (c * PH + kh) * PW + kw;
        }
    }
    // MFMA i, row r  <->  channel 8*(r>>2) + 4*i + (r&3)   (NM == 2)
    bf16x8 wf[NM], wl[NM];
    f32x4 bv[NM];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int ch = NM == 2 ? 8 * (n >> 2) + 4 * i + (n & 3) : n;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 8 * q + j;
            const float wv = k < K ? wgt[k * COUT + ch] : 0.f;
            wf[i][j] = (__bf16)wv;
            wl[i][j] = (__bf16)(wv - (float)wf[i][j]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) bv[i][t] = bias[NM == 2 ? 8 * q + 4 * i + t : 4 * q + t];
    }
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, (int)kRawBufferFlags);

    const __amdgpu_buffer_rsrc_t rd =
        __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, (int)dst_bytes, (int)kRawBufferFlags);

    // The patch elements this thread stages (the same four for every tile):
    // offset inside the image relative to the patch origin, and patch row/column.
    const int per_image = tiles_x * tiles_y, ntiles = per_image * B;
    const int total = Cin * PLANE;
    int pre[4], prc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = min((int)threadIdx.x + 256 * u, total - 1);
        const int c = e / PLANE, r2 = e - c * PLANE;
        const int r = r2 / PW, col = r2 - r * PW;
        pre[u] = (c * H + r) * W + col;
        prc[u] = (r << 16) | col;
    }
    // Loads of a tile's patch into registers.  Every load and store of the loop
    // below is issued unconditionally (out-of-range offsets instead of
    // predication) so the number of memory operations in flight is static and
    // the next tile's patch streams in while this tile is computed.
    uint32_t stage[4];
    auto fetch = [&](int tile, uint32_t (&st)[4]) {
        const int b = tile / per_image, t2 = tile - b * per_image;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        const int x0 = tx * FIRST_TW - 1, y0 = ty * FIRST_TH - 1;
        const int origin = (b * Cin * H + y0) * W + x0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int gy = y0 + (prc[u] >> 16), gx = x0 + (prc[u] & 0xffff);
            const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            const uint32_t o = ok ? (uint32_t)((origin + pre[u]) * 4) : 0x80000000u;
            st[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o, 0, 0);
        }
    };
    auto dummy_stores = [&]() {
        // Four range-dropped stores: they keep the sequence of outstanding
        // memory operations at loop entry equal to the one the back edge
        // carries, so the counted waits for a patch never include the stores
        // of the tiles in between.
#pragma unroll
        for (int u = 0; u < (F32 ? 4 * NM : 4); ++u) {
            if constexpr (NM == 2 || F32)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{0u, 0u, 0u, 0u}, rd, (int)(0xFFFFFF00u + 16u * (u & 3)), 0, 0);
            else
                __builtin_amdgcn_raw_buffer_store_b64(u32x2{0u, 0u}, rd, (int)(0xFFFFFF00u + 16u * u), 0, 0);
        }
    };
    const TileWalk tw = tile_walk(ntiles);
    if (tw.first >= tw.end) return;
    const int step = tw.step, last = tw.end - 1;
    fetch(tw.first, stage);
    dummy_stores();
    for (int tile = tw.first; tile < tw.end; tile += step) {
        const int b = tile / per_image, t2 = tile - b * per_image;
        const int ty = t2 / tiles_x, tx = t2 - ty * tiles_x;
        __syncthreads();                                   // previous tile's gathers are done
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if ((int)threadIdx.x + 256 * u < total) patch[threadIdx.x + 256 * u] = __uint_as_float(stage[u]);
        __syncthreads();
        fetch(min(tile + step, last), stage);              // next tile streams in meanwhile
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int row = 2 * wave + (sidx >> 1), colp = (sidx & 1) * 16 + n;
            const float* pp = patch + row * PW + colp;
            bf16x8 xf, xl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = pp[offl[j]];
                xf[j] = (__bf16)t;
                xl[j] = (__bf16)(t - (float)xf[j]);
            }
            f32x4 acc[NM];
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[i], xf, bv[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], xf, acc[i], 0, 0, 0);
            }
            // The MFMA leaves lane (n, q) with chunk q of pixel n; a lane
            // permutation (through the LDS crossbar, no memory) gives lane l
            // chunk l & 3 of pixel l >> 2, so the wave's store is 1 KiB in lane
            // order: whole 64-byte sectors per group of four lanes.
            const int src_lane4 = (16 * (lane & 3) + (lane >> 2)) * 4;
            const int y = ty * FIRST_TH + row, x = tx * FIRST_TW + (sidx & 1) * 16 + (lane >> 2);
            const uint32_t pixel = (uint32_t)((b * H + y) * W + x);
            const bool ok = y < H && x < W;
            if constexpr (F32) {
                // 4 * NM floats per lane: chunk lane & 3 of pixel lane >> 2, as below
#pragma unroll
                for (int i = 0; i < NM; ++i) {
                    u32x4 w;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
                        w[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(
                            src_lane4, (int)__float_as_uint(leaky_max(acc[i][t], slope)));
                    const uint32_t o = ok ? pixel * (uint32_t)(COUT * 4) + (uint32_t)(16 * NM) * (lane & 3) + 16u * i
                                          : 0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(w, rd, (int)o, 0, 0);
                }
            } else if constexpr (NM == 2) {
                u32x4 w;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        w[2 * i + t] = pack_bf16x2(leaky_max(acc[i][2 * t], slope), leaky_max(acc[i][2 * t + 1], slope));
#pragma unroll
                for (int t = 0; t < 4; ++t) w[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)w[t]);
                const uint32_t o = ok ? pixel * (uint32_t)(COUT * 2) + 16u * (lane & 3) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b128(w, rd, (int)o, 0, 0);
            } else {
                u32x2 w;
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    w[t] = pack_bf16x2(leaky_max(acc[0][2 * t], slope), leaky_max(acc[0][2 * t + 1], slope));
#pragma unroll
                for (int t = 0; t < 2; ++t) w[t] = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)w[t]);
                const uint32_t o = ok ? pixel * (uint32_t)(COUT * 2) + 8u * (lane & 3) : 0xFFFFFFF0u;
                __builtin_amdgcn_raw_buffer_store_b64(w, rd, (int)o, 0, 0);
            }
        }
    }
}

// ---------------------------------------------------------------- fused stem
// The first layer writes 64 bytes per pixel of the full-resolution image
// (354 MB at batch 32 x 416^2) only for the stride-2 layer behind it to read
// them back: both are bound by that round trip.  This kernel computes the two
// layers together.  A workgroup takes 8x16-pixel tiles of the SECOND layer's
// output; per tile it
//   A. stages the 19x35 input patch (fp32, every channel) in LDS, as
//      conv_first_kernel does (range-checked loads, next tile prefetched);
//   B. computes the 17x33 first-layer pixels the tile needs, 16 at a time, with
//      exactly conv_first_kernel's MFMA sequence (so the values are the same
//      bits), and parks them as bf16 in LDS, zero where the second layer pads;
//      layout [8-channel chunk q][row][column parity][column/2] x 16 B, so the
//      stride-2 reads of phase C are 16 consecutive slots;
//   C. runs the second layer's nine taps from that patch: wave = 4 output rows x
//      32 channels, its weight fragments (9 taps x 2) live in registers for the
//      whole kernel, K-steps in the im2col kernel's order (same accumulators);
//   D. bias + LeakyReLU + one rounding, output tile through LDS, 16-byte stores.
// First-layer activations are never written to memory.
constexpr int STEM_TY = 8, STEM_TX = 16;                   // second-layer output tile
constexpr int STEM_PR = 2 * STEM_TY + 1, STEM_PC = 2 * STEM_TX + 1;      // 17 x 33 first-layer pixels
constexpr int STEM_IR = STEM_PR + 2, STEM_IC = STEM_PC + 2;              // 19 x 35 input pixels
constexpr int STEM_NPIX = STEM_PR * STEM_PC;               // 561
constexpr int STEM_STRIPS = (STEM_NPIX + 15) / 16;         // 36
constexpr int STEM_HALF = STEM_TX + 1;                     // 17 columns per parity
constexpr int STEM_IN_BYTES = 8192;                        // >= 3 * 19 * 35 * 4
constexpr int STEM_L0_BYTES = 4 * STEM_PR * 2 * STEM_HALF * 16;          // 36992

struct StemArgs {
    const float* src;
    const float* w0;
    const float* b0;
    const __bf16* w1;      // [9][64][32] bf16 (K-tile major)
    const float* b1;
    __bf16* dst;
    int B, Cin, H, W, Ho, Wo;
    float slope0;
    int leaky1;
    uint32_t src_bytes;
    int tiles_x, tiles_y;
    // TAIL: the 1x1 layer (64 -> 32) that reads the second layer's output
    const __bf16* w2;      // [2][32][32] bf16 (K-tile major)
    const float* b2;
    __bf16* dst2;
    int leaky2;
};

// TAIL adds a phase E: the 1x1 layer behind the second layer (64 -> 32 channels, the first
// residual block's squeeze) is computed from the output tile while it sits in LDS, with the
// im2col kernel's two K-steps in its order (same accumulators, same bits), and leaves through
// the input patch's LDS (idle until the next tile).  The second layer's output is still
// written: the block's shortcut reads it.  It saves that 1x1 layer's launch, which read the
// 64 channels back from memory.
constexpr int STEM_TAB_BYTES = (STEM_STRIPS / 4) * 256 * 8;
constexpr int STEM_FIRST_BYTES = 6 * 64 * 16;              // first layer: weight fragments (hi, lo) and bias per lane
constexpr int STEM_W2_BYTES = 2 * 32 * 32 * 2;

template <bool TAIL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_stem_kernel(const StemArgs a) {
    constexpr int PLANE_IN = STEM_IR * STEM_IC;
    __shared__ __attribute__((aligned(1024))) char lds[STEM_IN_BYTES + STEM_L0_BYTES + STEM_TAB_BYTES + STEM_FIRST_BYTES +
                                                       (TAIL ? STEM_W2_BYTES : 0)];
    static_assert(STEM_TY * STEM_TX * 32 * 2 <= STEM_IN_BYTES, "the 1x1 layer's tile fits where the input patch was");
    uint32_t* const inwords = reinterpret_cast<uint32_t*>(lds);   // input patch: bf16 hi << 16 | bf16 lo per value
    char* const l0 = lds + STEM_IN_BYTES;                  // first-layer patch; later the output tile
    static_assert(STEM_TY * STEM_TX * 64 * 2 <= STEM_L0_BYTES, "output tile fits where the patch was");
    static_assert(STEM_STRIPS % 4 == 0, "strips split evenly over the four waves");

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int K0 = 9 * a.Cin;

    // ---- first layer: im2col offsets inside the input patch, weights (hi/lo), bias
    int offl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * q + j;
        offl[j] = 0;
        if (k < K0) {
            const int tap = k / a.Cin, c = k - tap * a.Cin;
            const int kh = tap / 3, kw = tap - 3 * kh;
            offl[j] = (c * STEM_IR + kh) * STEM_IC + kw;
        }
    }
    // (weight fragments hi / lo and bias of the first layer: worked out once, parked in LDS per lane and read
    // back at the top of every tile's phase B, so that their 24 registers are free in the other phases)
    u32x4* const first_tab = reinterpret_cast<u32x4*>(lds + STEM_IN_BYTES + STEM_L0_BYTES + STEM_TAB_BYTES) + lane;
    if (w == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = 8 * (n >> 2) + 4 * i + (n & 3);     // MFMA i, row r <-> channel 8*(r>>2) + 4*i + (r&3)
            bf16x8 f, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 8 * q + j;
                const float wv = k < K0 ? a.w0[k * 32 + ch] : 0.f;
                f[j] = (__bf16)wv;
                l[j] = (__bf16)(wv - (float)f[j]);
            }
            f32x4 bb;
#pragma unroll
            for (int t = 0; t < 4; ++t) bb[t] = a.b0[8 * q + 4 * i + t];
            first_tab[(0 + i) * 64] = __builtin_bit_cast(u32x4, f);
            first_tab[(2 + i) * 64] = __builtin_bit_cast(u32x4, l);
            first_tab[(4 + i) * 64] = __builtin_bit_cast(u32x4, bb);
        }
    }
    // ---- second layer: this wave's weight fragments, all nine taps
    const int wpx = w & 1, wch = w >> 1;                   // pixel half (4 rows), channel half (32)
    bf16x8 w1f[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            w1f[tap][tc] = *reinterpret_cast<const bf16x8*>(a.w1 + ((tap * 64 + wch * 32 + tc * 16 + n) * 32 + q * 8));

    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc((void*)a.src, 0, (int)a.src_bytes, (int)kRawBufferFlags);

    // ---- input patch staging: the elements this thread loads, the same for every tile
    const int total = a.Cin * PLANE_IN;
    uint32_t prc[8];                                       // channel << 24 | row << 8 | column
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = min((int)threadIdx.x + 256 * u, total - 1);
        const int c = e / PLANE_IN, r2 = e - c * PLANE_IN;
        const int r = r2 / STEM_IC, col = r2 - r * STEM_IC;
        prc[u] = ((uint32_t)c << 24) | ((uint32_t)r << 8) | (uint32_t)col;
    }
    const int per_image = a.tiles_x * a.tiles_y, ntiles = per_image * a.B;
    uint32_t stage[8];
    auto fetch = [&](int tile) {
        const int b = tile / per_image, t2 = tile - b * per_image;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int x0 = 2 * tx * STEM_TX - 2, y0 = 2 * ty * STEM_TY - 2;
        const int origin = (b * a.Cin * a.H + y0) * a.W + x0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = (int)(prc[u] >> 24), r = (int)((prc[u] >> 8) & 0xffu), col = (int)(prc[u] & 0xffu);
            const int gy = y0 + r, gx = x0 + col;
            const bool ok = (unsigned)gy < (unsigned)a.H && (unsigned)gx < (unsigned)a.W;
            const uint32_t o = ok ? (uint32_t)((origin + (c * a.H + r) * a.W + col) * 4) : 0x80000000u;
            stage[u] = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o, 0, 0);
        }
    };
    // ---- phase B: strip i of this wave holds patch pixels p = (w + 4 i) * 16 + n; per strip and lane the
    // word of the pixel's top-left input value and the byte offset of its 16 output bytes.  (Kept in LDS,
    // one 8-byte read per strip: as registers the eighteen values push the 72 weight fragments of the
    // second layer into scratch, worked out per strip they are a fifth of the strip's vector instructions.)
    u32x2* const strip_tab = reinterpret_cast<u32x2*>(lds + STEM_IN_BYTES + STEM_L0_BYTES) + threadIdx.x;
#pragma unroll
    for (int i = 0; i < STEM_STRIPS / 4; ++i) {
        const int pp = min((w + 4 * i) * 16 + n, STEM_NPIX - 1);
        const int pr = pp / STEM_PC, pc = pp - pr * STEM_PC;
        strip_tab[i * 256] = u32x2{(uint32_t)(pr * STEM_IC + pc),
                                   (uint32_t)((((q * STEM_PR + pr) * 2 + (pc & 1)) * STEM_HALF + (pc >> 1)) * 16)};
    }
    const bool last_strip_ok = (w + 4 * (STEM_STRIPS / 4 - 1)) * 16 + n < STEM_NPIX;   // only the last strip runs over
    char* const w2l = lds + STEM_IN_BYTES + STEM_L0_BYTES + STEM_TAB_BYTES + STEM_FIRST_BYTES;   // TAIL: the 1x1 layer's weights
    if constexpr (TAIL)
        *reinterpret_cast<u32x4*>(w2l + threadIdx.x * 16) = *reinterpret_cast<const u32x4*>(a.w2 + threadIdx.x * 8);
    const TileWalk tw = tile_walk(ntiles);
    if (tw.first >= tw.end) return;
    const int step = tw.step, last = tw.end - 1;
    fetch(tw.first);
    for (int tile = tw.first; tile < tw.end; tile += step) {
        const int b = tile / per_image, t2 = tile - b * per_image;
        const int tyi = t2 / a.tiles_x, txi = t2 - tyi * a.tiles_x;
        const int oy0 = tyi * STEM_TY, ox0 = txi * STEM_TX;
        // the previous tile's output tile has been read out of LDS by every wave (its stores to memory are
        // not waited for: nothing here depends on them)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if ((int)threadIdx.x + 256 * u < total) {
                // split once here into bf16 hi and lo halves of one word (every input value is
                // gathered by up to nine taps below, which used to split it nine times)
                const float t = __uint_as_float(stage[u]);
                const __bf16 h = (__bf16)t;
                const __bf16 l = (__bf16)(t - (float)h);
                inwords[threadIdx.x + 256 * u] = ((uint32_t)__builtin_bit_cast(unsigned short, h) << 16) |
                                                 (uint32_t)__builtin_bit_cast(unsigned short, l);
            }
        __syncthreads();
        fetch(min(tile + step, last));                     // next tile streams in meanwhile

        // ---- B: first-layer pixels of the patch, 16 per strip.  What a strip's lanes read and where they
        // write does not depend on the tile (strip_tab, worked out once per kernel), so a strip is one table
        // read, the eight gathers, eight byte permutes, six MFMAs, LeakyReLU as packed multiplies and bare
        // maxima, four packed conversions and one 16-byte LDS write.  Pixels outside the image must read as
        // the second layer's zero padding: tiles that reach over the image's edge zero them afterwards (same
        // wave, same LDS addresses, so in order), instead of every strip of every tile selecting per value.
        const int y_first = 2 * oy0 - 1, x_first = 2 * ox0 - 1;
        bf16x8 wf[2], wl[2];
        f32x4 bv[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            wf[i] = __builtin_bit_cast(bf16x8, first_tab[(0 + i) * 64]);
            wl[i] = __builtin_bit_cast(bf16x8, first_tab[(2 + i) * 64]);
            bv[i] = __builtin_bit_cast(f32x4, first_tab[(4 + i) * 64]);
        }
#pragma unroll 1
        for (int i = 0; i < STEM_STRIPS / 4; ++i) {
            const u32x2 st = strip_tab[i * 256];
            const uint32_t* ip = inwords + st[0];
            uint32_t wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[j] = ip[offl[j]];
            u32x4 xfw, xlw;                                // element j: half (j & 1) of word j / 2
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xfw[k] = __builtin_amdgcn_perm(wv[2 * k + 1], wv[2 * k], 0x07060302u);   // the two hi halves
                xlw[k] = __builtin_amdgcn_perm(wv[2 * k + 1], wv[2 * k], 0x05040100u);   // the two lo halves
            }
            const bf16x8 xf = __builtin_bit_cast(bf16x8, xfw), xl = __builtin_bit_cast(bf16x8, xlw);
            f32x4 acc0[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                acc0[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[m], xf, bv[m], 0, 0, 0);
                acc0[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xl, acc0[m], 0, 0, 0);
                acc0[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[m], xf, acc0[m], 0, 0, 0);
            }
            u32x4 pk;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x2 v = f32x2{acc0[m][2 * t], acc0[m][2 * t + 1]};
                    const f32x2 sv = v * f32x2{a.slope0, a.slope0};
                    float r0, r1;
                    asm("v_max_f32 %0, %1, %2" : "=v"(r0) : "v"(v[0]), "v"(sv[0]));
                    asm("v_max_f32 %0, %1, %2" : "=v"(r1) : "v"(v[1]), "v"(sv[1]));
                    pk[2 * m + t] = pack_bf16x2(r0, r1);
                }
            if (i + 1 < STEM_STRIPS / 4 || last_strip_ok) *reinterpret_cast<u32x4*>(l0 + st[1]) = pk;
        }
        if (y_first < 0 || x_first < 0 || y_first + STEM_PR > a.H || x_first + STEM_PC > a.W) {
#pragma unroll 1
            for (int i = 0; i < STEM_STRIPS / 4; ++i) {
                const u32x2 st = strip_tab[i * 256];
                const int pr = (int)st[0] / STEM_IC, pc = (int)st[0] - pr * STEM_IC;
                const bool outside = (unsigned)(y_first + pr) >= (unsigned)a.H || (unsigned)(x_first + pc) >= (unsigned)a.W;
                if (outside && (i + 1 < STEM_STRIPS / 4 || last_strip_ok)) *reinterpret_cast<u32x4*>(l0 + st[1]) = u32x4{0u, 0u, 0u, 0u};
            }
        }
        __syncthreads();

        // ---- C: second layer, nine taps, K-steps in tap order
        f32x4 acc[2][4];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) acc[tc][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            bf16x8 xf[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int pr = 2 * (wpx * 4 + tp) + kh;
                xf[tp] = *reinterpret_cast<const bf16x8*>(
                    l0 + (((q * STEM_PR + pr) * 2 + (kw & 1)) * STEM_HALF + n + (kw >> 1)) * 16);
            }
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[tap][tc], xf[tp], acc[tc][tp], 0, 0, 0);
        }
        __syncthreads();                                   // everyone is done reading the patch

        // ---- D: output tile (128 pixels x 64 channels) through LDS
        const int cq = q * 4;
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int slot = (wpx * 4 + tp) * STEM_TX + n;
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const int chl = wch * 32 + tc * 16 + cq;
                const f32x4 bs = *reinterpret_cast<const f32x4*>(a.b1 + chl);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float xx = acc[tc][tp][r] + bs[r];
                    if (a.leaky1) xx = xx > 0.f ? xx : 0.1f * xx;
                    v[r] = xx;
                }
                tile_put<64>(l0, slot, chl, v, false);
            }
        }
        __syncthreads();
        auto pix = [&](int slot) -> long long {
            const int y = oy0 + slot / STEM_TX, x = ox0 + slot % STEM_TX;
            return (y < a.Ho && x < a.Wo) ? ((long long)b * a.Ho + y) * a.Wo + x : -1;
        };
        if constexpr (!TAIL) {
            tile_store<STEM_TY * STEM_TX, 64, 256>(l0, a.dst, 64, 0, pix);
        } else {
            // ---- E: 1x1 layer on the parked tile; wave w takes pixel groups 2w and 2w + 1, both channel groups
#pragma unroll 1
            for (int g = 0; g < 2; ++g) {
                const int slot = (2 * w + g) * 16 + n;
                f32x4 acc2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 yf = *reinterpret_cast<const bf16x8*>(l0 + tile_addr<64>(slot, ks * 32 + q * 8));
#pragma unroll
                    for (int cg = 0; cg < 2; ++cg) {
                        const bf16x8 wq = *reinterpret_cast<const bf16x8*>(w2l + ((ks * 32 + cg * 16 + n) * 32 + q * 8) * 2);
                        acc2[cg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq, yf, acc2[cg], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int cg = 0; cg < 2; ++cg) {
                    const int chl = cg * 16 + cq;
                    const f32x4 bs = *reinterpret_cast<const f32x4*>(a.b2 + chl);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float xx = acc2[cg][r] + bs[r];
                        if (a.leaky2) xx = xx > 0.f ? xx : 0.1f * xx;
                        v[r] = xx;
                    }
                    tile_put<32>(lds, slot, chl, v, false);
                }
            }
            tile_store<STEM_TY * STEM_TX, 64, 256>(l0, a.dst, 64, 0, pix);
            __syncthreads();
            tile_store<STEM_TY * STEM_TX, 32, 256>(lds, a.dst2, 32, 0, pix);
        }
    }
}

// ---------------------------------------------------------------- thin 3x3 (32 -> 64 channels)
// With 32 input channels a 3x3 layer has nine K-steps in all: as an im2col GEMM
// every workgroup pays set-up and epilogue for a handful of MFMAs and re-reads
// each input pixel nine times through the vector-memory path.  Here a persistent
// workgroup keeps the layer's whole weight tensor (64 x 288 bf16) as MFMA
// fragments in registers (72 per lane) and walks 8x16-pixel output tiles: the
// 10x18-pixel input patch comes in once by LDS-DMA (64 bytes per pixel, whole
// pixels per lane quad so every line is fetched once; the four 16-byte chunks
// of pixel slot s sit at position chunk ^ ((s >> 2) & 3), which spreads the 16
// pixels a tap reads over the banks), double buffered so the next tile streams
// in behind the MFMAs; the residual tile is
// fetched by LDS-DMA during the taps; output as in the other kernels (one
// rounding, 16-byte stores).  K-steps in tap order: the same accumulators as the
// im2col kernel gives.  The layer is bound by its HBM traffic.
// Measured and left out: the residual tile fetched a tile ahead into a second output tile (one
// barrier less per tile, 56 KiB of LDS): 90.8 against 90.6 us inside the network.
constexpr int THIN_TY = 8, THIN_TX = 16;
constexpr int THIN_PR = THIN_TY + 2, THIN_PC = THIN_TX + 2;          // 10 x 18 input pixels
constexpr int THIN_SLOTS = 192;                                      // 180 used: twelve LDS-DMA instructions of 16 pixels
constexpr int THIN_PATCH = THIN_SLOTS * 64;                          // 12 KiB
constexpr int THIN_TILE = THIN_TY * THIN_TX * 64 * 2;                // 16 KiB output tile

struct ThinArgs {
    const __bf16* src;     // (B, H, W, 32)
    const __bf16* wgt;     // [9][64][32]
    const float* bias;
    const __bf16* res;     // (B, H, W, 64) or null
    __bf16* dst;           // (B, H, W, 64)
    const __bf16* zero;
    int B, H, W, leaky;
    int tiles_x, tiles_y;
};

__global__ void __launch_bounds__(256)
conv3x3_thin_kernel(const ThinArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * THIN_PATCH + THIN_TILE];
    char* const tile = lds + 2 * THIN_PATCH;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wpx = w & 1, wch = w >> 1;               // pixel half (4 rows), channel half (32)

    bf16x8 wf[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            wf[tap][tc] = *reinterpret_cast<const bf16x8*>(a.wgt + ((tap * 64 + wch * 32 + tc * 16 + n) * 32 + q * 8));

    // patch fill: instruction j (of 12) covers pixel slots 16j..16j+15, four lanes per pixel;
    // wave w issues j = w, w+4, w+8
    int prow[3], pcol[3], pchunk[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int sl = 16 * (w + 4 * i) + (lane >> 2);
        prow[i] = sl < THIN_PR * THIN_PC ? sl / THIN_PC : -100000;   // unused slots read the zero page
        pcol[i] = sl - (sl / THIN_PC) * THIN_PC;
        pchunk[i] = ((lane & 3) ^ ((sl >> 2) & 3)) * 8;
    }
    const int per_image = a.tiles_x * a.tiles_y, ntiles = per_image * a.B;
    auto issue_patch = [&](int t, int buf) {
        const int b = t / per_image, t2 = t - b * per_image;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int y0 = ty * THIN_TY - 1, x0 = tx * THIN_TX - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int iy = y0 + prow[i], ix = x0 + pcol[i];
            const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const __bf16* p = ok ? a.src + (((size_t)b * a.H + iy) * a.W + ix) * 32 + pchunk[i] : a.zero;
            lds_dma16(p, lds + buf * THIN_PATCH + (w + 4 * i) * 1024);
        }
    };
    const TileWalk tw = tile_walk(ntiles);
    if (tw.first >= tw.end) return;
    const int step = tw.step;
    const int cq = q * 4;
    f32x4 bsv[2];
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) bsv[tc] = *reinterpret_cast<const f32x4*>(a.bias + wch * 32 + tc * 16 + cq);
    // The weight fragments and the bias are waited for HERE.  Left to the compiler, the wait sits at their first
    // use inside the tile loop and is executed for every tile -- s_waitcnt vmcnt(17) ... vmcnt(0) through the first
    // taps, which after the first tile waits for nothing of theirs but does wait for the LDS-DMA fetches the loop has
    // just issued (the next tile's patch, this tile's residual): every tile then stalls in its MFMAs for a trip
    // to memory.  (An empty asm that names a register as read and written pins the wait in front of it.)  With
    // warm caches, stand-alone, the difference drowns in the noise; inside the network the layer went from
    // 99.0 to 90.4 us.
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc) asm volatile("" : "+v"(wf[tap][tc]));
#pragma unroll
    for (int tc = 0; tc < 2; ++tc) asm volatile("" : "+v"(bsv[tc]));
    issue_patch(tw.first, 0);
    int buf = 0;
    for (int t = tw.first; t < tw.end; t += step, buf ^= 1) {
        const int b = t / per_image, t2 = t - b * per_image;
        const int oy0 = (t2 / a.tiles_x) * THIN_TY, ox0 = (t2 % a.tiles_x) * THIN_TX;
        auto pix = [&](int slot) -> long long {
            const int y = oy0 + slot / THIN_TX, x = ox0 + slot % THIN_TX;
            return (y < a.H && x < a.W) ? ((long long)b * a.H + y) * a.W + x : -1;
        };
        // This tile's patch has landed (waited for in front of the last tile's stores, below; before the
        // first tile here) -- the stores themselves are not waited for: a wait at this point would expose
        // their whole trip to memory once per tile, and nothing depends on it.
        if (t == tw.first) wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // everyone's patch pieces are in; the output tile region is free again
        asm volatile("" ::: "memory");
        const bool more = t + step < tw.end;
        if (a.res) {
            // residual tile first (awaited first), then the next tile's patch
            constexpr int NINST = THIN_TY * THIN_TX * 8 / 64;            // 16 instructions, 4 per wave
#pragma unroll
            for (int i = 0; i < NINST / 4; ++i) {
                const int j = w + 4 * i;
                const int row = j * 8 + lane / 8;
                const int c = (lane % 8) ^ (row & 7);
                const long long m = pix(row);
                lds_dma16(m >= 0 ? a.res + (size_t)m * 64 + c * 8 : a.zero, tile + j * 1024);
            }
        }
        if (more) issue_patch(t + step, buf ^ 1);

        const char* pb = lds + buf * THIN_PATCH;
        f32x4 acc[2][4];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) acc[tc][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            bf16x8 xf[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int sl = (wpx * 4 + tp + kh) * THIN_PC + n + kw;
                xf[tp] = *reinterpret_cast<const bf16x8*>(pb + sl * 64 + ((q ^ ((sl >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[tap][tc], xf[tp], acc[tc][tp], 0, 0, 0);
        }
        if (a.res) {
            if (more) wait_vmcnt<3>();                 // the residual has landed; the next patch may still fly
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int slot = (wpx * 4 + tp) * THIN_TX + n;
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const int chl = wch * 32 + tc * 16 + cq;
                const f32x4 bs = bsv[tc];
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[tc][tp][r] + bs[r];
                    if (a.leaky) x = x > 0.f ? x : 0.1f * x;
                    v[r] = x;
                }
                tile_put<64>(tile, slot, chl, v, a.res != nullptr);
            }
        }
        // the next tile's patch pieces (issued at the top of this tile) and the last tile's stores: both old by now
        wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        tile_store<THIN_TY * THIN_TX, 64, 256>(tile, a.dst, 64, 0, pix);
    }
}

// ---------------------------------------------------------------- first residual block in one launch
// Block = 1x1 conv 64 -> 32, 3x3 conv 32 -> 64, shortcut from the block's input
// (/root/reference/yolo_v3.py:452-473 over a "conv 32/1x1, conv 64/3x3, shortcut from=-3" group of
// the cfg).  As two launches the 1x1 layer reads the 64-channel input (128 bytes per pixel), writes
// its 32 channels, and the 3x3 layer reads those back plus the 64-channel input AGAIN as the
// shortcut operand.  Here a persistent workgroup walks 8x16-pixel output tiles like the thin
// kernel; per tile the 10x18-pixel patch of the 64-channel INPUT comes in once by LDS-DMA
// (double buffered), the 1x1 layer is computed for all 180 patch pixels (27 % more than the
// tile's own 128: the halo) into the 32-channel patch the nine taps read (zero where the 3x3
// layer pads: outside the image the 1x1 layer's output does not exist), and the shortcut
// operand is the centre of the input patch, already in LDS.  The 1x1 layer's output never goes
// to memory and the block's input is read once: 402 MB instead of 732 MB at batch 32 x 208^2.
// Same bits as the two launches: the 1x1 sums run over K-steps of 32 in order from zero, then
// bias, LeakyReLU, ONE rounding to bf16; the 3x3 part is the thin kernel's sequence.
//   in64  two buffers of 192 pixel slots x 128 B; the 16-byte chunk c of slot s sits at
//         position c ^ ((s >> 1) & 7): the 16 consecutive slots a fragment read covers hit 16
//         distinct 16-byte bank slots.  Filled by 24 LDS-DMA instructions of 8 slots each.
//   p32   192 slots x 64 B in the thin kernel's layout (chunk q at q ^ ((s >> 2) & 3)); the
//         output tile (16 KiB) takes its place once the taps are done.
constexpr int BLK_IN = THIN_SLOTS * 128;                   // 24 KiB per input patch buffer

struct BlockArgs {
    const __bf16* src;     // (B, H, W, 64): the block's input
    const __bf16* w1;      // [2][32][32]: 1x1 weights, K-tile major
    const float* b1;
    const __bf16* w3;      // [9][64][32]
    const float* b3;
    __bf16* dst;           // (B, H, W, 64)
    const __bf16* zero;
    int B, H, W, leaky1, leaky3;
    int tiles_x, tiles_y;
};

__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
conv_block64_kernel(const BlockArgs a) {
    __shared__ __attribute__((aligned(1024))) char lds[2 * BLK_IN + THIN_TILE];
    char* const p32 = lds + 2 * BLK_IN;                    // 32-channel patch, later the output tile
    static_assert(THIN_PATCH <= THIN_TILE, "the output tile covers the 32-channel patch it replaces");
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, q = lane >> 4;
    const int wpx = w & 1, wch = w >> 1;                   // 3x3 part: pixel half (4 rows), channel half (32)

    // weights as MFMA A fragments, resident for the whole kernel
    bf16x8 w3f[9][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
            w3f[tap][tc] = *reinterpret_cast<const bf16x8*>(a.w3 + ((tap * 64 + wch * 32 + tc * 16 + n) * 32 + q * 8));
    bf16x8 w1f[2][2];                                      // [K-step][channel group of 16]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int cg = 0; cg < 2; ++cg)
            w1f[ks][cg] = *reinterpret_cast<const bf16x8*>(a.w1 + ((ks * 32 + cg * 16 + n) * 32 + q * 8));
    f32x4 b1v[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) b1v[cg] = *reinterpret_cast<const f32x4*>(a.b1 + cg * 16 + q * 4);

    // input patch fill: instruction j (of 24) covers slots 8j..8j+7, eight lanes per pixel;
    // wave w issues j = w, w+4, ..., w+20
    // (packed: column | row << 8 | chunk << 16, bit 31 = unused slot, which reads the zero page)
    uint32_t pinfo[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int sl = 8 * (w + 4 * i) + (lane >> 3);
        const int py = sl / THIN_PC;
        pinfo[i] = (uint32_t)(sl - py * THIN_PC) | ((uint32_t)py << 8) | ((uint32_t)((lane & 7) ^ ((sl >> 1) & 7)) << 16) |
                   (sl < THIN_PR * THIN_PC ? 0u : 0x80000000u);
    }
    const int per_image = a.tiles_x * a.tiles_y, ntiles = per_image * a.B;
    auto issue_patch = [&](int t, int buf) {
        const int b = t / per_image, t2 = t - b * per_image;
        const int ty = t2 / a.tiles_x, tx = t2 - ty * a.tiles_x;
        const int y0 = ty * THIN_TY - 1, x0 = tx * THIN_TX - 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int iy = y0 + (int)((pinfo[i] >> 8) & 255u), ix = x0 + (int)(pinfo[i] & 255u);
            const bool ok = (int)pinfo[i] >= 0 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const __bf16* p = ok ? a.src + (((size_t)b * a.H + iy) * a.W + ix) * 64 + ((pinfo[i] >> 16) & 7u) * 8 : a.zero;
            lds_dma16(p, lds + buf * BLK_IN + (w + 4 * i) * 1024);
        }
    };
    const TileWalk tw = tile_walk(ntiles);
    if (tw.first >= tw.end) return;
    const int step = tw.step;
    issue_patch(tw.first, 0);
    int buf = 0;
    const int cq = q * 4;
    const float slope1 = a.leaky1 ? 0.1f : 1.0f;
    for (int t = tw.first; t < tw.end; t += step, buf ^= 1) {
        const int b = t / per_image, t2 = t - b * per_image;
        const int oy0 = (t2 / a.tiles_x) * THIN_TY, ox0 = (t2 % a.tiles_x) * THIN_TX;
        auto pix = [&](int slot) -> long long {
            const int y = oy0 + slot / THIN_TX, x = ox0 + slot % THIN_TX;
            return (y < a.H && x < a.W) ? ((long long)b * a.H + y) * a.W + x : -1;
        };
        wait_vmcnt<0>();                               // this tile's input patch (and the stores of the last tile)
        __builtin_amdgcn_s_barrier();                  // ... everyone's; the last tile's output tile has been read out
        asm volatile("" ::: "memory");
        if (t + step < tw.end) issue_patch(t + step, buf ^ 1);
        const char* const ib = lds + buf * BLK_IN;

        // ---- the 1x1 layer on the whole patch: pixel groups w, w+4, w+8
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int sl = 16 * (w + 4 * i) + n;
            bf16x8 xf[2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                xf[ks] = *reinterpret_cast<const bf16x8*>(ib + sl * 128 + (((ks * 4 + q) ^ ((sl >> 1) & 7)) << 4));
            const int py = sl / THIN_PC, px = sl - py * THIN_PC;
            const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
            const bool inside = sl < THIN_PR * THIN_PC && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[0][cg], xf[0], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[1][cg], xf[1], acc, 0, 0, 0);
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[e] + b1v[cg][e];
                    x = x > 0.f ? x : slope1 * x;
                    pk[e] = inside ? (__bf16)x : (__bf16)0.f;
                }
                // channels cg*16 + 4q .. +3 of pixel slot sl: chunk 2 cg + (q >> 1), half q & 1
                *reinterpret_cast<bf16x4*>(p32 + sl * 64 + (((cg * 2 + (q >> 1)) ^ ((sl >> 2) & 3)) << 4) + (q & 1) * 8) = pk;
            }
        }
        __syncthreads();

        // ---- the 3x3 layer: nine taps from the 32-channel patch
        f32x4 acc[2][4];
#pragma unroll
        for (int tc = 0; tc < 2; ++tc)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) acc[tc][tp] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int kh = tap / 3, kw = tap - 3 * kh;
            bf16x8 xf[4];
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const int sl = (wpx * 4 + tp + kh) * THIN_PC + n + kw;
                xf[tp] = *reinterpret_cast<const bf16x8*>(p32 + sl * 64 + ((q ^ ((sl >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
#pragma unroll
                for (int tp = 0; tp < 4; ++tp)
                    acc[tc][tp] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w3f[tap][tc], xf[tp], acc[tc][tp], 0, 0, 0);
        }
        __syncthreads();                               // everyone is done with the 32-channel patch

        // ---- bias, LeakyReLU, shortcut (the centre of the input patch), one rounding
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const int slot = (wpx * 4 + tp) * THIN_TX + n;
            const int sc = (wpx * 4 + tp + 1) * THIN_PC + n + 1;          // the same pixel in the input patch
#pragma unroll
            for (int tc = 0; tc < 2; ++tc) {
                const int chl = wch * 32 + tc * 16 + cq;
                const f32x4 bs = *reinterpret_cast<const f32x4*>(a.b3 + chl);
                const bf16x4 rr = *reinterpret_cast<const bf16x4*>(ib + sc * 128 + (((chl >> 3) ^ ((sc >> 1) & 7)) << 4) +
                                                                   (chl & 7) * 2);
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = acc[tc][tp][r] + bs[r];
                    if (a.leaky3) x = x > 0.f ? x : 0.1f * x;
                    v[r] = x + (float)rr[r];
                }
                tile_put<64>(p32, slot, chl, v, false);
            }
        }
        __syncthreads();
        tile_store<THIN_TY * THIN_TX, 64, 256>(p32, a.dst, 64, 0, pix);
    }
}

// ---------------------------------------------------------------- launch
static int pick_bk(int c0, int c1) { return (c0 % 64 == 0 && c1 % 64 == 0) ? 64 : 32; }
static int pick_bn(int cout) { return cout <= 32 ? 32 : (cout <= 64 ? 64 : 128); }

// ---------------------------------------------------------------- split K: the second half
// At small batch the 13x13 and 26x26 layers have a handful of output tiles and a
// reduction of up to 4608: a few workgroups walk 144 K-steps while most of the chip
// idles.  Their K range is cut into slices (grid.y), every slice writes its raw fp32
// sums, and this kernel adds the slices in ascending order and applies what the
// single-pass epilogue applies: bias, LeakyReLU, residual, ONE rounding to bf16 --
// or, for a detection convolution, the head decode into the prediction tensor.
struct FinishArgs {
    const float* partial;      // [slices][M][Cout]
    const float* bias;
    const __bf16* res;
    void* dst;
    int M, Cout, slices, leaky, dst_fp32;
    int HoWo, Wo;
    FastDiv divCout, divHoWo, divWo, divD;
    int dec_D, dec_rows, dec_row_offset;
    float dec_stride;
    float dec_aw[YOLO_AMD_MAX_ANCHORS], dec_ah[YOLO_AMD_MAX_ANCHORS];
};

__global__ void __launch_bounds__(256)
conv_splitk_finish_kernel(const FinishArgs a) {
    const uint32_t total = (uint32_t)a.M * (uint32_t)a.Cout;
    const uint32_t e = blockIdx.x * 256u + threadIdx.x;
    if (e >= total) return;
    const uint32_t m = fdiv_u32(e, a.divCout);
    const int ch = (int)(e - m * (uint32_t)a.Cout);
    float x = a.partial[e];
    for (int z = 1; z < a.slices; ++z) x += a.partial[(size_t)z * total + e];
    x += a.bias[ch];
    if (a.leaky) x = x > 0.f ? x : 0.1f * x;
    if (a.dec_D) {
        const uint32_t b = fdiv_u32(m, a.divHoWo);
        const uint32_t pix = m - b * (uint32_t)a.HoWo;
        const uint32_t oy = fdiv_u32(pix, a.divWo);
        const float gx = (float)(int)(pix - oy * (uint32_t)a.Wo), gy = (float)(int)oy;
        const int ai = (int)fdiv_u32((uint32_t)ch, a.divD);
        const int d = ch - ai * a.dec_D;
        const float anc = d == 2 ? a.dec_aw[ai] : a.dec_ah[ai];
        const long long off = ((long long)b * a.dec_rows + a.dec_row_offset) * a.dec_D + (long long)pix * a.Cout;
        reinterpret_cast<float*>(a.dst)[off + ch] = decode_value(x, d, gx, gy, anc, a.dec_stride);
        return;
    }
    if (a.res) x += (float)a.res[e];
    if (a.dst_fp32) reinterpret_cast<float*>(a.dst)[e] = x;
    else reinterpret_cast<__bf16*>(a.dst)[e] = (__bf16)x;
}

template <int BM, int BN, int BK, int WAVES_M, int NSTAGE, int AMUL = 1, bool TAIL = false>
static void launch_igemm(ConvArgs a, hipStream_t stream) {
    a.tiles_m = (a.M + BM - 1) / BM;
    if (a.kslices > 1) a.ksplit = a.ksplit * 32 / BK;          // planned in steps of 32
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WAVES_M, NSTAGE, AMUL, TAIL>),
                       dim3(a.tiles_m * a.tiles_n, a.kslices), dim3(256), 0, stream, a);
}

// Development knob: YOLO_AMD_CONV_NSTAGE / YOLO_AMD_CONV_BK override the ring
// depth / K tile of the 128-channel-tile kernel so variants can be timed
// against each other in one process (tools/conv_bench.py).
// The environment is read once per knob (a forward pass makes several hundred
// look-ups, which at batch 1 would cost as much host time as the launches);
// yolo_amd_reload_knobs() makes the next look-ups read it again.
struct KnobSlot {
    const char* name;      // the call site's string literal
    bool present;
    int value;
};
static KnobSlot g_knobs[64];
static int g_nknobs = 0;
static int g_knob_gen = 0;             // bumped by yolo_amd_reload_knobs()
static std::mutex g_knob_mutex;

static int env_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lock(g_knob_mutex);
    for (int i = 0; i < g_nknobs; ++i)
        if (g_knobs[i].name == name) return g_knobs[i].present ? g_knobs[i].value : dflt;
    const char* v = getenv(name);
    if (g_nknobs < 64) g_knobs[g_nknobs++] = KnobSlot{name, v != nullptr, v ? atoi(v) : 0};
    return v ? atoi(v) : dflt;
}

#ifdef YOLO_AMD_DEV_TRACE
// a device address handed over in the environment (tools/conv3_trace.py builds its own
// copy of the library with this enabled; the shipped library never reads an address
// from the environment); cached like the other knobs
static void* env_ptr(const char* name) {
    static std::mutex m;
    static const char* cached_name = nullptr;
    static void* cached = nullptr;
    static int seen_gen = -1;
    std::lock_guard<std::mutex> lock(m);
    if (cached_name != name || seen_gen != g_knob_gen) {
        const char* v = getenv(name);
        cached = v ? (void*)strtoull(v, nullptr, 0) : nullptr;
        cached_name = name;
        seen_gen = g_knob_gen;
    }
    return cached;
}
#endif

// K-steps per slice and number of slices for a layer of `wgs` output tiles and KT
// K-steps (of 32), or slices = 1: split only when the tiles alone leave most CUs
// idle and the reduction is long enough to pay for the second launch (measured at
// batch 1: a K-step costs a lone workgroup about 0.3 us, the finishing launch 7-8 us;
// 3x3 layers on >= 256 channels gain 8-24 us each, 36-step layers and every 1x1 layer
// lose), keep at least four steps per slice, an even number of them (the two-step
// activation tile), and no more slices than fill the chip about twice.
static void splitk_plan(long long wgs, int KT, int* steps, int* slices) {
    *steps = KT; *slices = 1;
    // (YOLO_AMD_SPLITK_MAX_WGS / _MIN_KT / _FILL: development knobs for these three numbers)
    if (wgs > env_int("YOLO_AMD_SPLITK_MAX_WGS", 128) || KT < env_int("YOLO_AMD_SPLITK_MIN_KT", 64)) return;
    int want = (int)std::min<long long>(16, std::max<long long>(1, env_int("YOLO_AMD_SPLITK_FILL", 512) / wgs));
    int len = (KT + want - 1) / want;
    len = std::max(4, (len + 1) & ~1);
    const int n = (KT + len - 1) / len;
    if (n >= 2) { *steps = len; *slices = n; }
}

// Channel tile of the im2col kernel for a layer with M output pixels.
static int choose_bn(const yolo_amd_conv_desc& d, long long M) {
    int BN = pick_bn(d.cout);
    // On the small feature maps 128x128 tiles give fewer than ~1.5 workgroups
    // per CU; 64-channel tiles double the count (measured 5-8 % faster on the
    // 13x13 layers).  YOLO_AMD_CONV_BN forces a narrower tile (development knob).
    const long long groups128 = ((M + 127) / 128) * (d.cout_pad / 128);
    int want_bn = env_int("YOLO_AMD_CONV_BN", 0);
    if (want_bn == 0 && BN == 128 && groups128 < env_int("YOLO_AMD_CONV_BN128_MIN", 384)) want_bn = 64;
    // One or two images: even 64-channel tiles leave most CUs idle, and a lone workgroup's
    // K loop is bound by the bytes it pulls through its own CU, so 32-channel tiles (twice
    // the workgroups, each loading less) shorten it: 7-9 % on the batch-1 forward pass,
    // neutral from batch 2 on at this threshold, 3-9 % slower at batch 2-4 with 128 or more
    // (YOLO_AMD_CONV_BN64_MIN moves it, counted in 128 x 64 tiles).
    if ((want_bn == 0 || want_bn == 64) && BN >= 64 && env_int("YOLO_AMD_CONV_BN", 0) == 0 &&
        ((M + 127) / 128) * (d.cout_pad / 64) < env_int("YOLO_AMD_CONV_BN64_MIN", 64))
        want_bn = 32;
    if ((want_bn == 32 || want_bn == 64) && want_bn < BN && d.cout_pad % want_bn == 0) BN = want_bn;
    return BN;
}

static long long out_pixels(const yolo_amd_conv_desc& d) {
    const int pad = d.ksize / 2;
    const int Ho = (d.H + 2 * pad - d.ksize) / d.stride + 1;
    const int Wo = (d.W + 2 * pad - d.ksize) / d.stride + 1;
    return (long long)d.B * Ho * Wo;
}

// Slices the launcher would cut this layer's reduction into (1: no split).
static void splitk_for(const yolo_amd_conv_desc& d, int* steps, int* slices) {
    *steps = 0; *slices = 1;
    if (d.first_layer || d.cout_pad <= 0 || (d.ksize != 1 && d.ksize != 3) || !env_int("YOLO_AMD_SPLITK", 1)) return;
    const long long M = out_pixels(d);
    if (M <= 0) return;
    const int BN = choose_bn(d, M);
    if (d.cout_pad % BN) return;
    splitk_plan(((M + 127) / 128) * (d.cout_pad / BN), d.ksize * d.ksize * (d.c0 + d.c1) / 32, steps, slices);
}

// What launch_conv will run for a layer: the kernel family and its tile
// parameters.  yolo_amd_conv_query hands this to callers, so a test can assert
// which kernel a shape exercises.
struct ConvChoice {
    int kernel = 0;                 // YOLO_AMD_KERNEL_*
    int bm = 0, bn = 0, bk = 0;     // output pixels x channels per workgroup, K step
    int amul = 1, nstage = 0;       // im2col kernel: activation tile depth in K steps, LDS ring depth
    int waves = 4;                  // waves per workgroup
    int stg = 0;                    // patch kernel: taps per weight stage
    int th = 0, tw = 0;             // patch kernel: output tile, rows x columns
    int flat = 0;                   // patch kernel: flat patch planes (conv3x3_flat_kernel)
    int ksteps = 0, kslices = 1;    // split reduction: K-steps (of 32) per slice, slices
    int Ho = 0, Wo = 0;
    long long M = 0;
};

// Validates the descriptor and decides the kernel.  with_ptrs = false checks the
// shape fields only (the query entry point works on descriptors without buffers).
static int choose_conv(const yolo_amd_conv_desc& d, bool with_ptrs, ConvChoice* out) {
    ConvChoice c;
    if (with_ptrs) YA_REQUIRE(d.src0 && d.weight && d.bias && d.dst, "conv: null pointer");
    YA_REQUIRE(d.B > 0 && d.H > 0 && d.W > 0 && d.cout > 0, "conv: bad shape");
    YA_REQUIRE((d.ksize == 1 || d.ksize == 3) && (d.stride == 1 || d.stride == 2),
               "conv: ksize=%d stride=%d unsupported", d.ksize, d.stride);
    const int pad = d.ksize / 2;
    c.Ho = (d.H + 2 * pad - d.ksize) / d.stride + 1;
    c.Wo = (d.W + 2 * pad - d.ksize) / d.stride + 1;
    c.M = (long long)d.B * c.Ho * c.Wo;
    YA_REQUIRE(c.M < (1ll << 31), "conv: too many output pixels");
    if (d.first_layer) {
        YA_REQUIRE(d.ksize == 3 && d.stride == 1 && d.c0 >= 1 && d.c0 <= 3 && d.c1 == 0 && !d.residual &&
                       !(d.dst_fp32 && d.fuse_next) && d.up0 == 0 && !d.decode,
                   "conv: first layer must be 3x3 stride 1 with at most 3 input channels");
        YA_REQUIRE(d.cout == 32 || d.cout == 16,
                   "conv: first layer with %d output channels is not instantiated (16 or 32)", d.cout);
        c.kernel = d.fuse_next ? YOLO_AMD_KERNEL_STEM : YOLO_AMD_KERNEL_FIRST;
        c.bm = d.fuse_next ? STEM_TY * STEM_TX : FIRST_TH * FIRST_TW;
        c.bn = d.fuse_next ? 64 : d.cout;
        c.bk = 32;
        *out = c;
        return YA_OK;
    }
    YA_REQUIRE(d.c0 % 32 == 0 && d.c1 % 32 == 0 && d.c0 > 0 && d.c1 >= 0,
               "conv: input channels (%d, %d) must be multiples of 32", d.c0, d.c1);
    if (with_ptrs) YA_REQUIRE(d.c1 == 0 || d.src1, "conv: src1 missing");
    YA_REQUIRE(d.up0 == 0 || (d.up0 == 1 && d.ksize == 1 && d.H % 2 == 0 && d.W % 2 == 0),
               "conv: an upsampled source is supported for 1x1 convolutions only");
    if (d.fuse_next && d.ksize == 1) {
        // the first residual block as one launch: this 1x1 layer in front of the 3x3 layer of the next
        // descriptor (launch_block64 checks the pair)
        YA_REQUIRE(d.ksize == 1 && d.stride == 1 && d.c0 == 64 && d.c1 == 0 && d.up0 == 0 && d.cout == 32 &&
                       d.cout_pad == 32 && !d.dst_fp32 && !d.decode && !d.residual,
                   "conv: fuse_next on a layer that is not the first layer needs a 1x1 stride-1 64 -> 32 layer");
        c.kernel = YOLO_AMD_KERNEL_BLOCK;
        c.bm = THIN_TY * THIN_TX; c.bn = 64; c.bk = 32;
        *out = c;
        return YA_OK;
    }
    int BK = pick_bk(d.c0, d.c1);
    const int BN = choose_bn(d, c.M);
    // K tile 32 keeps LDS at 32 KiB per workgroup (more resident workgroups);
    // 64 only pays for the widest inputs
    const int want_bk = env_int("YOLO_AMD_CONV_BK", (d.c0 + d.c1 >= 512) ? 64 : 32);
    if (BK == 64 && want_bk == 32) BK = 32;
    const int K = d.ksize * d.ksize * (d.c0 + d.c1);
    YA_REQUIRE(d.k_pad == K && K % BK == 0, "conv: k_pad=%d, expected %d", d.k_pad, K);
    YA_REQUIRE(d.cout_pad % BN == 0 && d.cout_pad >= d.cout, "conv: cout_pad=%d not a multiple of %d", d.cout_pad, BN);
    YA_REQUIRE((long long)d.B * d.H * d.W * (d.c0 + d.c1) < (1ll << 31) && (long long)d.cout_pad * K < (1ll << 31),
               "conv: tensor too large for 32-bit element offsets");
    if (d.decode) {
        // the head decode (yolo_v3.predict) lives in the im2col kernel's fp32 epilogue
        // and in the split reduction's finishing kernel only: a detection layer never
        // takes the thin or the patch kernel
        const yolo_amd_head_desc& hd = *d.decode;
        const int A = d.decode_anchors;
        YA_REQUIRE(d.dst_fp32 && !d.residual && A >= 1 && A <= YOLO_AMD_MAX_ANCHORS && d.cout % A == 0 &&
                       d.cout / A >= 5 && c.Ho == c.Wo && hd.G == c.Ho,
                   "conv: fused decode needs an fp32, residual-free layer with cout = A*D on a GxG map");
        YA_REQUIRE(hd.row_offset >= 0 && hd.row_offset + c.Ho * c.Wo * A <= d.decode_rows,
                   "conv: fused decode rows [%d, %d) fall outside %d", hd.row_offset,
                   hd.row_offset + c.Ho * c.Wo * A, d.decode_rows);
    }
    // 3x3 stride-1 32 -> 64 channels: the thin kernel (YOLO_AMD_THIN=0: development knob)
    if (d.ksize == 3 && d.stride == 1 && d.c0 == 32 && d.c1 == 0 && d.up0 == 0 && d.cout == 64 &&
        d.cout_pad == 64 && !d.dst_fp32 && !d.decode && env_int("YOLO_AMD_THIN", 1)) {
        c.kernel = YOLO_AMD_KERNEL_THIN;
        c.bm = THIN_TY * THIN_TX; c.bn = 64; c.bk = 32;
        *out = c;
        return YA_OK;
    }
    // Stride-1 3x3 layers on 13*2^n feature maps can take the patch-reuse kernel.
    // Measured at batch 32 (tools/conv_bench.py): its eight-wave form with row-deep
    // weight stages wins on every map from 13x13 (one workgroup per CU: 59.5 us
    // against 75.3 us for the im2col kernel) to 104x104 (76.9 against 79.5 us for the
    // four-wave form); it keeps winning down to about one workgroup per two CUs
    // (batch 4-8 at 416: 2-4 % on the whole forward pass), below that the im2col
    // kernel's smaller tiles and its split reductions fill the chip better
    // (YOLO_AMD_CONV3_MIN moves the threshold).  With tap-deep stages (YOLO_AMD_CONV3_STG=1) the four-wave
    // form takes the layers with more than six tiles per CU.  YOLO_AMD_CONV3=0/1 and
    // YOLO_AMD_CONV3_WAVES=4/8 force the choice (development knobs).
    // Tile shapes: 13 x 13 on 13 * 2^n maps (416-pixel inputs), 19 x 10 on 19 * 2^n maps (608).
    int th = 0, tw = 0;
    if (d.H % 13 == 0 && d.W % 13 == 0) { th = 13; tw = 13; }
    else if (d.H % 19 == 0 && d.W % 19 == 0) { th = 19; tw = 10; }
    const long long p3_groups = th ? (long long)d.B * ((d.H + th - 1) / th) * ((d.W + tw - 1) / tw) * (d.cout_pad / 128) : 0;
    const int p3 = env_int("YOLO_AMD_CONV3", -1);
    if (d.ksize == 3 && d.stride == 1 && d.c1 == 0 && d.up0 == 0 && th != 0 &&
        d.cout_pad % 128 == 0 && !d.decode &&
        (p3 == 1 || (p3 < 0 && p3_groups >= env_int("YOLO_AMD_CONV3_MIN", 128)))) {
        c.kernel = YOLO_AMD_KERNEL_PATCH;
        c.th = th; c.tw = tw;
        // Private weight rings come with flat patch planes (conv3x3_flat_kernel: no address arithmetic in
        // the tap loop; equal on the 13 x 13 tiles, 4-8 % faster on the 19 x 10 ones, whose planes shrink
        // from 6 to 4 KiB: rings three taps deep and the barrier-free epilogue fit) unless
        // YOLO_AMD_CONV3_FLAT=0 asks for the rotated 16-slot rows, a ring depth or a timing variant of the
        // older kernel is selected, or the input is too large for the kernel's 32-bit buffer offsets.
        const bool flat_wanted = env_int("YOLO_AMD_CONV3_FLAT", 1) != 0 && env_int("YOLO_AMD_CONV3_RING", 3) == 3 &&
                                 env_int("YOLO_AMD_CONV3_VAR", 0) == 0 &&
                                 (long long)d.B * d.H * d.W * d.c0 * 2 < (long long)kBufferNowhere;
        c.bm = th * tw; c.bn = 128; c.bk = 32;
        if (th != 13) {
            // the 21-row patch buffers leave room for a tap-deep weight ring only
            // (64 KiB, two workgroups per CU), or for private rings two taps deep (80 KiB)
            c.stg = env_int("YOLO_AMD_CONV3_STG", 9) == 9 ? 9 : 1; c.waves = 8;
            c.flat = c.stg == 9 && flat_wanted;
            *out = c;
            return YA_OK;
        }
        {
            // Private weight rings (9: counted waits, one barrier per nine taps, the wave's own
            // residual pieces fetched during the last chunk) against row-deep shared stages (3):
            // 75.0 / 79.1 us on 64 -> 128 @ 104x104, 62.2 / 65.1 on 128 -> 256 @ 52x52, 54.8 / 57.9
            // on 256 -> 512 @ 26x26, 56.0 / 61.8 on 512 -> 1024 @ 13x13.
            const int stg = env_int("YOLO_AMD_CONV3_STG", 9);
            c.stg = stg == 9 ? 9 : (stg == 3 ? 3 : 1);
        }
        c.waves = env_int("YOLO_AMD_CONV3_WAVES", (c.stg >= 3 || p3_groups <= 1536) ? 8 : 4) == 8 ? 8 : 4;
        if (c.waves == 4) c.stg = 1;
        // 256-channel workgroups (one per CU, 24 MFMAs per 10 fragment reads instead of 12 per 8):
        // measured no faster on 26x26 (55.4 against 56.5 us) and slower on 52x52 and 13x13
        // (66.7 against 61.4, 81 against 61 us), so only YOLO_AMD_CONV3_TC=4 selects them
        // (development knob; the variant stays under test)
        if (c.waves == 8 && (c.stg == 3 || c.stg == 9) && d.cout_pad % 256 == 0 && env_int("YOLO_AMD_CONV3_TC", 2) == 4)
            c.bn = 256;
        c.flat = c.stg == 9 && c.waves == 8 && c.bn == 128 && flat_wanted;
        *out = c;
        return YA_OK;
    }
    c.kernel = YOLO_AMD_KERNEL_IGEMM;
    c.bm = 128; c.bn = BN; c.bk = BK;
    // Defaults from A/B timing on MI355X (tools/conv_bench.py): a two-stage ring
    // with three or four resident workgroups per CU beats deeper rings at one
    // workgroup per CU on every layer shape of the network.
    c.nstage = env_int("YOLO_AMD_CONV_NSTAGE", 2);
    // two-step activation tile (whole 128-byte lines per pixel): needs both
    // sources in whole 64-channel groups; not worth it for the shortest K loops
    const int amul = env_int("YOLO_AMD_CONV_AMUL", K >= 256 ? 2 : 1);
    c.amul = (amul == 2 && BK == 32 && c.nstage == 2 && d.c0 % 64 == 0 && d.c1 % 64 == 0) ? 2 : 1;
    if (c.amul == 1) {
        // instantiated ring depths per tile shape
        const bool deep3 = (BK == 32 && BN != 32) || (BN >= 64 && BK == 64);
        const bool deep4 = !(BN == 32 && BK == 32);
        if (!((c.nstage == 3 && deep3) || (c.nstage == 4 && deep4))) c.nstage = 2;
    }
    // Split reduction (see conv_splitk_finish_kernel): when the caller lent a
    // workspace and the layer has too few output tiles to occupy the chip.
    if (d.splitk_ws) {
        int steps, slices;
        splitk_for(d, &steps, &slices);
        if (slices > 1 && (size_t)slices * (size_t)c.M * d.cout * sizeof(float) <= d.splitk_ws_bytes) {
            YA_REQUIRE(((uintptr_t)d.splitk_ws & 15) == 0, "conv: split-K workspace must be 16-byte aligned");
            c.ksteps = steps; c.kslices = slices;
        }
    }
    *out = c;
    return YA_OK;
}

// tail: a 1x1 layer to run in this layer's epilogue (launch_conv_tail has checked that the kernel can)
static int launch_conv(const yolo_amd_conv_desc& d, const void* zero_page, hipStream_t stream,
                       const yolo_amd_conv_desc* tail = nullptr) {
    ConvChoice c;
    const int rc = choose_conv(d, true, &c);
    if (rc != YA_OK) return rc;
    const int Ho = c.Ho, Wo = c.Wo;
    const long long M = c.M;
    if (c.kernel == YOLO_AMD_KERNEL_FIRST) {
        const long long src_bytes = (long long)d.B * d.c0 * d.H * d.W * 4, dst_bytes = M * d.cout * (d.dst_fp32 ? 4 : 2);
        YA_REQUIRE(src_bytes < (1ll << 31) && dst_bytes < 0xFFFFFF00ll,
                   "conv: first layer input must be smaller than 2 GiB and its output smaller than 4 GiB");
        const int tiles_x = (d.W + FIRST_TW - 1) / FIRST_TW, tiles_y = (d.H + FIRST_TH - 1) / FIRST_TH;
        const long long ntiles = (long long)tiles_x * tiles_y * d.B;
        dim3 grid((unsigned)std::min<long long>(ntiles, 256 * 8 * 4));
        const float slope = d.leaky ? 0.1f : 1.0f;
#define YA_FIRST(CO, F32)                                                                               \
    hipLaunchKernelGGL((conv_first_kernel<CO, F32>), grid, dim3(256), 0, stream, (const float*)d.src0, \
                       (const float*)d.weight, d.bias, d.dst, d.B, d.c0, d.H, d.W, slope,              \
                       (uint32_t)src_bytes, (uint32_t)dst_bytes, tiles_x, tiles_y)
        if (d.cout == 32 && d.dst_fp32) YA_FIRST(32, true);
        else if (d.cout == 32) YA_FIRST(32, false);
        else if (d.dst_fp32) YA_FIRST(16, true);
        else YA_FIRST(16, false);
#undef YA_FIRST
        return YA_OK;
    }
    YA_REQUIRE(c.kernel != YOLO_AMD_KERNEL_STEM && c.kernel != YOLO_AMD_KERNEL_BLOCK,
               "conv: fuse_next is handled by yolo_amd_conv_forward");
    YA_REQUIRE(zero_page, "conv: zero page missing");
    const int K = d.ksize * d.ksize * (d.c0 + d.c1);
    if (c.kernel == YOLO_AMD_KERNEL_THIN) {
        ThinArgs t;
        t.src = (const __bf16*)d.src0; t.wgt = (const __bf16*)d.weight; t.bias = d.bias;
        t.res = (const __bf16*)d.residual; t.dst = (__bf16*)d.dst; t.zero = (const __bf16*)zero_page;
        t.B = d.B; t.H = d.H; t.W = d.W; t.leaky = d.leaky;
        t.tiles_x = (d.W + THIN_TX - 1) / THIN_TX; t.tiles_y = (d.H + THIN_TY - 1) / THIN_TY;
        const long long ntiles = (long long)t.tiles_x * t.tiles_y * d.B;
        hipLaunchKernelGGL(conv3x3_thin_kernel, dim3((unsigned)std::min<long long>(ntiles, 256 * 2)), dim3(256), 0,
                           stream, t);
        return YA_OK;
    }
    if (c.kernel == YOLO_AMD_KERNEL_PATCH) {
        Conv3Args p;
        p.src = (const __bf16*)d.src0; p.wgt = (const __bf16*)d.weight; p.bias = d.bias;
        p.res = (const __bf16*)d.residual; p.dst = d.dst; p.zero = (const __bf16*)zero_page;
        p.C = d.c0; p.H = d.H; p.W = d.W; p.Cout = d.cout; p.K = K; p.leaky = d.leaky; p.dst_fp32 = d.dst_fp32;
        p.tiles_y = (d.H + c.th - 1) / c.th; p.tiles_x = (d.W + c.tw - 1) / c.tw;
        p.ntiles = d.B * p.tiles_y * p.tiles_x;
        p.tiles_n = d.cout_pad / c.bn;
        p.prio = env_int("YOLO_AMD_CONV3_PRIO", 3);
        p.trace = nullptr;
#ifdef YOLO_AMD_DEV_TRACE
        p.trace = (long long*)env_ptr("YOLO_AMD_CONV3_TRACE");
#endif
        p.w2 = nullptr; p.b2 = nullptr; p.dst2 = nullptr; p.leaky2 = 0;
        if (tail) {
            YA_REQUIRE(c.flat && p.tiles_n == 1 && d.cout == 128 && !d.dst_fp32,
                       "conv: this layer's kernel cannot take a 1x1 layer in its epilogue");
            p.w2 = (const __bf16*)tail->weight; p.b2 = tail->bias; p.dst2 = (__bf16*)tail->dst; p.leaky2 = tail->leaky;
            if (c.th == 19)
                hipLaunchKernelGGL((conv3x3_flat_kernel<19, 10, 12, 1, 7, true>), dim3(p.ntiles), dim3(512), 0, stream, p);
            else
                hipLaunchKernelGGL((conv3x3_flat_kernel<13, 13, 15, 3, 7, true>), dim3(p.ntiles), dim3(512), 0, stream, p);
            return YA_OK;
        }
        if (c.flat && c.th == 19)
            hipLaunchKernelGGL((conv3x3_flat_kernel<19, 10, 12, 1, 7>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.flat && env_int("YOLO_AMD_CONV3_SPLIT", 7) == 6)
            hipLaunchKernelGGL((conv3x3_flat_kernel<13, 13, 15, 3, 6>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.flat)
            hipLaunchKernelGGL((conv3x3_flat_kernel<13, 13, 15, 3, 7>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.th == 19 && c.stg == 9)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9, 2, 19, 10, 2>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.th == 19)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 1, 2, 19, 10>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.bn == 256 && c.stg == 9)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9, 4, 13, 13, 3>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.bn == 256)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 3, 4>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.waves == 8 && c.stg == 9 && env_int("YOLO_AMD_CONV3_RING", 3) == 4)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9, 2, 13, 13, 4>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.waves == 8 && c.stg == 9 && env_int("YOLO_AMD_CONV3_RING", 3) == 2)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9, 2, 13, 13, 2>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.waves == 8 && c.stg == 9) {
            switch (env_int("YOLO_AMD_CONV3_VAR", 0)) {
#define YA_P3VAR(V) case V: hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9, 2, 13, 13, 3, V>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p); break
                YA_P3VAR(0x100); YA_P3VAR(0x200); YA_P3VAR(0x300); YA_P3VAR(0x700); YA_P3VAR(0x800); YA_P3VAR(0x900);
#undef YA_P3VAR
                default:
                    hipLaunchKernelGGL((conv3x3_patch_kernel<8, 9>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
            }
        }
        else if (c.waves == 8 && c.stg == 3)
            hipLaunchKernelGGL((conv3x3_patch_kernel<8, 3>), dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else if (c.waves == 8)
            hipLaunchKernelGGL(conv3x3_patch_kernel<8>, dim3(p.ntiles * p.tiles_n), dim3(512), 0, stream, p);
        else
            hipLaunchKernelGGL(conv3x3_patch_kernel<4>, dim3(p.ntiles * p.tiles_n), dim3(256), 0, stream, p);
        return YA_OK;
    }
    const int BN = c.bn, BK = c.bk, ns = c.nstage;
    ConvArgs a;
    a.src0 = (const __bf16*)d.src0; a.src1 = (const __bf16*)d.src1; a.wgt = (const __bf16*)d.weight;
    a.bias = d.bias; a.res = (const __bf16*)d.residual; a.dst = d.dst; a.zero = (const __bf16*)zero_page;
    a.C0 = d.c0; a.C1 = d.c1; a.up0 = d.up0; a.H = d.H; a.W = d.W; a.Ho = Ho; a.Wo = Wo; a.M = (int)M;
    a.Cout = d.cout; a.Kpad = K; a.ksize = d.ksize; a.stride = d.stride; a.leaky = d.leaky;
    a.dst_fp32 = d.dst_fp32;
    a.dec_D = 0; a.dec_rows = 0; a.dec_row_offset = 0; a.dec_stride = 0.f;
    a.divD = make_fastdiv(1u);
    for (int i = 0; i < YOLO_AMD_MAX_ANCHORS; ++i) a.dec_aw[i] = a.dec_ah[i] = 0.f;
    if (d.decode) {
        const yolo_amd_head_desc& hd = *d.decode;
        const int A = d.decode_anchors;
        a.dec_D = d.cout / A; a.dec_rows = d.decode_rows; a.dec_row_offset = hd.row_offset;
        a.dec_stride = hd.stride;
        for (int i = 0; i < A; ++i) { a.dec_aw[i] = hd.anchor_w[i]; a.dec_ah[i] = hd.anchor_h[i]; }
        a.divD = make_fastdiv((uint32_t)a.dec_D);
    }
    constexpr int BM = 128;
    a.tiles_m = (int)((M + BM - 1) / BM);
    a.tiles_n = d.cout_pad / BN;
    a.divHoWo = make_fastdiv((uint32_t)(Ho * Wo));
    a.divWo = make_fastdiv((uint32_t)Wo);

    a.ksplit = 0; a.kslices = 1;
    a.trace = nullptr;
#ifdef YOLO_AMD_DEV_TRACE
    a.trace = (long long*)env_ptr("YOLO_AMD_CONV3_TRACE");
#endif
    a.w2 = nullptr; a.b2 = nullptr; a.dst2 = nullptr; a.leaky2 = 0;
    if (tail) {
        YA_REQUIRE(c.amul == 2 && BN == 128 && c.kslices == 1 && a.tiles_n == 1 && !d.dst_fp32,
                   "conv: this layer's kernel cannot take a 1x1 layer in its epilogue");
        a.w2 = (const __bf16*)tail->weight; a.b2 = tail->bias; a.dst2 = (__bf16*)tail->dst; a.leaky2 = tail->leaky;
        launch_igemm<128, 128, 32, 2, 2, 2, true>(a, stream);
        return YA_OK;
    }
    FinishArgs f{};
    if (c.kslices > 1) {
        f.partial = (const float*)d.splitk_ws; f.bias = a.bias; f.res = a.res; f.dst = a.dst;
        f.M = a.M; f.Cout = a.Cout; f.slices = c.kslices; f.leaky = a.leaky; f.dst_fp32 = a.dst_fp32;
        f.HoWo = Ho * Wo; f.Wo = Wo;
        f.divCout = make_fastdiv((uint32_t)a.Cout); f.divHoWo = a.divHoWo; f.divWo = a.divWo; f.divD = a.divD;
        f.dec_D = a.dec_D; f.dec_rows = a.dec_rows; f.dec_row_offset = a.dec_row_offset;
        f.dec_stride = a.dec_stride;
        for (int i = 0; i < YOLO_AMD_MAX_ANCHORS; ++i) { f.dec_aw[i] = a.dec_aw[i]; f.dec_ah[i] = a.dec_ah[i]; }
        a.ksplit = c.ksteps; a.kslices = c.kslices;
        a.dst = d.splitk_ws; a.dst_fp32 = 1; a.res = nullptr; a.leaky = 0; a.dec_D = 0;
    }
    auto finish = [&]() {
        if (a.kslices > 1) {
            const long long total = (long long)f.M * f.Cout;
            hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                               stream, f);
        }
        return YA_OK;
    };
    if (c.amul == 2) {
        if (BN == 128) launch_igemm<128, 128, 32, 2, 2, 2>(a, stream);
        else if (BN == 64) launch_igemm<128, 64, 32, 2, 2, 2>(a, stream);
        else launch_igemm<128, 32, 32, 4, 2, 2>(a, stream);
        return finish();
    }
    if (BN == 128 && BK == 64) {
        if (ns == 3) launch_igemm<128, 128, 64, 2, 3>(a, stream);
        else if (ns == 4) launch_igemm<128, 128, 64, 2, 4>(a, stream);
        else launch_igemm<128, 128, 64, 2, 2>(a, stream);
    } else if (BN == 128 && BK == 32) {
        if (ns == 3) launch_igemm<128, 128, 32, 2, 3>(a, stream);
        else if (ns == 4) launch_igemm<128, 128, 32, 2, 4>(a, stream);
        else launch_igemm<128, 128, 32, 2, 2>(a, stream);
    } else if (BN == 64 && BK == 64) {
        if (ns == 4) launch_igemm<128, 64, 64, 2, 4>(a, stream);
        else if (ns == 3) launch_igemm<128, 64, 64, 2, 3>(a, stream);
        else launch_igemm<128, 64, 64, 2, 2>(a, stream);
    } else if (BN == 64 && BK == 32) {
        if (ns == 3) launch_igemm<128, 64, 32, 2, 3>(a, stream);
        else if (ns == 4) launch_igemm<128, 64, 32, 2, 4>(a, stream);
        else launch_igemm<128, 64, 32, 2, 2>(a, stream);
    } else if (BN == 32 && BK == 64) {
        if (ns == 4) launch_igemm<128, 32, 64, 4, 4>(a, stream);
        else launch_igemm<128, 32, 64, 4, 2>(a, stream);
    } else {
        launch_igemm<128, 32, 32, 4, 2>(a, stream);
    }
    return finish();
}

}  // namespace yolo_amd

using namespace yolo_amd;

extern "C" int yolo_amd_conv_cout_pad(int cout) {
    if (cout <= 0) return 0;
    const int bn = pick_bn(cout);
    return (cout + bn - 1) / bn * bn;
}

extern "C" void yolo_amd_reload_knobs(void) {
    std::lock_guard<std::mutex> lock(g_knob_mutex);
    g_nknobs = 0;
    ++g_knob_gen;
}

extern "C" size_t yolo_amd_conv_splitk_bytes(const yolo_amd_conv_desc* d) {
    if (!d) return 0;
    int steps, slices;
    splitk_for(*d, &steps, &slices);
    if (slices <= 1) return 0;
    return align_up((size_t)slices * (size_t)out_pixels(*d) * d->cout * sizeof(float), 256);
}

extern "C" int yolo_amd_conv_query(const yolo_amd_conv_desc* d, yolo_amd_conv_choice* out) {
    YA_REQUIRE(d && out, "conv_query: null pointer");
    ConvChoice c;
    yolo_amd_conv_desc q = *d;
    if (q.bn_train) {            // the convolution of such a layer writes the fp32 scratch
        q.dst_fp32 = 1; q.leaky = 0; q.residual = nullptr;
    }
    const int rc = choose_conv(q, false, &c);
    if (rc != YA_OK) return rc;
    out->kernel = c.kernel; out->bm = c.bm; out->bn = c.bn; out->bk = c.bk;
    out->act_ksteps = c.amul; out->lds_stages = c.nstage; out->waves = c.waves;
    out->taps_per_stage = c.stg; out->k_slices = c.kslices; out->patch_flat = c.flat;
    return YA_OK;
}

// descs[0] (first layer, fuse_next set) and descs[1] as one launch
static int launch_stem(const yolo_amd_conv_desc& d0, const yolo_amd_conv_desc& d1, const yolo_amd_conv_desc* d2,
                       hipStream_t stream) {
    YA_REQUIRE(d0.first_layer && d0.ksize == 3 && d0.stride == 1 && d0.c0 >= 1 && d0.c0 <= 3 && d0.c1 == 0 &&
                   d0.cout == 32 && !d0.residual && !d0.dst_fp32 && d0.up0 == 0,
               "conv: the fused stem needs a 3x3 stride-1 first layer with <= 3 input and 32 output channels");
    YA_REQUIRE(!d1.first_layer && d1.ksize == 3 && d1.stride == 2 && d1.c0 == 32 && d1.c1 == 0 && d1.cout == 64 &&
                   d1.cout_pad == 64 && !d1.residual && !d1.dst_fp32 && d1.up0 == 0 && d1.B == d0.B &&
                   d1.H == d0.H && d1.W == d0.W && d1.k_pad == 288 && !d1.decode,
               "conv: the fused stem needs a 3x3 stride-2 32 -> 64 layer on the first layer's output");
    YA_REQUIRE(d0.src0 && d0.weight && d0.bias && d1.weight && d1.bias && d1.dst, "conv: null pointer");
    const long long src_bytes = (long long)d0.B * d0.c0 * d0.H * d0.W * 4;
    YA_REQUIRE(src_bytes < (1ll << 31), "conv: first layer input must be smaller than 2 GiB");
    StemArgs a;
    a.src = (const float*)d0.src0; a.w0 = (const float*)d0.weight; a.b0 = d0.bias;
    a.w1 = (const __bf16*)d1.weight; a.b1 = d1.bias; a.dst = (__bf16*)d1.dst;
    a.B = d0.B; a.Cin = d0.c0; a.H = d0.H; a.W = d0.W;
    a.Ho = (d0.H + 2 - 3) / 2 + 1; a.Wo = (d0.W + 2 - 3) / 2 + 1;
    YA_REQUIRE((long long)a.B * a.Ho * a.Wo * 64 < (1ll << 31), "conv: stem output too large");
    a.slope0 = d0.leaky ? 0.1f : 1.0f; a.leaky1 = d1.leaky;
    a.src_bytes = (uint32_t)src_bytes;
    a.tiles_x = (a.Wo + STEM_TX - 1) / STEM_TX; a.tiles_y = (a.Ho + STEM_TY - 1) / STEM_TY;
    const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.B;
    dim3 grid((unsigned)std::min<long long>(ntiles, 256 * 2));
    a.w2 = nullptr; a.b2 = nullptr; a.dst2 = nullptr; a.leaky2 = 0;
    if (d2) {
        YA_REQUIRE(!d2->first_layer && d2->ksize == 1 && d2->stride == 1 && d2->c0 == 64 && d2->c1 == 0 &&
                       d2->cout == 32 && d2->cout_pad == 32 && d2->k_pad == 64 && !d2->residual && !d2->dst_fp32 &&
                       d2->up0 == 0 && !d2->decode && !d2->fuse_next && !d2->bn_train && d2->B == d0.B &&
                       d2->H == a.Ho && d2->W == a.Wo && d2->src0 == d1.dst,
                   "conv: the fused stem's third layer must be a 1x1 stride-1 64 -> 32 layer on the second layer's output");
        YA_REQUIRE(d2->weight && d2->bias && d2->dst, "conv: null pointer");
        a.w2 = (const __bf16*)d2->weight; a.b2 = d2->bias; a.dst2 = (__bf16*)d2->dst; a.leaky2 = d2->leaky;
        hipLaunchKernelGGL(conv_stem_kernel<true>, grid, dim3(256), 0, stream, a);
    } else {
        hipLaunchKernelGGL(conv_stem_kernel<false>, grid, dim3(256), 0, stream, a);
    }
    return YA_OK;
}

// descs[0] (1x1, 64 -> 32, fuse_next set) and descs[1] (3x3, 32 -> 64, shortcut from descs[0]'s
// input) as one launch: the first residual block of the network
static int launch_block64(const yolo_amd_conv_desc& d0, const yolo_amd_conv_desc& d1, const void* zero_page,
                          hipStream_t stream) {
    YA_REQUIRE(!d0.first_layer && d0.ksize == 1 && d0.stride == 1 && d0.c0 == 64 && d0.c1 == 0 && d0.up0 == 0 &&
                   d0.cout == 32 && d0.cout_pad == 32 && d0.k_pad == 64 && !d0.residual && !d0.dst_fp32 && !d0.decode &&
                   !d0.bn_train,
               "conv: fuse_next on a layer that is not the first layer needs a 1x1 stride-1 64 -> 32 layer");
    YA_REQUIRE(!d1.first_layer && d1.ksize == 3 && d1.stride == 1 && d1.c0 == 32 && d1.c1 == 0 && d1.up0 == 0 &&
                   d1.cout == 64 && d1.cout_pad == 64 && d1.k_pad == 288 && !d1.dst_fp32 && !d1.decode && !d1.bn_train &&
                   !d1.fuse_next && d1.B == d0.B && d1.H == d0.H && d1.W == d0.W,
               "conv: the fused block needs a 3x3 stride-1 32 -> 64 layer behind the 1x1 layer");
    YA_REQUIRE(d1.residual && d1.residual == d0.src0,
               "conv: the fused block's shortcut operand must be the 1x1 layer's input");
    YA_REQUIRE(d0.src0 && d0.weight && d0.bias && d1.weight && d1.bias && d1.dst && zero_page, "conv: null pointer");
    YA_REQUIRE((long long)d0.B * d0.H * d0.W * 64 < (1ll << 31), "conv: tensor too large for 32-bit element offsets");
    BlockArgs a;
    a.src = (const __bf16*)d0.src0; a.w1 = (const __bf16*)d0.weight; a.b1 = d0.bias;
    a.w3 = (const __bf16*)d1.weight; a.b3 = d1.bias; a.dst = (__bf16*)d1.dst; a.zero = (const __bf16*)zero_page;
    a.B = d0.B; a.H = d0.H; a.W = d0.W; a.leaky1 = d0.leaky; a.leaky3 = d1.leaky;
    a.tiles_x = (d0.W + THIN_TX - 1) / THIN_TX; a.tiles_y = (d0.H + THIN_TY - 1) / THIN_TY;
    const long long ntiles = (long long)a.tiles_x * a.tiles_y * a.B;
    hipLaunchKernelGGL(conv_block64_kernel, dim3((unsigned)std::min<long long>(ntiles, 256 * 2)), dim3(256), 0,
                       stream, a);
    return YA_OK;
}

// A layer whose BatchNorm runs on batch statistics: the convolution alone into the fp32
// scratch, then the statistics and the normalising pass of bn.hip.
static int launch_conv_bn_train(const yolo_amd_conv_desc& d, const void* zero_page, hipStream_t stream);

// descs[0] (3x3, 128 output channels, fuse_next set) and descs[1] (1x1, 128 -> 64, on descs[0]'s
// output): one launch where descs[0] runs a kernel that holds all 128 channels of a pixel tile
// in one workgroup (the im2col kernel's 128-channel tile with the two-step activation tile, the
// patch-reuse kernel's flat form), else two launches -- the results are the same bits either way.
// YOLO_AMD_TAIL (development knob): bit 0 allows the im2col form, bit 1 the patch-reuse form.
static int launch_conv_tail(const yolo_amd_conv_desc& d0, const yolo_amd_conv_desc& d1, const void* zero_page,
                            hipStream_t stream) {
    ConvChoice c;
    const int rc = choose_conv(d0, true, &c);
    if (rc != YA_OK) return rc;
    YA_REQUIRE(!d1.first_layer && d1.ksize == 1 && d1.stride == 1 && d1.c0 == d0.cout && d1.c1 == 0 && d1.up0 == 0 &&
                   !d1.fuse_next && d1.B == d0.B && d1.H == c.Ho && d1.W == c.Wo && d1.src0 == d0.dst,
               "conv: fuse_next on a 3x3 layer needs a 1x1 stride-1 layer on its output as the next descriptor");
    const bool holds_all = (c.kernel == YOLO_AMD_KERNEL_IGEMM && c.amul == 2 && c.bn == 128 && c.kslices == 1) ||
                           (c.kernel == YOLO_AMD_KERNEL_PATCH && c.flat && c.bn == 128);
    const bool can = holds_all && env_int("YOLO_AMD_TAIL", 3) & (c.kernel == YOLO_AMD_KERNEL_PATCH ? 2 : 1) &&
                     d0.cout == 128 && d0.cout_pad == 128 && !d0.dst_fp32 && !d0.decode && !d0.bn_train &&
                     d1.cout == 64 && d1.cout_pad == 64 && d1.k_pad == 128 && !d1.residual && !d1.dst_fp32 &&
                     !d1.decode && !d1.bn_train && d1.weight && d1.bias && d1.dst;
    if (can) return launch_conv(d0, zero_page, stream, &d1);
    yolo_amd_conv_desc plain = d0;
    plain.fuse_next = 0;
    const int r0 = d0.bn_train ? launch_conv_bn_train(plain, zero_page, stream) : launch_conv(plain, zero_page, stream);
    if (r0 != YA_OK) return r0;
    return d1.bn_train ? launch_conv_bn_train(d1, zero_page, stream) : launch_conv(d1, zero_page, stream);
}

static int launch_conv_bn_train(const yolo_amd_conv_desc& d, const void* zero_page, hipStream_t stream) {
    YA_REQUIRE(!d.dst_fp32 && !d.decode && !d.fuse_next, "conv: a bn_train layer needs a bf16 dst, no decode, no fuse_next");
    yolo_amd_conv_desc c = d;
    c.bn_train = nullptr;
    c.dst = d.bn_train->raw;
    c.dst_fp32 = 1;
    c.leaky = 0;
    c.residual = nullptr;
    YA_REQUIRE(c.dst, "conv: bn_train without a raw buffer");
    const int rc = launch_conv(c, zero_page, stream);
    if (rc != YA_OK) return rc;
    return launch_bn_train(*d.bn_train, out_pixels(d), d.cout, d.residual, d.dst, d.leaky, stream);
}

extern "C" int yolo_amd_conv_forward(const yolo_amd_conv_desc* descs, int n, const void* zero_page,
                                     void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(descs && n >= 0, "conv_forward: null descriptor array");
    for (int i = 0; i < n; ++i) {
        int rc;
        if (descs[i].first_layer && descs[i].fuse_next) {
            const int more = descs[i].fuse_next == 2 ? 2 : 1;      // 2: the 1x1 layer behind as well
            YA_REQUIRE(i + more < n, "conv_forward: fuse_next reaches past the last descriptor");
            rc = launch_stem(descs[i], descs[i + 1], more == 2 ? &descs[i + 2] : nullptr, stream);
            i += more;
        } else if (descs[i].fuse_next && descs[i].ksize == 3) {
            YA_REQUIRE(i + 1 < n, "conv_forward: fuse_next on the last descriptor");
            rc = launch_conv_tail(descs[i], descs[i + 1], zero_page, stream);
            ++i;
        } else if (descs[i].fuse_next) {
            YA_REQUIRE(i + 1 < n, "conv_forward: fuse_next on the last descriptor");
            rc = launch_block64(descs[i], descs[i + 1], zero_page, stream);
            ++i;
        } else if (descs[i].bn_train) {
            rc = launch_conv_bn_train(descs[i], zero_page, stream);
        } else {
            rc = launch_conv(descs[i], zero_page, stream);
        }
        if (rc != YA_OK) return rc;
    }
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}
