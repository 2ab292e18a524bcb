// Confidence filter + per-class greedy NMS (filter_results) and box_iou for
// gfx950.  Restates /root/reference/utilis.py:298-323 (box_iou), :355-390
// (filter_img_class) and :393-465 (filter_results) as six small kernels:
//
//   compact  : stream the (B,N,D) tensor once through LDS; for every row with
//              objectness > conf compute corners, first-index class argmax and
//              a 64-bit sort key; wave-ballot compaction into per-image lists
//   sort     : candidates are compacted into eight lists per image by class
//              range; one workgroup per list, LDS bitonic sort of the keys
//              (class asc, objectness desc, original row asc), written at the
//              list's place in the image's key array, and segment boundaries
//              per (image, class); a small plan kernel lays out the tiles
//   mask     : 64x64 tiles of the per-segment suppression matrix, one wave per
//              tile, spread over the whole GPU, stored tile by tile: bit j of
//              word i of tile (t, w) says box 64w+j (later in the visiting
//              order) is suppressed by box 64t+i (IoU in the reference's fp32
//              operation order)
//   reduce   : one workgroup per (image, class) segment walks the boxes in
//              order, 64 at a time: resolves the diagonal tile (serial only
//              over boxes that suppress something), then ORs the survivors'
//              words of the later tiles into the "removed" bit vector
//   scan     : exclusive scan of kept counts -> output row offsets, total K
//   gather   : write the (K, 9) rows
//
// All arithmetic that decides a keep/suppress outcome is plain IEEE fp32 with
// no contraction (the file is built with -ffp-contract=off) and correctly
// rounded division, so kept rows are bit-identical to the reference's.
#include <mutex>

#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

// torch.max / torch.min / torch.clamp propagate NaN; fmaxf/fminf do not.
__device__ __forceinline__ float tmax(float a, float b) {
    return (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
}
__device__ __forceinline__ float tmin(float a, float b) {
    return (a != a) ? a : ((b != b) ? b : (a < b ? a : b));
}

// utilis.py:298-323, same operation order, +1 pixel convention.
__device__ __forceinline__ float iou_ref(float4 p, float4 q) {
    float ix1 = tmax(p.x, q.x);
    float iy1 = tmax(p.y, q.y);
    float ix2 = tmin(p.z, q.z);
    float iy2 = tmin(p.w, q.w);
    float iw = tmax(__fadd_rn(__fsub_rn(ix2, ix1), 1.0f), 0.0f);
    float ih = tmax(__fadd_rn(__fsub_rn(iy2, iy1), 1.0f), 0.0f);
    float inter = __fmul_rn(iw, ih);
    float a1 = __fmul_rn(__fadd_rn(__fsub_rn(p.z, p.x), 1.0f), __fadd_rn(__fsub_rn(p.w, p.y), 1.0f));
    float a2 = __fmul_rn(__fadd_rn(__fsub_rn(q.z, q.x), 1.0f), __fadd_rn(__fsub_rn(q.w, q.y), 1.0f));
    float uni = __fsub_rn(__fadd_rn(a1, a2), inter);
    return __fdiv_rn(inter, uni);
}

// The suppression matrix needs one IoU per pair of boxes of a segment, and the
// NaN-propagating max/min/clamp above cost four instructions each.  For "tame"
// boxes (every coordinate finite and below 1e18 in magnitude) no intermediate
// can overflow or turn NaN before the division, so the plain v_max/v_min give
// the same bits as tmax/tmin; the arithmetic and its order are unchanged.
__device__ __forceinline__ bool box_tame(float4 p) {
    const float L = 1e18f;
    return fabsf(p.x) < L && fabsf(p.y) < L && fabsf(p.z) < L && fabsf(p.w) < L;
}
__device__ __forceinline__ float box_area(float4 p) {
    return __fmul_rn(__fadd_rn(__fsub_rn(p.z, p.x), 1.0f), __fadd_rn(__fsub_rn(p.w, p.y), 1.0f));
}
// single instructions, without the NaN canonicalisation fmaxf/fminf put in front
__device__ __forceinline__ float vmax(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float iou_tame(float4 p, float a1, float4 q, float a2) {
    const float ix1 = vmax(p.x, q.x), iy1 = vmax(p.y, q.y);
    const float ix2 = vmin(p.z, q.z), iy2 = vmin(p.w, q.w);
    const float iw = vmax(__fadd_rn(__fsub_rn(ix2, ix1), 1.0f), 0.0f);
    const float ih = vmax(__fadd_rn(__fsub_rn(iy2, iy1), 1.0f), 0.0f);
    const float inter = __fmul_rn(iw, ih);
    return __fdiv_rn(inter, __fsub_rn(__fadd_rn(a1, a2), inter));
}

// `iou < thresh` without the division.  The reference compares the fp32 quotient
// fl(inter / uni) with thresh.  For uni > 0 and a positive normal thresh, rounding to
// nearest is monotone, so fl(inter / uni) < thresh exactly when the true quotient lies
// below mid = (pred(thresh) + thresh) / 2, i.e. when inter < uni * mid.  mid has 25
// significant bits and uni 24: the product is exact in float64, so the comparison is
// exact.  (A tie, inter == uni * mid, cannot happen: mid's last bit is set, so the product
// needs 25 significant bits and inter has 24.)  A float64 multiply and compare take the
// place of the IEEE division's dozen instructions.  Pairs with uni <= 0 or NaN are flagged
// and their word is recomputed with the division.
__device__ __forceinline__ bool below_tame(float4 p, float a1, float4 q, float a2, double mid, bool& flagged) {
    const float ix1 = vmax(p.x, q.x), iy1 = vmax(p.y, q.y);
    const float ix2 = vmin(p.z, q.z), iy2 = vmin(p.w, q.w);
    const float iw = vmax(__fadd_rn(__fsub_rn(ix2, ix1), 1.0f), 0.0f);
    const float ih = vmax(__fadd_rn(__fsub_rn(iy2, iy1), 1.0f), 0.0f);
    const float inter = __fmul_rn(iw, ih);
    const float uni = __fsub_rn(__fadd_rn(a1, a2), inter);
    flagged |= !(uni > 0.f);
    return (double)inter < (double)uni * mid;
}

// mid for below_tame, or 0 when thresh is not a positive normal number (the kernels then
// keep the division)
static double thresh_midpoint(float t) {
    if (!(t >= 1.17549435e-38f) || !(t <= 3.402823466e38f)) return 0.0;
    uint32_t u;
    memcpy(&u, &t, 4);
    --u;
    float prev;
    memcpy(&prev, &u, 4);
    return ((double)prev + (double)t) * 0.5;
}

// ---------------------------------------------------------------- box_iou
__global__ void __launch_bounds__(256)
box_iou_kernel(const float* __restrict__ b1, int n1, const float* __restrict__ b2, int m,
               float* __restrict__ out) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const float* p = b1 + (n1 == 1 ? 0 : 4 * (size_t)i);
    const float* q = b2 + 4 * (size_t)i;
    out[i] = iou_ref(make_float4(p[0], p[1], p[2], p[3]), make_float4(q[0], q[1], q[2], q[3]));
}

// ---------------------------------------------------------------- keys
// key = class << 52 | ~ordered(objectness) << 20 | row, where ordered() is the
// usual order-preserving map from fp32 to uint32 (flip the sign bit of
// non-negative values, all bits of negative ones); inverting it sorts
// descending.  row < 2^20, class < 2^12.
constexpr int KEY_ROW_BITS = 20;
constexpr int KEY_CLS_SHIFT = 52;
__device__ __forceinline__ unsigned long long make_key(int cls, float obj, int row) {
    unsigned int u = __float_as_uint(obj);
    u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;
    unsigned int ob = ~u;
    return ((unsigned long long)cls << KEY_CLS_SHIFT) | ((unsigned long long)ob << KEY_ROW_BITS) |
           (unsigned long long)row;
}
__device__ __forceinline__ int key_cls(unsigned long long k) { return (int)(k >> KEY_CLS_SHIFT); }
__device__ __forceinline__ int key_row(unsigned long long k) { return (int)(k & ((1u << KEY_ROW_BITS) - 1)); }
__device__ __forceinline__ float key_obj(unsigned long long k) {
    unsigned int u = ~(unsigned int)(k >> KEY_ROW_BITS);
    u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
    return __uint_as_float(u);
}

// ---------------------------------------------------------------- compact
// Block = WAVES waves; each wave owns a tile of 64 consecutive rows of one
// image and copies its 64*D floats into its own LDS slice by LDS-DMA: every
// 16-byte piece of the tile is in flight at once and no register is used.
// Rows are only 4-byte aligned (D = 85), so the copy starts at the 16-byte
// boundary at or below the tile and the slice keeps the same misalignment.
__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// The per-image candidate counters take one atomic per wave; each sits in its
// own 128-byte line so that the images' atomics do not queue on one L2 line.
constexpr int CNT_STRIDE = 32;

template <int WAVES>
__global__ void __launch_bounds__(WAVES * 64)
nms_compact_kernel(const float* __restrict__ pred, int N, int D, int C, float conf,
                   int raw_stride, int G, int slice_bytes, int* __restrict__ cand_count,
                   unsigned long long* __restrict__ raw_keys, float4* __restrict__ boxes,
                   float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int row0 = (blockIdx.x * WAVES + wave) * 64;
    char* slice = lds + (size_t)wave * slice_bytes;
    const int rows = min(64, N - row0);            // may be <= 0 for the tail wave
    int mis = 0;
    if (rows > 0) {
        const char* src = reinterpret_cast<const char*>(pred + ((size_t)b * N + row0) * D);
        const int nbytes = rows * D * 4;
        mis = (int)((uintptr_t)src & 15);
        const char* g0 = src - mis;
        if (g0 < reinterpret_cast<const char*>(pred)) {
            // first tile of a tensor that does not start on a 16-byte boundary
            for (int i = lane * 4; i < nbytes; i += 256)
                *reinterpret_cast<float*>(slice + mis + i) = *reinterpret_cast<const float*>(src + i);
        } else {
            const int total = mis + nbytes, full = total >> 4;
            for (int kb = 0; kb < full; kb += 64)
                if (kb + lane < full) lds_dma16(g0 + (size_t)(kb + lane) * 16, slice + kb * 16);
            for (int i = (full << 4) + lane * 4; i < total; i += 256)
                *reinterpret_cast<float*>(slice + i) = *reinterpret_cast<const float*>(g0 + i);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const float* tile = reinterpret_cast<const float*>(slice + mis);
    __syncthreads();
    bool pass = false;
    float obj = 0.f;
    const float* r = tile + lane * D;
    if (lane < rows) {
        obj = r[4];
        pass = (obj > conf) && (obj > 0.0f);
    }
    if (__ballot(pass) == 0) return;
    int bi = 0;
    float best = 0.f;
    if (pass) {
        // first-index maximum; a NaN beats every number and the first NaN wins
        best = r[5];
#pragma unroll 8
        for (int j = 1; j < C; ++j) {
            float v = r[5 + j];
            if ((v > best) || ((v != v) && !(best != best))) { best = v; bi = j; }
        }
    }
    // Candidates go to one of G lists per image by class range (list g holds
    // the classes c with c*G/C == g), so that G workgroups can sort an image.
    const int grp = (int)(((long long)bi * G) / C);
    // One atomic instruction per wave: lane g reserves the slots of list g
    // (G <= 64), then every candidate reads its list's base from that lane.
    unsigned long long mymask = 0;
    int mycount = 0;
    for (int g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(pass && grp == g);
        if (grp == g) mymask = m;
        if (lane == g) mycount = __popcll(m);
    }
    int base = 0;
    if (lane < G && mycount) base = atomicAdd(&cand_count[(b * G + lane) * CNT_STRIDE], mycount);
    base = __shfl(base, grp);
    if (pass) {
        const int slot = base + __popcll(mymask & ((1ull << lane) - 1ull));
        raw_keys[((size_t)b * G + grp) * raw_stride + slot] = make_key(bi, obj, row0 + lane);
    }
    if (pass) {
        float cx = r[0], cy = r[1], w = r[2], h = r[3];
        float hw = __fdiv_rn(w, 2.0f), hh = __fdiv_rn(h, 2.0f);
        float4 bx = make_float4(__fsub_rn(cx, hw), __fsub_rn(cy, hh), __fadd_rn(cx, hw), __fadd_rn(cy, hh));
        const int row = row0 + lane;
        boxes[(size_t)b * N + row] = bx;
        scores[(size_t)b * N + row] = best;
    }
}

// Four lanes per row: the default form.  A wave stages 16 rows (5.4 KB at D = 85) instead of
// 64, a block of four waves 64 rows, so six blocks = 24 waves fit a CU where the 64-row form has
// six waves: the kernel is a chain of memory round trips per wave (tile in, counter atomic out,
// lists out) and only waves in flight hide them.  Lane = (row, part): part p scans classes p,
// p + 4, ... and the four partial first-index maxima are merged by two quad swaps.  The merge
// keeps what the sequential scan returns: the first NaN if there is one, else the first
// occurrence of the largest score.  The block's four waves reserve their list slots with ONE
// atomic per class list (counts through LDS), as many atomics per row as the 64-row form issues.
constexpr int COMPACT4_ROWS = 16, COMPACT4_WAVES = 4;

__device__ __forceinline__ void merge_best(float& va, int& ia, float vb, int ib) {
    const bool a_ok = ia >= 0, b_ok = ib >= 0;
    const bool a_nan = va != va, b_nan = vb != vb;
    const bool take_b = b_ok && (!a_ok || (b_nan && (!a_nan || ib < ia)) ||
                                 (!a_nan && !b_nan && (vb > va || (vb == va && ib < ia))));
    if (take_b) { va = vb; ia = ib; }
}

__global__ void __launch_bounds__(COMPACT4_WAVES * 64)
nms_compact4_kernel(const float* __restrict__ pred, int N, int D, int C, float conf,
                    int raw_stride, int G, int slice_bytes, int* __restrict__ cand_count,
                    unsigned long long* __restrict__ raw_keys, float4* __restrict__ boxes,
                    float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int* const wcnt = reinterpret_cast<int*>(lds + COMPACT4_WAVES * slice_bytes);   // [waves][64] counts per list
    int* const bases = wcnt + COMPACT4_WAVES * 64;                                  // [64] the block's slots per list
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int row0 = (blockIdx.x * COMPACT4_WAVES + wave) * COMPACT4_ROWS;
    char* slice = lds + (size_t)wave * slice_bytes;
    const int rows = min(COMPACT4_ROWS, N - row0);            // may be <= 0 for the tail waves
    int mis = 0;
    if (rows > 0) {
        const char* src = reinterpret_cast<const char*>(pred + ((size_t)b * N + row0) * D);
        const int nbytes = rows * D * 4;
        mis = (int)((uintptr_t)src & 15);
        const char* g0 = src - mis;
        if (g0 < reinterpret_cast<const char*>(pred)) {
            // first tile of a tensor that does not start on a 16-byte boundary
            for (int i = lane * 4; i < nbytes; i += 256)
                *reinterpret_cast<float*>(slice + mis + i) = *reinterpret_cast<const float*>(src + i);
        } else {
            const int total = mis + nbytes, full = total >> 4;
            for (int kb = 0; kb < full; kb += 64)
                if (kb + lane < full) lds_dma16(g0 + (size_t)(kb + lane) * 16, slice + kb * 16);
            for (int i = (full << 4) + lane * 4; i < total; i += 256)
                *reinterpret_cast<float*>(slice + i) = *reinterpret_cast<const float*>(g0 + i);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_wave_barrier();
    const float* tile = reinterpret_cast<const float*>(slice + mis);
    const int rr = lane >> 2, part = lane & 3;
    const float* r = tile + rr * D;
    bool pass = false;
    float obj = 0.f;
    if (rr < rows) {
        obj = r[4];
        pass = (obj > conf) && (obj > 0.0f);
    }
    int bi = -1;
    float best = 0.f;
    if (pass && part < C) {
        // first-index maximum of this lane's classes; a NaN beats every number, the first NaN wins
        best = r[5 + part];
        bi = part;
#pragma unroll 4
        for (int j = part + 4; j < C; j += 4) {
            const float v = r[5 + j];
            if ((v > best) || ((v != v) && !(best != best))) { best = v; bi = j; }
        }
    }
    // merge the four parts of a row (every lane of the quad ends up with the row's result)
    merge_best(best, bi, __shfl_xor(best, 1, 64), __shfl_xor(bi, 1, 64));
    merge_best(best, bi, __shfl_xor(best, 2, 64), __shfl_xor(bi, 2, 64));
    const bool rep = pass && part == 0;                       // one lane speaks for the row
    const int grp = rep ? (int)(((long long)bi * G) / C) : -1;
    unsigned long long mymask = 0;
    int mycount = 0;
    for (int g = 0; g < G; ++g) {
        const unsigned long long m = __ballot(grp == g);
        if (grp == g) mymask = m;
        if (lane == g) mycount = __popcll(m);
    }
    if (lane < G) wcnt[wave * 64 + lane] = mycount;
    __syncthreads();
    if (wave == 0 && lane < G) {
        int tot = 0;
#pragma unroll
        for (int w2 = 0; w2 < COMPACT4_WAVES; ++w2) tot += wcnt[w2 * 64 + lane];
        bases[lane] = tot ? atomicAdd(&cand_count[(b * G + lane) * CNT_STRIDE], tot) : 0;
    }
    __syncthreads();
    if (rep) {
        int base = bases[grp];
        for (int w2 = 0; w2 < wave; ++w2) base += wcnt[w2 * 64 + grp];
        const int slot = base + __popcll(mymask & ((1ull << lane) - 1ull));
        const int row = row0 + rr;
        raw_keys[((size_t)b * G + grp) * raw_stride + slot] = make_key(bi, obj, row);
        const float cx = r[0], cy = r[1], w = r[2], h = r[3];
        const float hw = __fdiv_rn(w, 2.0f), hh = __fdiv_rn(h, 2.0f);
        boxes[(size_t)b * N + row] = make_float4(__fsub_rn(cx, hw), __fsub_rn(cy, hh), __fadd_rn(cx, hw), __fadd_rn(cy, hh));
        scores[(size_t)b * N + row] = best;
    }
}

// ---------------------------------------------------------------- sort
constexpr int SORT_THREADS = 1024;
constexpr int SORT_LDS_KEYS = 16384;     // 128 KiB of the CU's 160 KiB LDS

template <typename Ptr>
__device__ __forceinline__ void bitonic_sort(Ptr s, int P) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (P >> 1); t += SORT_THREADS) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                int l = i | j;
                bool up = (i & k) == 0;
                unsigned long long a = s[i], c = s[l];
                if ((a > c) == up) { s[i] = c; s[l] = a; }
            }
            __syncthreads();
        }
    }
}

// Lists of at most 4 * SORT_THREADS keys (the usual case: a class group of one image at
// the callers' confidence thresholds): E = P / SORT_THREADS keys per lane, key number
// E * threadIdx.x + e in v[e].  Compare-exchange steps whose partner lies in the lane
// (j < E) or in the wave (j < 64 E) trade keys in registers or by lane shuffles, without
// LDS traffic or a workgroup barrier: 45 of the 55 steps of a 1024-key sort, 56 of 66 at
// 2048 keys.  The others go through `s` (write, barrier, read the partners, barrier).
// Lanes whose keys lie beyond P are idle but take part in the barriers.
template <int E>
__device__ __forceinline__ void bitonic_sort_lanes(unsigned long long* s, int P, unsigned long long (&v)[E]) {
    const int t = threadIdx.x;
    const bool live = t * E < P;
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            unsigned long long other[E];
            if (j >= 64 * E) {
                if (live) {
#pragma unroll
                    for (int e = 0; e < E; ++e) s[t * E + e] = v[e];
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) other[e] = live ? s[(t * E + e) ^ j] : 0ull;
                __syncthreads();
            } else if (j >= E) {
#pragma unroll
                for (int e = 0; e < E; ++e) other[e] = __shfl_xor(v[e], j / E);
            } else if (j == 1) {             // (constant register indices: no run-time indexing of v)
#pragma unroll
                for (int e = 0; e < E; ++e) other[e] = v[(e ^ 1) & (E - 1)];
            } else {
#pragma unroll
                for (int e = 0; e < E; ++e) other[e] = v[(e ^ 2) & (E - 1)];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int i = t * E + e;
                const bool keep_min = ((i & j) == 0) == ((i & k) == 0);
                const unsigned long long lo = v[e] < other[e] ? v[e] : other[e];
                const unsigned long long hi = v[e] < other[e] ? other[e] : v[e];
                v[e] = keep_min ? lo : hi;
            }
        }
    }
}

template <int E>
__device__ __forceinline__ void sort_list_lanes(unsigned long long* s, int P, const unsigned long long* raw, int cnt) {
    const int t = threadIdx.x;
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = t * E + e < cnt ? raw[t * E + e] : ~0ull;
    bitonic_sort_lanes<E>(s, P, v);
    if (t * E < P) {
#pragma unroll
        for (int e = 0; e < E; ++e) s[t * E + e] = v[e];
    }
}

// One workgroup per (class group, image): sorts the group's candidate list and
// writes it at its place in the image's sorted key array (the groups hold
// ascending class ranges, so group g starts where the groups before it end),
// and records where each of its classes starts and ends.
__global__ void __launch_bounds__(SORT_THREADS)
nms_sort_kernel(const int* __restrict__ cand_count, int G, int raw_stride,
                unsigned long long* __restrict__ raw_keys, int key_stride, int C,
                unsigned long long* __restrict__ keys, int* __restrict__ seg_start,
                int* __restrict__ seg_end) {
    __shared__ unsigned long long skeys[SORT_LDS_KEYS];
    const int g = blockIdx.x, b = blockIdx.y;
    const int cnt = cand_count[(b * G + g) * CNT_STRIDE];
    int offset = 0;
    for (int h = 0; h < g; ++h) offset += cand_count[(b * G + h) * CNT_STRIDE];
    unsigned long long* raw = raw_keys + ((size_t)b * G + g) * raw_stride;
    unsigned long long* gk = keys + (size_t)b * key_stride + offset;
    // this group's classes: c*G/C == g
    const int c_lo = (int)(((long long)g * C + G - 1) / G), c_hi = (int)(((long long)(g + 1) * C + G - 1) / G);
    for (int c = c_lo + threadIdx.x; c < c_hi; c += SORT_THREADS) {
        seg_start[b * C + c] = 0;
        seg_end[b * C + c] = 0;
    }
    if (cnt == 0) return;
    int P = 1;
    while (P < cnt) P <<= 1;
    const unsigned long long* sorted;
    if (P <= 4 * SORT_THREADS) {
        if (P <= SORT_THREADS) sort_list_lanes<1>(skeys, P, raw, cnt);
        else if (P == 2 * SORT_THREADS) sort_list_lanes<2>(skeys, P, raw, cnt);
        else sort_list_lanes<4>(skeys, P, raw, cnt);
        sorted = skeys;
    } else if (P <= SORT_LDS_KEYS) {
        for (int i = threadIdx.x; i < P; i += SORT_THREADS) skeys[i] = (i < cnt) ? raw[i] : ~0ull;
        __syncthreads();
        bitonic_sort(skeys, P);
        sorted = skeys;
    } else {
        // more candidates than the LDS holds: same network in global memory
        // (raw_stride >= P is guaranteed by the host)
        for (int i = cnt + threadIdx.x; i < P; i += SORT_THREADS) raw[i] = ~0ull;
        __threadfence_block();
        __syncthreads();
        bitonic_sort(raw, P);
        sorted = raw;
    }
    // (the zeroing above is ordered before the writes below by the barriers
    // inside bitonic_sort, or by this one when P == 1)
    __syncthreads();
    for (int i = threadIdx.x; i < cnt; i += SORT_THREADS) {
        const unsigned long long k = sorted[i];
        gk[i] = k;
        const int c = key_cls(k);
        if (i == 0 || key_cls(sorted[i - 1]) != c) seg_start[b * C + c] = offset + i;
        if (i == cnt - 1 || key_cls(sorted[i + 1]) != c) seg_end[b * C + c] = offset + i + 1;
    }
}

// ---------------------------------------------------------------- plan
// One workgroup per image plans the suppression-matrix tiles: a segment of n
// boxes has T = ceil(n/64) word columns and T(T+1)/2 upper-triangular tiles of
// 64 mask words each.  Exclusive prefix sums over the classes give every
// segment its first tile number and its mask offset.
__global__ void __launch_bounds__(SORT_THREADS)
nms_plan_kernel(int C, const int* __restrict__ seg_start, const int* __restrict__ seg_end,
                int* __restrict__ tile_base, int* __restrict__ mask_off, int* __restrict__ total_tiles) {
    __shared__ int sa[SORT_THREADS], sb[SORT_THREADS];
    const int b = blockIdx.x;
    int carry_t = 0, carry_w = 0;
    for (int c0 = 0; c0 < C; c0 += SORT_THREADS) {
        const int c = c0 + threadIdx.x;
        int tiles = 0, words = 0;
        if (c < C) {
            const int n = seg_end[b * C + c] - seg_start[b * C + c];
            const int T = (n + 63) >> 6;
            tiles = T * (T + 1) / 2;
            words = tiles * 64;
        }
        sa[threadIdx.x] = tiles;
        sb[threadIdx.x] = words;
        __syncthreads();
        for (int d = 1; d < SORT_THREADS; d <<= 1) {
            int va = 0, vb = 0;
            if ((int)threadIdx.x >= d) { va = sa[threadIdx.x - d]; vb = sb[threadIdx.x - d]; }
            __syncthreads();
            sa[threadIdx.x] += va;
            sb[threadIdx.x] += vb;
            __syncthreads();
        }
        if (c < C) {
            tile_base[b * C + c] = carry_t + sa[threadIdx.x] - tiles;
            mask_off[b * C + c] = carry_w + sb[threadIdx.x] - words;
        }
        carry_t += sa[SORT_THREADS - 1];
        carry_w += sb[SORT_THREADS - 1];
        __syncthreads();
    }
    if (threadIdx.x == 0) total_tiles[b] = carry_t;
}

// ---------------------------------------------------------------- mask
// One wave per 64x64 tile (ti <= tj) of a segment's suppression matrix.
// Lane = row box i = 64*ti + lane; the 64 column boxes go through LDS.
// Bit jj of the word: box j = 64*tj + jj comes later than i in the visiting
// order and `iou(i, j) < thresh` is false (so iou >= thresh or NaN), which is
// when the reference zeroes row j while box i is the pivot (utilis.py:379-381).
// The matrix is stored tile by tile (tile (ti, tj) of a segment at number
// ti*T - ti*(ti-1)/2 + tj - ti, 64 words each, lane order), so a wave writes
// and later reads a whole tile as one contiguous 512-byte run.
constexpr int MASK_GRID_X = 256;

__global__ void __launch_bounds__(256)
nms_mask_kernel(const unsigned long long* __restrict__ keys, int key_stride, int N, int C,
                const int* __restrict__ seg_start, const int* __restrict__ seg_end,
                const int* __restrict__ tile_base, const int* __restrict__ mask_off,
                const int* __restrict__ total_tiles, const float4* __restrict__ boxes, int box_stride,
                float thresh, double mid, unsigned long long* __restrict__ mask, size_t mask_stride) {
    __shared__ float4 colbox[4][64];
    __shared__ float colarea[4][64];
    const int b = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nt = total_tiles[b];
    const unsigned long long* gkb = keys + (size_t)b * key_stride;
    const float4* gb = boxes + (size_t)b * N * box_stride;
    for (int q = blockIdx.x * 4 + wave; q < nt; q += gridDim.x * 4) {
        // segment that owns tile q: the last class whose first tile is <= q
        int lo = 0, hi = C - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tile_base[b * C + mid] <= q) lo = mid; else hi = mid - 1;
        }
        const int c = lo;
        const int s = seg_start[b * C + c];
        const int n = seg_end[b * C + c] - s;
        const int T = (n + 63) >> 6;
        const int q_local = q - tile_base[b * C + c];
        int rem = q_local, ti = 0;
        while (rem >= T - ti) { rem -= T - ti; ++ti; }
        const int tj = ti + rem;
        const int i = ti * 64 + lane, j0 = tj * 64;
        float4 mine = make_float4(0.f, 0.f, 0.f, 0.f), col = mine;
        if (i < n) mine = gb[(size_t)key_row(gkb[s + i]) * box_stride];
        if (j0 + lane < n) col = gb[(size_t)key_row(gkb[s + j0 + lane]) * box_stride];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // earlier reads of colbox are done
        __builtin_amdgcn_wave_barrier();
        colbox[wave][lane] = col;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        const int ncol = __builtin_amdgcn_readfirstlane(min(64, n - j0));
        unsigned long long word = 0;
        if (__ballot(!box_tame(mine) || !box_tame(col)) == 0ull) {
            colarea[wave][lane] = box_area(col);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const float my_area = box_area(mine);
            bool flagged = false;
            if (mid != 0.0) {
                auto pair = [&](int jj) {
                    const bool sup = !below_tame(mine, my_area, colbox[wave][jj], colarea[wave][jj], mid, flagged);
                    word |= (unsigned long long)sup << jj;
                };
                if (ncol == 64) {            // a full tile: constant trip count, four pairs' LDS reads in flight
#pragma unroll 4
                    for (int jj = 0; jj < 64; ++jj) pair(jj);
                } else {
                    for (int jj = 0; jj < ncol; ++jj) pair(jj);
                }
            }
            if (mid == 0.0 || __ballot(flagged) != 0ull) {   // a union that is not positive: divide
                word = 0;
                for (int jj = 0; jj < ncol; ++jj) {
                    const bool sup = !(iou_tame(mine, my_area, colbox[wave][jj], colarea[wave][jj]) < thresh);
                    word |= (unsigned long long)sup << jj;
                }
            }
        } else {
            for (int jj = 0; jj < ncol; ++jj) {
                const bool sup = !(iou_ref(mine, colbox[wave][jj]) < thresh);
                word |= (unsigned long long)sup << jj;
            }
        }
        if (ti == tj) word &= ~0ull << lane << 1;            // only boxes later in the order
        if (i >= n) word = 0;
        mask[(size_t)b * mask_stride + mask_off[b * C + c] + (size_t)q_local * 64 + lane] = word;
    }
}

// ---------------------------------------------------------------- reduce
// One workgroup (8 waves) per (image, class).  For each group g of 64 boxes in
// visiting order a box survives iff no earlier survivor suppresses it (the
// reference's greedy loop, utilis.py:367-385): `removed` bit of box j = OR over the
// earlier survivors i of bit j of i's mask row.
//
// Wave 0 walks the chain.  At step g it owns the group's removed word, resolves
// the group with its diagonal tile (serial only over boxes that suppress
// something: a box with an empty word changes nothing), writes the survivors'
// positions, and folds the survivors' rows of tile (g, g+1) -- the only tile the
// NEXT step depends on, fetched a step ahead -- into a register it carries over.
// Waves 1-7 run one step behind: they fold the survivors of group g-1 into the
// words g+1, g+2, ... (several tile loads in flight per wave), which nobody needs
// before step g+1.  One barrier per step; the chain's step is a resolve plus one
// wave-wide OR instead of a resolve plus a round of dependent tile loads.
constexpr int REDUCE_MAX_WORDS = 512;    // 32768 boxes per segment
constexpr int REDUCE_THREADS = 512;

// OR over the 64 lanes, the same value in every lane afterwards.  Four DPP steps
// (row_shr 1, 2, 4, 8: lanes shifted in from outside a row of 16 read as zero) leave
// the OR of each row in its last lane, four v_readlane fetch those into scalar
// registers: about twenty VALU/SALU instructions and no trip through the LDS
// crossbar, which a butterfly of six __shfl_xor pays six times over (and the chain
// below runs one of these per step).
__device__ __forceinline__ unsigned int wave_or32(unsigned int v) {
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, true);
    v |= (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, true);
    return __builtin_amdgcn_readlane(v, 15) | __builtin_amdgcn_readlane(v, 31) |
           __builtin_amdgcn_readlane(v, 47) | __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ unsigned long long wave_or(unsigned long long v) {
    return ((unsigned long long)wave_or32((unsigned int)(v >> 32)) << 32) | wave_or32((unsigned int)v);
}

__global__ void __launch_bounds__(REDUCE_THREADS)
nms_reduce_kernel(int N, int C, const int* __restrict__ seg_start, const int* __restrict__ seg_end,
                  const int* __restrict__ mask_off, const unsigned long long* __restrict__ mask,
                  size_t mask_stride, int* __restrict__ kept_pos, int* __restrict__ seg_kept) {
    __shared__ unsigned long long removed[REDUCE_MAX_WORDS];
    __shared__ unsigned long long keepw[REDUCE_MAX_WORDS];
    const int c = blockIdx.x, b = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int s = seg_start[b * C + c];
    const int n = seg_end[b * C + c] - s;
    if (n <= 0) {
        if (threadIdx.x == 0) seg_kept[b * C + c] = 0;
        return;
    }
    const int T = (n + 63) >> 6;
    if (T == 1 && wave != 0) return;                     // a one-group segment is wave 0's alone: no barrier below
    const unsigned long long* tiles = mask + (size_t)b * mask_stride + mask_off[b * C + c];
    auto tile = [&](int ti, int tj) { return tiles + (size_t)(ti * T - ti * (ti - 1) / 2 + tj - ti) * 64; };
    int* kp = kept_pos + (size_t)b * N + s;
    if (T > 1) {
        for (int w = threadIdx.x; w < T; w += REDUCE_THREADS) removed[w] = 0;
        __syncthreads();
    }
    int nk = 0;
    unsigned long long d = 0, nxt = 0, near = 0;
    constexpr int H = REDUCE_THREADS / 64 - 1, U = 4;        // helper waves, tiles a helper keeps in flight
    unsigned long long pf[U];                                // helpers: row t+1's tiles, fetched a step ahead
    auto prefetch = [&](int t) {                             // words t+2+(wave-1), +H, ... of tile row t
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int w = t + 2 + (wave - 1) + u * H;
            pf[u] = w < T ? tile(t, w)[lane] : 0ull;
        }
    };
    if (wave == 0) {
        d = tile(0, 0)[lane];
        if (T > 1) nxt = tile(0, 1)[lane];
    } else if (T > 2) {
        prefetch(0);
    }
    for (int g = 0; g < T; ++g) {
        if (wave == 0) {
            unsigned long long d_next = 0, nxt_next = 0;                // the next step's two tiles, a step ahead
            if (g + 1 < T) d_next = tile(g + 1, g + 1)[lane];
            if (g + 2 < T) nxt_next = tile(g + 1, g + 2)[lane];
            const int left = n - g * 64;
            const unsigned long long valid = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            unsigned long long rem = (T > 1 ? removed[g] : 0ull) | near | ~valid;
            unsigned long long act = __ballot(d != 0ull) & ~rem;
            while (act) {
                // the first box still standing survives; what it suppresses leaves the
                // list at once, so the loop takes one step per SURVIVOR with a non-empty
                // row, not one per box
                const int i = __ffsll((long long)act) - 1;
                const unsigned int dlo = __builtin_amdgcn_readlane((unsigned int)d, i);
                const unsigned int dhi = __builtin_amdgcn_readlane((unsigned int)(d >> 32), i);
                rem |= ((unsigned long long)dhi << 32) | dlo;
                act &= ~rem & ~(1ull << i);
            }
            const unsigned long long keep = ~rem;
            if ((keep >> lane) & 1ull) kp[nk + __popcll(keep & ((1ull << lane) - 1ull))] = g * 64 + lane;
            nk += __popcll(keep);
            if (g + 1 < T) {
                if (lane == 0) keepw[g] = keep;
                near = wave_or(((keep >> lane) & 1ull) ? nxt : 0ull);
            }
            d = d_next;
            nxt = nxt_next;
        } else if (g >= 1) {
            const int t = g - 1;                                         // published before the last barrier
            const bool mine = (keepw[t] >> lane) & 1ull;
            // the first H*U words behind the chain's own come from the registers filled a step ago
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int w = t + 2 + (wave - 1) + u * H;
                if (w < T) {                                             // wave-uniform
                    const unsigned long long r = wave_or(mine ? pf[u] : 0ull);
                    if (lane == 0 && r) removed[w] |= r;
                }
            }
            // segments of more than (H*U + 2) * 64 boxes: the words further out are fetched here
            for (int w0 = t + 2 + (wave - 1) + H * U; w0 < T; w0 += H * U) {
                unsigned long long v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int w = w0 + u * H;
                    v[u] = (w < T && mine) ? tile(t, w)[lane] : 0ull;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int w = w0 + u * H;
                    if (w < T) {
                        const unsigned long long r = wave_or(v[u]);
                        if (lane == 0 && r) removed[w] |= r;
                    }
                }
            }
            if (g + 2 < T) prefetch(g);                                  // row t+1, consumed at the next step
        }
        if (T > 1) {
            // LDS traffic only crosses this barrier: a bare s_barrier behind an LDS wait.
            // (__syncthreads() would also wait for wave 0's tile loads of the next step,
            // putting a global-memory round trip back into every step of the chain.)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
    }
    if (threadIdx.x == 0) seg_kept[b * C + c] = nk;
}

// ---------------------------------------------------------------- scan
__global__ void __launch_bounds__(1024)
nms_scan_kernel(const int* __restrict__ seg_kept, int S, int* __restrict__ seg_off,
                int* __restrict__ total) {
    __shared__ int part[1024];
    const int t = threadIdx.x;
    const int per = (S + 1023) / 1024;
    const int lo = min(S, t * per), hi = min(S, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += seg_kept[i];
    part[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    int run = part[t] - sum;                 // exclusive prefix of this thread's range
    for (int i = lo; i < hi; ++i) {
        seg_off[i] = run;
        run += seg_kept[i];
    }
    if (t == 1023) *total = part[1023];
}

// ---------------------------------------------------------------- gather
// Row = [image, x1, y1, x2, y2, objectness, class score, class index, y1]
// (utilis.py:397-411 makes the 9th value alias ymin).
// One lane per slot of the images' sorted key arrays.  Slot i of image b lies in the
// segment of class c = class of key i; it is slot r = i - seg_start of that segment's kept
// list and stands for an output row iff r < seg_kept.  (One workgroup per (image, class)
// walked the longest kept list 64 rows at a time: 15 us for a 600-row list.)
__global__ void __launch_bounds__(256)
nms_gather_kernel(const unsigned long long* __restrict__ keys, int key_stride, int N, int C,
                  const int* __restrict__ cand_count, int G,
                  const int* __restrict__ seg_start, const int* __restrict__ seg_kept,
                  const int* __restrict__ seg_off, const int* __restrict__ kept_pos,
                  const float4* __restrict__ boxes, const float* __restrict__ scores,
                  float* __restrict__ out) {
    const int b = blockIdx.y;
    int total = 0;
    for (int g = 0; g < G; ++g) total += cand_count[(b * G + g) * CNT_STRIDE];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total || i >= N) return;
    const unsigned long long* gk = keys + (size_t)b * key_stride;
    const int pos = kept_pos[(size_t)b * N + i];
    const int c = key_cls(gk[i]);
    if (c >= C) return;
    const int s = seg_start[b * C + c];
    const int r = i - s;
    if (r < 0 || r >= seg_kept[b * C + c] || pos < 0 || s + pos >= total) return;
    const unsigned long long k = gk[s + pos];
    const int row = key_row(k);
    if (row >= N) return;
    const float4 bx = boxes[(size_t)b * N + row];
    float* o = out + (size_t)(seg_off[b * C + c] + r) * 9;
    o[0] = (float)b;
    o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
    o[5] = key_obj(k);
    o[6] = scores[(size_t)b * N + row];
    o[7] = (float)c;
    o[8] = bx.y;
}


// ---------------------------------------------------------------- filter_img_class
// Single-class entry: rows are (m, 8) fp32 [x1,y1,x2,y2,obj,score,cls,ymin]
// (utilis.py:355-390).  Rows whose class column equals `cls` are selected;
// rows with objectness == 0 never survive the reference's nonzero() pass.
__global__ void __launch_bounds__(256)
rows8_keys_kernel(const float* __restrict__ rows8, int m, float cls, int* __restrict__ count,
                  unsigned long long* __restrict__ keys) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    bool pass = false;
    float obj = 0.f;
    if (r < m) {
        obj = rows8[(size_t)r * 8 + 4];
        pass = (rows8[(size_t)r * 8 + 6] == cls) && (obj != 0.0f);
    }
    unsigned long long mk = __ballot(pass);
    if (mk == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(count, __popcll(mk));
    base = __shfl(base, 0);
    if (pass) keys[base + __popcll(mk & ((1ull << lane) - 1ull))] = make_key(0, obj, r);
}

__global__ void __launch_bounds__(64)
rows8_gather_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ seg_kept,
                    const int* __restrict__ kept_pos, const float* __restrict__ rows8,
                    float* __restrict__ out) {
    const int nk = seg_kept[0];
    for (int r = blockIdx.x * 64 + threadIdx.x; r < nk; r += gridDim.x * 64) {
        const int row = key_row(keys[kept_pos[r]]);
        const float4* src = reinterpret_cast<const float4*>(rows8 + (size_t)row * 8);
        float4* dst = reinterpret_cast<float4*>(out + (size_t)r * 8);
        dst[0] = src[0];
        dst[1] = src[1];
    }
}

// ---------------------------------------------------------------- workspace
struct NmsWorkspace {
    int* cand_count;     // [B][G][CNT_STRIDE], one counter per 128-byte line
    unsigned long long* raw_keys;  // [B][G][raw_stride] candidate keys as compacted, per class group
    int G, raw_stride;
    int* total;          // [1]  (lives in the caller's out_count instead when given)
    int* seg_start;      // [B*C]
    int* seg_end;        // [B*C]
    int* seg_kept;       // [B*C]
    int* seg_off;        // [B*C]
    unsigned long long* keys;  // [B][key_stride]
    float4* boxes;       // [B][N]
    float* scores;       // [B][N]
    int* kept_pos;       // [B][N]
    int* tile_base;      // [B*C] first tile number of each segment
    int* mask_off;       // [B*C] first mask word of each segment (inside the image's region)
    int* total_tiles;    // [B]
    unsigned long long* mask;   // [B][mask_stride] suppression matrices, tile by tile per segment
    size_t mask_stride;
    int key_stride;
    size_t bytes;
    size_t zero_bytes;   // leading region that must be zeroed before every call
};

// candidate lists (= sort workgroups) per image
static int class_groups(int C) { return C < 8 ? 1 : 8; }

static NmsWorkspace carve(void* base, int B, int N, int C) {
    NmsWorkspace w;
    w.G = class_groups(C);
    char* p = (char*)base;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* r = p + off; off = align_up(off + bytes, 256); return r; };
    w.cand_count = (int*)take(sizeof(int) * (size_t)B * w.G * CNT_STRIDE);
    w.zero_bytes = off;
    w.total = (int*)take(sizeof(int));
    const size_t S = (size_t)B * C;
    w.seg_start = (int*)take(sizeof(int) * S);
    w.seg_end = (int*)take(sizeof(int) * S);
    w.seg_kept = (int*)take(sizeof(int) * S);
    w.seg_off = (int*)take(sizeof(int) * S);
    w.key_stride = (N <= SORT_LDS_KEYS) ? N : (int)next_pow2((uint32_t)N);
    w.keys = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)B * w.key_stride);
    w.raw_stride = w.key_stride;
    w.raw_keys = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)B * w.G * w.raw_stride);
    w.boxes = (float4*)take(sizeof(float4) * (size_t)B * N);
    w.scores = (float*)take(sizeof(float) * (size_t)B * N);
    w.kept_pos = (int*)take(sizeof(int) * (size_t)B * N);
    w.tile_base = (int*)take(sizeof(int) * S);
    w.mask_off = (int*)take(sizeof(int) * S);
    w.total_tiles = (int*)take(sizeof(int) * (size_t)B);
    // per image: sum over segments of 32*T*(T+1) words, T = ceil(n/64), sum of n <= N
    //   <= N*N/128 + 1.5*N + 64*min(N, C)
    w.mask_stride = ((size_t)N * N + 127) / 128 + 2 * (size_t)N + 64 * (size_t)(N < C ? N : C);
    w.mask = (unsigned long long*)take(sizeof(unsigned long long) * (size_t)B * w.mask_stride);
    w.bytes = off;
    return w;
}

}  // namespace yolo_amd

using namespace yolo_amd;

extern "C" size_t yolo_amd_filter_results_workspace_bytes(int B, int N, int C) {
    if (B <= 0 || N <= 0 || C <= 0) return 0;
    return carve(nullptr, B, N, C).bytes;
}

// The kernels of filter_results on a workspace the caller of this function
// provides (the library user's buffer, or this library's static scratch).
static int run_filter_results(const float* pred, int B, int N, int D, int C, float conf, float nms_thresh,
                              void* workspace, size_t workspace_bytes, float* out_rows, int* out_count,
                              hipStream_t stream) {
    YA_REQUIRE(pred && workspace && out_rows && out_count, "filter_results: null pointer");
    YA_REQUIRE(B > 0 && N > 0 && C > 0, "filter_results: B, N, C must be positive (got %d, %d, %d)", B, N, C);
    YA_REQUIRE(D >= 5 + C, "filter_results: D=%d is smaller than 5 + C=%d", D, 5 + C);
    YA_REQUIRE(N <= REDUCE_MAX_WORDS * 64, "filter_results: N=%d exceeds %d boxes per image", N,
               REDUCE_MAX_WORDS * 64);
    YA_REQUIRE(C < (1 << (64 - KEY_CLS_SHIFT)), "filter_results: C=%d exceeds the class field", C);
    YA_REQUIRE(((uintptr_t)pred & 3) == 0 && ((uintptr_t)workspace & 255) == 0,
               "filter_results: pred must be 4-byte and workspace 256-byte aligned");
    YA_REQUIRE((size_t)B * C < (1u << 30) && B <= 65535, "filter_results: B*C too large");
    NmsWorkspace w = carve(workspace, B, N, C);
    if (workspace_bytes < w.bytes) {
        set_error("filter_results: workspace holds %zu bytes, %zu needed", workspace_bytes, w.bytes);
        return YA_ERR_WORKSPACE;
    }
    YA_CHECK_HIP(hipMemsetAsync(w.cand_count, 0, w.zero_bytes, stream));

    // compact: 16 rows per wave and four lanes per row where G lists fit the 64-lane
    // bookkeeping (always: G <= 8) and the block's tiles fit 64 KiB; else 64 rows per wave,
    // LDS per wave = 64*D floats, fewer waves per block for wide rows
    const size_t wave_lds = align_up((size_t)64 * D * sizeof(float) + 16, 16);
    const size_t wave4_lds = align_up((size_t)COMPACT4_ROWS * D * sizeof(float) + 16, 16);
    const size_t block4_lds = wave4_lds * COMPACT4_WAVES + (COMPACT4_WAVES + 1) * 64 * sizeof(int);
    YA_REQUIRE(wave_lds <= 150 * 1024, "filter_results: D=%d rows do not fit the LDS tile", D);
    if (block4_lds <= 64 * 1024 && w.G <= 64) {
        dim3 grid((N + COMPACT4_ROWS * COMPACT4_WAVES - 1) / (COMPACT4_ROWS * COMPACT4_WAVES), B);
        hipLaunchKernelGGL(nms_compact4_kernel, grid, dim3(COMPACT4_WAVES * 64), block4_lds, stream, pred, N, D, C,
                           conf, w.raw_stride, w.G, (int)wave4_lds, w.cand_count, w.raw_keys, w.boxes, w.scores);
    } else if (wave_lds * 2 <= 64 * 1024) {
        dim3 grid((N + 127) / 128, B);
        hipLaunchKernelGGL(nms_compact_kernel<2>, grid, dim3(128), wave_lds * 2, stream, pred, N, D, C,
                           conf, w.raw_stride, w.G, (int)wave_lds, w.cand_count, w.raw_keys, w.boxes, w.scores);
    } else {
        dim3 grid((N + 63) / 64, B);
        hipLaunchKernelGGL(nms_compact_kernel<1>, grid, dim3(64), wave_lds, stream, pred, N, D, C,
                           conf, w.raw_stride, w.G, (int)wave_lds, w.cand_count, w.raw_keys, w.boxes, w.scores);
    }
    hipLaunchKernelGGL(nms_sort_kernel, dim3(w.G, B), dim3(SORT_THREADS), 0, stream, w.cand_count, w.G,
                       w.raw_stride, w.raw_keys, w.key_stride, C, w.keys, w.seg_start, w.seg_end);
    hipLaunchKernelGGL(nms_plan_kernel, dim3(B), dim3(SORT_THREADS), 0, stream, C, w.seg_start, w.seg_end,
                       w.tile_base, w.mask_off, w.total_tiles);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(MASK_GRID_X, B), dim3(256), 0, stream, w.keys, w.key_stride,
                       N, C, w.seg_start, w.seg_end, w.tile_base, w.mask_off, w.total_tiles, w.boxes, 1,
                       nms_thresh, thresh_midpoint(nms_thresh), w.mask, w.mask_stride);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(C, B), dim3(REDUCE_THREADS), 0, stream, N, C, w.seg_start, w.seg_end,
                       w.mask_off, w.mask, w.mask_stride, w.kept_pos, w.seg_kept);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), 0, stream, w.seg_kept, B * C, w.seg_off,
                       out_count);
    hipLaunchKernelGGL(nms_gather_kernel, dim3((N + 255) / 256, B), dim3(256), 0, stream, w.keys, w.key_stride, N, C,
                       w.cand_count, w.G, w.seg_start, w.seg_kept, w.seg_off, w.kept_pos, w.boxes, w.scores, out_rows);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}

extern "C" int yolo_amd_filter_results(const float* pred, int B, int N, int D, int C, float conf,
                                       float nms_thresh, void* workspace, size_t workspace_bytes,
                                       float* out_rows, int* out_count, void* stream_) {
    return run_filter_results(pred, B, N, D, C, conf, nms_thresh, workspace, workspace_bytes, out_rows,
                              out_count, (hipStream_t)stream_);
}

// ---------------------------------------------------------------- filter_results without a caller workspace
// For callers that cannot allocate (code under HIP graph capture, tight serving
// loops): the workspace of filter_results lives in a file-scope device array
// that every GPU the library runs on holds until the process ends.  In host
// code the array's name is only a stand-in for the symbol, not a device address:
// the address on a device comes from hipGetSymbolAddress, taken once per device
// with that device current, and only that pointer is ever handed to a kernel
// or to hipMemsetAsync.
static __device__ __align__(256) unsigned char scratch[YOLO_AMD_STATIC_SCRATCH_BYTES];

constexpr int MAX_DEVICES = 64;
static void* g_scratch_ptr[MAX_DEVICES];           // per device: where `scratch` lives there
// Per device: the stream of the last call and an event recorded behind its kernels, so
// that a call on ANOTHER stream while that work may still be running (the one way two
// users of the one array can overlap) is refused instead of silently corrupting both.
static hipStream_t g_scratch_stream[MAX_DEVICES];
static hipEvent_t g_scratch_event[MAX_DEVICES];
static bool g_scratch_pending[MAX_DEVICES];
static std::mutex g_scratch_mutex;

extern "C" int yolo_amd_filter_results_static(const float* pred, int B, int N, int D, int C, float conf,
                                              float nms_thresh, float* out_rows, int* out_count,
                                              void* stream_) {
    YA_REQUIRE(B > 0 && N > 0 && C > 0, "filter_results_static: B, N, C must be positive (got %d, %d, %d)", B, N, C);
    const size_t need = carve(nullptr, B, N, C).bytes;
    if (need > (size_t)YOLO_AMD_STATIC_SCRATCH_BYTES) {
        set_error("filter_results_static: B=%d, N=%d, C=%d needs %zu bytes of workspace, the static scratch holds "
                  "%zu; use yolo_amd_filter_results with a caller workspace", B, N, C, need,
                  (size_t)YOLO_AMD_STATIC_SCRATCH_BYTES);
        return YA_ERR_WORKSPACE;
    }
    int dev = -1;
    YA_CHECK_HIP(hipGetDevice(&dev));
    YA_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "filter_results_static: device %d out of range", dev);
    hipStream_t stream = (hipStream_t)stream_;
    // (held over the launches: two host threads cannot interleave their kernels on the array)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        // set-up, once per device (make the first call outside a graph capture)
        void* p = nullptr;
        YA_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        YA_REQUIRE(p && ((uintptr_t)p & 255) == 0, "filter_results_static: bad address for the static scratch");
        YA_CHECK_HIP(hipEventCreateWithFlags(&g_scratch_event[dev], hipEventDisableTiming));
        g_scratch_ptr[dev] = p;
    }
    void* ptr = g_scratch_ptr[dev];
    if (g_scratch_pending[dev] && g_scratch_stream[dev] != stream) {
        const hipError_t q = hipEventQuery(g_scratch_event[dev]);
        if (q == hipErrorNotReady) {
            set_error("filter_results_static: the previous call on this GPU was queued on another stream and has "
                      "not finished; the one static scratch cannot serve two streams at once (queue the calls on "
                      "one stream, or use yolo_amd_filter_results with a workspace per stream)");
            return YA_ERR_BUSY;
        }
        YA_CHECK_HIP(q);
        g_scratch_pending[dev] = false;
    }
    const int rc = run_filter_results(pred, B, N, D, C, conf, nms_thresh, ptr, (size_t)YOLO_AMD_STATIC_SCRATCH_BYTES,
                                      out_rows, out_count, stream);
    // Remember where this call's kernels are queued.  A stream under graph capture cannot
    // carry the guard (an event recorded into a graph cannot be queried): calls replayed
    // from a graph are the caller's to keep on one stream.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone) {
        if (hipEventRecord(g_scratch_event[dev], stream) == hipSuccess) {
            g_scratch_stream[dev] = stream;
            g_scratch_pending[dev] = true;
        }
    } else {
        (void)hipGetLastError();
    }
    return rc;
}


extern "C" size_t yolo_amd_filter_img_class_workspace_bytes(int m) {
    if (m <= 0) return 0;
    return carve(nullptr, 1, m, 1).bytes;
}

extern "C" int yolo_amd_filter_img_class(const float* rows8, int m, float cls, float nms_thresh,
                                         void* workspace, size_t workspace_bytes, float* out_rows8,
                                         int* out_count, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(rows8 && workspace && out_rows8 && out_count, "filter_img_class: null pointer");
    YA_REQUIRE(m > 0 && m <= REDUCE_MAX_WORDS * 64, "filter_img_class: m=%d out of range", m);
    YA_REQUIRE(((uintptr_t)rows8 & 15) == 0 && ((uintptr_t)out_rows8 & 15) == 0 &&
                   ((uintptr_t)workspace & 255) == 0,
               "filter_img_class: rows must be 16-byte and workspace 256-byte aligned");
    NmsWorkspace w = carve(workspace, 1, m, 1);
    if (workspace_bytes < w.bytes) {
        set_error("filter_img_class: workspace holds %zu bytes, %zu needed", workspace_bytes, w.bytes);
        return YA_ERR_WORKSPACE;
    }
    YA_CHECK_HIP(hipMemsetAsync(w.cand_count, 0, w.zero_bytes, stream));
    hipLaunchKernelGGL(rows8_keys_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, rows8, m, cls,
                       w.cand_count, w.raw_keys);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(1, 1), dim3(SORT_THREADS), 0, stream, w.cand_count, 1,
                       w.raw_stride, w.raw_keys, w.key_stride, 1, w.keys, w.seg_start, w.seg_end);
    hipLaunchKernelGGL(nms_plan_kernel, dim3(1), dim3(SORT_THREADS), 0, stream, 1, w.seg_start, w.seg_end,
                       w.tile_base, w.mask_off, w.total_tiles);
    hipLaunchKernelGGL(nms_mask_kernel, dim3(MASK_GRID_X, 1), dim3(256), 0, stream, w.keys, w.key_stride,
                       m, 1, w.seg_start, w.seg_end, w.tile_base, w.mask_off, w.total_tiles,
                       reinterpret_cast<const float4*>(rows8), 2, nms_thresh, thresh_midpoint(nms_thresh), w.mask, w.mask_stride);
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(1, 1), dim3(REDUCE_THREADS), 0, stream, m, 1, w.seg_start, w.seg_end,
                       w.mask_off, w.mask, w.mask_stride, w.kept_pos, w.seg_kept);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(1024), 0, stream, w.seg_kept, 1, w.seg_off,
                       out_count);
    hipLaunchKernelGGL(rows8_gather_kernel, dim3(min(256, (m + 63) / 64)), dim3(64), 0, stream, w.keys,
                       w.seg_kept, w.kept_pos, rows8, out_rows8);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}

extern "C" int yolo_amd_box_iou(const float* box1, int n1, const float* box2, int m, float* out,
                                void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(m >= 0 && (n1 == 1 || n1 == m), "box_iou: box1 must hold 1 or m=%d boxes (got %d)", m, n1);
    if (m == 0) return YA_OK;
    YA_REQUIRE(box1 && box2 && out, "box_iou: null pointer");
    hipLaunchKernelGGL(box_iou_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, box1, n1, box2, m, out);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}
