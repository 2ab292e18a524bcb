// LDS image of one operand tile of the implicit-GEMM convolution, shared by
// the device code (conv.hip) and a host-side checker (tests/test_conv_layout.py
// drives csrc/conv_layout_check.cpp).
//
// A tile is R rows (pixels, or output channels) x BK bf16 of the reduction
// dimension.  It is cut into 16-byte chunks (8 bf16); CH = BK/8 chunks per row.
// The image is filled by LDS-DMA: wave-instruction j writes the 64 slots
// [64*j, 64*j + 64) in lane order, so slot p = 64*j + lane.  Which (row,
// logical chunk) a slot holds is free, because each lane supplies its own
// global source address; it is chosen so that the MFMA fragment reads
// (ds_read_b128: lane l reads row base + (l & 15), chunk 4*ks + (l >> 4)) hit
// 16 distinct 16-byte bank slots in every 16-lane service group.
#pragma once

#ifdef __HIPCC__
#define YA_HD __host__ __device__ __forceinline__
#else
#define YA_HD inline
#endif

namespace yolo_amd {

// XOR mask applied to the chunk index of a row.
template <int BK>
YA_HD int chunk_swizzle(int row);

// BK = 64: 8 chunks (128 B) per row, two rows per 256-byte bank row.
template <>
YA_HD int chunk_swizzle<64>(int row) { return (row >> 1) & 7; }

// BK = 32: 4 chunks (64 B) per row, four rows per bank row.  g = {0,2,3,1}
// indexed by (row >> 2) & 3 (a plain (row >> 2) & 3 leaves a 2-way conflict).
template <>
YA_HD int chunk_swizzle<32>(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }

// slot (16-byte units from the tile base) of logical chunk `c` of row `row`
template <int BK>
YA_HD int slot_of(int row, int c) { return row * (BK / 8) + (c ^ chunk_swizzle<BK>(row)); }

// inverse: which (row, logical chunk) does slot p hold
template <int BK>
YA_HD void slot_to_row_chunk(int p, int& row, int& c) {
    constexpr int CH = BK / 8;
    row = p / CH;
    c = (p % CH) ^ chunk_swizzle<BK>(row);
}

// Tile walk of a persistent grid of G workgroups over ntiles tiles, XCD-aware.  Workgroups are
// dealt round-robin over the eight XCDs (blockIdx % 8 says which share an L2; a speed assumption,
// never a correctness one), and neighbouring spatial tiles share halo pixels and cache lines:
// every group of workgroups therefore takes one CONTIGUOUS eighth of the tiles and walks it
// side by side (tile = first + i * step while < end), so that the tiles in flight on an XCD at
// any moment are a compact run of the image and their shared lines are fetched from HBM once.
// Every tile is visited by exactly one workgroup for any G >= 1 (checked on the host).
struct TileWalk {
    int first, end, step;
};
YA_HD TileWalk tile_walk_of(int ntiles, int G, int block) {
    const int ng = G < 8 ? G : 8;                              // groups (a grid of fewer than eight: one workgroup each)
    const int g = block % ng, loc = block / ng;
    const int nb = (G - g + ng - 1) / ng;                      // workgroups in this group
    const int lo = (int)(((long long)ntiles * g) / ng), hi = (int)(((long long)ntiles * (g + 1)) / ng);
    return TileWalk{lo + loc, hi, nb};
}

}  // namespace yolo_amd
