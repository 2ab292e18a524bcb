// Host-side check of the LDS image used by the implicit-GEMM convolution
// (yolo_amd/csrc/conv_layout.h): the slot map is a bijection, and every
// 16-lane service group of a ds_read_b128 fragment read touches 16 distinct
// 16-byte slots of the 256-byte bank row (bank = (addr / 4) % 64).
#include <cstdio>
#include <set>
#include <vector>

#include "../../yolo_amd/csrc/conv_layout.h"
#include "../../yolo_amd/csrc/common_host.h"

using namespace yolo_amd;

// lanes serviced together by ds_read_b128 (MI355X LDS)
static const int GROUPS[4][16] = {
    {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};

template <int BK>
static int check(int rows) {
    constexpr int CH = BK / 8;
    int bad = 0;
    std::set<int> seen;
    for (int p = 0; p < rows * CH; ++p) {
        int r, c;
        slot_to_row_chunk<BK>(p, r, c);
        if (r < 0 || r >= rows || c < 0 || c >= CH) { ++bad; continue; }
        if (slot_of<BK>(r, c) != p) ++bad;
        seen.insert(r * CH + c);
    }
    if ((int)seen.size() != rows * CH) ++bad;
    // the fill-side claim: the logical chunk of a lane does not depend on which
    // of its wave's instructions (j = 4*i + w) it is executing
    for (int w = 0; w < 4; ++w)
        for (int lane = 0; lane < 64; ++lane) {
            int first = -1;
            for (int j = w; j * 64 < rows * CH; j += 4) {
                int r, c;
                slot_to_row_chunk<BK>(j * 64 + lane, r, c);
                if (r != j * (64 / CH) + lane / CH) ++bad;
                if (first < 0) first = c; else if (c != first) ++bad;
            }
        }
    // fragment reads: lane l reads row base + (l & 15), chunk 4*ks + (l >> 4)
    for (int base = 0; base + 16 <= rows; base += 16)
        for (int ks = 0; ks < BK / 32; ++ks)
            for (int g = 0; g < 4; ++g) {
                std::set<int> slots;
                for (int t = 0; t < 16; ++t) {
                    int l = GROUPS[g][t];
                    int addr = slot_of<BK>(base + (l & 15), ks * 4 + (l >> 4)) * 16;
                    slots.insert((addr / 16) % 16);
                }
                if (slots.size() != 16) ++bad;
            }
    return bad;
}

static int check_fastdiv() {
    int bad = 0;
    const unsigned ds[] = {1, 2, 3, 5, 7, 9, 13, 26, 52, 85, 169, 255, 676, 2704, 43264, 173056, 369664, 1u << 20, 3000001};
    for (unsigned d : ds) {
        FastDiv f = make_fastdiv(d);
        for (unsigned long long n = 0; n < (1ull << 31); n += 104729) {
            if (fdiv_u32_host((unsigned)n, f) != (unsigned)n / d) ++bad;
        }
        for (unsigned n : {0u, 1u, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1u << 31) - 1, (1u << 31) - d})
            if (fdiv_u32_host(n, f) != n / d) ++bad;
    }
    return bad;
}

// every tile of a persistent grid's walk is visited by exactly one workgroup
static int check_tile_walk() {
    int bad = 0;
    const int grids[] = {1, 2, 3, 5, 6, 7, 8, 9, 15, 16, 17, 63, 64, 100, 256, 512, 8192};
    std::vector<int> seen;
    for (int ntiles = 1; ntiles <= 12000; ntiles += (ntiles < 300 ? 1 : 487)) {
        for (int G : grids) {
            if (G > ntiles) continue;                    // the launchers size the grid min(ntiles, cap)
            seen.assign(ntiles, 0);
            for (int b = 0; b < G; ++b) {
                const TileWalk w = tile_walk_of(ntiles, G, b);
                if (w.step < 1) { ++bad; continue; }
                for (int t = w.first; t < w.end; t += w.step) {
                    if (t < 0 || t >= ntiles) { ++bad; break; }
                    ++seen[t];
                }
            }
            for (int t = 0; t < ntiles; ++t) bad += seen[t] != 1;
        }
    }
    return bad;
}

int main() {
    int bad = 0;
    bad += check<64>(128);
    bad += check<64>(32);
    bad += check<32>(128);
    bad += check<32>(32);
    int fd = check_fastdiv();
    int tw = check_tile_walk();
    printf("layout_bad=%d fastdiv_bad=%d tilewalk_bad=%d\n", bad, fd, tw);
    return (bad || fd || tw) ? 1 : 0;
}
