// YOLO head decode for gfx950: all heads of a forward pass in ONE launch,
// written straight into the final (B, N, D) prediction tensor at each head's
// row offset.  Restates yolo_v3.predict (/root/reference/yolo_v3.py:205-262):
//
//   row = (gy*G + gx)*A + a                      (yolo_v3.py:214-216)
//   bx = (sigmoid(tx) + gx) * stride             (:218, :246, :261)
//   by = (sigmoid(ty) + gy) * stride
//   bw = (exp(tw) * aw') * stride                (:255, :261)  aw' = float32(anchor_w / stride)
//   bh = (exp(th) * ah') * stride
//   obj = sigmoid(to), cls_j = sigmoid(tc_j)     (:220, :258)
//
// fp32 throughout, no clamping of exp.  Pure HBM streaming: each input element
// is read once and each output element written once.
#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

struct HeadArg {
    const float* raw;
    int G;
    int layout;          // YOLO_AMD_LAYOUT_NCHW or _NHWC
    int row_offset;
    int elems;           // G*G*A*D, elements of this head per image
    float stride;
    float aw[YOLO_AMD_MAX_ANCHORS];
    float ah[YOLO_AMD_MAX_ANCHORS];
    FastDiv divG;
};

struct DecodeArgs {
    HeadArg head[YOLO_AMD_MAX_HEADS];
    int n_heads;
    int A, D, N;
    FastDiv divD, divA;
    int blocks_per_head[YOLO_AMD_MAX_HEADS + 1];   // prefix sums of blocks along grid.x
};

constexpr int DEC_THREADS = 256;
constexpr int DEC_PER_THREAD = 4;

__global__ void __launch_bounds__(DEC_THREADS)
decode_kernel(DecodeArgs args, float* __restrict__ out) {
    // which head does this block belong to (uniform)
    int h = 0;
    while (h + 1 < args.n_heads && (int)blockIdx.x >= args.blocks_per_head[h + 1]) ++h;
    const HeadArg& hd = args.head[h];
    const int b = blockIdx.y;
    const int blk = blockIdx.x - args.blocks_per_head[h];
    const int A = args.A, D = args.D;
    float* o = out + ((size_t)b * args.N + hd.row_offset) * D;
    const float* in = hd.raw + (size_t)b * hd.elems;
    const int GG = hd.G * hd.G;
#pragma unroll
    for (int u = 0; u < DEC_PER_THREAD; ++u) {
        const int t = (blk * DEC_PER_THREAD + u) * DEC_THREADS + threadIdx.x;
        if (t >= hd.elems) break;
        const uint32_t r = fdiv_u32((uint32_t)t, args.divD);      // row inside the head
        const int d = t - (int)r * D;
        const uint32_t pix = fdiv_u32(r, args.divA);
        const int a = (int)r - (int)pix * A;
        float x;
        if (hd.layout == YOLO_AMD_LAYOUT_NHWC) {
            x = in[t];
        } else {
            x = in[(size_t)(a * D + d) * GG + pix];
        }
        float gxf = 0.f, gyf = 0.f;
        if (d < 2) {
            const uint32_t gy = fdiv_u32(pix, hd.divG);
            gxf = (float)((int)pix - (int)gy * hd.G);
            gyf = (float)(int)gy;
        }
        const float v = decode_value(x, d, gxf, gyf, d == 2 ? hd.aw[a] : hd.ah[a], hd.stride);
        o[t] = v;
    }
}

}  // namespace yolo_amd

using namespace yolo_amd;

extern "C" int yolo_amd_decode(const yolo_amd_head_desc* heads, int n_heads, int B, int A, int D,
                               int N_total, float* out, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(heads && out, "decode: null pointer");
    YA_REQUIRE(n_heads >= 1 && n_heads <= YOLO_AMD_MAX_HEADS, "decode: n_heads=%d out of range", n_heads);
    YA_REQUIRE(A >= 1 && A <= YOLO_AMD_MAX_ANCHORS, "decode: A=%d out of range (max %d)", A, YOLO_AMD_MAX_ANCHORS);
    YA_REQUIRE(B >= 1 && B <= 65535 && D >= 5, "decode: bad B=%d or D=%d", B, D);
    DecodeArgs args;
    memset(&args, 0, sizeof(args));
    args.n_heads = n_heads;
    args.A = A; args.D = D; args.N = N_total;
    args.divD = make_fastdiv((uint32_t)D);
    args.divA = make_fastdiv((uint32_t)A);
    int rows = 0, blocks = 0;
    for (int h = 0; h < n_heads; ++h) {
        const yolo_amd_head_desc& s = heads[h];
        YA_REQUIRE(s.raw && s.G >= 1, "decode: head %d has a null tensor or G < 1", h);
        YA_REQUIRE(s.layout == YOLO_AMD_LAYOUT_NCHW || s.layout == YOLO_AMD_LAYOUT_NHWC,
                   "decode: head %d has unknown layout %d", h, s.layout);
        const long long elems = (long long)s.G * s.G * A * D;
        YA_REQUIRE(elems < (1ll << 31), "decode: head %d too large", h);
        YA_REQUIRE(s.row_offset >= 0 && s.row_offset + s.G * s.G * A <= N_total,
                   "decode: head %d rows [%d, %d) fall outside N_total=%d", h, s.row_offset,
                   s.row_offset + s.G * s.G * A, N_total);
        HeadArg& d = args.head[h];
        d.raw = s.raw; d.G = s.G; d.layout = s.layout; d.row_offset = s.row_offset;
        d.elems = (int)elems; d.stride = s.stride;
        for (int a = 0; a < A; ++a) { d.aw[a] = s.anchor_w[a]; d.ah[a] = s.anchor_h[a]; }
        d.divG = make_fastdiv((uint32_t)s.G);
        args.blocks_per_head[h] = blocks;
        blocks += (int)((elems + DEC_THREADS * DEC_PER_THREAD - 1) / (DEC_THREADS * DEC_PER_THREAD));
        rows += s.G * s.G * A;
    }
    args.blocks_per_head[n_heads] = blocks;
    YA_REQUIRE(rows <= N_total, "decode: heads hold %d rows, N_total=%d", rows, N_total);
    hipLaunchKernelGGL(decode_kernel, dim3(blocks, B), dim3(DEC_THREADS), 0, stream, args, out);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}
