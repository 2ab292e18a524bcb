// Image pre-processing of the callers for gfx950: raw BGR frames in device memory
// to the (B, 3, H, H) fp32 network input.  Restates, per image,
//
//   LetterBoxImage_cv / letterbox_image   /root/reference/utilis.py:69-86, 495-509
//     new_w, new_h = int(w * min(W/w, H/h)), int(h * min(W/w, H/h))   (host, doubles)
//     resized = cv2.resize(img, (new_w, new_h), INTER_CUBIC)
//     canvas  = 128 everywhere, resized in the middle
//     canvas  = (canvas - canvas.min()) / (canvas.max() - canvas.min())   float64
//   ImgToTensorCv                          /root/reference/utilis.py:107-116
//     BGR -> RGB, HWC -> CHW, float32
//   transforms.Normalize(mean, std)        /root/reference/detection.py:34-35
//     float32 subtract, float32 divide
//   prep_image's tail                      /root/reference/utilis.py:514-515
//     float32 divide by 255 instead of the min-max step
//
// Two kernels, both bound by HBM traffic:
//   letterbox_kernel   one lane per canvas pixel: 4x4-tap bicubic of the source
//                      (OpenCV's 8-bit algorithm: 11-bit fixed-point weights from
//                      the Keys kernel with A = -0.75, horizontal then vertical
//                      integer sums, one rounding shift by 22, saturation; border
//                      pixels repeat) or the grey 128 outside the resized image;
//                      writes the uint8 canvas and folds the per-image minimum and
//                      maximum into two words (at most one atomic pair per block).
//   to_input_kernel    a canvas byte can take 256 values, so the whole float64 /
//                      float32 chain of the reference is evaluated once per value
//                      and channel into a 768-entry table in LDS; then four pixels
//                      per lane: 12 canvas bytes in, three float4 out (one per
//                      colour plane, channels flipped).
#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

constexpr int PRE_CHUNK = 32;   // images per launch: their geometry travels as a kernel argument

struct PreImage {
    const uint8_t* src;
    long long pitch;
    double scale_x, scale_y;    // source pixels per destination pixel
    int w, h;                   // source size
    int new_w, new_h;           // size of the resized image inside the canvas
    int left, top;              // its offset
};
struct PreBatch {
    PreImage img[PRE_CHUNK];
};

// Source coordinate of destination index d and the four fixed-point weights of the
// taps s-1 .. s+2.  Every float operation rounds on its own (the library is built
// without contraction), in the order OpenCV's interpolateCubic writes them.
__device__ __forceinline__ int cubic_taps(int d, double scale, int w[4]) {
    const float f = (float)(((double)d + 0.5) * scale - 0.5);
    const float fl = floorf(f);
    const float x = f - fl;
    const float A = -0.75f;
    const float xp = x + 1.0f, om = 1.0f - x;
    const float c0 = ((A * xp - 5.0f * A) * xp + 8.0f * A) * xp - 4.0f * A;
    const float c1 = ((A + 2.0f) * x - (A + 3.0f)) * x * x + 1.0f;
    const float c2 = ((A + 2.0f) * om - (A + 3.0f)) * om * om + 1.0f;
    const float c3 = 1.0f - c0 - c1 - c2;
    w[0] = (int)rintf(c0 * 2048.0f);
    w[1] = (int)rintf(c1 * 2048.0f);
    w[2] = (int)rintf(c2 * 2048.0f);
    w[3] = (int)rintf(c3 * 2048.0f);
    return (int)fl;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Source column (or row) the taps of destination index d are centred on.
__device__ __forceinline__ int cubic_origin(int d, double scale) {
    return (int)floorf((float)(((double)d + 0.5) * scale - 0.5));
}

constexpr int PATCH_BYTES = 40 * 1024 - 512;   // largest source footprint of a tile staged through LDS (4 blocks per CU)

// Row sums of the four taps when they are 12 consecutive bytes held in three words.
__device__ __forceinline__ void row_sums(const uint32_t q[3], const int wx[4], int hs[3]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = 3 * k + c;
            hs[c] += __mul24((int)((q[byte >> 2] >> (8 * (byte & 3))) & 255u), wx[k]);
        }
    }
}

// grid: (tiles of 32 x 8 canvas pixels, images of the chunk); block 256
//
// The bicubic gather is bound by the number of vector-memory instructions: every
// lane has its own address, and such an instruction occupies the CU's address unit
// for 16 to 27 cycles whatever its width.  So the source rectangle a tile needs
// (the footprint of its 32 x 8 destination pixels plus the tap border) is first
// copied to LDS with 16-byte loads, a whole wave reading consecutive chunks, and
// the 48 taps of a pixel then come out of LDS.  A tile whose footprint does not fit
// PATCH_BYTES (shrinking by more than about 6.5 x, where neighbouring pixels share
// no taps anyway) gathers from global memory instead.  Products fit 24-bit
// multiplies: bytes times weights (|w| < 2^12) in the row sums, row sums (< 2^21)
// times weights.
__global__ void __launch_bounds__(256)
letterbox_kernel(const PreBatch batch, int side, int tiles_x, uint8_t* __restrict__ canvas,
                 unsigned int* __restrict__ minmax) {
    __shared__ __align__(16) uint8_t patch[PATCH_BYTES + 32];
    const PreImage& im = batch.img[blockIdx.y];
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int cx = tx * 32 + (threadIdx.x & 31), cy = ty * 8 + (threadIdx.x >> 5);

    // block-uniform: destination range of the tile inside the resized image and its
    // source footprint [x_lo, x_hi] x [y_lo, y_hi]
    const int dxa = max(tx * 32 - im.left, 0), dxb = min(tx * 32 + 31 - im.left, im.new_w - 1);
    const int dya = max(ty * 8 - im.top, 0), dyb = min(ty * 8 + 7 - im.top, im.new_h - 1);
    const bool any = dxa <= dxb && dya <= dyb;
    int x_lo = 0, y_lo = 0, rowb = 0;
    bool staged = false;
    if (any) {
        x_lo = clampi(cubic_origin(dxa, im.scale_x) - 1, 0, im.w - 1);
        y_lo = clampi(cubic_origin(dya, im.scale_y) - 1, 0, im.h - 1);
        const int x_hi = clampi(cubic_origin(dxb, im.scale_x) + 2, 0, im.w - 1);
        const int y_hi = clampi(cubic_origin(dyb, im.scale_y) + 2, 0, im.h - 1);
        const int nx = x_hi - x_lo + 1, ny = y_hi - y_lo + 1;
        // a row's first byte sits 0..15 bytes into its first 16-byte chunk
        rowb = (3 * nx + 15 + 15) & ~15;
        staged = (long long)rowb * ny <= PATCH_BYTES;
        if (staged) {
            const int per_row = rowb >> 4;
            for (int i = threadIdx.x; i < per_row * ny; i += 256) {
                const int ry = i / per_row, j = i - ry * per_row;
                const uint8_t* first = im.src + (long long)(y_lo + ry) * im.pitch + 3 * x_lo;
                const uint8_t* chunk = first - ((uintptr_t)first & 15) + 16 * j;
                // an aligned chunk that holds a byte of the row stays inside the row's
                // memory page; one wholly behind the row's last needed byte is skipped
                if (chunk < first + 3 * nx)
                    *reinterpret_cast<uint4*>(patch + ry * rowb + 16 * j) = *reinterpret_cast<const uint4*>(chunk);
            }
        }
    }
    __syncthreads();

    unsigned int lo = 255u, hi = 0u;
    if (cx < side && cy < side) {
        int v[3] = {128, 128, 128};
        const int dx = cx - im.left, dy = cy - im.top;
        if (dx >= 0 && dx < im.new_w && dy >= 0 && dy < im.new_h) {
            int wx[4], wy[4];
            const int sx = cubic_taps(dx, im.scale_x, wx);
            const int sy = cubic_taps(dy, im.scale_y, wy);
            int xo[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) xo[k] = 3 * clampi(sx + k - 1, 0, im.w - 1);
            // away from the left and right border the four taps of a row are 12
            // consecutive bytes: fetched as the aligned words that cover them
            const bool inner = sx >= 1 && sx + 2 <= im.w - 1;
            int acc[3] = {0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = clampi(sy + r - 1, 0, im.h - 1);
                const uint8_t* row = im.src + (long long)y * im.pitch;
                int hs[3] = {0, 0, 0};
                if (staged) {
                    // offset of source byte 3 * x_lo of this row inside its LDS row
                    const int lead = (int)(((uintptr_t)row + 3 * x_lo) & 15);
                    const int base = (y - y_lo) * rowb + lead - 3 * x_lo;
                    if (inner) {
                        const int b = base + xo[0];
                        const uint32_t* wp = reinterpret_cast<const uint32_t*>(patch + (b & ~3));
                        const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3];
                        uint32_t q[3];
                        q[0] = __builtin_amdgcn_alignbyte(d1, d0, b & 3);
                        q[1] = __builtin_amdgcn_alignbyte(d2, d1, b & 3);
                        q[2] = __builtin_amdgcn_alignbyte(d3, d2, b & 3);
                        row_sums(q, wx, hs);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
#pragma unroll
                            for (int c = 0; c < 3; ++c) hs[c] += __mul24((int)patch[base + xo[k] + c], wx[k]);
                        }
                    }
                } else if (inner) {
                    // An aligned word that holds a byte of the row cannot leave the
                    // row's memory page; the fourth word is only needed when the span
                    // is unaligned and repeats the third otherwise.
                    const uintptr_t addr = (uintptr_t)(row + xo[0]);
                    const unsigned int shift = (unsigned int)(addr & 3);
                    const uint32_t* wp = reinterpret_cast<const uint32_t*>(addr - shift);
                    const uint32_t d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[shift ? 3 : 2];
                    uint32_t q[3];
                    q[0] = __builtin_amdgcn_alignbyte(d1, d0, shift);
                    q[1] = __builtin_amdgcn_alignbyte(d2, d1, shift);
                    q[2] = __builtin_amdgcn_alignbyte(d3, d2, shift);
                    row_sums(q, wx, hs);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) hs[c] += __mul24((int)row[xo[k] + c], wx[k]);
                    }
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += __mul24(hs[c], wy[r]);
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = clampi((acc[c] + (1 << 21)) >> 22, 0, 255);
        }
        uint8_t* out = canvas + ((size_t)blockIdx.y * side * side + (size_t)cy * side + cx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            out[c] = (uint8_t)v[c];
            lo = min(lo, (unsigned int)v[c]);
            hi = max(hi, (unsigned int)v[c]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo = min(lo, (unsigned int)__shfl_xor((int)lo, off));
        hi = max(hi, (unsigned int)__shfl_xor((int)hi, off));
    }
    // One update per block, and only when it would change the word: same-address
    // atomics are served one at a time, and after the first few blocks of an image
    // hardly any block still lowers either word.  Both words start as all-ones:
    // word 0 keeps the minimum, word 1 the minimum of 255 - value.  The plain reads
    // may be stale, which only costs an atomic that changes nothing.
    __shared__ unsigned int part[2][4];
    if ((threadIdx.x & 63) == 0) {
        part[0][threadIdx.x >> 6] = lo;
        part[1][threadIdx.x >> 6] = 255u - hi;
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned int v = min(min(part[threadIdx.x][0], part[threadIdx.x][1]),
                                   min(part[threadIdx.x][2], part[threadIdx.x][3]));
        unsigned int* word = &minmax[2 * blockIdx.y + threadIdx.x];
        if (v < __builtin_nontemporal_load(word)) atomicMin(word, v);
    }
}

struct PreNorm {
    float mean[3], stdv[3];
    int minmax, div255, normalise;
};

// grid: (groups of 256 x 4 pixels, images of the chunk); block 256
__global__ void __launch_bounds__(256)
to_input_kernel(const uint8_t* __restrict__ canvas, const unsigned int* __restrict__ minmax, int pixels,
                PreNorm p, float* __restrict__ dst) {
    __shared__ float table[3][256];
    const int img = blockIdx.y;
    const int mn = (int)minmax[2 * img], mx = 255 - (int)minmax[2 * img + 1];
    for (int e = threadIdx.x; e < 768; e += 256) {
        const int ch = e >> 8, v = e & 255;     // ch: output (RGB) channel
        float f;
        if (p.minmax) f = (float)((double)(v - mn) / (double)(mx - mn));
        else f = (float)v;
        if (p.div255) f = __fdiv_rn(f, 255.0f);
        if (p.normalise) f = __fdiv_rn(__fsub_rn(f, p.mean[ch]), p.stdv[ch]);
        table[ch][v] = f;
    }
    __syncthreads();
    const int q = blockIdx.x * 256 + threadIdx.x;      // group of four pixels
    if (4 * q >= pixels) return;
    const uint32_t* in = reinterpret_cast<const uint32_t*>(canvas + (size_t)img * pixels * 3) + 3 * (size_t)q;
    const uint32_t w0 = in[0], w1 = in[1], w2 = in[2];
    uint8_t b[12];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        b[k] = (w0 >> (8 * k)) & 255;
        b[4 + k] = (w1 >> (8 * k)) & 255;
        b[8 + k] = (w2 >> (8 * k)) & 255;
    }
    float* out = dst + (size_t)img * 3 * pixels + 4 * (size_t)q;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float4 o;      // output channel ch is canvas channel 2 - ch
        o.x = table[ch][b[0 + 2 - ch]];
        o.y = table[ch][b[3 + 2 - ch]];
        o.z = table[ch][b[6 + 2 - ch]];
        o.w = table[ch][b[9 + 2 - ch]];
        *reinterpret_cast<float4*>(out + (size_t)ch * pixels) = o;
    }
}

static size_t canvas_bytes(int batch, int side) { return align_up((size_t)batch * side * side * 3, 256); }

}  // namespace yolo_amd

using namespace yolo_amd;

extern "C" size_t yolo_amd_preprocess_workspace_bytes(int batch, int inp_dim) {
    if (batch < 1 || inp_dim < 1) return 0;
    return canvas_bytes(batch, inp_dim) + align_up((size_t)batch * 2 * sizeof(unsigned int), 256);
}

extern "C" int yolo_amd_preprocess(const yolo_amd_image* images, int batch, int inp_dim, int flags,
                                   const float* mean3, const float* std3, float* dst, uint8_t* canvas_out,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(images && batch >= 1, "preprocess: no images");
    YA_REQUIRE(inp_dim >= 4 && inp_dim % 4 == 0 && inp_dim <= 8192,
               "preprocess: input side %d must be a multiple of 4 in [4, 8192]", inp_dim);
    YA_REQUIRE((mean3 == nullptr) == (std3 == nullptr), "preprocess: mean and std come together");
    YA_REQUIRE(!((flags & YOLO_AMD_PRE_MINMAX) && (flags & YOLO_AMD_PRE_DIV255)),
               "preprocess: min-max and divide-by-255 exclude each other");
    YA_REQUIRE(dst || canvas_out, "preprocess: neither an input tensor nor a canvas was asked for");
    YA_REQUIRE(workspace, "preprocess: null workspace");
    const size_t need = yolo_amd_preprocess_workspace_bytes(batch, inp_dim);
    if (workspace_bytes < need) {
        set_error("preprocess: workspace of %zu bytes, %zu needed", workspace_bytes, need);
        return YA_ERR_WORKSPACE;
    }
    YA_REQUIRE(((uintptr_t)workspace & 255) == 0, "preprocess: workspace must be 256-byte aligned");
    YA_REQUIRE(!dst || ((uintptr_t)dst & 15) == 0, "preprocess: dst must be 16-byte aligned");
    uint8_t* canvas = canvas_out ? canvas_out : (uint8_t*)workspace;
    YA_REQUIRE(((uintptr_t)canvas & 3) == 0, "preprocess: canvas must be 4-byte aligned");
    unsigned int* minmax = (unsigned int*)((uint8_t*)workspace + canvas_bytes(batch, inp_dim));

    for (int i = 0; i < batch; ++i) {
        const yolo_amd_image& s = images[i];
        YA_REQUIRE(s.data && s.width >= 1 && s.height >= 1 && s.pitch_bytes >= 3LL * s.width,
                   "preprocess: image %d: bad pointer, size %dx%d or pitch %lld", i, s.width, s.height,
                   (long long)s.pitch_bytes);
    }

    PreNorm norm{};
    norm.minmax = (flags & YOLO_AMD_PRE_MINMAX) ? 1 : 0;
    norm.div255 = (flags & YOLO_AMD_PRE_DIV255) ? 1 : 0;
    norm.normalise = mean3 ? 1 : 0;
    for (int c = 0; c < 3; ++c) {
        norm.mean[c] = mean3 ? mean3[c] : 0.0f;
        norm.stdv[c] = std3 ? std3[c] : 1.0f;
    }

    YA_CHECK_HIP(hipMemsetAsync(minmax, 0xFF, (size_t)batch * 2 * sizeof(unsigned int), stream));
    const int pixels = inp_dim * inp_dim;
    const int tiles_x = (inp_dim + 31) / 32, tiles_y = (inp_dim + 7) / 8;
    for (int first = 0; first < batch; first += PRE_CHUNK) {
        const int n = batch - first < PRE_CHUNK ? batch - first : PRE_CHUNK;
        PreBatch pb{};
        for (int i = 0; i < n; ++i) {
            const yolo_amd_image& s = images[first + i];
            PreImage& d = pb.img[i];
            // utilis.py:75-78, evaluated in doubles as Python does
            const double W = inp_dim, H = inp_dim;
            const double rw = W / (double)s.width, rh = H / (double)s.height;
            const double r = rw < rh ? rw : rh;
            d.new_w = (int)((double)s.width * r);
            d.new_h = (int)((double)s.height * r);
            YA_REQUIRE(d.new_w >= 1 && d.new_h >= 1,
                       "preprocess: image %d (%dx%d) shrinks to nothing at input side %d", first + i, s.width,
                       s.height, inp_dim);
            d.left = (inp_dim - d.new_w) / 2;
            d.top = (inp_dim - d.new_h) / 2;
            d.src = (const uint8_t*)s.data;
            d.pitch = s.pitch_bytes;
            d.w = s.width;
            d.h = s.height;
            // cv::resize: inv_scale = dsize / ssize, scale = 1 / inv_scale
            d.scale_x = 1.0 / ((double)d.new_w / (double)s.width);
            d.scale_y = 1.0 / ((double)d.new_h / (double)s.height);
        }
        hipLaunchKernelGGL(letterbox_kernel, dim3(tiles_x * tiles_y, n), dim3(256), 0, stream, pb, inp_dim,
                           tiles_x, canvas + (size_t)first * pixels * 3, minmax + 2 * first);
        YA_CHECK_HIP(hipGetLastError());
        if (dst) {
            hipLaunchKernelGGL(to_input_kernel, dim3((pixels / 4 + 255) / 256, n), dim3(256), 0, stream,
                               canvas + (size_t)first * pixels * 3, minmax + 2 * first, pixels, norm,
                               dst + (size_t)first * 3 * pixels);
            YA_CHECK_HIP(hipGetLastError());
        }
    }
    return YA_OK;
}
