// BatchNorm2d on batch statistics for gfx950: what nn.BatchNorm2d(out_channels) as built by
// utilis.create_module (/root/reference/utilis.py:186-189) computes while the model is in
// train() mode -- the mode /root/reference/video.py:21-25 and get_fps.py:25-29 run
// yolo_v3.forward in, since they never call eval().  Batch statistics cannot be folded into
// the convolution's weights, so such a layer runs as
//
//   conv (unfolded bf16 weights, fp32 accumulation)  ->  raw [M][C] fp32        (conv.hip)
//   bn_partial_kernel    per-channel sum and sum of squares of row blocks, float64
//   bn_finalize_kernel   blocks added in ascending order; mean, biased variance,
//                        scale = gamma / sqrt(var + eps), shift = beta - mean * scale;
//                        running_mean / running_var updated as torch does
//                        (momentum, unbiased variance M / (M - 1))
//   bn_apply_kernel      raw * scale + shift, LeakyReLU(0.1), + residual, one rounding to
//                        bf16, NHWC; num_batches_tracked += 1
//
// Every sum has a fixed order, so the statistics (and with them every output bit) are the
// same from run to run.  All three kernels are bound by the raw tensor's bytes.
#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

typedef __bf16 bn_bf16x8 __attribute__((ext_vector_type(8)));

// Block = qpb channel quads x rl row lanes (qpb * rl <= 256); thread (tq, tr) adds rows
// r0 + tr, r0 + tr + rl, ... of its four channels, then the row lanes are added in lane
// order through LDS.  partial[block][channel][0 = sum, 1 = sum of squares].
__global__ void __launch_bounds__(256)
bn_partial_kernel(const float* __restrict__ raw, long long M, int C, int qpb, int rl, long long rows_per_block,
                  double* __restrict__ partial) {
    __shared__ double red[256 * 8];
    const int t = threadIdx.x;
    const int tq = t % qpb, tr = t / qpb;
    const int quad = blockIdx.y * qpb + tq;
    const bool active = tr < rl && quad * 4 < C;
    double s[4] = {0.0, 0.0, 0.0, 0.0}, ss[4] = {0.0, 0.0, 0.0, 0.0};
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    if (active) {
#pragma unroll 4
        for (long long r = r0 + tr; r < r1; r += rl) {
            const float4 v = *reinterpret_cast<const float4*>(raw + r * C + quad * 4);
            const double x0 = v.x, x1 = v.y, x2 = v.z, x3 = v.w;
            s[0] += x0; s[1] += x1; s[2] += x2; s[3] += x3;
            ss[0] += x0 * x0; ss[1] += x1 * x1; ss[2] += x2 * x2; ss[3] += x3 * x3;
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[t * 8 + k] = s[k];
        red[t * 8 + 4 + k] = ss[k];
    }
    __syncthreads();
    if (tr == 0 && quad * 4 < C) {
        for (int j = 1; j < rl; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s[k] += red[(j * qpb + tq) * 8 + k];
                ss[k] += red[(j * qpb + tq) * 8 + 4 + k];
            }
        }
        double* out = partial + ((size_t)blockIdx.x * C + quad * 4) * 2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out[2 * k] = s[k];
            out[2 * k + 1] = ss[k];
        }
    }
}

__global__ void __launch_bounds__(64)
bn_finalize_kernel(const double* __restrict__ partial, int S, int C, long long M, const float* __restrict__ gamma,
                   const float* __restrict__ beta, float* __restrict__ running_mean,
                   float* __restrict__ running_var, const long long* __restrict__ nbt, float eps, float momentum,
                   float* __restrict__ scale_shift) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, ss = 0.0;
    for (int i = 0; i < S; ++i) {
        s += partial[((size_t)i * C + c) * 2];
        ss += partial[((size_t)i * C + c) * 2 + 1];
    }
    const double mean = s / (double)M;
    double var = ss / (double)M - mean * mean;
    if (!(var > 0.0)) var = 0.0;
    const double scale = (double)gamma[c] / sqrt(var + (double)eps);
    scale_shift[c] = (float)scale;
    scale_shift[C + c] = (float)((double)beta[c] - mean * scale);
    // exponential moving average factor; momentum=None averages over all batches seen
    // (the counter is incremented by the apply kernel, after every block here has read it)
    float f = momentum;
    if (momentum < 0.f) f = 1.0f / (float)((nbt ? *nbt : 0) + 1);
    if (running_mean) running_mean[c] = (1.0f - f) * running_mean[c] + f * (float)mean;
    if (running_var) {
        const double unbiased = var * ((double)M / (double)(M - 1));
        running_var[c] = (1.0f - f) * running_var[c] + f * (float)unbiased;
    }
}

// One lane = eight channels of one pixel: 32 bytes of raw in, 16 bytes out.
__global__ void __launch_bounds__(256)
bn_apply_kernel(const float* __restrict__ raw, const float* __restrict__ scale_shift,
                const __bf16* __restrict__ res, __bf16* __restrict__ dst, long long total8, int C, int leaky,
                long long* __restrict__ nbt) {
    const int C8 = C >> 3;
    const long long step = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total8; i += step) {
        const int c = (int)(i % C8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(raw + i * 8);
        const float4 b = *reinterpret_cast<const float4*>(raw + i * 8 + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(scale_shift + c);
        const float4 s1 = *reinterpret_cast<const float4*>(scale_shift + c + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(scale_shift + C + c);
        const float4 h1 = *reinterpret_cast<const float4*>(scale_shift + C + c + 4);
        float v[8] = {fmaf(a.x, s0.x, h0.x), fmaf(a.y, s0.y, h0.y), fmaf(a.z, s0.z, h0.z), fmaf(a.w, s0.w, h0.w),
                      fmaf(b.x, s1.x, h1.x), fmaf(b.y, s1.y, h1.y), fmaf(b.z, s1.z, h1.z), fmaf(b.w, s1.w, h1.w)};
        if (leaky) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] > 0.f ? v[k] : 0.1f * v[k];
        }
        if (res) {
            const bn_bf16x8 rv = *reinterpret_cast<const bn_bf16x8*>(res + i * 8);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += (float)rv[k];
        }
        bn_bf16x8 o;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (__bf16)v[k];
        *reinterpret_cast<bn_bf16x8*>(dst + i * 8) = o;
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// Statistics + normalisation of a layer whose convolution has already been queued on
// `stream` into bn.raw ([M][C] fp32).  dst: [M][C] bf16; residual: the same shape or NULL.
int launch_bn_train(const yolo_amd_bn_train& bn, long long M, int C, const void* residual, void* dst, int leaky,
                    hipStream_t stream) {
    YA_REQUIRE(bn.raw && bn.partial && bn.scale_shift && bn.gamma && bn.beta && dst,
               "bn_train: null pointer (raw, partial, scale_shift, gamma, beta and dst are required)");
    YA_REQUIRE(C > 0 && C % 8 == 0, "bn_train: %d channels (a multiple of 8 is required)", C);
    YA_REQUIRE(M > 1, "bn_train: expected more than 1 value per channel when training, got %lld", M);
    YA_REQUIRE((((uintptr_t)bn.raw | (uintptr_t)dst | (uintptr_t)residual | (uintptr_t)bn.scale_shift) & 15) == 0 &&
                   ((uintptr_t)bn.partial & 7) == 0,
               "bn_train: raw, dst, residual and scale_shift must be 16-byte aligned");
    YA_REQUIRE(bn.eps >= 0.f && bn.momentum <= 1.f, "bn_train: eps=%g momentum=%g", (double)bn.eps, (double)bn.momentum);
    const int quads = C / 4;
    const int qpb = quads < 256 ? quads : 256;
    const int rl = 256 / qpb;
    long long S = (M + (long long)rl * 8 - 1) / ((long long)rl * 8);
    if (S > YOLO_AMD_BN_MAX_PARTIALS) S = YOLO_AMD_BN_MAX_PARTIALS;
    const long long rows_per_block = (M + S - 1) / S;
    S = (M + rows_per_block - 1) / rows_per_block;
    hipLaunchKernelGGL(bn_partial_kernel, dim3((unsigned)S, (unsigned)((quads + qpb - 1) / qpb)), dim3(256), 0, stream,
                       bn.raw, M, C, qpb, rl, rows_per_block, bn.partial);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, stream, bn.partial, (int)S,
                       C, M, bn.gamma, bn.beta, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.eps,
                       bn.momentum, bn.scale_shift);
    const long long total8 = M * (C / 8);
    long long blocks = (total8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bn.raw, bn.scale_shift,
                       (const __bf16*)residual, (__bf16*)dst, total8, C, leaky, bn.num_batches_tracked);
    return YA_OK;
}

}  // namespace yolo_amd
