// Box post-processing of the callers for gfx950: map detections from the
// letterboxed network frame back to each original image and clamp them to it.
// Restates the block after filter_results in /root/reference/detection.py:52-70
// (the same block is in /root/reference/video.py:78-91), one fp32 operation per
// reference operation, in its order:
//
//   dims = im_dim_list[image]                       (w, h)       detection.py:53-54
//   sf   = min((1 / w) * height, (1 / h) * height)               :55-56
//          (`int / tensor` is Tensor.__rtruediv__: reciprocal, then multiply)
//   x   -= (height - sf * w) / 2     (columns 1, 3)              :62-63
//   y   -= (height - sf * h) / 2     (columns 2, 4)              :64-65
//   x, y /= sf                                                   :66
//   x = clamp(x, 0, w),  y = clamp(y, 0, h)                      :67-71
//
// Column 0 (image), 5-7 (scores, class) and 8 (the ymin copy filter_results
// leaves there) are not touched, as in the reference.
#include "common.h"
#include "../../include/yolo_amd.h"

namespace yolo_amd {

// torch.min / torch.clamp propagate NaN
__device__ __forceinline__ float pmin(float a, float b) { return (a != a) ? a : ((b != b) ? b : (a < b ? a : b)); }
__device__ __forceinline__ float pclamp(float v, float lo, float hi) {
    if (v != v) return v;
    v = v < lo ? lo : v;
    return v > hi ? hi : v;
}

__global__ void __launch_bounds__(256)
rescale_boxes_kernel(float* __restrict__ rows, int K, const float* __restrict__ dims, int B, float height) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= K) return;
    float* r = rows + (size_t)i * 9;
    const long long img = (long long)r[0];               // output[:, 0].long()
    if (img < 0 || img >= B) return;                      // index_select would have raised; leave the row
    const float w = dims[2 * img], h = dims[2 * img + 1];
    const float sf = pmin(__fmul_rn(__fdiv_rn(1.0f, w), height), __fmul_rn(__fdiv_rn(1.0f, h), height));
    const float px = __fdiv_rn(__fsub_rn(height, __fmul_rn(sf, w)), 2.0f);
    const float py = __fdiv_rn(__fsub_rn(height, __fmul_rn(sf, h)), 2.0f);
    r[1] = pclamp(__fdiv_rn(__fsub_rn(r[1], px), sf), 0.0f, w);
    r[3] = pclamp(__fdiv_rn(__fsub_rn(r[3], px), sf), 0.0f, w);
    r[2] = pclamp(__fdiv_rn(__fsub_rn(r[2], py), sf), 0.0f, h);
    r[4] = pclamp(__fdiv_rn(__fsub_rn(r[4], py), sf), 0.0f, h);
}

// One wave that does nothing for about `ticks` ticks of the 100 MHz wall clock: occupies a stream without
// occupying the GPU (the harness uses it to find streams that run side by side).
__global__ void __launch_bounds__(64) spin_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

}  // namespace yolo_amd

using namespace yolo_amd;

extern "C" int yolo_amd_spin(int microseconds, void* stream_) {
    YA_REQUIRE(microseconds >= 0 && microseconds <= 100000, "spin: %d us is outside [0, 100000]", microseconds);
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_, (long long)microseconds * 100);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}

extern "C" int yolo_amd_rescale_boxes(float* rows9, int K, const float* im_dims, int B, float height,
                                      void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    YA_REQUIRE(K >= 0 && B >= 1, "rescale_boxes: bad K=%d or B=%d", K, B);
    if (K == 0) return YA_OK;
    YA_REQUIRE(rows9 && im_dims, "rescale_boxes: null pointer");
    hipLaunchKernelGGL(rescale_boxes_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, rows9, K, im_dims,
                       B, height);
    YA_CHECK_HIP(hipGetLastError());
    return YA_OK;
}
