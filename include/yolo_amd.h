/*
 * libyolo_amd.so -- C ABI of the MI355X (gfx950) YOLOv3 inference hot path.
 *
 * The reference (jayden199012/yolo) is pure Python with no FFI of its own; the
 * Python functions named beside each entry point are the interfaces it
 * replaces.  The drop-in Python modules (yolo_amd/utilis.py, yolo_amd/yolo_v3.py)
 * bind these symbols with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all
 *     work is enqueued asynchronously on it and nothing synchronises;
 *   - the caller allocates every buffer, including worst-case outputs and the
 *     workspace; the library allocates nothing.  Process-wide state is limited
 *     to the thread-local last-error string and a cache of the YOLO_AMD_*
 *     environment knobs (integers; see yolo_amd_reload_knobs);
 *   - calls on different streams may run concurrently as long as they do not
 *     share a workspace, a split-K scratch or an output buffer;
 *   - return value: 0 on success, negative on error (see yolo_amd_last_error).
 */
#ifndef YOLO_AMD_H
#define YOLO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YOLO_AMD_ABI_VERSION 8
#define YOLO_AMD_MAX_HEADS 4
#define YOLO_AMD_MAX_ANCHORS 8

#define YOLO_AMD_LAYOUT_NCHW 0 /* (B, A*D, G, G): what yolo_v3.predict receives */
#define YOLO_AMD_LAYOUT_NHWC 1 /* (B, G, G, A*D): what the fused conv stack emits */

/* Message describing the last failing call on this thread (host string). */
const char* yolo_amd_last_error(void);
int yolo_amd_abi_version(void);

/* ---------------------------------------------------------------- box_iou
 * Replaces utilis.box_iou(box1, box2)      /root/reference/utilis.py:298-323
 * box1: n1 x 4 (n1 == 1 broadcasts, else n1 == m), box2: m x 4, corner form,
 * fp32.  out: m IoUs with the +1 pixel convention, bit-identical to the
 * reference's fp32 result (same operation order, IEEE division). */
int yolo_amd_box_iou(const float* box1, int n1, const float* box2, int m, float* out,
                     void* stream);

/* ---------------------------------------------------------------- decode
 * Replaces yolo_v3.predict(x, layer, batch_size) for every head at once
 *                                           /root/reference/yolo_v3.py:205-262
 * and the two torch.cat calls that assemble the heads (yolo_v3.py:488-493). */
typedef struct {
    const float* raw;   /* raw head tensor, fp32, layout below                */
    int G;              /* grid side                                          */
    int layout;         /* YOLO_AMD_LAYOUT_NCHW or YOLO_AMD_LAYOUT_NHWC       */
    int row_offset;     /* first row of this head inside the N_total rows     */
    float stride;       /* height // G                                        */
    float anchor_w[YOLO_AMD_MAX_ANCHORS]; /* float32(anchor_w / stride), the  */
    float anchor_h[YOLO_AMD_MAX_ANCHORS]; /* division done in float64 (host)  */
} yolo_amd_head_desc;

/* heads: host array.  out: (B, N_total, D) fp32; row (gy*G+gx)*A + a of each
 * head lands at row_offset + that row. */
int yolo_amd_decode(const yolo_amd_head_desc* heads, int n_heads, int B, int A, int D,
                    int N_total, float* out, void* stream);

/* ---------------------------------------------------------------- NMS
 * Replaces utilis.filter_results(prediction, params) and the
 * filter_img_class calls inside it      /root/reference/utilis.py:355-465
 *
 * pred: (B, N, D) fp32 decoded predictions, D >= 5 + C (never written).
 * out_rows: room for B*N rows of 9 fp32
 *           [image, x1, y1, x2, y2, objectness, class score, class index, y1]
 *           ordered image asc, class asc, objectness desc, original row asc.
 * out_count: one device int32 that receives K, the number of rows written
 *            (K == 0 is the reference's "return 0" case).
 * workspace: 256-byte aligned, at least
 *            yolo_amd_filter_results_workspace_bytes(B, N, C) bytes. */
size_t yolo_amd_filter_results_workspace_bytes(int B, int N, int C);
int yolo_amd_filter_results(const float* pred, int B, int N, int D, int C, float conf,
                            float nms_thresh, void* workspace, size_t workspace_bytes,
                            float* out_rows, int* out_count, void* stream);

/* yolo_amd_filter_results without a workspace argument, for callers that cannot
 * allocate device memory (code under HIP graph capture, serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  The
 * workspace is a static device array of YOLO_AMD_STATIC_SCRATCH_BYTES inside
 * the library (each GPU the library runs on holds one copy until the process
 * ends); shapes whose yolo_amd_filter_results_workspace_bytes(B, N, C) exceeds
 * it are rejected with a negative code and a message (16 MiB covers one 416 or
 * 320 image, two 416 images at C = 80, not a 608 image).  The call itself
 * allocates and frees nothing; the first call on a device looks up the array's
 * address (hipGetSymbolAddress) -- make that call outside a graph capture.
 * Same kernels, same rows, bit for bit, as yolo_amd_filter_results.
 * Every call on one GPU uses that GPU's one array: calls that overlap in time
 * on one GPU (two streams) would race; queue them on one stream.  The library
 * guards the common case: a call on a stream other than the previous call's,
 * while that call's kernels have not finished, returns -5 (busy) and launches
 * nothing.  (Set-up on the first call per device also creates the event the
 * guard uses; calls queued under graph capture are not tracked by it.) */
#define YOLO_AMD_STATIC_SCRATCH_BYTES (16u << 20)
int yolo_amd_filter_results_static(const float* pred, int B, int N, int D, int C, float conf,
                                   float nms_thresh, float* out_rows, int* out_count, void* stream);

/* Replaces utilis.filter_img_class(cls, image_pred, ind, nms_thesh)
 *                                         /root/reference/utilis.py:355-390
 * rows8: (m, 8) fp32 [x1, y1, x2, y2, objectness, class score, class index,
 * ymin], 16-byte aligned.  Rows whose class index equals `cls` are sorted by
 * objectness (descending, ties by row) and greedily suppressed; survivors are
 * written to out_rows8 (room for m rows) and their number to *out_count. */
size_t yolo_amd_filter_img_class_workspace_bytes(int m);
int yolo_amd_filter_img_class(const float* rows8, int m, float cls, float nms_thresh,
                              void* workspace, size_t workspace_bytes, float* out_rows8,
                              int* out_count, void* stream);

/* ---------------------------------------------------------------- caller post-processing
 * Replaces the rescale-and-clamp block the callers run on filter_results'
 * output           /root/reference/detection.py:52-70, /root/reference/video.py:78-91
 *
 * rows9: (K, 9) fp32 rows as written by yolo_amd_filter_results, modified in
 * place: columns 1-4 go from the letterboxed height x height frame to the
 * original image of the row (column 0) and are clamped to it.
 * im_dims: (B, 2) fp32 on the device, original (width, height) per image. */
int yolo_amd_rescale_boxes(float* rows9, int K, const float* im_dims, int B, float height,
                           void* stream);

/* Keeps `stream` busy for about `microseconds` (at most 100 000) with one idle wave.  No
 * counterpart in the reference: the host side uses it to find HIP streams whose kernels run
 * side by side (streams that share a hardware queue run one after the other). */
int yolo_amd_spin(int microseconds, void* stream);

/* ---------------------------------------------------------------- caller pre-processing
 * Replaces the image transforms the callers apply before model(images):
 *   LetterBoxImage_cv / LetterboxImage / letterbox_image
 *                              /root/reference/utilis.py:69-86, 89-104, 495-509
 *   ImgToTensorCv              /root/reference/utilis.py:107-116
 *   transforms.Normalize       /root/reference/detection.py:34-35
 *   prep_image                 /root/reference/utilis.py:512-516
 *
 * images: HOST array of `batch` descriptors of frames in DEVICE memory, 8-bit,
 * three interleaved channels in cv2.imread's order (BGR).  Each frame is resized
 * with its aspect kept (bicubic, OpenCV's 8-bit fixed-point INTER_CUBIC, no
 * antialiasing) onto an inp_dim x inp_dim canvas of grey 128.
 * flags: YOLO_AMD_PRE_MINMAX  (canvas - min) / (max - min) in float64 per image,
 *                             as LetterBoxImage_cv does (a flat canvas gives NaN);
 *        YOLO_AMD_PRE_DIV255  float32 / 255 as prep_image does; not both.
 * mean3/std3: HOST, three floats each in RGB order, or both NULL: (x - mean) / std.
 * dst: (batch, 3, inp_dim, inp_dim) fp32 on the device, channels RGB; may be NULL
 *      when only the canvas is wanted.
 * canvas_out: optional (batch, inp_dim, inp_dim, 3) uint8 on the device, the
 *      letterboxed frames before normalisation (what LetterboxImage returns);
 *      NULL keeps them in the workspace.
 * Fails with a message when a frame shrinks to zero width or height. */
typedef struct yolo_amd_image {
    const void* data;      /* device pointer to the first pixel */
    int32_t width, height;
    int64_t pitch_bytes;   /* distance between rows, >= 3 * width */
} yolo_amd_image;

#define YOLO_AMD_PRE_MINMAX 1
#define YOLO_AMD_PRE_DIV255 2

size_t yolo_amd_preprocess_workspace_bytes(int batch, int inp_dim);
int yolo_amd_preprocess(const yolo_amd_image* images, int batch, int inp_dim, int flags,
                        const float* mean3, const float* std3, float* dst, uint8_t* canvas_out,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------- conv stack
 * Replaces the conv / upsample / route / shortcut arms of yolo_v3.forward
 *                                          /root/reference/yolo_v3.py:449-473
 * executed over nn.Sequential(Conv2d, BatchNorm2d, LeakyReLU) modules built by
 * utilis.create_module                     /root/reference/utilis.py:169-236
 *
 * One descriptor = one fused layer: implicit-GEMM convolution (bf16 inputs,
 * fp32 accumulation on MFMA) + bias (BatchNorm folded on the host) +
 * optional LeakyReLU(0.1) + optional residual add (the following shortcut
 * block) on an input that may be the channel concatenation of a nearest-2x
 * upsampled tensor and a second tensor (the route block in front of it).
 * Activations are NHWC bf16.  Weights are bf16, zero padded to cout_pad
 * output channels, stored K-tile major as [K/32][cout_pad][32] with
 * k = (kh*ksize + kw)*Cin + c (Cin a multiple of 32): the 32-wide slice of
 * reduction indices a workgroup stages for its channel tile is one contiguous
 * run of whole cache lines. */

/* BatchNorm2d on BATCH statistics behind a convolution: what the modules of
 * utilis.create_module (/root/reference/utilis.py:186-189) compute while the model is
 * in train() mode, which is how /root/reference/video.py:21-25 and get_fps.py:25-29 run
 * yolo_v3.forward (they never call eval()).  A layer whose descriptor points at one of
 * these runs as four launches: the convolution with the UNFOLDED weights into `raw`
 * (fp32, bias added, no activation), per-channel sums over all B*Ho*Wo pixels
 * (float64, added in a fixed order: results do not vary from run to run), the
 * statistics (biased variance; running_mean / running_var / num_batches_tracked
 * updated in place as torch.nn.BatchNorm2d does, the running variance with the
 * unbiased estimate), then (raw - mean) / sqrt(var + eps) * gamma + beta, LeakyReLU,
 * the residual add and the one rounding to bf16 into the descriptor's dst.
 * All pointers are device pointers; the three scratch buffers may be shared by the
 * layers of one stream. */
#define YOLO_AMD_BN_MAX_PARTIALS 256
typedef struct {
    float* raw;               /* scratch: >= B*Ho*Wo*cout floats                       */
    double* partial;          /* scratch: >= YOLO_AMD_BN_MAX_PARTIALS * cout * 2 doubles */
    float* scale_shift;       /* scratch: >= 2 * cout floats                            */
    const float* gamma;       /* BatchNorm2d.weight [cout]                              */
    const float* beta;        /* BatchNorm2d.bias [cout]                                */
    float* running_mean;      /* [cout], updated in place; NULL: not tracked            */
    float* running_var;       /* [cout], updated in place; NULL: not tracked            */
    long long* num_batches_tracked; /* one int64, incremented; NULL: not tracked        */
    float eps;                /* BatchNorm2d.eps                                        */
    float momentum;           /* BatchNorm2d.momentum; < 0: cumulative moving average   */
                              /* (momentum=None: factor 1 / num_batches_tracked)        */
} yolo_amd_bn_train;

typedef struct {
    const void* src0;     /* NHWC bf16, (B, H >> up0, W >> up0, c0); for the   */
                          /* first layer: NCHW fp32 (B, c0, H, W)               */
    const void* src1;     /* NHWC bf16 (B, H, W, c1) or NULL                    */
    const void* weight;   /* packed bf16 (first layer: fp32 [k][cout])          */
    const float* bias;    /* [cout_pad] fp32                                    */
    const void* residual; /* NHWC bf16 (B, Ho, Wo, cout) or NULL                */
    void* dst;            /* NHWC (B, Ho, Wo, cout), bf16 or fp32               */
    int c0, c1, up0;      /* channels of src0/src1, log2 of src0's upsampling   */
    int B, H, W;          /* input spatial size seen by the convolution         */
    int cout, cout_pad;   /* real and padded output channels                    */
    int ksize, stride;    /* 1 or 3; 1 or 2 (padding = ksize / 2)               */
    int k_pad;            /* reduction length K = ksize*ksize*(c0 + c1)         */
    int leaky;            /* apply LeakyReLU(0.1)                               */
    int dst_fp32;         /* 1: dst is fp32 (detection convs), 0: bf16          */
    int first_layer;      /* 1: src0 is the NCHW fp32 network input (dst_fp32 is  */
                          /* allowed there only without fuse_next)              */
    /* Detection convolutions can run yolo_v3.predict in their epilogue
     * (/root/reference/yolo_v3.py:205-262) instead of leaving it to
     * yolo_amd_decode: with decode != NULL (host pointer; needs dst_fp32,
     * cout == decode_anchors * D, H == W == decode->G, no residual), dst is the
     * (B, decode_rows, D) prediction tensor and this layer writes its rows
     * decode->row_offset .. + G*G*decode_anchors, already decoded;
     * decode->raw and decode->layout are not used. */
    const yolo_amd_head_desc* decode;
    int decode_rows;      /* N_total of the prediction tensor                   */
    int decode_anchors;   /* A                                                  */
    int fuse_next;        /* run this layer together with the next descriptor as */
                          /* one launch; this layer's dst is then not written.   */
                          /* Two pairs exist: the first layer + a 3x3 stride-2   */
                          /* 32 -> 64 layer, and a 1x1 64 -> 32 layer + a 3x3    */
                          /* stride-1 32 -> 64 layer whose residual operand is   */
                          /* the 1x1 layer's src0 (the network's first residual  */
                          /* block: conv, conv, shortcut from=-3,                */
                          /* /root/reference/yolo_v3.py:452-473).  On the first  */
                          /* layer the value 2 takes the descriptor after the    */
                          /* pair as well, a 1x1 stride-1 64 -> 32 layer on the  */
                          /* second layer's dst: three layers in one launch, the */
                          /* second and third layers' dst both written.  On a    */
                          /* 3x3 layer with 128 output channels, 1 asks for the  */
                          /* next descriptor (a 1x1 stride-1 128 -> 64 layer on  */
                          /* this layer's dst) to be computed in this layer's    */
                          /* epilogue: both dst are written; one launch where    */
                          /* the kernel picked for the 3x3 layer can do that,    */
                          /* two otherwise, the same bits either way.            */
    /* Optional device scratch for a split reduction: at small batch a layer
     * with few output tiles and a long K is cut into K slices whose fp32
     * partial sums pass through this buffer (>= yolo_amd_conv_splitk_bytes,
     * 16-byte aligned; layers of one stream may share it).  NULL: never split.
     * The slices are added in a fixed order, so results are deterministic, but
     * they differ in the last bits from the unsplit layer. */
    void* splitk_ws;
    size_t splitk_ws_bytes;
    /* Host pointer or NULL.  Not NULL: `weight` holds the convolution's own weights
     * (no BatchNorm folded), `bias` its bias (zeros without one), and the layer is
     * normalised on batch statistics as described at yolo_amd_bn_train; needs a
     * bf16 dst, cout a multiple of 8, more than one output pixel per channel, no
     * decode and no fuse_next. */
    const yolo_amd_bn_train* bn_train;
} yolo_amd_conv_desc;

/* Scratch bytes the split reduction of this layer wants (0: it would not be
 * split).  Depends on the shape fields only. */
size_t yolo_amd_conv_splitk_bytes(const yolo_amd_conv_desc* desc);

/* Output channels are padded to a whole channel tile: returns the padded
 * count a weight/bias packer must allocate for `cout` real channels. */
int yolo_amd_conv_cout_pad(int cout);

/* The YOLO_AMD_* environment knobs (kernel variants for A/B timing,
 * YOLO_AMD_SPLITK=0 to keep every reduction in one pass) are read once and
 * cached; call this after changing the environment of a running process. */
void yolo_amd_reload_knobs(void);

/* Which kernel yolo_amd_conv_forward runs for a layer, and with what tiling.
 * No counterpart in the reference (torch picks its own kernels); it exists so
 * that a test can assert which kernel a shape exercises, and for tuning tools.
 * Only the shape fields of the descriptor are read, plus splitk_ws /
 * splitk_ws_bytes (a layer is split only when it is lent enough scratch). */
#define YOLO_AMD_KERNEL_FIRST 1 /* first layer alone (NCHW fp32 in)                   */
#define YOLO_AMD_KERNEL_STEM  2 /* first layer + the stride-2 layer behind it          */
#define YOLO_AMD_KERNEL_THIN  3 /* 3x3 stride 1, 32 -> 64 channels, weights in registers */
#define YOLO_AMD_KERNEL_PATCH 4 /* 3x3 stride 1 with the input patch reused over the taps */
#define YOLO_AMD_KERNEL_IGEMM 5 /* im2col implicit GEMM (every other layer)            */
#define YOLO_AMD_KERNEL_BLOCK 6 /* 1x1 64 -> 32 + 3x3 32 -> 64 + shortcut in one launch */
typedef struct {
    int kernel;          /* YOLO_AMD_KERNEL_*                                   */
    int bm, bn, bk;      /* output pixels x output channels per workgroup; K step */
    int act_ksteps;      /* IGEMM: depth of the activation tile in K steps (1, 2) */
    int lds_stages;      /* IGEMM: depth of the LDS ring                         */
    int waves;           /* waves per workgroup                                  */
    int taps_per_stage;  /* PATCH: taps per shared weight stage (1 or 3); 9: every */
                         /* wave stages its own channels' weights, one barrier per */
                         /* nine taps                                              */
    int k_slices;        /* > 1: split reduction + finishing launch              */
    int patch_flat;      /* PATCH with private rings: 1 = flat patch planes (fragment */
                         /* addresses are constants), 0 = rotated 16-slot rows      */
} yolo_amd_conv_choice;
int yolo_amd_conv_query(const yolo_amd_conv_desc* desc, yolo_amd_conv_choice* out);

/* Run `n` fused layers back to back on `stream`.  descs: host array.
 * zero_page: >= 256 bytes of device memory holding zeros (padding source). */
int yolo_amd_conv_forward(const yolo_amd_conv_desc* descs, int n, const void* zero_page,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_AMD_H */
