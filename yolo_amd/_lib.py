"""ctypes binding of libyolo_amd.so (the C ABI declared in include/yolo_amd.h).

The library is the product: there is no Python or CPU fallback.  If the shared
object is missing or a symbol cannot be resolved, importing this module fails
with instructions to build it (``python -c "import __graft_entry__ as g; g.build()"``
or ``make -C yolo_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YOLO_AMD_LIB: another build of the same library, for A/B timing of kernel changes
LIB_PATH = os.environ.get("YOLO_AMD_LIB") or os.path.join(_HERE, "libyolo_amd.so")

MAX_HEADS = 4
MAX_ANCHORS = 8
LAYOUT_NCHW = 0
LAYOUT_NHWC = 1


class HeadDesc(ctypes.Structure):
    _fields_ = [
        ("raw", ctypes.c_void_p),
        ("G", ctypes.c_int),
        ("layout", ctypes.c_int),
        ("row_offset", ctypes.c_int),
        ("stride", ctypes.c_float),
        ("anchor_w", ctypes.c_float * MAX_ANCHORS),
        ("anchor_h", ctypes.c_float * MAX_ANCHORS),
    ]


class ImageDesc(ctypes.Structure):
    """yolo_amd_image of include/yolo_amd.h."""
    _fields_ = [("data", ctypes.c_void_p), ("width", ctypes.c_int32), ("height", ctypes.c_int32),
                ("pitch_bytes", ctypes.c_int64)]


class BnTrain(ctypes.Structure):
    """yolo_amd_bn_train: BatchNorm2d on batch statistics behind a convolution."""
    _fields_ = [
        ("raw", ctypes.c_void_p),
        ("partial", ctypes.c_void_p),
        ("scale_shift", ctypes.c_void_p),
        ("gamma", ctypes.c_void_p),
        ("beta", ctypes.c_void_p),
        ("running_mean", ctypes.c_void_p),
        ("running_var", ctypes.c_void_p),
        ("num_batches_tracked", ctypes.c_void_p),
        ("eps", ctypes.c_float),
        ("momentum", ctypes.c_float),
    ]


BN_MAX_PARTIALS = 256


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("src0", ctypes.c_void_p),
        ("src1", ctypes.c_void_p),
        ("weight", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("residual", ctypes.c_void_p),
        ("dst", ctypes.c_void_p),
        ("c0", ctypes.c_int), ("c1", ctypes.c_int), ("up0", ctypes.c_int),
        ("B", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int),
        ("cout", ctypes.c_int), ("cout_pad", ctypes.c_int),
        ("ksize", ctypes.c_int), ("stride", ctypes.c_int),
        ("k_pad", ctypes.c_int),
        ("leaky", ctypes.c_int),
        ("dst_fp32", ctypes.c_int),
        ("first_layer", ctypes.c_int),
        ("decode", ctypes.POINTER(HeadDesc)),
        ("decode_rows", ctypes.c_int),
        ("decode_anchors", ctypes.c_int),
        ("fuse_next", ctypes.c_int),
        ("splitk_ws", ctypes.c_void_p),
        ("splitk_ws_bytes", ctypes.c_size_t),
        ("bn_train", ctypes.POINTER(BnTrain)),
    ]


class ConvChoice(ctypes.Structure):
    """yolo_amd_conv_choice of include/yolo_amd.h."""
    _fields_ = [("kernel", ctypes.c_int), ("bm", ctypes.c_int), ("bn", ctypes.c_int), ("bk", ctypes.c_int),
                ("act_ksteps", ctypes.c_int), ("lds_stages", ctypes.c_int), ("waves", ctypes.c_int),
                ("taps_per_stage", ctypes.c_int), ("k_slices", ctypes.c_int), ("patch_flat", ctypes.c_int)]


KERNEL_FIRST, KERNEL_STEM, KERNEL_THIN, KERNEL_PATCH, KERNEL_IGEMM = 1, 2, 3, 4, 5
KERNEL_NAMES = {1: "first", 2: "stem", 3: "thin", 4: "patch", 5: "igemm", 6: "block"}


# name -> (restype, argtypes); every symbol include/yolo_amd.h declares
SYMBOLS = {
    "yolo_amd_last_error": (ctypes.c_char_p, []),
    "yolo_amd_abi_version": (ctypes.c_int, []),
    "yolo_amd_box_iou": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
    "yolo_amd_decode": (ctypes.c_int, [ctypes.POINTER(HeadDesc), ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_void_p]),
    "yolo_amd_filter_results_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int,
                                                                  ctypes.c_int]),
    "yolo_amd_filter_results": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "yolo_amd_filter_results_static": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                                       ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                                       ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                                       ctypes.c_void_p]),
    "yolo_amd_filter_img_class_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "yolo_amd_filter_img_class": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t,
                                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "yolo_amd_spin": (ctypes.c_int, [ctypes.c_int, ctypes.c_void_p]),
    "yolo_amd_rescale_boxes": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_float, ctypes.c_void_p]),
    "yolo_amd_preprocess_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "yolo_amd_preprocess": (ctypes.c_int, [ctypes.POINTER(ImageDesc), ctypes.c_int, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_size_t, ctypes.c_void_p]),
    "yolo_amd_reload_knobs": (None, []),
    "yolo_amd_conv_splitk_bytes": (ctypes.c_size_t, [ctypes.POINTER(ConvDesc)]),
    "yolo_amd_conv_cout_pad": (ctypes.c_int, [ctypes.c_int]),
    "yolo_amd_conv_query": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.POINTER(ConvChoice)]),
    "yolo_amd_conv_forward": (ctypes.c_int, [ctypes.POINTER(ConvDesc), ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p]),
}


class YoloAmdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library is the only implementation of this path. "
            "Build it with `make -C yolo_amd/csrc` (hipcc, --offload-arch=gfx950).")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as exc:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    want = 8
    got = lib.yolo_amd_abi_version()
    if got != want:
        raise ImportError(f"{LIB_PATH} has ABI version {got}, the Python side expects {want}")
    return lib


lib = _load()


def conv_choice(desc):
    """The kernel the library would run for one ConvDesc, as a dict."""
    c = ConvChoice()
    check(lib.yolo_amd_conv_query(ctypes.byref(desc), ctypes.byref(c)), "conv_query")
    return {"kernel": KERNEL_NAMES[c.kernel], "bm": c.bm, "bn": c.bn, "bk": c.bk, "act_ksteps": c.act_ksteps,
            "lds_stages": c.lds_stages, "waves": c.waves, "taps_per_stage": c.taps_per_stage,
            "k_slices": c.k_slices, "patch_flat": c.patch_flat}


def check(rc, what):
    if rc != 0:
        msg = lib.yolo_amd_last_error()
        raise YoloAmdError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


def current_stream_ptr(device=None):
    """hipStream_t of torch's current stream on ``device`` (as an int)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise YoloAmdError("no ROCm device is visible: this path runs on the MI355X HIP kernels "
                           "only and has no CPU fallback")
