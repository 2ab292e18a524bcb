"""The C-ABI library loads on a machine without a GPU, exports every symbol
include/yolo_amd.h declares, and its structs have the layout the ctypes
mirror assumes.  No compute calls here."""
import ctypes
import os
import re
import subprocess
import tempfile

from helpers import ROOT

HEADER = os.path.join(ROOT, "include", "yolo_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yolo_amd_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from yolo_amd import _lib
    names = declared_functions()
    assert len(names) >= 9
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in names:
        assert hasattr(raw, name), f"{name} is declared in yolo_amd.h but not exported"
        assert name in _lib.SYMBOLS, f"{name} has no ctypes prototype in yolo_amd/_lib.py"
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.lib.yolo_amd_abi_version() == 8


def test_struct_layouts_match_the_header():
    from yolo_amd import _lib
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "yolo_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(yolo_amd_head_desc), offsetof(yolo_amd_head_desc, stride),
         offsetof(yolo_amd_head_desc, anchor_h), sizeof(yolo_amd_conv_desc),
         offsetof(yolo_amd_conv_desc, c0), offsetof(yolo_amd_conv_desc, first_layer),
         offsetof(yolo_amd_conv_desc, decode), offsetof(yolo_amd_conv_desc, fuse_next),
         offsetof(yolo_amd_conv_desc, bn_train), sizeof(yolo_amd_bn_train),
         offsetof(yolo_amd_bn_train, num_batches_tracked), offsetof(yolo_amd_bn_train, momentum));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as tmp:
        c = os.path.join(tmp, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(tmp, "t")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        got = [int(v) for v in subprocess.check_output([exe]).split()]
    H, C, N = _lib.HeadDesc, _lib.ConvDesc, _lib.BnTrain
    want = [ctypes.sizeof(H), H.stride.offset, H.anchor_h.offset, ctypes.sizeof(C), C.c0.offset,
            C.first_layer.offset, C.decode.offset, C.fuse_next.offset, C.bn_train.offset, ctypes.sizeof(N),
            N.num_batches_tracked.offset, N.momentum.offset]
    assert _lib.BN_MAX_PARTIALS == int(re.search(r"#define YOLO_AMD_BN_MAX_PARTIALS\s+(\d+)", open(HEADER).read()).group(1))
    assert got == want


def test_conv_query_reports_the_kernel_without_a_gpu():
    """yolo_amd_conv_query reads shape fields only: the dispatch of the benchmark's
    layer shapes can be checked on the host."""
    from yolo_amd import _lib
    def desc(B, H, c0, cout, k, stride, first=0, fuse=0):
        d = _lib.ConvDesc()
        d.B, d.H, d.W, d.c0, d.cout, d.ksize, d.stride = B, H, H, c0, cout, k, stride
        d.cout_pad = cout if first else _lib.lib.yolo_amd_conv_cout_pad(cout)
        d.k_pad = k * k * c0
        d.first_layer, d.fuse_next = first, fuse
        return d
    q = _lib.conv_choice
    assert q(desc(32, 416, 3, 32, 3, 1, first=1, fuse=1))["kernel"] == "stem"
    assert q(desc(32, 416, 3, 32, 3, 1, first=1))["kernel"] == "first"
    assert q(desc(32, 208, 32, 64, 3, 1))["kernel"] == "thin"
    c = q(desc(32, 26, 256, 512, 3, 1))
    assert (c["kernel"], c["waves"], c["taps_per_stage"], c["patch_flat"]) == ("patch", 8, 9, 1)    # private weight rings, flat patch
    c = q(desc(32, 52, 128, 256, 3, 1))
    assert (c["kernel"], c["waves"], c["taps_per_stage"], c["patch_flat"]) == ("patch", 8, 9, 1)
    c = q(desc(32, 38, 256, 512, 3, 1))                                   # 608: 19 * 2^n maps take 19 x 10 tiles
    assert (c["kernel"], c["bm"], c["taps_per_stage"], c["patch_flat"]) == ("patch", 190, 9, 1)
    assert q(desc(32, 40, 256, 512, 3, 1))["kernel"] == "igemm"          # 320: no tile shape for 10 * 2^n maps
    c = q(desc(32, 52, 256, 128, 1, 1))
    assert (c["kernel"], c["bn"], c["bk"], c["act_ksteps"], c["lds_stages"]) == ("igemm", 128, 32, 2, 2)
    c = q(desc(4, 52, 256, 128, 1, 1))
    assert (c["kernel"], c["bn"]) == ("igemm", 64)                      # few tiles: narrower channel tile
    c = q(desc(1, 52, 256, 128, 1, 1))
    assert (c["kernel"], c["bn"]) == ("igemm", 32)                      # a single image: narrower still
    c = q(desc(32, 13, 1024, 512, 1, 1))
    assert (c["kernel"], c["bn"], c["bk"], c["act_ksteps"]) == ("igemm", 64, 64, 1)
    bad = desc(1, 13, 48, 64, 5, 1)
    ch = _lib.ConvChoice()
    import ctypes
    assert _lib.lib.yolo_amd_conv_query(ctypes.byref(bad), ctypes.byref(ch)) < 0


def test_host_side_argument_checks_need_no_gpu():
    from yolo_amd._lib import lib
    assert lib.yolo_amd_filter_results_workspace_bytes(0, 10, 80) == 0
    # per image: keys, boxes, scores, kept positions (32 B per box), eight candidate
    # lists of N keys, plus the suppression-matrix tiles
    # (N*N/128 + 2N + 64*min(N, C) words of 8 bytes)
    N = 10647
    nbytes = lib.yolo_amd_filter_results_workspace_bytes(32, N, 80)
    per_image = N * 32 + 8 * N * 8 + (N * N // 128 + 2 * N + 64 * 80) * 8
    assert 32 * per_image <= nbytes < 32 * per_image * 1.02 + (1 << 16)
    assert lib.yolo_amd_filter_results_workspace_bytes(1, 22743, 80) > 32768 * 8   # keys padded to 32768
    assert lib.yolo_amd_conv_cout_pad(255) == 256 and lib.yolo_amd_conv_cout_pad(32) == 32
    assert lib.yolo_amd_conv_cout_pad(64) == 64 and lib.yolo_amd_conv_cout_pad(9) == 32
    # a null pointer is rejected before anything touches the device
    rc = lib.yolo_amd_filter_results(None, 1, 10, 85, 80, 0.5, 0.4, None, 0, None, None, None)
    assert rc < 0 and b"null" in lib.yolo_amd_last_error()
    rc = lib.yolo_amd_box_iou(None, 3, None, 5, None, None)
    assert rc < 0 and b"box1" in lib.yolo_amd_last_error()
