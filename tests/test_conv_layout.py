"""Host-side check of the LDS operand image and the exact division used by the
kernels (compiled with g++, no GPU)."""
import os
import subprocess
import tempfile

from helpers import ROOT


def test_lds_image_is_bijective_and_conflict_free_fastdiv_is_exact_and_tile_walks_cover():
    src = os.path.join(ROOT, "tests", "src", "conv_layout_check.cpp")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "check")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe]).decode()
    assert "layout_bad=0" in out and "fastdiv_bad=0" in out and "tilewalk_bad=0" in out
