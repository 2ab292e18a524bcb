"""Sustained bf16 MFMA rate of the chip (register-only loop of v_mfma_f32_16x16x32_bf16):
what the matrix pipes deliver at the clock the part holds under that load, as a
reference point beside the nominal 2.5 PFLOP/s.
    make -C tools/probe && python tools/mfma_peak.py"""
import ctypes
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "probe", "libmfma_probe.so"))
lib.mfma_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]


def main():
    out = torch.zeros(1 << 22, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for acc in (12,):
        for blocks_per_cu in (1, 2, 4):
            blocks, iters = 256 * blocks_per_cu, 20000
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert lib.mfma_probe_launch(out.data_ptr(), blocks, iters, acc, st) == 0
                e1.record()
                torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            flop = blocks * 4 * iters * acc * 2.0 * 16 * 16 * 32
            print(f"{acc:2d} accumulators, {blocks_per_cu} workgroups (x4 waves) per CU: {ms:7.2f} ms, "
                  f"{flop / ms / 1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
