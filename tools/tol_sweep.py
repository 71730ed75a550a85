#!/usr/bin/env python
"""Where tests/test_kernels.py's TOL comes from: the conv parity cases over many seeds per precision, worst
error per output with the metric the tests use (max|a-b| / max(max|ref|, natural scale of a reduction output)).

    python tools/tol_sweep.py --seeds 200 > profiles/r4_tol_sweep_emu.txt          # host emulator (no GPU)
    python tools/tol_sweep.py --seeds 200 --gpu > gpurun_out/r4_tol_sweep_gpu.txt  # MI355X

TOL[prec] in the tests = 2 x the worst value seen here (rounded up to one digit).  The cases are the cancellation-prone ones (1-3
output channels: the bias gradient is 1-3 sums), a ragged generic-kernel case, an LDS-DMA case and a three-tap weight-gradient case.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import vqgan_training_amd as vq   # noqa: E402
import conftest                   # noqa: E402
import test_kernels as T          # noqa: E402

SHAPES = [  # N, H, W, Cin, Cout, R, stride, pad, up, relu, out_hw
    (1, 4, 64, 128, 3, 3, 1, 1, 1, False, None),      # decoder.conv_out: 3-element bias gradient (the round-3 failure)
    (1, 4, 4, 24, 1, 2, 2, 0, 1, False, None),        # 1 output channel
    (2, 6, 6, 128, 72, 3, 1, 1, 1, True, None),       # LDS-DMA kernel, ragged cout tile, ReLU
    (1, 8, 8, 128, 128, 3, 1, 1, 1, False, None),     # three-tap weight gradient
    (2, 6, 10, 24, 136, 3, 1, 1, 1, False, None),     # generic kernel, ragged M
    (1, 4, 4, 64, 64, 3, 1, 1, 2, False, None),       # sub-pixel Upsample
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--precs", default="fp32x6,fp32x3,fp32,bf16,fp16")
    a = ap.parse_args()
    if a.gpu:
        lib = vq._lib.VqLibrary(conftest.HIP_LIB)
        be = conftest.Backend("gpu", "cuda:0", lib)
    else:
        conftest._build("emu", conftest.EMU_LIB)
        lib = vq._lib.VqLibrary(conftest.EMU_LIB)
        be = conftest.Backend("emu", "cpu", lib)
    vq._lib._set_library_for_tests(lib)
    print(f"# backend={be.name} seeds={a.seeds}; worst rel. error over the seeds, per output; TOL in tests/test_kernels.py")
    for prec in a.precs.split(","):
        worst_prec = 0.0
        for shape in SHAPES:
            if prec in ("fp32x6", "fp32x3", "fp32") and shape[3] % 64 == 0 and shape[3] >= 128 and be.name == "emu" and a.seeds > 50:
                seeds = 50        # the 3-term split is 3x the emulated MFMAs
            else:
                seeds = a.seeds
            worst = {}
            for s in range(seeds):
                vq.ops.clear_caches()
                rep = {}
                T._conv_case(be, (prec,) + shape, seed=1000 + s, report=rep)
                for k, v in rep.items():
                    worst[k] = max(worst.get(k, 0.0), v)
            worst_prec = max(worst_prec, *worst.values())
            print(f"{prec:7s} {'-'.join(map(str, shape)):45s} seeds={seeds:4d} " + " ".join(f"{k}={v:.3e}" for k, v in worst.items()), flush=True)
        print(f"{prec:7s} WORST {worst_prec:.3e}  TOL {T.TOL[prec]:.1e}  margin x{T.TOL[prec] / worst_prec:.2f}", flush=True)


if __name__ == "__main__":
    main()
