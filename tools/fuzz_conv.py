"""Random-shape sweep of the conv entry points (forward, data gradient, weight + bias gradient) on the HOST EMULATOR build against the
fp32 oracle — test tooling (same harness as tests/test_kernels.py::_conv_case), for shapes the fixed case lists do not name: odd
image sizes, channel counts that are not multiples of 8 / 64, one-pixel images, every storage type.

    [FUZZ_DEVICE=cuda] python tools/fuzz_conv.py [n_cases] [seed]
"""
import os
import random
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.conftest as cf                                          # noqa: E402
import tests.test_kernels as tk                                      # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    import vqgan_training_amd as vqm
    if os.environ.get("FUZZ_DEVICE", "cpu") == "cuda":     # the product library on the GPU instead of the host emulator
        lib = vqm._lib.VqLibrary(cf.HIP_LIB)
        vqm._lib._set_library_for_tests(lib)
        be = cf.Backend("gpu", "cuda:0", lib)
    else:
        cf._build("emu", cf.EMU_LIB)
        lib = vqm._lib.VqLibrary(cf.EMU_LIB)
        vqm._lib._set_library_for_tests(lib)
        be = cf.Backend("emu", "cpu", lib)
    chans = [1, 3, 5, 8, 12, 24, 40, 64, 72, 128, 136, 192] + ([256, 320, 512] if os.environ.get("FUZZ_DEVICE") == "cuda" else [])
    bad = 0
    for i in range(n):
        prec = rnd.choice(["bf16", "fp16", "fp32", "fp32x3", "f16x3", "fp32x6"])
        form = rnd.choice(["same3", "same3", "up", "down", "patch", "one"])
        N = rnd.choice([1, 1, 2, 3])
        ci, co = rnd.choice(chans), rnd.choice(chans)
        H, W = (rnd.randint(1, 70), rnd.randint(1, 70)) if os.environ.get("FUZZ_DEVICE") == "cuda" else (rnd.randint(1, 19), rnd.randint(1, 19))
        relu = rnd.random() < 0.3
        if form == "same3":
            case = (prec, N, H, W, ci, co, 3, 1, 1, 1, relu, None)
        elif form == "up":
            H, W = min(H, 9), min(W, 9)
            case = (prec, N, H, W, ci, co, 3, 1, 1, 2, relu, None)
        elif form == "down":
            H, W = 2 * max(1, H // 2), 2 * max(1, W // 2)
            case = (prec, N, H, W, ci, co, 3, 2, 0, 1, relu, (H // 2, W // 2))
        elif form == "patch":
            r = rnd.choice([2, 4])
            H, W = r * max(1, H // r), r * max(1, W // r)
            case = (prec, N, H, W, ci, co, r, r, 0, 1, relu, None)
        else:
            case = (prec, N, H, W, ci, co, 1, 1, 0, 1, relu, None)
        vqm.ops.clear_caches()
        try:
            tk._conv_case(be, case, seed=1000 + i)
            print("ok  ", case, flush=True)
        except Exception as e:                       # noqa: BLE001
            bad += 1
            print("FAIL", case, repr(e)[:300], flush=True)
            if not isinstance(e, AssertionError):
                traceback.print_exc(limit=3)
    print(f"{n - bad} / {n} ok")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
