import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
import conftest
import test_kernels as T
import vqgan_training_amd as vq
lib = vq._lib.lib()
be = conftest.Backend("gpu", "cuda:0", lib)
for knob in (0, 8197 << 4):
    bad = 0
    for case in T._random_conv_cases(36, seed=20260925):
        vq.ops.clear_caches()
        lib.dll.vq_debug_set_conv_tile(knob)
        try:
            T._conv_case(be, case)
        except AssertionError as e:
            bad += 1; print("FAIL knob", knob, case, str(e)[:160].replace("\n", " "))
        finally:
            lib.dll.vq_debug_set_conv_tile(0)
    print("knob", knob, "failures:", bad)
