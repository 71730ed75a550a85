"""The drop-in boundary itself (CPU only, no compute): the product library libvqhip.so loads, and exports — and the ctypes
binding declares — exactly the entry points include/vqhip.h declares."""
import ctypes
import os
import re

import pytest

import vqgan_training_amd as vq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, "include", "vqhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                      # comments
    text = re.sub(r"typedef\s+struct\s+\w*\s*\{.*?\}\s*\w+\s*;", "", text, flags=re.S)   # struct bodies
    names = re.findall(r"\b(vq_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_header_binding_and_library_agree():
    declared = _header_functions()
    assert len(declared) >= 40, declared
    bound = sorted(vq._lib.EXPORTED_SYMBOLS)
    assert declared == bound, (sorted(set(declared) - set(bound)), sorted(set(bound) - set(declared)))
    path = vq._lib._LIB_PATH
    if not os.path.exists(path):
        pytest.fail(f"{path} is not built: run `make` (the product never falls back to anything else)")
    dll = ctypes.CDLL(path)                                                 # loads without a GPU (no HIP call is made)
    missing = [n for n in declared if not hasattr(dll, n)]
    assert not missing, missing
    dll.vq_abi_version.restype = ctypes.c_int
    assert dll.vq_abi_version() == vq._lib.ABI_VERSION


def test_exported_symbols_are_exactly_the_header():
    """The boundary is THIN: built with -fvisibility=hidden and a linker version script (csrc/libvqhip.map), the product library's
    dynamic symbol table holds the entry points include/vqhip.h declares and nothing else — no mangled C++ helpers, no kernel
    handles or device stubs (round 3 exported 80 symbols for a 49-symbol header)."""
    import subprocess
    path = vq._lib._LIB_PATH
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    out = subprocess.run([nm, "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _header_functions(), (sorted(set(exported) - set(_header_functions())), sorted(set(_header_functions()) - set(exported)))


def test_missing_library_fails_loudly(tmp_path):
    """No fallback path: a missing or incomplete shared object raises at load time."""
    with pytest.raises((OSError, RuntimeError)):
        vq._lib.VqLibrary(str(tmp_path / "libvqhip.so"))


def test_product_never_touches_the_oracle_or_the_emulator():
    """oracle/ and tests/emu/ are test infrastructure: nothing under the product package may import or load them."""
    pkg = os.path.join(ROOT, "vqgan-training_amd")
    offenders = []
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "libvqhip_emu" in src or "hip_emu.h" in src:
                    offenders.append(os.path.join(dirpath, f))
    assert not offenders, offenders


def test_error_contract_without_a_gpu():
    """Argument validation happens before any HIP call, so the product library can be asked on a GPU-less machine: invalid
    requests return a negative VqStatus and leave a message in the calling thread's vq_last_error() — they are never silently
    accepted (include/vqhip.h)."""
    import threading
    L = vq._lib.VqLibrary(vq._lib._LIB_PATH)
    C = ctypes
    buf = (C.c_float * 64)()
    p = C.cast(buf, C.c_void_p)
    d = vq._lib.VqConvDesc()
    (d.N, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout, d.Cin_w, d.Cout_w, d.R, d.S, d.stride, d.dil_in, d.up, d.pad_t, d.pad_l, d.dtype,
     d.split, d.relu, d.subpix) = (1, 4, 4, 8, 4, 4, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1, 0, 1, 0, 0)
    cases = []
    cases.append(("null pointers", L.dll.vq_conv2d_fwd(C.byref(d), None, None, None, None, None, None, None, 0, None)))
    d.Cin = 12                                              # not a multiple of 8
    cases.append(("channel padding", L.dll.vq_conv2d_fwd(C.byref(d), p, p, None, None, None, p, None, 0, None)))
    d.Cin, d.N = 8, 0                                       # empty batch
    cases.append(("empty tensor", L.dll.vq_conv2d_fwd(C.byref(d), p, p, None, None, None, p, None, 0, None)))
    d.N, d.up = 1, 3                                        # only nearest-2x is folded into the gather
    cases.append(("up = 3", L.dll.vq_conv2d_fwd(C.byref(d), p, p, None, None, None, p, None, 0, None)))
    d.up, d.subpix, d.Cout, d.Cout_w = 1, 2, 64, 64         # 16 rows per phase block: below the 32-row tile
    cases.append(("sub-pixel rows", L.dll.vq_conv2d_fwd(C.byref(d), p, p, None, None, None, p, None, 0, None)))
    d.subpix, d.Cout, d.Cout_w = 2, 128, 128
    cases.append(("sub-pixel wgrad", L.dll.vq_conv2d_wgrad(C.byref(d), p, p, p, None, 0, p, 1 << 30, None)))
    cases.append(("weight transform mode", L.dll.vq_subpixel_weights(p, p, 4, 4, 7, None)))
    cases.append(("GroupNorm channels", L.dll.vq_gn_stats(p, 1, 16, 12, 4, 1e-6, 0, p, p, p, 1 << 20, None)))
    cases.append(("GroupNorm workspace", L.dll.vq_gn_stats(p, 1, 16, 64, 32, 1e-6, 0, p, p, p, 8, None)))
    cases.append(("attention head width", L.dll.vq_attention_fwd(p, p, p, 1, 4, 48, 24, 0, None)))
    cases.append(("pack layout", L.dll.vq_pack_weight_fwd(p, 8, 8, 3, 3, 8, 8, 1, 9, 0, None, p, None)))
    cases.append(("binary16 pack without a scale slot", L.dll.vq_pack_weight_fwd(p, 8, 8, 3, 3, 8, 8, 1, 0, 2, None, p, None)))
    cases.append(("binary16 pack with the 3-term split", L.dll.vq_pack_weight_fwd(p, 8, 8, 3, 3, 8, 8, 3, 0, 2, p, p, None)))
    for what, rc in cases:
        assert rc < 0, what
        assert L.last_error(), what
    # thread-local: a fresh thread has no error pending, and its own error does not overwrite ours
    mine = L.last_error()
    seen = {}

    def other():
        seen["before"] = L.last_error()
        seen["rc"] = L.dll.vq_subpixel_weights(None, None, 1, 1, 0, None)
        seen["after"] = L.last_error()
    t = threading.Thread(target=other)
    t.start(); t.join()
    assert seen["before"] == "" and seen["rc"] < 0 and "vq_subpixel_weights" in seen["after"]
    assert L.last_error() == mine
