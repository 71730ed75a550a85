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
