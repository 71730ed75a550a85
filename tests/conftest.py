import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

EMU_LIB = os.path.join(ROOT, "tests", "emu", "libvqhip_emu.so")
HIP_LIB = os.path.join(ROOT, "vqgan-training_amd", "libvqhip.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(session, config, items):
    """Row-level evidence first (tests/test_rows.py::ROWS: one canonical oracle-parity test per SURVEY §8 row, in row order), kernel-
    variant / forced-tile / fuzz cases last; everything else keeps its collection order.  Runs before the `-m` deselection, so the
    recorded list (checked by test_rows.py) holds both the emulator and the GPU instances."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_rows import row_rank
    config._vq_all_items = [(it.nodeid, it.get_closest_marker("gpu") is not None) for it in items]
    keyed = sorted(enumerate(items), key=lambda t: (row_rank(t[1].nodeid), t[0]))
    items[:] = [it for _, it in keyed]
    config._vq_order = [it.nodeid for it in items]


def _build(target, product):
    if os.path.exists(product):
        srcs = [os.path.join(ROOT, "vqgan-training_amd", "csrc"), os.path.join(ROOT, "tests", "emu"),
                os.path.join(ROOT, "include")]
        newest = max(os.path.getmtime(os.path.join(d, f)) for d in srcs for f in os.listdir(d)
                     if f.endswith((".hip", ".h", ".cpp")))
        if os.path.getmtime(product) >= newest:
            return
    subprocess.run(["make", "-C", ROOT, target, "-j8"], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def emu_library():
    """Host-emulated build of the kernel sources (tests/emu) — test infrastructure only."""
    _build("emu", EMU_LIB)
    import vqgan_training_amd as vq
    return vq._lib.VqLibrary(EMU_LIB)


@pytest.fixture(scope="session")
def hip_library():
    """The product library; must already be built (it travels to the GPU box in-tree)."""
    import vqgan_training_amd as vq
    return vq._lib.VqLibrary(HIP_LIB)


class Backend:
    def __init__(self, name, device, library):
        self.name, self.device, self.library = name, torch.device(device), library


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def backend(request):
    """Runs a kernel test twice: on host cores through the emulator (not gpu) and on cuda:0 (gpu)."""
    import vqgan_training_amd as vq
    if request.param == "emu":
        library = request.getfixturevalue("emu_library")
        be = Backend("emu", "cpu", library)
    else:
        assert torch.cuda.is_available(), "gpu test selected without a GPU"
        library = request.getfixturevalue("hip_library")
        be = Backend("gpu", "cuda:0", library)
    vq._lib._set_library_for_tests(library)
    vq.ops.clear_caches()
    yield be
    vq._lib._set_library_for_tests(None)
    vq.ops.clear_caches()
