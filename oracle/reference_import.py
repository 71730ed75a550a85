"""ORACLE — TEST INFRASTRUCTURE ONLY.  Imports the REAL reference (/root/reference) on CPU.

Only usable in the build container (the GPU box has no /root/reference): it pins the restatement in
oracle/{ops_ref,model_ref}.py and generates the golden fixtures under tests/golden/.  Recipe =
SURVEY §8(c): stub the three absent third-party modules (torchvision, webdataset, wandb), put a seeded
synthetic vgg.pth into the cwd (utils.py:24-37 would otherwise try wget), gloo world_size=1 for
GradNorm's scalar all-reduce (vae_trainer.py:42-44,59).
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import tempfile
import types

import torch
import torch.nn as nn

REFERENCE_DIR = "/root/reference"
_VGG_CFG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def available() -> bool:
    return os.path.exists(os.path.join(REFERENCE_DIR, "ae.py"))


def _vgg16_features():
    layers, cin = [], 3
    for v in _VGG_CFG_D:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


def _install_stubs():
    from transformers import get_cosine_schedule_with_warmup  # noqa: F401  must precede the torchvision stub
    if "torchvision" not in sys.modules or getattr(sys.modules["torchvision"], "_vq_stub", False) is False:
        tv = types.ModuleType("torchvision")
        tv._vq_stub = True
        models = types.ModuleType("torchvision.models")

        class _VGG(nn.Module):
            def __init__(self):
                super().__init__()
                self.features = _vgg16_features()

        models.vgg16 = lambda pretrained=False, **kw: _VGG()
        transforms = types.ModuleType("torchvision.transforms")
        for name in ("Compose", "ToTensor", "Normalize", "CenterCrop", "Resize", "RandomCrop", "GaussianBlur"):
            setattr(transforms, name, type(name, (), {"__init__": lambda self, *a, **k: None,
                                                       "__call__": lambda self, x: x}))
        tv.models, tv.transforms = models, transforms
        sys.modules.update({"torchvision": tv, "torchvision.models": models, "torchvision.transforms": transforms})
    if "webdataset" not in sys.modules:
        wds = types.ModuleType("webdataset")
        wds.split_by_node = wds.split_by_worker = None
        sys.modules["webdataset"] = wds
    if "wandb" not in sys.modules:
        sys.modules["wandb"] = types.ModuleType("wandb")


_cache = {}


def _ensure_world_of_one():
    """The reference's GradNorm all-reduces inside its backward (vae_trainer.py:59): a gloo world of one rank must be up whenever its
    modules run — also after another test (the train_ddp CLI test) brought a process group up and tore it down again."""
    import torch.distributed as dist
    if not dist.is_initialized():
        # a file rendezvous: no port to collide on when several test processes import the reference at once (pytest -n)
        import tempfile
        fd, path = tempfile.mkstemp(prefix="vq_ref_world_")
        os.close(fd)
        os.unlink(path)                                   # (the file store creates it)
        dist.init_process_group("gloo", init_method="file://" + path, rank=0, world_size=1)


def load():
    """-> (ae, utils, vae_trainer) modules of the reference, imported under private names."""
    _ensure_world_of_one()
    if "mods" in _cache:
        return _cache["mods"]
    assert available(), "reference not mounted"
    _install_stubs()
    saved = {k: sys.modules.get(k) for k in ("ae", "utils", "vae_trainer")}
    sys.path.insert(0, REFERENCE_DIR)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="vq_ref_")
    try:
        os.chdir(tmp)
        torch.save({f"lin{i}.model.1.weight": torch.rand(1, c, 1, 1, generator=torch.Generator().manual_seed(1 + i))
                    for i, c in enumerate((64, 128, 256, 512, 512))}, "vgg.pth")
        for k in ("ae", "utils", "vae_trainer"):
            sys.modules.pop(k, None)
        utils = importlib.import_module("utils")
        utils.os = os                       # utils.py:29 uses os without importing it
        ae = importlib.import_module("ae")
        vt = importlib.import_module("vae_trainer")
        _cache["tmp"] = tmp
    finally:
        os.chdir(cwd)
        sys.path.remove(REFERENCE_DIR)
        for k, v in saved.items():
            if v is not None:
                sys.modules[k] = v
            else:
                sys.modules.pop(k, None)
    _cache["mods"] = (ae, utils, vt)
    return _cache["mods"]


def load_tae():
    """-> the reference's tae.py module (3-D TVAE; imports only torch and einops), under a private name."""
    if "tae" in _cache:
        return _cache["tae"]
    assert available(), "reference not mounted"
    spec = importlib.util.spec_from_file_location("_vq_reference_tae", os.path.join(REFERENCE_DIR, "tae.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    _cache["tae"] = mod
    return mod


def in_ref_cwd(fn):
    """Run `fn` with cwd = the directory holding the synthetic vgg.pth (LPIPS() loads it from cwd)."""
    cwd = os.getcwd()
    os.chdir(_cache["tmp"])
    try:
        return fn()
    finally:
        os.chdir(cwd)
