"""Shims that let the READ-ONLY reference at /root/reference import and run on CPU in the
build container.  Used ONLY by tests/golden/make_golden.py (fixture generation); nothing in
the product, the GPU tests, smoke() or bench.py imports this file or the reference.

Shims: stub modules for omegaconf / pycocotools / skimage, identity
`.cuda()`, `'cuda*' -> 'cpu'` in Tensor.to / Module.to, no-op torch.cuda.empty_cache.
"""
import sys
import types

import torch
import torch.nn as nn

REF = "/root/reference"


def install():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import matplotlib
    matplotlib.use("Agg")

    om = types.ModuleType("omegaconf")
    lc = types.ModuleType("omegaconf.listconfig")

    class ListConfig(list):
        pass

    lc.ListConfig = ListConfig
    om.listconfig = lc
    om.OmegaConf = object
    sys.modules.setdefault("omegaconf", om)
    sys.modules.setdefault("omegaconf.listconfig", lc)
    for name in ("pycocotools", "pycocotools.coco", "skimage", "skimage.transform", "skimage.io"):
        m = types.ModuleType(name)
        m.COCO = object
        m.resize = None
        sys.modules.setdefault(name, m)

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    torch.cuda.set_device = lambda *a, **k: None

    def _fix(a):
        if isinstance(a, str) and a.startswith("cuda"):
            return "cpu"
        if isinstance(a, torch.device) and a.type == "cuda":
            return torch.device("cpu")
        return a

    _orig_to = torch.Tensor.to

    def _to(self, *a, **k):
        a = tuple(_fix(x) for x in a)
        k = {kk: _fix(v) for kk, v in k.items()}
        return _orig_to(self, *a, **k)

    torch.Tensor.to = _to
    _orig_mto = nn.Module.to

    def _mto(self, *a, **k):
        a = tuple(_fix(x) for x in a)
        k = {kk: _fix(v) for kk, v in k.items()}
        return _orig_mto(self, *a, **k)

    nn.Module.to = _mto
