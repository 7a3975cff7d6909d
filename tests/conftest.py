import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _poison_uninitialised_memory():
    """OCCF_TEST_POISON=1: every ``torch.empty`` / ``empty_like`` / ``new_empty`` / ``empty_strided`` buffer is filled with NaN
    (floating point) or 0x5A bytes (integers) before it is handed out, so that a kernel or a host routine that READS a
    buffer it was supposed to overwrite first shows up as a wrong / NaN result instead of depending on what the
    allocator happened to return (on the GPU: on which tests ran before).  Debug mode of the CPU (emulation) suite."""
    def poison(t):
        if t.numel() == 0 or t.is_meta:
            return t
        with torch.no_grad():
            if t.is_floating_point() or t.is_complex():
                t.fill_(float("nan"))
            elif t.dtype == torch.bool:
                t.fill_(True)
            else:
                t.fill_(0x5A5A5A5A if t.dtype in (torch.int32, torch.int64) else 0x5A)
        return t

    def wrap(fn):
        def inner(*a, **k):
            return poison(fn(*a, **k))
        inner.__name__ = getattr(fn, "__name__", "empty")
        return inner
    for name in ("empty", "empty_like", "empty_strided"):
        setattr(torch, name, wrap(getattr(torch, name)))
    torch.Tensor.new_empty = wrap(torch.Tensor.new_empty)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    if os.environ.get("OCCF_TEST_POISON", "0") == "1":
        _poison_uninitialised_memory()


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (-m "not gpu") runs every kernel through the host emulation: ~35 min of single-core work, 8-10 min on
    four to six pytest-xdist workers (the two-rank DDP test alone takes 7.7 min).  So a plain ``pytest tests -m "not gpu"`` distributes itself over 4 workers when xdist is
    installed (OCCF_TEST_SERIAL=1 or an explicit -n / -p no:xdist keeps it serial).  GPU runs are never distributed: one
    process owns the device and the loaded in-tree library stays visible in that process."""
    opt = config.option
    if hasattr(config, "workerinput") or os.environ.get("PYTEST_XDIST_WORKER"):
        return None                     # inside an xdist worker: never nest
    if os.environ.get("OCCF_TEST_SERIAL", "0") == "1" or "not gpu" not in (getattr(opt, "markexpr", "") or ""):
        return None
    if getattr(opt, "numprocesses", None) or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(opt, "collectonly", False) or getattr(opt, "usepdb", False):
        return None
    opt.numprocesses = max(1, min(6, (os.cpu_count() or 1) - 2))
    return None


@pytest.fixture(autouse=True)
def _dcnv2_cpu_statement():
    """CPU tensors through ``ModulatedDeformConv2dPack`` (the R101-DCN image backbone built on the host in the wiring
    and GPU-vs-CPU tests) take the oracle's grid_sample statement; the product raises without it"""
    from occformer_amd.detector import ModulatedDeformConv2dPack
    from oracle import dcn_ref
    prev = ModulatedDeformConv2dPack.cpu_reference
    ModulatedDeformConv2dPack.cpu_reference = staticmethod(dcn_ref.module_reference)
    yield
    ModulatedDeformConv2dPack.cpu_reference = prev


def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


class Backend:
    """Either the real gfx950 library on cuda:0 ('hip', -m gpu) or the TEST-ONLY host
    emulation build of the same kernel sources ('emu', CPU)."""

    def __init__(self, kind):
        self.kind = kind
        if kind == "hip":
            from occformer_amd.ops import get_ops
            self.ops = get_ops()
            self.device = torch.device("cuda:0")
        else:
            from occformer_amd import _lib
            from occformer_amd.ops import HipOps
            from tests.hipemu import build as emu_build
            self.ops = HipOps(_lib.bind(emu_build.build()), strict=False)
            self.device = torch.device("cpu")

    def to(self, *ts):
        out = tuple(t.to(self.device) if torch.is_tensor(t) else t for t in ts)
        return out if len(out) > 1 else out[0]


_backends = {}


def _get_backend(kind):
    if kind not in _backends:
        _backends[kind] = Backend(kind)
    return _backends[kind]


@pytest.fixture(params=["emu", pytest.param("hip", marks=pytest.mark.gpu)])
def be(request):
    if request.param == "hip" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _get_backend(request.param)


@pytest.fixture
def hip():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _get_backend("hip")
