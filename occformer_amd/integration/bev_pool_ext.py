"""Drop-in for the reference's compiled module ``mmdet3d.ops.bev_pool.bev_pool_ext`` (pybind11,
mmdetection3d/mmdet3d/ops/bev_pool/src/bev_pool.cpp:89-94): copy this file to ``mmdet3d/ops/bev_pool/bev_pool_ext.py``
and the reference's ``bev_pool.py`` (``bev_pool()`` -> ``QuickCumsumCuda.apply`` -> ``.backward()``, :37-97) runs
unchanged on liboccformer_hip.so.  A ctypes binding of two C-ABI entry points (include/occformer_hip.h:24-38); argument
order as ``bev_pool.cpp:22-28 / 58-64`` (interval LENGTHS before STARTS).  No torch extension, no pybind.

The library is ``$OCCFORMER_HIP_LIB`` or the in-tree ``occformer_amd/liboccformer_hip.so``; ``import torch`` precedes the
dlopen so both share torch's HIP runtime.  There is no CPU implementation: host tensors raise (the test suite binds the
x86 emulation build of the same kernel sources through ``use_library`` to run the reference's wrapper without a GPU).
"""
import ctypes
import os

import torch

_vp, _i = ctypes.c_void_p, ctypes.c_int
_state = {"lib": None, "host_ok": False}


def use_library(lib, host_tensors=False):
    """bind an already opened library (``ctypes.CDLL``); ``host_tensors``: it is the host emulation build"""
    for f in (lib.occf_bev_pool_fwd, lib.occf_bev_pool_bwd):
        f.argtypes = [_vp] * 5 + [_i] * 7 + [_vp]
        f.restype = _i
    _state.update(lib=lib, host_ok=bool(host_tensors))


def _lib():
    if _state["lib"] is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.environ.get("OCCFORMER_HIP_LIB") or os.path.join(os.path.dirname(here), "liboccformer_hip.so")
        use_library(ctypes.CDLL(path))
    return _state["lib"]


def _stream(t):
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream      # honoured (the reference uses the default stream)
    if not _state["host_ok"]:
        raise RuntimeError("bev_pool_ext: liboccformer_hip.so has no CPU path; pass device tensors")
    return 0


def _chk(rc):
    if rc:
        raise RuntimeError(f"occformer_hip error {rc}")


def bev_pool_forward(x, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """x [n, c] fp32, geom_feats [n, 4] int32 (x, y, z, batch), int32 intervals -> out [b, d, h, w, c]
    (zero where no interval lands; bev_pool_cuda.cu:20-49)"""
    lib = _lib()
    x, geom_feats = x.contiguous(), geom_feats.contiguous()
    out = torch.empty((int(b), int(d), int(h), int(w), x.shape[1]), dtype=x.dtype, device=x.device)
    _chk(lib.occf_bev_pool_fwd(x.data_ptr(), geom_feats.data_ptr(), interval_starts.data_ptr(),
                               interval_lengths.data_ptr(), out.data_ptr(), int(b), int(d), int(h), int(w),
                               x.shape[0], x.shape[1], interval_starts.numel(), _stream(x)))
    return out


def bev_pool_backward(out_grad, geom_feats, interval_lengths, interval_starts, b, d, h, w):
    """out_grad [b, d, h, w, c] -> x_grad [n, c]: every row of an interval receives its voxel's gradient
    (bev_pool_cuda.cu:52-84)"""
    lib = _lib()
    n, c = geom_feats.shape[0], out_grad.shape[4]
    out_grad = out_grad.contiguous()
    x_grad = torch.empty((n, c), dtype=out_grad.dtype, device=out_grad.device)
    _chk(lib.occf_bev_pool_bwd(out_grad.data_ptr(), geom_feats.data_ptr(), interval_starts.data_ptr(),
                               interval_lengths.data_ptr(), x_grad.data_ptr(), int(b), int(d), int(h), int(w),
                               n, c, interval_starts.numel(), _stream(out_grad)))
    return x_grad
