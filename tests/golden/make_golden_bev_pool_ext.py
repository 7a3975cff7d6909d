"""Records what the REFERENCE's own ``mmdet3d/ops/bev_pool/bev_pool.py`` (imported unmodified from /root/reference)
passes to and expects from its FFI module ``bev_pool_ext`` (bev_pool.cpp:22-87) on a seeded case, so that the GPU box
(no /root/reference) can replay the very calls through ``occformer_amd/integration/bev_pool_ext.py``:

    python tests/golden/make_golden_bev_pool_ext.py      ->  tests/golden/bev_pool_ext_calls.npz

The FFI under the wrapper while recording is the literal sequential restatement of bev_pool_cuda.cu:20-84 below
(one thread per (interval, channel), rows added in order) -- the outputs stored are the reference kernel's, bit for bit.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import paramgen  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
REF_DIR = "/root/reference/mmdetection3d/mmdet3d/ops/bev_pool"


def load_reference_wrapper(ext_module, name="occf_ref_bev_pool"):
    """import the reference's bev_pool.py UNMODIFIED with ``from . import bev_pool_ext`` resolving to ``ext_module``"""
    import importlib
    import types
    for k in [k for k in sys.modules if k == name or k.startswith(name + ".")]:
        del sys.modules[k]
    pkg = types.ModuleType(name)
    pkg.__path__ = [REF_DIR]
    sys.modules[name] = pkg
    sys.modules[name + ".bev_pool_ext"] = ext_module
    pkg.bev_pool_ext = ext_module
    return importlib.import_module(name + ".bev_pool")


class SequentialExt:
    """bev_pool_cuda.cu:20-49 / 52-84 restated literally; records every call"""

    def __init__(self):
        self.calls = {}

    def bev_pool_forward(self, x, geom, lengths, starts, b, d, h, w):
        out = torch.zeros(int(b), int(d), int(h), int(w), x.shape[1])
        for s, l in zip(starts.tolist(), lengths.tolist()):
            acc = torch.zeros(x.shape[1])
            for r in range(s, s + l):
                acc = acc + x[r]
            gx, gy, gz, gb = geom[s].tolist()
            out[gb, gz, gx, gy] = acc
        self.calls.update(x=x, geom=geom, lengths=lengths, starts=starts, bdhw=torch.tensor([b, d, h, w]), out=out)
        return out

    def bev_pool_backward(self, out_grad, geom, lengths, starts, b, d, h, w):
        xg = torch.zeros(geom.shape[0], out_grad.shape[4])
        for s, l in zip(starts.tolist(), lengths.tolist()):
            gx, gy, gz, gb = geom[s].tolist()
            xg[s:s + l] = out_grad[gb, gz, gx, gy]
        self.calls.update(out_grad=out_grad, x_grad=xg)
        return xg


def case(seed=5, n=900, c=12, B=2, D=4, H=9, W=7):
    g = torch.Generator().manual_seed(seed)
    coords = torch.stack([torch.randint(0, H, (n,), generator=g), torch.randint(0, W, (n,), generator=g),
                          torch.randint(0, D, (n,), generator=g), torch.randint(0, B, (n,), generator=g)], 1)
    coords[:60] = coords[0]                                   # one long interval
    feats = paramgen.tensor("bpx.feats", (n, c), seed)
    gout = paramgen.tensor("bpx.gout", (B, c, D, H, W), seed)
    return feats, coords, gout, (B, D, H, W)


def main():
    ext = SequentialExt()
    mod = load_reference_wrapper(ext)
    feats, coords, gout, dims = case()
    f = feats.clone().requires_grad_(True)
    y = mod.bev_pool(f, coords, *dims)                       # [B, C, D, H, W]
    (y * gout).sum().backward()
    c = ext.calls
    np.savez_compressed(os.path.join(OUT, "bev_pool_ext_calls.npz"), y=y.detach().numpy(), feats_grad=f.grad.numpy(),
                        **{k: v.detach().numpy() for k, v in c.items()})
    print("wrote bev_pool_ext_calls.npz:", {k: tuple(v.shape) for k, v in c.items()})


if __name__ == "__main__":
    main()
