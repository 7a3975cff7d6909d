"""Which ATen kernels a training step launches, by call site: a TorchDispatchMode over one step of the (shrunk, full
layer count) detector on the host-emulation backend.  Counts only ops that launch a kernel on a GPU (views, empty and
metadata ops excluded); the call site is the innermost frame inside occformer_amd/ (forward code and custom backward
nodes), or the autograd node for ATen's own backward formulas.
    python scripts/glue_census2.py [min_count]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VIEWS = {"view", "_unsafe_view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze",
         "t", "detach", "alias", "as_strided", "unbind", "split", "split_with_sizes", "chunk", "empty", "empty_like",
         "empty_strided", "new_empty", "new_empty_strided", "_reshape_alias", "unfold", "diagonal", "view_as_real",
         "lift_fresh", "is_same_size", "sym_size", "sym_stride", "sym_numel", "_local_scalar_dense", "item",
         "is_nonzero", "set_", "resize_", "narrow", "movedim", "flatten", "unflatten", "result_type", "stride", "size",
         "numel", "dim", "is_contiguous", "_version", "prim", "contiguous"}


class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()
        self.big = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = func.overloadpacket.__name__
        if name in VIEWS:
            return out
        if name == "_to_copy":
            src = args[0]
            if isinstance(out, torch.Tensor) and out.dtype == src.dtype and out.device == src.device:
                pass
        site = "?"
        for fr in reversed(traceback.extract_stack(limit=40)):
            if "/occformer_amd/" in fr.filename and "ops.py" not in fr.filename:
                site = "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
                break
        n = out.numel() if isinstance(out, torch.Tensor) else 0
        self.agg[(name, site)] += 1
        if n >= 1 << 20:
            self.big[(name, site)] += n
        return out


def main():
    import occformer_amd  # noqa: F401
    import occformer_amd.ops as ops_mod
    from occformer_amd import noise
    from occformer_amd.registry import build_model
    from occformer_amd.training import DeviceRNG
    from tests import paramgen, tinycfg
    from tests.conftest import Backend
    from tests.golden.make_golden_train import inputs, train_cfg
    be = Backend("emu")
    ops_mod._ops = be.ops
    cfg, meta = tinycfg.tiny_nusc(ncams=2)
    cfg["pts_bbox_head"]["transformer_decoder"]["num_layers"] = 9
    cfg["img_bev_encoder_neck"]["encoder"]["num_layers"] = 6
    cfg["train_cfg"] = dict(pts=train_cfg(num_points=64))
    cfg["test_cfg"] = None
    model = build_model(cfg)
    model.train()
    B, N = 1, 2
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=30)
    x = paramgen.tensor("gc_x", (B, N, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, pts = inputs("nusc")
    H, W = meta["input_size"]
    gd = paramgen.uniform("gc_d", (B, N, H, W), 5) * 12.0
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    kw = dict(img_metas=metas, img_inputs=[x, *cams, gd], gt_occ=gt_occ[:1], points_occ=[pts[0]])
    params = [p for p in model.parameters() if p.requires_grad]
    noise.set_rng(DeviceRNG("cpu", seed=1))
    opt = torch.optim.AdamW(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **kw)
        sum(v for k, v in losses.items() if "loss" in k).backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()

    step()
    step()
    with Census() as c:
        step()
    minc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    tot = sum(c.agg.values())
    print("kernel-launching ATen calls in one step: %d" % tot)
    by_site = collections.Counter()
    for (name, site), n in c.agg.items():
        by_site[site] += n
    print("-- by call site")
    for site, n in by_site.most_common(60):
        ops = sorted(((k[0], v) for k, v in c.agg.items() if k[1] == site), key=lambda kv: -kv[1])
        print("%5d  %-48s %s" % (n, site, " ".join("%s:%d" % kv for kv in ops[:8])))
    print("-- (op, site) >= %d" % minc)
    for (name, site), n in c.agg.most_common():
        if n >= minc:
            print("%5d  %-22s %s" % (n, name, site))


if __name__ == "__main__":
    main()
