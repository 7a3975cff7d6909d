"""Which host-side tensor ops of one TRAINING step launch the small ATen kernels (adds, copies, fills, cats)?
Runs on the CPU through the host emulation of the kernels (the ATen ops a step issues are the same on the GPU):
    python scripts/glue_census_cpu.py  ->  table: count, op, first occformer_amd frame / autograd node"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occformer_amd  # noqa: E402,F401
import occformer_amd.ops as ops_mod  # noqa: E402
from occformer_amd import noise  # noqa: E402
from occformer_amd.registry import build_model  # noqa: E402
from occformer_amd.training import DeviceRNG  # noqa: E402
from tests import paramgen, tinycfg  # noqa: E402
from tests.conftest import Backend  # noqa: E402
from tests.golden.make_golden_train import inputs, train_cfg  # noqa: E402


def main():
    be = Backend("emu")
    ops_mod._ops = be.ops
    cfg, meta = tinycfg.tiny_nusc(ncams=2)
    # the full model's layer counts (launch counts per step then match the bench workload)
    cfg["pts_bbox_head"]["transformer_decoder"]["num_layers"] = 9
    cfg["img_bev_encoder_neck"]["encoder"]["num_layers"] = 6
    cfg["train_cfg"] = dict(pts=train_cfg(num_points=64))
    cfg["test_cfg"] = None
    model = build_model(cfg)
    model.load_state_dict(paramgen.fill_state_dict(model.state_dict(), 77))
    model.train()
    B, N = 1, 2
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=5)
    x = paramgen.tensor("gc_x", (B, N, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, pts = inputs("nusc")
    H, W = meta["input_size"]
    gd = paramgen.uniform("gc_d", (B, N, H, W), 5) * 12.0
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    kw = dict(img_metas=metas, img_inputs=[x, *cams, gd], gt_occ=gt_occ[:1], points_occ=[pts[0]])
    params = [p for p in model.parameters() if p.requires_grad]
    noise.set_rng(DeviceRNG("cpu", seed=1))

    def step():
        for p in params:
            p.grad = None
        losses = model(return_loss=True, **kw)
        sum(v for k, v in losses.items() if "loss" in k).backward()

    step()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
        step()
    agg = collections.defaultdict(int)
    LAUNCH = ("add", "mul", "div", "sub", "cat", "stack", "copy_", "contiguous", "clone", "zeros", "fill_", "zero_",
              "index", "gather", "sum", "mean", "where", "sort", "neg", "sigmoid", "softmax", "exp", "log", "clamp",
              "rsub", "full", "ones", "arange", "eq", "ne", "lt", "gt", "ge", "le", "max", "min", "scatter", "select_backward",
              "slice_backward", "index_put", "masked_fill", "relu", "threshold", "bitwise", "logical", "floor", "rand",
              "randperm", "exponential", "topk", "cumsum", "flip", "repeat", "expand_copy", "to", "_to_copy", "float", "long",
              "int", "abs", "sqrt", "pow", "reciprocal", "nan_to_num", "any", "all", "nonzero", "unique", "bincount",
              "argmax", "one_hot", "binary_cross_entropy", "cross_entropy", "nll_loss", "log_softmax", "linalg")
    for e in prof.events():
        if not e.name.startswith("aten::"):
            continue
        par = e.cpu_parent
        if par is not None and par.name.startswith("aten::"):
            continue
        base = e.name[6:].split(".")[0]
        if not any(base == k or base == k + "_" or base.startswith(k + "_") or base.startswith(k) and k in ("zeros", "ones", "full", "linalg") for k in LAUNCH):
            continue
        st = [s for s in (e.stack or []) if "occformer_amd" in s]
        if st:
            loc = st[0].split("/repo/")[-1][:90]
        else:
            q = par
            while q is not None and not q.name.startswith("autograd::engine") and q.cpu_parent is not None:
                q = q.cpu_parent
            loc = q.name[:90] if q is not None else "?"
        agg[(e.name, loc)] += 1
    rows = sorted(agg.items(), key=lambda kv: -kv[1])
    want = sys.argv[1:] or None
    tot = collections.Counter()
    for (n, loc), c in rows:
        tot[n] += c
    print("top-level aten ops per step (tiny config):", dict(tot.most_common(25)))
    for (n, loc), c in rows[:120]:
        if want is None or any(w in n for w in want):
            print(f"{c:5d}  {n:28s} {loc}")


if __name__ == "__main__":
    main()
