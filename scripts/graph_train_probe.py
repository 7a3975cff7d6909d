"""Can the WHOLE training step (forward_train + backward + clip + fused AdamW) be captured in one HIP graph, and what
does replay save?  The step is free of host synchronisation (e8bf3fc), ~5 600 launches of which ~4 000 run < 10 us.
    python scripts/graph_train_probe.py [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occformer_amd  # noqa: E402,F401
from occformer_amd import configs, noise  # noqa: E402
from occformer_amd.registry import build_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(dev).train()
img_inputs, metas, _ = configs.synthetic_sample(meta, dev, seed=0)
gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, dev, seed=0)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True, capturable=True)
kw = dict(img_metas=metas, img_inputs=list(img_inputs) + [gt_depths], gt_occ=gt_occ, points_occ=gt_points)
head = model.pts_bbox_head
gt_prepared = head.preprocess_gt(gt_occ, metas)          # (host reads n_present once, outside the graph)
rng = noise.get_rng(dev)


def body():
    losses = model(return_loss=True, gt_prepared=gt_prepared, **kw)
    total = sum(v for k, v in losses.items() if "loss" in k)
    total.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()
    return total


def eager_step():
    opt.zero_grad(set_to_none=True)
    return body()


def timeit(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        eager_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
ms_eager, l_eager = timeit(eager_step, steps)
print("eager: %.2f ms/step, loss %.5f" % (ms_eager, float(l_eager)), flush=True)

head._infeasible_pending = None
g = torch.cuda.CUDAGraph()
if hasattr(rng, "gen"):
    g.register_generator_state(rng.gen)
opt.zero_grad(set_to_none=True)
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        static_loss = body()
except Exception as e:  # noqa: BLE001
    import traceback
    traceback.print_exc()
    print("capture failed:", type(e).__name__, str(e)[:600], flush=True)
    sys.exit(0)
head._infeasible_pending = None
g.replay()
torch.cuda.synchronize()
print("captured; loss after one replay %.5f" % float(static_loss), flush=True)
ms_graph, _ = timeit(g.replay, steps)
print("graph replay: %.2f ms/step, loss %.5f" % (ms_graph, float(static_loss)), flush=True)
