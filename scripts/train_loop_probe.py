"""N plain training steps of the bench workload (nothing else in the process): the target of `rocprofv3 --kernel-trace`
runs whose trace scripts/gap_analysis.py turns into busy / idle time per step.
    python scripts/train_loop_probe.py [steps] [warmup]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occformer_amd  # noqa: E402,F401
from occformer_amd import configs  # noqa: E402
from occformer_amd.registry import build_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
warmup = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(dev).train()
img_inputs, metas, _ = configs.synthetic_sample(meta, dev, seed=0)
gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, dev, seed=0)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
kw = dict(img_metas=metas, img_inputs=list(img_inputs) + [gt_depths], gt_occ=gt_occ, points_occ=gt_points)


def step():
    opt.zero_grad(set_to_none=True)
    if os.environ.get("PROBE_PREFETCH", "1") == "1":
        model.prefetch_gt(gt_occ, ready=True)
    losses = model(return_loss=True, **kw)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()


for _ in range(warmup):
    step()
torch.cuda.synchronize()
# marker kernels around the timed region: a 1-element fill of a recognisable dtype
mark = torch.zeros(7, dtype=torch.float64, device=dev)
t0 = time.perf_counter()
for i in range(steps):
    mark.fill_(float(i))
    step()
mark.fill_(-1.0)
torch.cuda.synchronize()
print(f"{steps} steps: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms/step (host wall)")
if os.environ.get("PROBE_CPROFILE", "0") == "1":
    # where the HOST spends a step: launch-side cost only (nothing synchronises inside a step)
    import cProfile
    import pstats
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for i in range(3):
        step()
    pr.disable()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"host time to ISSUE a step (profiled, no sync): {t_host / 3 * 1e3:.1f} ms")
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(45)
    st.sort_stats("cumulative").print_stats(60)
if os.environ.get("PROBE_HOSTTIME", "0") == "1":
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(3):
        step()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"host time to ISSUE a step: {t_issue / 3 * 1e3:.1f} ms; until the GPU is done: {t_all / 3 * 1e3:.1f} ms")
