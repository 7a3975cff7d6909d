"""Which ATen / library operators (not this package's kernels) does one training step launch, how often and for how
long?  torch.profiler over two steps of bench.py's training step; rows = operator x input shapes, sorted by device time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occformer_amd  # noqa
from occformer_amd import configs
from occformer_amd.registry import build_model
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(dev)
img_inputs, metas, points = configs.synthetic_sample(meta, dev, seed=0)
gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, dev, seed=0)
kw = dict(img_metas=metas, img_inputs=list(img_inputs) + [gt_depths], gt_occ=gt_occ, points_occ=gt_points)
model.train()
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)


def step():
    opt.zero_grad(set_to_none=True)
    model.prefetch_gt(gt_occ, ready=True)
    l = model(return_loss=True, **kw)
    sum(v for k, v in l.items() if "loss" in k).backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
ev = prof.key_averages(group_by_input_shape=True)
rows = [(e.self_device_time_total / 2e3, e.count / 2, e.key, str(e.input_shapes)[:110]) for e in ev if e.self_device_time_total > 0]
rows.sort(reverse=True)
print(f"{'ms/step':>8s} {'calls':>6s}  operator  input shapes")
tot = 0.0
for ms, n, k, shp in rows[:70]:
    print(f"{ms:8.3f} {n:6.1f}  {k}  {shp}")
for ms, n, k, shp in rows:
    tot += ms
print("total self device time of all profiled operators:", round(tot, 2), "ms per step")
by = {}
for ms, n, k, shp in rows:
    a = by.setdefault(k, [0.0, 0.0]); a[0] += ms; a[1] += n
print("\nATen / library operators by shape:")
for ms, n, k, shp in [r for r in rows if r[2].startswith("aten::") or "miopen" in r[2].lower()][:80]:
    print(f"{ms:8.3f} {n:6.1f}  {k}  {shp}")
print("\nby operator:")
for k, (ms, n) in sorted(by.items(), key=lambda t: -t[1][0])[:60]:
    print(f"{ms:8.3f} {n:7.1f}  {k}")
