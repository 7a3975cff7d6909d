"""Which ATen ops (elementwise adds, copies, fills ...) surround the library's kernels in one training step:
torch.profiler over one step, grouped by op name and input shapes, sorted by device time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occformer_amd  # noqa
from occformer_amd import configs
from occformer_amd.registry import build_model
from torch.profiler import profile, ProfilerActivity

device = torch.device("cuda", 0)
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(device).train()
img_inputs, metas, points = configs.synthetic_sample(meta, device, seed=0)
gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, device, seed=0)
train_inputs = list(img_inputs) + [gt_depths]
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)

def step():
    opt.zero_grad(set_to_none=True)
    losses = model(return_loss=True, img_metas=metas, img_inputs=train_inputs, gt_occ=gt_occ, points_occ=gt_points)
    total = sum(v for k, v in losses.items() if "loss" in k)
    total.backward()
    torch.nn.utils.clip_grad_norm_(params, 5.0)
    opt.step()

for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key.startswith("aten::") or e.key.startswith("Memcpy") or e.key.startswith("Memset"):
        t = getattr(e, "self_device_time_total", None)
        if t is None:
            t = e.self_cuda_time_total
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"ATen / memcpy device time in one step: {tot / 1e3:.2f} ms")
for t, n, k, shp in rows[:70]:
    print(f"{t / 1e3:8.3f} ms {n:5d}x  {k:32s} {shp}")

print()
print("call sites of the copy / cat / add / fill ops (device time, count, innermost repo frames):")
sites = {}
for e in prof.events():
    if e.name not in ("aten::copy_", "aten::cat", "aten::add_", "aten::add", "aten::fill_", "aten::zeros", "aten::contiguous",
                      "aten::clone", "aten::index", "aten::mul", "aten::zero_"):
        continue
    t = getattr(e, "device_time_total", None)
    if t is None:
        t = e.cuda_time_total
    frames = [f for f in (e.stack or []) if "/occformer_amd/" in f or "/bench.py" in f or "aten_profile" in f]
    key = (e.name, tuple(f.split("/occformer_amd/")[-1][:70] for f in frames[:3]))
    a = sites.setdefault(key, [0.0, 0])
    a[0] += t
    a[1] += 1
for (name, fr), (t, n) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{t / 1e3:8.3f} ms {n:5d}x {name:18s} {' <- '.join(fr)}")
