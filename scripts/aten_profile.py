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
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=60,
                                                         max_name_column_width=40, max_shapes_column_width=70))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=40,
                                                  max_name_column_width=40, max_src_column_width=110))
