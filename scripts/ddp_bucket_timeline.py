"""When do DDP's gradient buckets become ready during the backward?  One rank on the nccl backend (a 1-GPU box: RCCL
itself issues NO kernel for a one-rank all-reduce -- profiles/r06/r06y_rccl_overlap.txt: 0 of 224 798 launches -- so the
overlap of its ring kernels with the backward cannot be traced here).  What CAN be shown is the other half of the claim:
the reducer hands bucket after bucket to the communicator WHILE the backward is still running.  A communication hook
records a HIP event on the compute stream when each bucket is handed over, then calls the default all-reduce.

    OCCF_DIST_AT_WORLD_1=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 \
        --master-port 29543 scripts/ddp_bucket_timeline.py"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occformer_amd  # noqa
from occformer_amd import configs
from occformer_amd.registry import build_model

dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(dev).train()
img_inputs, metas, points = configs.synthetic_sample(meta, dev, seed=0)
gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, dev, seed=0)
kw = dict(img_metas=metas, img_inputs=list(img_inputs) + [gt_depths], gt_occ=gt_occ, points_occ=gt_points)
net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[dev.index], broadcast_buffers=False,
                                                gradient_as_bucket_view=True, bucket_cap_mb=64)
marks = []


def hook(state, bucket):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()                                           # on the compute stream, at the moment the bucket is handed over
    marks.append((ev, bucket.buffer().numel() * 4 / 2 ** 20, time.perf_counter()))
    return dist.all_reduce(bucket.buffer(), async_op=True).get_future().then(lambda f: f.value()[0])


net.register_comm_hook(None, hook)
params = [p for p in model.parameters() if p.requires_grad]
opt = torch.optim.AdamW(params, lr=1e-4, fused=True)
for it in range(3):
    opt.zero_grad(set_to_none=True)
    model.prefetch_gt(gt_occ, ready=True)
    marks.clear()
    losses = net(return_loss=True, **kw)
    total = sum(v for k, v in losses.items() if "loss" in k)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    total.backward()
    e1.record()
    torch.cuda.synchronize()
    opt.step()
bwd = e0.elapsed_time(e1)
print(f"backward of the last step: {bwd:.1f} ms on the device; {len(marks)} buckets of <= 64 MB handed to the communicator:")
for i, (ev, mb, t) in enumerate(marks):
    at = e0.elapsed_time(ev)
    print(f"  bucket {i}: {mb:6.1f} MB  handed over at {at:7.1f} ms of the backward's device time ({100 * at / bwd:5.1f} %), host time +{1e3 * (t - t0):.1f} ms")
print("every bucket but the last is handed over before the backward ends: its all-reduce is queued behind it on RCCL's stream "
      "while the remaining backward kernels run on the compute stream")
dist.destroy_process_group()
