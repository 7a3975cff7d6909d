"""Multi-GPU training path on CPU: world_size 2, gloo, one process per rank, the kernels through the host emulation.
The reference's parallelism is DDP with one gradient all-reduce per step (P/occformer/apis/mmdet_train.py:72-80) plus the
``reduce_mean`` of the mask-loss normaliser (mask2former_nusc_occ.py:408).  Each rank runs ONE tiny training step of
``OccupancyFormer`` on its own sample under ``torch.nn.parallel.DistributedDataParallel``, applies grad-clip + ``AdamW(fused=True)`` (the bench's optimizer, which
does not bump ``param._version``), and runs a SECOND step on the updated weights; asserted:
  * every parameter received a gradient (DDP would stall otherwise) and the all-reduced gradients / post-step
    parameters are bit-identical on both ranks;
  * the all-reduced gradient equals the mean of the two per-sample gradients computed WITHOUT DDP (each rank runs its
    sample on a freshly built plain model; the mean is taken explicitly from an all_gather; the two samples carry the
    same label set, so the cross-rank normaliser equals the local one);
  * the same for the second step, against freshly built models loaded with the updated weights (stale weight
    layouts under DDP would show here; VERDICT r2 #1b)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
import occformer_amd, occformer_amd.ops as ops_mod
from occformer_amd import dist_utils, noise
from occformer_amd.registry import build_model
from occformer_amd.training import DeviceRNG
from tests import paramgen, tinycfg
from tests.conftest import Backend
from tests.golden.make_golden_train import inputs, train_cfg

torch.set_num_threads(2)
be = Backend("emu"); ops_mod._ops = be.ops
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])

def build():
    cfg, meta = tinycfg.tiny_nusc(ncams=2)
    cfg["pts_bbox_head"]["transformer_decoder"]["num_layers"] = 3
    cfg["img_bev_encoder_backbone"]["block_numbers"] = [1, 1, 1, 1]
    cfg["img_bev_encoder_neck"]["encoder"]["num_layers"] = 1
    cfg["train_cfg"] = dict(pts=train_cfg(num_points=64)); cfg["test_cfg"] = None
    m = build_model(cfg)
    m.load_state_dict(paramgen.fill_state_dict(m.state_dict(), 91))
    return m.train(), meta

def sample(r, meta):
    cams = paramgen.camera_rig(1, 2, *meta["input_size"], meta["focal"], seed=20 + r)
    x = paramgen.tensor(f"ddp_x{r}", (1, 2, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, pts = inputs("nusc")
    occ = gt_occ[:1] if r == 0 else gt_occ[:1].flip(1)        # same label set on both ranks
    H, W = meta["input_size"]
    gd = paramgen.uniform(f"ddp_d{r}", (1, 2, H, W), 5) * 12.0
    gd = torch.where(paramgen.uniform(f"ddp_k{r}", (1, 2, H, W), 6) < 0.05, gd, torch.zeros_like(gd))
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    return dict(img_metas=metas, img_inputs=[x, *cams, gd], gt_occ=occ, points_occ=[pts[r]])

def step(net, r, meta):
    noise.set_rng(DeviceRNG("cpu", seed=100 + r))
    losses = net(return_loss=True, **sample(r, meta))
    sum(v for k, v in losses.items() if "loss" in k).backward()

dist_utils.init("gloo")
model, meta = build()
ddp = torch.nn.parallel.DistributedDataParallel(model, broadcast_buffers=False, gradient_as_bucket_view=True)
step(ddp, rank, meta)
missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
assert not missing, missing
flat = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad])
both = [torch.empty_like(flat) for _ in range(world)]
dist.all_gather(both, flat)
assert torch.equal(both[0], both[1]), "all-reduced gradients differ between ranks"
# the bench's optimizer: fused AdamW updates the storage WITHOUT bumping param._version (fused.py, _EPOCH)
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=2e-3, weight_decay=0.01, fused=True)
torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 5.0)
opt.step()
chk = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
both = [torch.empty_like(chk) for _ in range(world)]
dist.all_gather(both, chk)
assert torch.equal(both[0], both[1]), "parameters diverged after the step"
# second step under DDP on the UPDATED weights
sd1 = {k: v.detach().clone() for k, v in model.state_dict().items()}
opt.zero_grad(set_to_none=True)
step(ddp, rank, meta)
flat2 = torch.cat([p.grad.reshape(-1) for p in model.parameters() if p.requires_grad])
# references WITHOUT DDP: rank r runs its own sample on a freshly built plain model (no hooks, no buckets); the mean over
# the ranks is taken explicitly from an all_gather.  (Both samples on rank 0, as before, cost two more emulated steps.)
def single(sd):
    m2, _ = build()
    if sd is not None:
        m2.load_state_dict(sd)
    step(m2, rank, meta)
    g = torch.cat([p.grad.reshape(-1) for p in m2.parameters() if p.requires_grad])
    parts = [torch.empty_like(g) for _ in range(world)]
    dist.all_gather(parts, g)
    return sum(parts) / world
ref = single(None)
ref2 = single(sd1)                     # freshly built models at the updated weights: no cache can be stale there
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    err = float((flat - ref).norm() / ref.norm())
    assert err < 1e-5, err
    err2 = float((flat2 - ref2).norm() / ref2.norm())
    moved = float((ref2 - ref).norm() / ref.norm())
    assert moved > 1e-2, ("the update is too small for the second step to tell stale from current weights", moved)
    assert err2 < 1e-5, ("second DDP step (after fused AdamW) disagrees with fresh models at the updated weights", err2)
    print("OK", f"{err:.1e}", f"{err2:.1e}", f"moved {moved:.1e}", flat.numel())
'''


def test_two_rank_ddp_training_step(tmp_path):
    script = tmp_path / "ddp_worker.py"
    script.write_text(_WORKER % ROOT)
    port = 29900 + os.getpid() % 90
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-3000:] for o in outs]
    assert outs[0][0].strip().splitlines()[-1].startswith("OK"), outs[0]
