"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" in the CPU tests).  The path shards by independent samples (the reference
runs batch 1 per GPU under DDP, occformer_nusc_r50_256x704.py:266; mmdet_train.py:72-80): the
forward has NO data-path collective; the training step keeps exactly the reference's two --
DDP's bucketed gradient all-reduce (wired in bench.py / tests/test_ddp_train.py) and the
scalar ``reduce_mean`` of the loss normalisers (mask2former_nusc_occ.py:408) -- plus the timing
barrier / max and, in evaluation, the 16x16 confusion-matrix sum (apis/test.py:206-210)."""
import os

import torch
import torch.distributed as dist


def init(backend, device=None):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    kw = {}
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]),
                            **kw)
    return dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard(n_samples, rank_=None, world_=None):
    """Contiguous per-rank block of sample indices (eval-style DistributedSampler split,
    mmdet3d/datasets/samplers/distributed_sampler.py:35-38 of the reference's fork)."""
    r = rank() if rank_ is None else rank_
    w = world() if world_ is None else world_
    per = (n_samples + w - 1) // w
    return list(range(min(r * per, n_samples), min((r + 1) * per, n_samples)))


def max_over_ranks(value, device="cpu"):
    if world() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_confusion(hist, device="cpu"):
    """apis/test.py:206-210: all-reduce(SUM) of the lidarseg confusion matrix."""
    t = torch.as_tensor(hist, dtype=torch.int64, device=device)
    if world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_mean(t):
    """mmdet.core.reduce_mean (mask2former_nusc_occ.py:408; mask2former_occ.py:425,437): all-reduce(SUM) / world
    of a loss normaliser; identity for a single process."""
    if world() == 1:
        return t
    t = t.clone()
    dist.all_reduce(t.div_(world()), op=dist.ReduceOp.SUM)
    return t
