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


# OCCF_DIST_AT_WORLD_1=1: an initialised process group of ONE rank still issues every collective of the path
# (reduce_mean, the timing max, SyncBatchNorm's gather / reduce) -- how a 1-GPU box exercises the RCCL call sites
# (profiles/r05: the N > 1 runs are the driver's)
_AT_WORLD_1 = os.environ.get("OCCF_DIST_AT_WORLD_1", "0") == "1"


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    """do the path's collectives run?  (more than one rank, or one rank with OCCF_DIST_AT_WORLD_1=1)"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _AT_WORLD_1)


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
    if not collectives_on():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_confusion(hist, device="cpu"):
    """apis/test.py:206-210: all-reduce(SUM) of the lidarseg confusion matrix."""
    t = torch.as_tensor(hist, dtype=torch.int64, device=device)
    if collectives_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_mean(t):
    """mmdet.core.reduce_mean (mask2former_nusc_occ.py:408; mask2former_occ.py:425,437): all-reduce(SUM) / world
    of a loss normaliser; identity for a single process."""
    if not collectives_on():
        return t
    t = t.clone()
    dist.all_reduce(t.div_(world()), op=dist.ReduceOp.SUM)
    return t


# --------------------------------------------------------------------------- SyncBatchNorm (the configs' ``sync_bn = True``)
class _SyncBatchNormFn(torch.autograd.Function):
    """Batch normalisation on the statistics of ALL ranks' batches.  Forward: per-rank (count, mean, biased variance)
    per channel, one all-gather of [world, 2C + 1] floats, Chan's pairwise combination (no E[x^2] - E[x]^2
    cancellation); backward: one all-reduce of [2C] (sum dy, sum dy * xhat).  Weight / bias gradients stay per-rank
    sums -- DDP's gradient all-reduce averages them like every other parameter, as with torch.nn.SyncBatchNorm."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, group):
        C = x.shape[1]
        dims = [0] + list(range(2, x.dim()))
        shape = [1, C] + [1] * (x.dim() - 2)
        xf = x.float()
        n = x.numel() // C
        if n > 0:
            var_r, mean_r = torch.var_mean(xf, dims, unbiased=False)
        else:
            var_r, mean_r = xf.new_zeros(C), xf.new_zeros(C)
        mine = torch.cat([mean_r, var_r, xf.new_tensor([float(n)])])
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
        dist.all_gather(every, mine, group=group)
        every = torch.stack(every)
        cnt = every[:, -1:]
        total = cnt.sum()
        mean = (every[:, :C] * cnt).sum(0) / total
        var = ((every[:, C:2 * C] + (every[:, :C] - mean) ** 2) * cnt).sum(0) / total
        invstd = torch.rsqrt(var + eps)
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(mean.to(running_mean.dtype), alpha=momentum)
            unbiased = var * (total / (total - 1).clamp_min(1.0))
            running_var.mul_(1 - momentum).add_(unbiased.to(running_var.dtype), alpha=momentum)
        xhat = (xf - mean.view(shape)) * invstd.view(shape)
        y = xhat
        if weight is not None:
            y = y * weight.float().view(shape)
        if bias is not None:
            y = y + bias.float().view(shape)
        ctx.save_for_backward(xhat, weight, invstd, total)
        ctx.group, ctx.has_bias = group, bias is not None
        return y.to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xhat, weight, invstd, total = ctx.saved_tensors
        C = xhat.shape[1]
        dims = [0] + list(range(2, xhat.dim()))
        shape = [1, C] + [1] * (xhat.dim() - 2)
        dyf = dy.float()
        s = torch.cat([dyf.sum(dims), (dyf * xhat).sum(dims)])
        dw = s[C:].clone().to(weight.dtype) if weight is not None and ctx.needs_input_grad[1] else None
        db = s[:C].clone().to(dy.dtype) if ctx.has_bias and ctx.needs_input_grad[2] else None
        dx = None
        if ctx.needs_input_grad[0]:
            dist.all_reduce(s, group=ctx.group)
            s = s / total
            scale = invstd if weight is None else invstd * weight.float()
            dx = ((dyf - s[:C].view(shape) - xhat * s[C:].view(shape)) * scale.view(shape)).to(dy.dtype)
        return dx, dw, db, None, None, None, None, None


class _SyncMixin:
    """forward of the converted BatchNorm classes; a single process, eval mode or ``track_running_stats=False`` in
    eval keep the parent's behaviour"""
    process_group = None

    def forward(self, x):
        self._check_input_dim(x)
        use_batch = self.training or (self.running_mean is None and self.running_var is None)
        if not use_batch or not collectives_on():
            return super().forward(x)
        momentum = 0.0 if self.momentum is None else self.momentum
        track = self.training and self.track_running_stats
        if track and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(1)
            if self.momentum is None:
                momentum = 1.0 / float(self.num_batches_tracked)
        return _SyncBatchNormFn.apply(x, self.weight, self.bias, self.running_mean if track else None,
                                      self.running_var if track else None, self.eps, momentum, self.process_group)


class SyncBatchNorm1d(_SyncMixin, torch.nn.BatchNorm1d):
    pass


class SyncBatchNorm2d(_SyncMixin, torch.nn.BatchNorm2d):
    pass


class SyncBatchNorm3d(_SyncMixin, torch.nn.BatchNorm3d):
    pass


_SYNC_CLASS = {torch.nn.BatchNorm1d: SyncBatchNorm1d, torch.nn.BatchNorm2d: SyncBatchNorm2d,
               torch.nn.BatchNorm3d: SyncBatchNorm3d}


def convert_sync_batchnorm(model, process_group=None):
    """tools/train.py:221-223 (``if distributed and cfg.get('sync_bn', False)``: every shipped config sets it): each
    BatchNorm of ``model`` normalises with the statistics of all ranks' batches.  The modules are re-classed IN PLACE
    (same parameters, buffers, state-dict keys and optimizer references; still instances of their BatchNormNd class, so
    the eval-mode folding and ``norm_eval`` code paths see them unchanged); works on RCCL and on gloo (CPU tests).
    Without it (the default, north_star: the gradient all-reduce is the only collective) BatchNorms use per-rank
    statistics."""
    for m in model.modules():
        cls = _SYNC_CLASS.get(type(m))
        if cls is not None:
            m.__class__ = cls
            m.process_group = process_group
    return model


def is_synced(bn):
    """does this BatchNorm see other ranks' samples (so that one vector per rank still has batch statistics)?"""
    return isinstance(bn, _SyncMixin) and collectives_on()
