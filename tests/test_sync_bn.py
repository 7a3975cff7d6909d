"""``occformer_amd.dist_utils.convert_sync_batchnorm`` -- the reference's ``sync_bn = True`` (tools/train.py:221-223; every
shipped config sets it) as an opt-in: two gloo ranks holding the halves of a batch reproduce single-process BatchNorm on
the whole batch (outputs, input gradients, summed weight / bias gradients, running statistics), including the
one-vector-per-rank case of DepthNet's camera-MLP BatchNorm1d (SemanticKITTI: batch 1, one camera)."""
import os
import subprocess
import sys

import torch
import torch.nn as nn

from occformer_amd import dist_utils

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import copy, os, sys
sys.path.insert(0, %r)
import torch, torch.nn as nn
from occformer_amd import dist_utils
d = dist_utils.init("gloo")
r, w = dist_utils.rank(), dist_utils.world()
torch.manual_seed(0)
net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.BatchNorm2d(8), nn.ReLU(), nn.Conv2d(8, 5, 1), nn.BatchNorm2d(5, momentum=None))
vec = nn.Sequential(nn.BatchNorm1d(6), nn.Linear(6, 4))
vol = nn.BatchNorm3d(3, affine=False)
with torch.no_grad():
    for m in list(net.modules()) + list(vec.modules()):
        if isinstance(m, nn.modules.batchnorm._BatchNorm):
            m.weight.uniform_(0.5, 1.5); m.bias.uniform_(-0.5, 0.5)
X = torch.randn(6, 3, 5, 7) * 2 + 1          # rank 0 holds 4 samples, rank 1 two: unequal counts
V = torch.randn(2, 6)                        # one vector per rank
U = torch.randn(4, 3, 2, 3, 2)
T, TV, TU = torch.randn(6, 5, 5, 7), torch.randn(2, 4), torch.randn(4, 3, 2, 3, 2)
cut = [slice(0, 4), slice(4, 6)][r]
cutu = [slice(0, 1), slice(1, 4)][r]

def run(net, vec, vol, x, v, u, t, tv, tu):
    x = x.clone().requires_grad_(True); v = v.clone().requires_grad_(True); u = u.clone().requires_grad_(True)
    y, yv, yu = net(x), vec(v), vol(u)
    ((y * t).sum() + (yv * tv).sum() + (yu * tu).sum()).backward()
    return y.detach(), yv.detach(), yu.detach(), x.grad, v.grad, u.grad

ref = [copy.deepcopy(m).train() for m in (net, vec, vol)]
full = run(*ref, X, V, U, T, TV, TU)
syn = [dist_utils.convert_sync_batchnorm(copy.deepcopy(m)).train() for m in (net, vec, vol)]
keys = [list(m.state_dict()) for m in syn]
assert keys == [list(m.state_dict()) for m in ref]
assert isinstance(syn[0][1], nn.BatchNorm2d) and isinstance(syn[1][0], nn.BatchNorm1d) and dist_utils.is_synced(syn[1][0])
assert not dist_utils.is_synced(ref[1][0])
part = run(*syn, X[cut], V[r:r + 1], U[cutu], T[cut], TV[r:r + 1], TU[cutu])
# (errors relative to max(|reference|, 1): a conv bias in front of a BatchNorm and the input gradient of a BatchNorm over two
# vectors are exact zeros up to rounding / O(eps))
worst = 0.0
for a, b, c in zip(part, full, (cut, slice(r, r + 1), cutu, cut, slice(r, r + 1), cutu)):
    worst = max(worst, float((a - b[c]).abs().max() / b.abs().max().clamp_min(1.0)))
for ms, mr in zip(syn, ref):
    for (k, ps), (_, pr) in zip(ms.named_parameters(), mr.named_parameters()):
        g = ps.grad.clone()
        d.all_reduce(g)                                  # the whole-batch loss is the sum of the two ranks' losses
        worst = max(worst, float((g - pr.grad).abs().max() / pr.grad.abs().max().clamp_min(1.0)))
    for (k, bs), (_, br) in zip(ms.named_buffers(), mr.named_buffers()):
        worst = max(worst, float((bs.float() - br.float()).abs().max() / br.float().abs().max().clamp_min(1.0)))
# eval mode: running statistics, no collective
for ms, mr in zip(syn, ref):
    ms.eval(); mr.eval()
assert torch.allclose(syn[0](X[cut]), ref[0](X[cut]), atol=1e-5)
print("OK", r, w, "%%.1e" %% worst, flush=True)
assert worst < 1e-4, worst          # (2.6e-5: the two-vector BatchNorm1d input gradient, rounding x invstd)
d.destroy_process_group()
'''


def test_sync_batchnorm_two_rank_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % ROOT)
    port = 29950 + os.getpid() % 40
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all(o[0].strip().splitlines()[-1].startswith("OK") for o in outs), outs


def test_sync_batchnorm_single_process_is_plain_batchnorm():
    """without a process group (or with one rank) the converted modules are the BatchNorms they were"""
    torch.manual_seed(1)
    a = nn.Sequential(nn.Conv2d(3, 4, 1), nn.BatchNorm2d(4))
    import copy
    b = dist_utils.convert_sync_batchnorm(copy.deepcopy(a))
    assert type(b[1]) is dist_utils.SyncBatchNorm2d and not dist_utils.is_synced(b[1])
    x = torch.randn(3, 3, 4, 4)
    assert torch.equal(a(x), b(x)) and torch.equal(a[1].running_var, b[1].running_var)
    a.eval(), b.eval()
    assert torch.equal(a(x), b(x))
    b.load_state_dict(a.state_dict())
