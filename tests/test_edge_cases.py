"""Empty / degenerate inputs through the C ABI: every entry point either does nothing and reports success or
refuses with an argument error -- never a launch with an empty grid (hipErrorInvalidConfiguration) or a crash."""
import pytest
import torch

from occformer_amd.ops import OccfError
from tests import paramgen


def test_point_sample_zero_points(be):
    vol = paramgen.tensor("e.vol", (1, 3, 4, 4, 2), 1)
    pts = torch.zeros((1, 0, 3))
    out = be.ops.point_sample_3d(be.to(vol), be.to(pts), True, "border")
    assert tuple(out.shape) == (1, 3, 0)


def test_lidarseg_zero_points(be):
    mp = paramgen.tensor("e.mp", (1, 5, 4, 4, 2), 1)
    cls = paramgen.tensor("e.cls", (1, 5, 18), 2)
    out = be.ops.lidarseg_sample(be.to(mp), be.to(cls), be.to(torch.zeros((0, 4))))
    assert tuple(out.shape) == (0, 17)


def test_bev_pool_no_points(be):
    """every point filtered out (an empty frustum): the pooled volume is all zeros"""
    C, B, Z, Y, X = 4, 1, 2, 3, 3
    feats = torch.zeros((0, C))
    coords = torch.zeros((0, 4), dtype=torch.int32)
    starts = torch.zeros((0,), dtype=torch.int32)
    lengths = torch.zeros((0,), dtype=torch.int32)
    out = be.ops.bev_pool_forward(be.to(feats), be.to(coords), be.to(lengths), be.to(starts), B, Z, Y, X)
    assert float(out.abs().sum()) == 0.0 and out.numel() == B * Z * Y * X * C


def test_hungarian_degenerate(be):
    # a single query and a single ground-truth row
    m, a = be.ops.hungarian(be.to(torch.tensor([[0.3]])))
    assert m.cpu().tolist() == [0] and a.cpu().tolist() == [1]
    # more ground-truth rows than queries: every query is used exactly once, the spare rows stay unmatched
    cost = paramgen.tensor("e.hc", (3, 5), 1)
    m, a = be.ops.hungarian(be.to(cost))
    m, a = m.cpu(), a.cpu()
    assert sorted(v for v in m.tolist() if v >= 0) == [0, 1, 2] and (m == -1).sum() == 2
    assert sorted(a.tolist()) == sorted(g + 1 for g, q in enumerate(m.tolist()) if q >= 0)


def test_linear_rejects_bad_shapes(be):
    x = paramgen.tensor("e.x", (4, 30), 1)          # K % 4 != 0 is not a layout the kernels accept silently
    w = paramgen.tensor("e.w", (8, 30), 2)
    try:
        out = be.ops.linear(be.to(x), be.to(w)).cpu()
    except OccfError:
        return
    assert torch.allclose(out, x @ w.t(), atol=1e-4, rtol=1e-4)


def test_sample_without_replacement_takes_everything(be):
    """k == number of positive-weight voxels: the sample is exactly that set"""
    w = torch.zeros(1, 257)
    w[0, ::3] = 1.0
    k = int((w > 0).sum())
    u = paramgen.uniform("e.u", (1, 257), 3).clamp_min(1e-12)
    out = be.ops.sample_without_replacement(be.to(w), be.to(u), k).cpu()
    assert sorted(out[0].tolist()) == torch.nonzero(w[0] > 0).flatten().tolist()
