"""Training-time sampling kernels vs PyTorch / the reference semantics."""
import pytest
import torch
import torch.nn.functional as F

from tests import paramgen


@pytest.mark.parametrize("align,pad", [(False, "border"), (True, "zeros"), (False, "zeros"), (True, "border")])
def test_point_sample_3d(be, align, pad):
    N, C, X, Y, Z, P = 5, 2, 6, 5, 4, 300
    vol = paramgen.tensor("psv", (N, C, X, Y, Z), 1)
    pts = paramgen.uniform("psp", (N, P, 3), 1) * 1.3 - 0.15           # some out of range
    ref = F.grid_sample(vol, (pts * 2 - 1).view(N, P, 1, 1, 3), mode="bilinear", padding_mode=pad,
                        align_corners=align).view(N, C, P)
    out = be.ops.point_sample_3d(be.to(vol), be.to(pts), align, pad).cpu()
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-4)
    shared = pts[:1].contiguous()
    ref2 = F.grid_sample(vol, (shared.expand(N, -1, -1) * 2 - 1).view(N, P, 1, 1, 3), mode="bilinear",
                         padding_mode=pad, align_corners=align).view(N, C, P)
    out2 = be.ops.point_sample_3d(be.to(vol), be.to(shared), align, pad).cpu()
    assert torch.allclose(out2, ref2, atol=2e-5, rtol=1e-4)



def test_point_sample_3d_rows_is_gather_then_sample(be):
    """occf_point_sample_3d_rows_fwd == point_sample_3d(vol[rows][:, None], pts)[:, 0] bit for bit (per-row and shared
    points, both padding modes, repeated and unordered rows); MaskRows.sample / .gather / .dense over two images"""
    from occformer_amd.training import MaskRows
    import occformer_amd.ops as ops_mod
    vol = paramgen.tensor("psr_v", (5, 6, 7, 4), 1)
    rows = torch.tensor([3, 0, 3, 4, 1, 1], dtype=torch.int64)
    pts = paramgen.uniform("psr_p", (6, 50, 3), 2) * 1.2 - 0.1
    for align in (False, True):
        for pad in ("zeros", "border"):
            for p in (pts, pts[:1]):
                ref = be.ops.point_sample_3d(be.to(vol[rows].unsqueeze(1).contiguous()), be.to(p), align, pad)[:, 0]
                out = be.ops.point_sample_3d_rows(be.to(vol), be.to(rows), be.to(p), align, pad)
                assert torch.equal(out.cpu(), ref.cpu())
    old = ops_mod._ops
    ops_mod._ops = be.ops
    try:
        vol2 = paramgen.tensor("psr_v2", (3, 6, 7, 4), 3)
        r2 = torch.tensor([2, 0], dtype=torch.int64)
        mr = MaskRows.cat([MaskRows([(be.to(vol), be.to(rows))]), MaskRows([(be.to(vol2), be.to(r2))])])
        dense = torch.cat((vol[rows], vol2[r2]), 0)
        assert mr.shape == (8, 6, 7, 4) and torch.equal(mr.dense().cpu(), dense)
        pts8 = paramgen.uniform("psr_p8", (8, 20, 3), 4)
        ref = be.ops.point_sample_3d(be.to(dense.unsqueeze(1).contiguous()), be.to(pts8), False, "border")[:, 0]
        assert torch.equal(mr.sample(be.to(pts8), False, "border").cpu(), ref.cpu())
        idx = torch.randint(0, 6 * 7 * 4, (8, 9), generator=torch.Generator().manual_seed(0))
        assert torch.equal(mr.gather(be.to(idx)).cpu(), torch.gather(dense.reshape(8, -1), 1, idx))
    finally:
        ops_mod._ops = old

@pytest.mark.parametrize("align,pad", [(False, "border"), (True, "zeros"), (False, "zeros")])
def test_point_sample_tokens_and_lazy_matching_logits(be, align, pad):
    """channels-last sampling == grid_sample of the channel-major volume; and sampling the mask FEATURES then
    contracting with mask_embed == sampling the query logits einsum('qc,cxyz->qxyz') (the matching cost of the
    training step, mask2former_nusc_occ.py:232-238)"""
    from occformer_amd.training import LazyMask
    import occformer_amd.ops as ops_mod
    X, Y, Z, E, Q, P = 6, 5, 4, 16, 7, 257
    feat = paramgen.tensor("pst_feat", (X * Y * Z, E), 1)
    embed = paramgen.tensor("pst_embed", (Q, E), 2)
    pts = paramgen.uniform("pst_pts", (P, 3), 1) * 1.3 - 0.15
    vol = feat.t().reshape(1, E, X, Y, Z)
    ref = F.grid_sample(vol, (pts * 2 - 1).view(1, P, 1, 1, 3), mode="bilinear", padding_mode=pad,
                        align_corners=align).view(E, P)
    out = be.ops.point_sample_tokens(be.to(feat), (X, Y, Z), be.to(pts), align, pad).cpu()
    assert torch.allclose(out.t(), ref, atol=2e-5, rtol=1e-4)
    # a column slice of a wider token buffer (row stride > C)
    wide = torch.cat((feat, feat * 2), 1).contiguous()
    out2 = be.ops.point_sample_tokens(be.to(wide)[:, E:], (X, Y, Z), be.to(pts), align, pad).cpu()
    assert torch.allclose(out2.t(), 2 * ref, atol=4e-5, rtol=1e-4)
    logits = (embed @ feat.t()).view(1, Q, X, Y, Z)
    ref_q = F.grid_sample(logits, (pts * 2 - 1).view(1, P, 1, 1, 3), mode="bilinear", padding_mode=pad,
                          align_corners=align).view(Q, P)
    saved = ops_mod._ops
    ops_mod._ops = be.ops
    try:
        lazy = LazyMask(be.to(logits)[0], be.to(embed), be.to(feat))
        got = lazy.sample_all(be.to(pts), align, pad).cpu()
    finally:
        ops_mod._ops = saved
    assert got.shape == (Q, P)
    assert float((got - ref_q).abs().max()) <= 1e-4 * float(ref_q.abs().max())


@pytest.mark.parametrize("R,V,k,shared", [(3, 5000, 700, True), (2, 4096, 4096, False), (1, 300, 1, True),
                                          (4, 20000, 15000, True),
                                          # V > 65 536: the multi-kernel pipeline (smaller rows: one workgroup per row)
                                          (2, 70000, 900, True)])
def test_sample_without_replacement_matches_exponential_race_topk(be, R, V, k, shared):
    """same selected SET as torch's own algorithm for multinomial(replacement=False):
    topk(w / q), q ~ Exp(1), with the exponential noise injected"""
    w = paramgen.uniform("sw", (1 if shared else R, V), 2) ** 3
    w[:, ::7] = 0.0                                                      # zero-weight voxels are never drawn
    u = paramgen.uniform("su", (R, V), 3).clamp_min(1e-12)
    keys = torch.where(w > 0, w / (-torch.log(u)).clamp_min(1e-38), torch.zeros(()))
    ref = torch.topk(keys.expand(R, V) if shared else keys, k, dim=1)[1]
    out = be.ops.sample_without_replacement(be.to(w), be.to(u), k).cpu()
    out_e = be.ops.sample_without_replacement(be.to(w), be.to(-torch.log(u)), k, exponential=True).cpu()
    for r in range(R):
        kr = keys[0] if shared else keys[r]
        for o in (out[r], out_e[r]):
            a, b = set(o.tolist()), set(ref[r].tolist())
            assert len(a) == k
            # zero-weight voxels tie at key 0 (only drawn once k exceeds the positive-weight count): any of
            # them is as good as another, so compare the positive-key part of the set
            assert {i for i in a if float(kr[i]) > 0} == {i for i in b if float(kr[i]) > 0}


def test_sample_without_replacement_distribution(be):
    """chi-square style check: empirical inclusion frequency follows the class weights"""
    V, k, R = 2000, 200, 64
    w = torch.ones(1, V)
    w[:, :500] = 4.0
    u = paramgen.uniform("sd", (R, V), 5).clamp_min(1e-12)
    out = be.ops.sample_without_replacement(be.to(w), be.to(u), k).cpu()
    frac_heavy = float((out < 500).float().mean())
    # expected share of heavy items among the first 10% drawn: ~ 4*500/(4*500+1500) = 0.571 (slightly less
    # without replacement)
    assert 0.50 < frac_heavy < 0.60


@pytest.mark.parametrize("V", [9000, 70000])
def test_topk_smallest_abs(be, V):
    R, k = 3, 2500
    v = paramgen.tensor("tka", (R, V), 3)
    ref = torch.topk(-v.abs(), k, dim=1)[1]
    out = be.ops.topk_smallest_abs(be.to(v), k).cpu()
    for r in range(R):
        assert set(out[r].tolist()) == set(ref[r].tolist())


@pytest.mark.parametrize("P", [1234, 8200])          # 8200: the 1024-thread workgroups of the long rows
def test_point_loss_rows(be, P):
    R = 7 if P < 8192 else 2
    x = paramgen.tensor("plx", (R, P), 4)
    t = paramgen.uniform("plt", (R, P), 4)
    out = be.ops.point_loss_rows(be.to(x), be.to(t)).cpu()
    s = x.sigmoid()
    ref = torch.stack([F.binary_cross_entropy_with_logits(x, t, reduction="none").sum(1), (s * t).sum(1), s.sum(1),
                       t.sum(1)], 1)
    assert torch.allclose(out, ref, rtol=2e-5, atol=1e-3)


@pytest.mark.parametrize("P,Q,G", [(1, 100, 17), (3, 100, 20), (2, 7, 7), (2, 5, 12), (1, 130, 1), (1, 64, 65),
                                   (1, 300, 40)])
def test_hungarian_matches_scipy(be, P, Q, G):
    """mask_hungarian_assigner.py:104-126: same assignment (and total cost) as scipy.optimize.linear_sum_assignment
    on the cost matrix the assigner builds (generic float costs: the optimum is unique)"""
    from scipy.optimize import linear_sum_assignment
    cost = paramgen.tensor("hc", (P, Q, G), 3, 2.0) + paramgen.uniform("hu", (P, Q, G), 4)
    match, assigned = be.ops.hungarian(be.to(cost))
    match, assigned = match.cpu(), assigned.cpu()
    for p in range(P):
        r, c = linear_sum_assignment(cost[p].numpy())
        ref = torch.zeros(Q, dtype=torch.int32)
        ref[torch.from_numpy(r)] = torch.from_numpy(c).int() + 1
        assert torch.equal(assigned[p], ref)
        for g in range(G):
            q = int(match[p, g])
            assert (q == -1 and g not in c) or int(ref[q]) == g + 1
    one, _ = be.ops.hungarian(be.to(cost[0]))                            # un-batched call
    assert torch.equal(one.cpu(), match[0])


def test_hungarian_structured_costs(be):
    """integer-valued costs with many ties: the assignment may differ from scipy's, its total cost may not"""
    from scipy.optimize import linear_sum_assignment
    g = torch.Generator().manual_seed(5)
    cost = torch.randint(0, 4, (4, 30, 9), generator=g).float()
    match, assigned = be.ops.hungarian(be.to(cost))
    match, assigned = match.cpu(), assigned.cpu()
    for p in range(cost.shape[0]):
        r, c = linear_sum_assignment(cost[p].numpy())
        q = match[p].long()
        assert len(set(q.tolist())) == cost.shape[2] and (q >= 0).all()          # a full, injective matching
        assert float(cost[p][q, torch.arange(cost.shape[2])].sum()) == float(cost[p].numpy()[r, c].sum())
        assert torch.equal(assigned[p][q], torch.arange(cost.shape[2], dtype=torch.int32) + 1)
        assert int((assigned[p] > 0).sum()) == cost.shape[2]


def test_lazy_mask_contracts_on_demand(be):
    """training.LazyMask without materialised logits: ``dense``, the matched ``rows`` and the per-image view are the
    einsum('bqc,bcxyz->bqxyz') contraction (mask2former_nusc_occ.py:455), computed only when asked for"""
    from occformer_amd.training import LazyMask
    import occformer_amd.ops as ops_mod
    B, Q, E, X, Y, Z = 2, 6, 32, 4, 3, 2
    embed = paramgen.tensor("lm_e", (B, Q, E), 1)
    feat = paramgen.tensor("lm_f", (B, X * Y * Z, E), 2)
    ref = torch.einsum("bqc,bvc->bqv", embed, feat).view(B, Q, X, Y, Z)
    saved = ops_mod._ops
    ops_mod._ops = be.ops
    try:
        fd = be.to(feat)
        split = None if be.ops.precision == "f32" else be.ops.split_bf16(fd)
        lazy = LazyMask(None, be.to(embed), fd, (X, Y, Z), split)
        assert lazy._dense is None and tuple(lazy.shape) == (B, Q, X, Y, Z)
        rows = lazy.rows([torch.tensor([1, 4], device=be.device), torch.tensor([0], device=be.device)])
        assert lazy._dense is None, "matched rows must not materialise the whole volume"
        assert rows.dense.shape == (3, X, Y, Z)
        got = rows.dense.cpu()
        want = torch.cat((ref[0][[1, 4]], ref[1][[0]]), 0)
        assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
        img1 = lazy[1]
        assert float((img1.dense.cpu() - ref[1]).abs().max()) <= 1e-4 * float(ref.abs().max())
        assert float((lazy.dense.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
        # an eagerly materialised LazyMask behaves the same
        eager = LazyMask(be.to(ref), be.to(embed), fd)
        assert torch.equal(eager.rows([torch.tensor([2], device=be.device), torch.tensor([5], device=be.device)]).dense.cpu(),
                           torch.cat((ref[0][[2]], ref[1][[5]]), 0))
    finally:
        ops_mod._ops = saved
