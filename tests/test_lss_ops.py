"""View-transform kernels against the oracle and the golden
vectors produced by the reference.  Runs on the host emulation of the kernel sources
(CPU) and, with -m gpu, on the real gfx950 library through the C ABI."""
import pytest
import torch

from oracle import occformer_ref as O
from tests import paramgen
from tests.conftest import golden


def _intervals(coords, B, Z, X, Y):
    order, geom, starts, lengths = O.bev_pool_intervals(coords.long(), B, Z, X, Y)
    return order, geom, starts, lengths


def test_bev_pool_golden_bit_exact(be):
    """reference's own bev_pool() output (argsort + QuickCumsumCuda + permute) on the golden case"""
    g = golden("bev_pool")
    B, Z, X, Y = g["B"], g["Z"], g["X"], g["Y"]
    order, geom, starts, lengths = _intervals(g["coords"], B, Z, X, Y)
    x = g["feats"][order].contiguous()
    out = be.ops.bev_pool_forward(*be.to(x, geom, lengths, starts), B, Z, X, Y).cpu()
    ref = g["out"].permute(0, 2, 3, 4, 1)                       # [B, Z, X, Y, C]
    assert torch.allclose(out, ref, atol=1e-5, rtol=0)
    # vs the oracle's literal kernel restatement: same order, same adds.  index_add_ may
    # re-associate, so compare against an explicit sequential sum for bit-exactness.
    seq = torch.zeros_like(out)
    for s, l in zip(starts.tolist(), lengths.tolist()):
        acc = torch.zeros(x.shape[1])
        for r in range(s, s + l):
            acc = acc + x[r]
        gx, gy, gz, gb = geom[s].tolist()
        seq[gb, gz, gx, gy] = acc
    assert torch.equal(out, seq)


@pytest.mark.parametrize("c", [4, 6, 128])
def test_bev_pool_fwd_bwd_vs_oracle(be, c):
    B, Z, X, Y = 2, 3, 7, 5
    n = 500
    gen = torch.Generator().manual_seed(c)
    coords = torch.stack([torch.randint(0, X, (n,), generator=gen), torch.randint(0, Y, (n,), generator=gen),
                          torch.randint(0, Z, (n,), generator=gen), torch.randint(0, B, (n,), generator=gen)], 1)
    coords[:40] = coords[0]                                     # one long interval
    order, geom, starts, lengths = _intervals(coords, B, Z, X, Y)
    x = paramgen.tensor("x", (n, c), c)[order].contiguous()
    out = be.ops.bev_pool_forward(*be.to(x, geom, lengths, starts), B, Z, X, Y).cpu()
    ref = O.bev_pool_forward(x, geom, starts, lengths, B, Z, X, Y)
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-6)
    og = paramgen.tensor("og", (B, Z, X, Y, c), c)
    xg = be.ops.bev_pool_backward(*be.to(og, geom, lengths, starts), B, Z, X, Y).cpu()
    assert torch.equal(xg, O.bev_pool_backward(og, geom, starts, lengths))


def test_bev_pool_empty(be):
    x = torch.zeros(0, 8)
    geom = torch.zeros(0, 4, dtype=torch.int32)
    e = torch.zeros(0, dtype=torch.int32)
    out = be.ops.bev_pool_forward(*be.to(x, geom, e, e), 1, 2, 3, 4).cpu()
    assert out.shape == (1, 2, 3, 4, 8) and float(out.abs().sum()) == 0.0


def _cam_pack(rots, trans, intrins, post_rots, post_trans, bda):
    from occformer_amd.view_transformer import pack_cameras
    return pack_cameras(rots, trans, intrins, post_rots, post_trans, bda)


@pytest.mark.parametrize("kitti", [False, True])
def test_voxel_index_vs_reference_geometry(be, kitti):
    """kernel geometry+quantisation == get_geometry + voxel_pooling index/mask of the reference"""
    B, N = 2, 3
    H, W, ds = 64, 176, 16
    frustum = O.make_frustum((H, W), ds, [2.0, 10.0, 0.5])
    cams = paramgen.camera_rig(B, N, H, W, 140.0, seed=7, kitti=kitti)
    dx, bx, nx = O.grid_constants([-8, 8, 1.0], [-8, 8, 1.0], [-2, 2, 0.5])
    geom = O.lss_geometry(frustum, *cams)
    coords, kept = O.lss_voxel_coords(geom, dx, bx, nx)
    X, Y, Z = 16, 16, 8
    ref = torch.where(kept, ((coords[:, 3] * X + coords[:, 0]) * Y + coords[:, 1]) * Z + coords[:, 2],
                      torch.full_like(coords[:, 0], -1)).int()
    cam, bda12 = _cam_pack(*cams)
    grid = torch.cat((bx - dx / 2.0, dx, nx)).float()
    vox = be.ops.lss_voxel_index(*be.to(frustum.reshape(-1, 3).contiguous(), cam, bda12, grid),
                                 B, N, X, Y, Z, kitti).cpu()
    mism = int((vox != ref).sum())
    # index work is bit-exact: with the SAME per-camera constants (computed by the same torch CPU ops as the
    # reference's get_geometry) the kernel's un-fused mul/add chain reproduces ATen's 3x3 matmul order
    assert mism == 0, f"{mism} of {ref.numel()} voxel ids differ"
    assert int((ref >= 0).sum()) > 0.2 * ref.numel()
    # the trunc-toward-zero quirk must be exercised
    raw = (geom - (bx - dx / 2.0)) / dx
    quirk = ((raw > -1) & (raw < 0)).any(-1).view(-1) & kept
    assert int(quirk.sum()) > 0 and int((vox[quirk] >= 0).sum()) >= int(quirk.sum()) - mism


def test_voxel_index_golden(be):
    g = golden("view_transformer")
    B, N = g["rots"].shape[:2]
    X, Y, Z = 16, 16, 8
    coords, kept = g["coords"].long(), g["kept"]
    ref = torch.where(kept, ((coords[:, 3] * X + coords[:, 0]) * Y + coords[:, 1]) * Z + coords[:, 2],
                      torch.full_like(coords[:, 0], -1)).int()
    frustum = O.make_frustum((64, 176), 16, [2.0, 10.0, 0.5])
    dx, bx, nx = O.grid_constants([-8, 8, 1.0], [-8, 8, 1.0], [-2, 2, 0.5])
    cam, bda12 = _cam_pack(g["rots"], g["trans"], g["intrins"], g["post_rots"], g["post_trans"], g["bda"])
    grid = torch.cat((bx - dx / 2.0, dx, nx)).float()
    vox = be.ops.lss_voxel_index(*be.to(frustum.reshape(-1, 3).contiguous(), cam, bda12, grid),
                                 B, N, X, Y, Z, False).cpu()
    assert int((vox != ref).sum()) <= 1


@pytest.mark.parametrize("C", [32, 6])
def test_lift_splat_vs_oracle(be, C):
    B, N, D, fH, fW = 2, 3, 16, 4, 11
    X, Y, Z = 16, 16, 8
    frustum = O.make_frustum((64, 176), 16, [2.0, 10.0, 0.5])
    cams = paramgen.camera_rig(B, N, 64, 176, 140.0, seed=3)
    dx, bx, nx = O.grid_constants([-8, 8, 1.0], [-8, 8, 1.0], [-2, 2, 0.5])
    geom = O.lss_geometry(frustum, *cams)
    depth = paramgen.tensor("depth", (B * N, D, fH, fW), 1).softmax(1)
    feat = paramgen.tensor("feat", (B * N, C, fH, fW), 1)
    ref = O.lift_splat(depth, feat, geom, dx, bx, nx)            # [B, C, X, Y, Z]
    coords, kept = O.lss_voxel_coords(geom, dx, bx, nx)
    vox = torch.where(kept, ((coords[:, 3] * X + coords[:, 0]) * Y + coords[:, 1]) * Z + coords[:, 2],
                      torch.full_like(coords[:, 0], -1)).int()
    from occformer_amd.view_transformer import build_voxel_csr
    offsets, pts = build_voxel_csr(vox, B * X * Y * Z)
    feat_cl = feat.permute(0, 2, 3, 1).reshape(B * N, fH * fW, C).contiguous()
    out = be.ops.lift_splat_forward(*be.to(depth.reshape(B * N, D, fH * fW).contiguous(), feat_cl,
                                           offsets, pts), B * X * Y * Z).cpu()
    out = out.view(B, X, Y, Z, C).permute(0, 4, 1, 2, 3)
    assert torch.allclose(out, ref, atol=1e-6, rtol=1e-6)
    # backward against autograd through the oracle
    depth_r = depth.clone().requires_grad_(True)
    feat_r = feat.clone().requires_grad_(True)
    og = paramgen.tensor("og", ref.shape, 2)
    O.lift_splat(depth_r, feat_r, geom, dx, bx, nx).backward(og)
    og_cl = og.permute(0, 2, 3, 4, 1).reshape(-1, C).contiguous()
    dd, df = be.ops.lift_splat_backward(*be.to(og_cl, depth.reshape(B * N, D, fH * fW).contiguous(),
                                               feat_cl, vox))
    assert torch.allclose(dd.cpu().view_as(depth), depth_r.grad, atol=1e-5, rtol=1e-4)
    df = df.cpu().view(B * N, fH, fW, C).permute(0, 3, 1, 2)
    assert torch.allclose(df, feat_r.grad, atol=1e-5, rtol=1e-4)


@pytest.mark.parametrize("groups,dg", [(4, 1), (2, 2)])
def test_deform_conv_vs_oracle(be, groups, dg):
    """csrc/dcn.hip + grouped GEMM == the oracle's restatement of mmcv deform_conv2d"""
    BN, C, H, W, Cout, k = 2, 32, 5, 7, 24, 3
    x = paramgen.tensor("dx", (BN, C, H, W), 1)
    off = paramgen.tensor("doff", (BN, dg * 2 * k * k, H, W), 2, 1.5)
    w = paramgen.tensor("dw", (Cout, C // groups, k, k), 3, 0.2)
    ref = O.deform_conv2d(x, off, w, 1, 1, 1, groups, dg)
    col = be.ops.deform_im2col(be.to(x.permute(0, 2, 3, 1).contiguous()), be.to(off), k, 1, 1, 1, groups, dg)
    out = torch.empty(BN * H * W, Cout)
    out = be.to(out)
    for g, wg in enumerate(w.chunk(groups, 0)):
        wt = be.to(wg.permute(0, 2, 3, 1).reshape(wg.shape[0], -1).contiguous())
        be.ops.linear(col[:, g].flatten(1), wt, out=out[:, g * (Cout // groups):(g + 1) * (Cout // groups)])
    out = out.cpu().view(BN, H, W, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(out, ref, atol=2e-5, rtol=1e-4), float((out - ref).abs().max())
