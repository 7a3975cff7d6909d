"""Image -> voxel view transform (LSS depth-splat voxel pooling), registry name
``ViewTransformerLiftSplatShootVoxel``.

Host-side mirror of projects/mmdet3d_plugin/occformer/image2bev/ViewTransformerLSSVoxel.py
(+ the pieces of ViewTransformerLSSBEVDepth.py it inherits): same constructor keys, same
``get_mlp_input`` / ``forward`` / ``get_depth_loss`` contract, same state-dict names
(SURVEY.md Appendix D).  The lift, the geometry/quantisation and the voxel pooling run in
the gfx950 kernels of csrc/lss.hip; the [B,N,D,fH,fW,C] volume is never materialised.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import noise
from .dist_utils import is_synced
from .ops import get_ops
from .registry import NECKS


# ------------------------------------------------------------------ host-side packing
def pack_cameras(rots, trans, intrins, post_rots, post_trans, bda):
    """Per-camera constants for occf_lss_voxel_index, computed with the same torch ops and
    in the same order as get_geometry (ViewTransformerLSSBEVDepth.py:126-141).
    Returns cam [B*N, 27] and bda12 [B, 12] (3x4 row-major)."""
    B, N = trans.shape[:2]
    # linalg.inv_ex = torch.inverse without the singularity check, whose `info` read-back is a device-to-host
    # synchronisation at the very start of every forward (same LU, same values)
    ipr = torch.linalg.inv_ex(post_rots)[0]
    if intrins.shape[-1] == 4:
        shift = intrins[:, :, :3, 3]
        K = intrins[:, :, :3, :3]
    else:
        shift = torch.zeros_like(trans)
        K = intrins
    comb = rots.matmul(torch.linalg.inv_ex(K)[0])
    cam = torch.cat((ipr.reshape(B, N, 9), post_trans.reshape(B, N, 3), comb.reshape(B, N, 9),
                     trans.reshape(B, N, 3), shift.reshape(B, N, 3)), -1).reshape(B * N, 27)
    bda12 = torch.zeros(B, 3, 4, dtype=torch.float32, device=bda.device)
    if bda.shape[-1] == 4:
        bda12.copy_(bda[:, :3, :4])
    else:
        bda12[:, :, :3] = bda
    return cam.float().contiguous(), bda12.reshape(B, 12).contiguous()


def build_voxel_csr(vox, n_vox):
    """Stable counting order of the kept points by voxel row: offsets [n_vox+1] i32 and the
    point indices sorted by (voxel, original index).  Out-of-range points (vox = -1) sort
    to the tail and are never referenced.  No host synchronisation."""
    key = torch.where(vox < 0, torch.full_like(vox, n_vox), vox)
    skey, order = torch.sort(key, stable=True)
    # offsets[v] = first position whose key >= v (binary search on the sorted keys)
    bounds = torch.arange(n_vox + 1, dtype=skey.dtype, device=vox.device)
    offsets = torch.searchsorted(skey, bounds, out_int32=True)
    return offsets, order.int()


class _LiftSplat(torch.autograd.Function):
    """Fused lift+splat with the hand-written backward (the reference's counterpart is
    QuickCumsumCuda, mmdet3d/ops/bev_pool/bev_pool.py:37-80)."""

    @staticmethod
    def forward(ctx, depth, feat_cl, vox, offsets, sorted_pts, n_vox):
        ctx.save_for_backward(depth, feat_cl, vox)
        return get_ops().lift_splat_forward(depth, feat_cl, offsets, sorted_pts, n_vox)

    @staticmethod
    def backward(ctx, grad):
        depth, feat_cl, vox = ctx.saved_tensors
        d_depth, d_feat = get_ops().lift_splat_backward(grad.contiguous(), depth, feat_cl, vox)
        return d_depth, d_feat, None, None, None, None


_DCN_CACHE = {}


# ------------------------------------------------------------------ DepthNet (dense 2-D part)
def _conv_train(conv, x):
    """Training mode: an nn.Conv2d of DepthNet on the library's kernel pair (channels-last implicit GEMM forward,
    wgrad / dgrad kernels backward) instead of ATen / MIOpen -- 7 ms of fp32 MIOpen kernels per training step at the
    nuScenes sizes, and MIOpen's per-box algorithm search out of the benchmark.  x NCHW (any strides) -> the NCHW
    view of the channels-last result; the BatchNorm that follows stays on ATen (train-mode batch statistics)."""
    from . import autograd as A
    y = A.conv(x.permute(0, 2, 3, 1).unsqueeze(3).contiguous(), conv)          # [BN, H, W, 1, Cout]
    return y.squeeze(3).permute(0, 3, 1, 2)


# OCCF_DEPTHNET_LIB (default auto): DepthNet's training-mode convolutions on the library's kernel pairs instead of ATen /
# MIOpen.  At the nuScenes sizes (M = 6 x 16 x 44 = 4 224 rows) 139.5 vs 141.0 ms per step (r04i; r03b: 170.4 vs 172.0)
# and no dependence on MIOpen's per-box algorithm search (SemanticKITTI, 640 channels / one camera: 149 ms with the
# library kernels, 139 ms with MIOpen on a box where its search settles on the igemm kernels, 278 ms on one where it did
# not -- r02 probes 8 / 27).  It stayed off while a non-ATen summation order could tip the tiny parity configuration
# over its bound through a flipped ReLU gate; the comparisons now force the heavy gates.
# Per shape: the SemanticKITTI workloads (one camera: 24 x 80 = 1 920 rows of 640 channels) run 104.1 ms per step with
# DepthNet on MIOpen and 108.0 ms on the library's kernels (r04j) -- too few rows to fill the chip with 128-row tiles -- so
# the default ("auto") takes the library's kernels from 4 096 rows up; 0 / 1 force either side.
_DEPTHNET_LIB = os.environ.get("OCCF_DEPTHNET_LIB", "auto")


def _relu(x):
    """every ReLU of DepthNet's TRAINING graph (ATen).  The one comparison tap of this module: DepthNet's ReLUs sit
    behind train-mode BatchNorms and weigh on its own parameters' gradients, so comparisons against the oracle take
    their gates like the head's (noise.relu_gate: a no-op unless a comparison records)."""
    return noise.relu_gate(F.relu(x))


def _conv(mod, conv, x):
    lib = _DEPTHNET_LIB == "1" or (_DEPTHNET_LIB != "0" and x.is_cuda and x.shape[0] * x.shape[2] * x.shape[3] >= 4096)
    return _conv_train(conv, x) if lib and mod.training and torch.is_grad_enabled() else conv(x)


class _BasicBlock(nn.Module):
    """mmdet ResNet BasicBlock as used by DepthNet (ViewTransformerLSSBEVDepth.py:475-477)."""

    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        y = _relu(self.bn1(_conv(self, self.conv1, x)))
        return _relu(self.bn2(_conv(self, self.conv2, y)) + x)

    def forward_cl(self, x_cl):
        """channels-last [BN, H, W, 1, C]; BatchNorms folded into the convolutions"""
        from . import fused
        y = fused.conv_bn(x_cl, self.conv1, self.bn1, act=1)
        return fused.conv_bn(y, self.conv2, self.bn2, act=0, residual=x_cl).relu_()


class _AtrousBranch(nn.Module):
    def __init__(self, cin, cout, k, dil):
        super().__init__()
        self.atrous_conv = nn.Conv2d(cin, cout, k, padding=0 if k == 1 else dil, dilation=dil, bias=False)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return _relu(self.bn(_conv(self, self.atrous_conv, x)))

    def forward_cl(self, x_cl):
        from . import fused
        return fused.conv_bn(x_cl, self.atrous_conv, self.bn, act=1)


class _ImageASPP(nn.Module):
    """ViewTransformerLSSBEVDepth.py:337-407."""

    def __init__(self, cin, mid):
        super().__init__()
        self.aspp1 = _AtrousBranch(cin, mid, 1, 1)
        self.aspp2 = _AtrousBranch(cin, mid, 3, 6)
        self.aspp3 = _AtrousBranch(cin, mid, 3, 12)
        self.aspp4 = _AtrousBranch(cin, mid, 3, 18)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(cin, mid, 1, bias=False),
                                             nn.BatchNorm2d(mid), nn.ReLU())
        self.conv1 = nn.Conv2d(mid * 5, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid)
        self.dropout = nn.Dropout(0.5)

    def forward(self, x):
        gp = self.global_avg_pool
        # (the 1 x 1 convolution of the pooled [BN, C, 1, 1] vector as a linear: see _SE.forward)
        g = F.linear(gp[0](x).flatten(1), gp[1].weight.flatten(1), gp[1].bias)[:, :, None, None]
        if self.training and g.numel() == g.shape[1] and not is_synced(gp[2]):
            # one pooled value per channel on this rank (SemanticKITTI: batch 1, one camera): batch statistics do
            # not exist -- the reference gets them from SyncBatchNorm over its 8 ranks (here:
            # dist_utils.convert_sync_batchnorm, opt-in); under the default per-rank policy this layer
            # normalises with its running statistics
            g = F.batch_norm(g, gp[2].running_mean, gp[2].running_var, gp[2].weight, gp[2].bias, False, 0.0, gp[2].eps)
        else:
            g = gp[2](g)
        branches = (self.aspp1(x), self.aspp2(x), self.aspp3(x), self.aspp4(x))      # (the reference's order: branches, then pool)
        g = _relu(g).expand(-1, -1, *x.shape[2:])                                        # gp[3] = nn.ReLU
        y = torch.cat((*branches, g), 1)
        y = _relu(self.bn1(_conv(self, self.conv1, y)))
        if self.training:
            # nn.Dropout(0.5) as an explicit mask from the injectable noise source (occformer_amd/noise.py)
            mask = noise.dropout_mask(tuple(y.shape), self.dropout.p, y.device)
            return y if mask is None else y * mask
        return y

    def forward_cl(self, x_cl):
        """channels-last [BN, H, W, 1, C] (eval: dropout is the identity)"""
        from . import fused
        BN_, H, W, _, C = x_cl.shape
        g = x_cl.mean((1, 2, 3)).view(BN_, C, 1, 1)                       # AdaptiveAvgPool2d((1, 1))
        g = self.global_avg_pool[3](self.global_avg_pool[2](self.global_avg_pool[1](g)))   # [BN, mid, 1, 1]
        g = g.view(BN_, 1, 1, 1, -1).expand(BN_, H, W, 1, -1)
        y = torch.cat((self.aspp1.forward_cl(x_cl), self.aspp2.forward_cl(x_cl), self.aspp3.forward_cl(x_cl),
                       self.aspp4.forward_cl(x_cl), g), -1)
        return fused.conv_bn(y, self.conv1, self.bn1, act=1)


class _CamMlp(nn.Module):
    def __init__(self, cin, hidden, cout):
        super().__init__()
        self.fc1 = nn.Linear(cin, hidden)
        self.fc2 = nn.Linear(hidden, cout)

    def forward(self, x):
        return self.fc2(_relu(self.fc1(x)))


class _SE(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv_reduce = nn.Conv2d(c, c, 1)
        self.conv_expand = nn.Conv2d(c, c, 1)

    def forward(self, x, x_se):
        # the two 1 x 1 convolutions act on ONE vector per camera ([BN, C, 1, 1]): as linears on [BN, C] -- the same
        # arithmetic, and the weight gradients come back in the parameters' own strides (as convolutions ATen returned
        # them channels_last, which DDP's bucket views refuse: a copy per gradient and step, r05v)
        v = x_se.flatten(1)
        h = _relu(F.linear(v, self.conv_reduce.weight.flatten(1), self.conv_reduce.bias))
        g = torch.sigmoid(F.linear(h, self.conv_expand.weight.flatten(1), self.conv_expand.bias))
        return x * g[:, :, None, None]

    def forward_cl(self, x_cl, x_se):
        """x_cl [BN, H, W, 1, C]; x_se [BN, C, 1, 1] (the gate is a per-camera channel vector)"""
        gate = torch.sigmoid(self.conv_expand(F.relu(self.conv_reduce(x_se))))
        return x_cl * gate.view(gate.shape[0], 1, 1, 1, -1)


class DeformConv2dPack(nn.Module):
    """DCNv1 as configured in DepthNet (ViewTransformerLSSBEVDepth.py:479-487: 3x3, pad 1,
    groups 4, deform_groups 1, no bias; mmcv zero-initialises conv_offset).  Bilinear
    im2col (zeros outside) followed by a grouped contraction."""

    def __init__(self, cin, cout, k=3, padding=1, groups=4, deform_groups=1):
        super().__init__()
        self.k, self.padding, self.groups, self.deform_groups = k, padding, groups, deform_groups
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k, k))
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        self.conv_offset = nn.Conv2d(cin, deform_groups * 2 * k * k, k, padding=padding)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def _group_weights(self):
        """weight [Cout, Cin/groups, k, k] -> per group [Cout/groups, k*k*Cin/groups] (tap-major K)."""
        from . import fused
        w = self.weight
        ops = get_ops()

        def make():
            ws = [wg.permute(0, 2, 3, 1).reshape(wg.shape[0], -1).contiguous() for wg in w.detach().chunk(self.groups, 0)]
            # (hi, lo) bf16 split per group: K = 9 * Cin/groups is sliced over the chip by the split-K GEMM
            return [(wg, None if ops.precision == "f32" else ops.split_bf16(wg)) for wg in ws], ops.precision
        hit = fused._versioned(_DCN_CACHE, w, make)
        if hit[1] != ops.precision:                  # precision mode switched since the cache entry was made
            _DCN_CACHE.pop(id(w), None)
            hit = fused._versioned(_DCN_CACHE, w, make)
        return hit[0]

    def forward(self, x):
        """x [BN, C, H, W] -> [BN, Cout, H, W]; bilinear im2col in csrc/dcn.hip, then one MFMA GEMM
        per conv group on its contiguous K-slab."""
        ops = get_ops()
        B, C, H, W = x.shape
        k, pad, G = self.k, self.padding, self.groups
        if self.training:
            from . import autograd as A
            out = A.DeformConv.apply(x.permute(0, 2, 3, 1), self.conv_offset(x), self.weight, k, pad, G,
                                     self.deform_groups)
            return out.view(B, H, W, -1).permute(0, 3, 1, 2)
        offset = self.conv_offset(x).contiguous()
        col = ops.deform_im2col(x.permute(0, 2, 3, 1).contiguous(), offset, k, 1, pad, 1, G, self.deform_groups)
        Cout = self.weight.shape[0]
        out = torch.empty((B * H * W, Cout), dtype=x.dtype, device=x.device)
        for g, (wg, sp) in enumerate(self._group_weights()):
            ops.linear(col[:, g].flatten(1), wg,       # row-strided view: the group's contiguous K-slab
                       out=out[:, g * (Cout // G):(g + 1) * (Cout // G)], w_split=sp)
        return out.view(B, H, W, Cout).permute(0, 3, 1, 2)

    def forward_cl(self, x_cl):
        """channels-last [BN, H, W, 1, C] -> [BN, H, W, 1, Cout]; conv_offset runs on the implicit-GEMM kernel"""
        from . import fused
        ops = get_ops()
        B, H, W, _, C = x_cl.shape
        k, pad, G = self.k, self.padding, self.groups
        offset = fused.conv_bn(x_cl, self.conv_offset).view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        col = ops.deform_im2col(x_cl.reshape(B, H, W, C), offset, k, 1, pad, 1, G, self.deform_groups)
        Cout = self.weight.shape[0]
        out = torch.empty((B * H * W, Cout), dtype=x_cl.dtype, device=x_cl.device)
        for g, (wg, sp) in enumerate(self._group_weights()):
            ops.linear(col[:, g].flatten(1), wg, out=out[:, g * (Cout // G):(g + 1) * (Cout // G)], w_split=sp)
        return out.view(B, H, W, 1, Cout)


class DepthNet(nn.Module):
    """ViewTransformerLSSBEVDepth.py:450-504.  Eval mode runs channels-last on the
    library's implicit-GEMM / split-K kernels with the BatchNorms folded in and the DCN on csrc/dcn.hip; only
    the [BN, C] camera-MLP vectors stay on ATen.  Training mode keeps the plain nn.Module graph."""

    def __init__(self, in_channels, mid_channels, context_channels, depth_channels, cam_channels=27):
        super().__init__()
        self.reduce_conv = nn.Sequential(nn.Conv2d(in_channels, mid_channels, 3, padding=1),
                                         nn.BatchNorm2d(mid_channels), nn.ReLU(inplace=True))
        self.context_conv = nn.Conv2d(mid_channels, context_channels, 1)
        self.bn = nn.BatchNorm1d(cam_channels)
        self.depth_mlp = _CamMlp(cam_channels, mid_channels, mid_channels)
        self.depth_se = _SE(mid_channels)
        self.context_mlp = _CamMlp(cam_channels, mid_channels, mid_channels)
        self.context_se = _SE(mid_channels)
        self.depth_conv = nn.Sequential(
            _BasicBlock(mid_channels), _BasicBlock(mid_channels), _BasicBlock(mid_channels),
            _ImageASPP(mid_channels, mid_channels),
            DeformConv2dPack(mid_channels, mid_channels, 3, 1, groups=4),
            nn.Conv2d(mid_channels, depth_channels, 1))

    def forward(self, x, mlp_input):
        """x [BN, C, H, W] -> [BN, D + Cctx, H, W].  Eval mode: channels-last, every convolution on the
        implicit-GEMM kernels with its BatchNorm folded in; training mode keeps the nn.Module graph."""
        if self.training:
            m = mlp_input.reshape(-1, mlp_input.shape[-1])
            if m.shape[0] > 1 or is_synced(self.bn):
                m = self.bn(m)
            else:
                # one camera vector per rank (SemanticKITTI: batch 1, one camera): batch statistics do not exist;
                # the reference gets them from SyncBatchNorm over the 8 ranks (tools/train.py:221-223; here opt-in:
                # dist_utils.convert_sync_batchnorm) -- with per-rank statistics (the default: the only collective kept
                # is the gradient all-reduce) this layer uses its running statistics
                m = F.batch_norm(m, self.bn.running_mean, self.bn.running_var, self.bn.weight, self.bn.bias, False,
                                 0.0, self.bn.eps)
            x = _relu(self.reduce_conv[1](_conv(self, self.reduce_conv[0], x)))                   # reduce_conv[2] = nn.ReLU
            ctx = _conv(self, self.context_conv, self.context_se(x, self.context_mlp(m)[..., None, None]))
            depth = self.depth_se(x, self.depth_mlp(m)[..., None, None])
            for layer in self.depth_conv:
                depth = _conv(self, layer, depth) if isinstance(layer, nn.Conv2d) else layer(depth)
            return torch.cat((depth, ctx), 1)
        from . import fused
        m = self.bn(mlp_input.reshape(-1, mlp_input.shape[-1]))
        x_cl = x.permute(0, 2, 3, 1).unsqueeze(3).contiguous()                      # [BN, H, W, 1, C]
        x_cl = fused.conv_bn(x_cl, self.reduce_conv[0], self.reduce_conv[1], act=1)
        ctx = fused.conv_bn(self.context_se.forward_cl(x_cl, self.context_mlp(m)[..., None, None]), self.context_conv)
        d = self.depth_se.forward_cl(x_cl, self.depth_mlp(m)[..., None, None])
        for blk in self.depth_conv[:3]:
            d = blk.forward_cl(d)
        d = self.depth_conv[3].forward_cl(d)
        d = self.depth_conv[4].forward_cl(d)
        d = fused.conv_bn(d, self.depth_conv[5])
        y = torch.cat((d, ctx), -1)                                                  # [BN, H, W, 1, D + Cctx]
        return y.squeeze(3).permute(0, 3, 1, 2)                                      # logical NCHW view


# ------------------------------------------------------------------ the registered module
@NECKS.register_module()
class ViewTransformerLiftSplatShootVoxel(nn.Module):
    def __init__(self, loss_depth_weight, grid_config=None, data_config=None, numC_input=512,
                 numC_Trans=64, downsample=16, cam_channels=27, point_cloud_range=None,
                 loss_depth_type="bce", loss_depth_reg_weight=0.0, use_voxel_net=False,
                 accelerate=False, use_bev_pool=True, vp_megvii=False, vp_stero=False, **kwargs):
        super().__init__()
        if grid_config is None:
            grid_config = dict(xbound=[-51.2, 51.2, 0.8], ybound=[-51.2, 51.2, 0.8],
                               zbound=[-10.0, 10.0, 20.0], dbound=[1.0, 60.0, 1.0])
        if vp_megvii or use_voxel_net:
            raise NotImplementedError("only the bev_pool voxel path of the OccFormer configs is built")
        self.grid_config = grid_config
        self.data_config = data_config or dict(input_size=(256, 704))
        self.downsample = downsample
        self.loss_depth_weight = loss_depth_weight
        self.loss_depth_type = loss_depth_type
        self.cam_channels = cam_channels
        self.point_cloud_range = point_cloud_range
        self.cam_depth_range = grid_config["dbound"]
        rows = [grid_config["xbound"], grid_config["ybound"], grid_config["zbound"]]
        # float nn.Parameters, as in the reference: they are part of released checkpoints
        self.dx = nn.Parameter(torch.tensor([r[2] for r in rows], dtype=torch.float32), requires_grad=False)
        self.bx = nn.Parameter(torch.tensor([r[0] + r[2] / 2.0 for r in rows], dtype=torch.float32),
                               requires_grad=False)
        self.nx = nn.Parameter(torch.tensor([(r[1] - r[0]) / r[2] for r in rows], dtype=torch.float32),
                               requires_grad=False)
        self.grid_size = tuple(int(round((r[1] - r[0]) / r[2])) for r in rows)
        H, W = self.data_config["input_size"]
        fH, fW = H // downsample, W // downsample
        d = torch.arange(*grid_config["dbound"], dtype=torch.float32)
        self.D = d.numel()
        u = torch.linspace(0, W - 1, fW, dtype=torch.float32).view(1, 1, fW).expand(self.D, fH, fW)
        v = torch.linspace(0, H - 1, fH, dtype=torch.float32).view(1, fH, 1).expand(self.D, fH, fW)
        self.frustum = nn.Parameter(torch.stack((u, v, d.view(-1, 1, 1).expand(self.D, fH, fW)), -1),
                                    requires_grad=False)
        self.numC_input = numC_input
        self.numC_Trans = numC_Trans
        self.depth_net = DepthNet(numC_input, numC_input, numC_Trans, self.D, cam_channels=cam_channels)

    # -- ViewTransformerLSSBEVDepth.py:591-646
    def get_mlp_input(self, rot, tran, intrin, post_rot, post_tran, bda=None):
        B, N = rot.shape[:2]
        if bda is None:
            bda = torch.eye(3).to(rot).view(1, 3, 3).repeat(B, 1, 1)
        bda_n = bda.view(B, 1, *bda.shape[-2:]).expand(B, N, *bda.shape[-2:])
        cols = [intrin[..., 0, 0], intrin[..., 1, 1], intrin[..., 0, 2], intrin[..., 1, 2]]
        if intrin.shape[-1] == 4:
            cols += [intrin[..., 0, 3], intrin[..., 1, 3], intrin[..., 2, 3]]
        cols += [post_rot[..., 0, 0], post_rot[..., 0, 1], post_tran[..., 0], post_rot[..., 1, 0],
                 post_rot[..., 1, 1], post_tran[..., 1], bda_n[..., 0, 0], bda_n[..., 0, 1],
                 bda_n[..., 1, 0], bda_n[..., 1, 1], bda_n[..., 2, 2]]
        feats = torch.stack(cols, -1)
        if intrin.shape[-1] == 4 and bda.shape[-1] == 4:
            feats = torch.cat((feats, bda_n[..., :3, 3]), -1)
        sensor2ego = torch.cat((rot, tran.reshape(B, N, 3, 1)), -1).reshape(B, N, -1)
        return torch.cat((feats, sensor2ego), -1)

    def get_depth_dist(self, x):
        return x.softmax(dim=1)

    def voxel_index(self, rots, trans, intrins, post_rots, post_trans, bda):
        """(vox [B*N*D*fH*fW] i32, offsets, sorted_pts): frustum -> voxel rows + CSR."""
        B, N = trans.shape[:2]
        X, Y, Z = self.grid_size
        cam, bda12 = pack_cameras(rots, trans, intrins, post_rots, post_trans, bda)
        grid = torch.cat((self.bx - self.dx / 2.0, self.dx, self.nx)).float()
        vox = get_ops().lss_voxel_index(self.frustum.reshape(-1, 3), cam, bda12, grid, B, N, X, Y, Z,
                                        bda.shape[-1] == 4)
        offsets, pts = build_voxel_csr(vox, B * X * Y * Z)
        return vox, offsets, pts

    def forward(self, input):
        x, rots, trans, intrins, post_rots, post_trans, bda, mlp_input = input[:8]
        B, N, C, H, W = x.shape
        if (H, W) != tuple(self.frustum.shape[1:3]):
            # the reference fails with a shape error here; the splat kernels index B*N*D*fH*fW points unchecked
            raise ValueError(f"image features are {H}x{W} but the frustum (data_config.input_size // downsample) "
                             f"is {tuple(self.frustum.shape[1:3])}")
        y = self.depth_net(x.view(B * N, C, H, W), mlp_input)
        depth_prob = self.get_depth_dist(y[:, :self.D])
        feat_cl = y[:, self.D:self.D + self.numC_Trans].permute(0, 2, 3, 1).reshape(B * N, H * W, -1)
        vox, offsets, pts = self.voxel_index(rots, trans, intrins, post_rots, post_trans, bda)
        X, Y, Z = self.grid_size
        out = _LiftSplat.apply(depth_prob.reshape(B * N, self.D, H * W).contiguous(),
                               feat_cl.contiguous(), vox, offsets, pts, B * X * Y * Z)
        # logical [B, C, X, Y, Z] over channels-last memory (what the 3-D encoder consumes)
        return out.view(B, X, Y, Z, self.numC_Trans).permute(0, 4, 1, 2, 3), depth_prob

    # -- ViewTransformerLSSVoxel.py:27-75
    def get_downsampled_gt_depth(self, gt_depths):
        B, N, H, W = gt_depths.shape
        ds = self.downsample
        g = gt_depths.view(B * N, H // ds, ds, W // ds, ds).permute(0, 1, 3, 2, 4).reshape(-1, ds * ds)
        g = torch.where(g == 0.0, torch.full_like(g, 1e5), g).min(-1).values
        g = g.view(B * N, H // ds, W // ds)
        db = self.grid_config["dbound"]
        g = (g - (db[0] - db[2] / 2)) / db[2]
        vals = g.clone()
        g = torch.where((g < self.D + 1) & (g >= 0.0), g, torch.zeros_like(g))
        onehot = F.one_hot(g.long(), num_classes=self.D + 1).view(-1, self.D + 1)[:, 1:]
        return vals, onehot.float()

    def get_depth_loss(self, depth_labels, depth_preds):
        if self.loss_depth_type != "bce":
            raise NotImplementedError(self.loss_depth_type)
        _, labels = self.get_downsampled_gt_depth(depth_labels)
        preds = depth_preds.float().permute(0, 2, 3, 1).reshape(-1, self.D)
        # The reference selects the foreground rows (``pred[fg_mask]``, ViewTransformerLSSVoxel.py:60-63) and divides by
        # ``max(1.0, fg_mask.sum())``: a boolean selection and a Python ``max`` on a device tensor, i.e. TWO host
        # synchronisations in the middle of the step -- the host then waits for everything queued so far (the whole view
        # transformer + encoder; measured r03d: 46 of the 145 ms the host needs to issue a training step) and the GPU
        # idles while the host catches up.  Same value without a data-dependent shape: the rows outside the mask
        # contribute exactly 0 (BCE is finite: torch clamps the logs at -100).
        fg = labels.max(1).values > 0.0
        # a NaN / Inf prediction in a row the reference never looks at must stay out of the loss AND out of every
        # gradient: the rows outside the mask are replaced by a constant BEFORE the BCE (a select after it would keep the
        # forward value clean but its backward is 0 * (p - t) / (p (1 - p)) = NaN for a NaN p -- ADVICE r4), so exactly
        # zero gradient reaches them, as with the reference's row selection
        preds = torch.where(fg[:, None], preds, preds.new_full((), 0.5))
        bce = F.binary_cross_entropy(preds, labels, reduction="none")
        loss = torch.where(fg, bce.sum(1), bce.new_zeros(())).sum() / fg.sum().to(preds.dtype).clamp_min(1.0)
        return self.loss_depth_weight * loss
