"""Detector glue, registry names ``OccupancyFormer`` (+ the 2-D image branch ``ResNet`` (optionally with DCNv2
stages) / ``SECONDFPN`` so the unchanged nuScenes configs build; ``CustomEfficientNet`` of the SemanticKITTI
configs lives in efficientnet.py).

Host-side mirror of projects/mmdet3d_plugin/occformer/detectors/occupancyformer.py
(:14-254) on top of BEVDet.__init__ (detectors/bevdepth.py:16-34) and
MVXTwoStageDetector.__init__ (mmdet3d/models/detectors/mvx_two_stage.py:23-70): same
constructor keys, sub-module attribute names, ``forward(return_loss=...)`` dispatch and
``simple_test`` output dict.  The image branch is dense 2-D convolution and stays on
PyTorch-ROCm/MIOpen (outside the hand-written kernel scope).
"""
import collections
import contextlib
import os
import time
import weakref

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .registry import BACKBONES, DETECTORS, MODELS, NECKS


# ------------------------------------------------------------------ 2-D image branch (non-hot-path)
class ModulatedDeformConv2dPack(nn.Module):
    """DCNv2 of the R101-DCN config's image backbone (``dcn=dict(type='DCNv2', deform_groups=1, ...)`` in
    occformer_nusc_r101_896x1600.py:78; mmcv.ops.ModulatedDeformConv2dPack, restated -- mmcv itself is not
    available offline): ``conv_offset`` (zero-initialised, with bias) predicts per output position and deform
    group 2*k*k offsets -- channel 2t = dy, 2t+1 = dx of tap t, after the op's chunk(3) / cat of the first two
    thirds -- and k*k modulation logits; tap t samples the input bilinearly (zero outside the image) at its
    regular position + offset, the sample is scaled by sigmoid(logit), then the ordinary weighted sum over taps and
    input channels.  GPU tensors without autograd run the hand-written bilinear im2col (csrc/dcn.hip, with the
    modulation) + the split-bf16 MFMA GEMM; the grid_sample formulation below is the differentiable / CPU statement
    of the same op."""

    def __init__(self, cin, cout, kernel_size=3, stride=1, padding=1, dilation=1, groups=1, deform_groups=1,
                 bias=False):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("grouped DCNv2 is not used by the OccFormer configs")
        self.k, self.stride, self.padding, self.dilation, self.dg = kernel_size, stride, padding, dilation, deform_groups
        self.force_hip = False          # tests: route CPU tensors through the bound library (the host emulation)
        self.weight = nn.Parameter(torch.empty(cout, cin, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        nn.init.kaiming_uniform_(self.weight, nonlinearity="relu")
        self.conv_offset = nn.Conv2d(cin, deform_groups * 3 * kernel_size * kernel_size, kernel_size, stride=stride,
                                     padding=padding, dilation=dilation, bias=True)
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def _forward_hip(self, x, offset, mask):
        """csrc/dcn.hip: x [B, C, H, W] fp32 on the GPU -> [B, Cout, Ho, Wo]"""
        from . import fused
        from .ops import get_ops
        ops = get_ops()
        B, C, H, W = x.shape
        Ho, Wo = offset.shape[-2:]
        col = ops.deform_im2col(x.permute(0, 2, 3, 1).contiguous(), offset.contiguous(), self.k, self.stride,
                                self.padding, self.dilation, 1, self.dg, mask=mask.contiguous())
        tap = lambda w: w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()      # k = tap * C + c
        w2 = fused._versioned(fused._TAP_CACHE, self.weight, lambda: tap(self.weight.detach()))
        out = ops.linear(col.flatten(1), w2, None if self.bias is None else self.bias.detach(),
                         w_split=fused.split_weight(self.weight, tap))
        return out.view(B, Ho, Wo, -1).permute(0, 3, 1, 2)

    # tests only: a callable (module, x, offset, mask) -> out that stands in for CPU tensors (oracle/dcn_ref.py: the
    # grid_sample statement of the op, installed by tests/conftest.py).  The product itself has no CPU path.
    cpu_reference = None

    def forward(self, x):
        B, C, H, W = x.shape
        k, dg = self.k, self.dg
        # a reduced-precision image branch (bench.py --image-dtype bf16, torch.autocast) crosses this layer as an fp32
        # ISLAND, and the island starts HERE: the sampling positions are fp32 quantities (a bf16 offset has 3
        # fractional bits at 8 pixels), so the offset convolution and the sigmoid run outside autocast on the fp32 input
        # (under autocast they would already return bf16 -- ADVICE r4), the gather / modulation / contraction on the
        # fp32 kernels, and the result returns in the caller's dtype: casts of [B, C, H, W] instead of a second
        # implementation
        dt = x.dtype
        autocast = x.is_cuda and torch.is_autocast_enabled()
        if dt != torch.float32:
            x = x.float()
        with torch.autocast("cuda", enabled=False) if autocast else contextlib.nullcontext():
            o1, o2, logit = torch.chunk(self.conv_offset(x), 3, dim=1)
            offset = torch.cat((o1, o2), 1).float()                      # [B, dg * 2 * k*k, Ho, Wo]
            mask = torch.sigmoid(logit).float()                          # [B, dg * k*k, Ho, Wo]
        if not (x.is_cuda or self.force_hip):
            if ModulatedDeformConv2dPack.cpu_reference is None:
                raise RuntimeError("ModulatedDeformConv2dPack: the deformable sampling runs on the library's kernels "
                                   "(csrc/dcn.hip); there is no CPU path")
            return ModulatedDeformConv2dPack.cpu_reference(self, x, offset, mask)
        if (C // dg) % 4 != 0 or self.dilation != 1:
            raise NotImplementedError(f"ModulatedDeformConv2dPack: {C // dg} channels per deform group / dilation "
                                      f"{self.dilation} (the kernels take multiples of 4 channels, dilation 1)")
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad):
            # training: im2col / col2im(+coord, +mask) kernels and the split-bf16 contractions as one autograd node
            # (occf_modulated_deform_col2im); conv_offset and the sigmoid stay on ATen autograd
            from . import autograd as A
            out = A.DeformConv.apply(x.permute(0, 2, 3, 1), offset, self.weight, k, self.padding, 1, dg, mask,
                                     self.stride)
            out = out.view(B, offset.shape[-2], offset.shape[-1], -1).permute(0, 3, 1, 2)
            out = out if self.bias is None else out + self.bias.view(1, -1, 1, 1)
        else:
            out = self._forward_hip(x, offset, mask)
        # (under autocast an fp32 input means the caller's ops produce the autocast dtype: hand that back)
        dt = torch.get_autocast_dtype("cuda") if autocast and dt == torch.float32 else dt
        return out if dt == torch.float32 else out.to(dt)


# ---- inference route of the image branch: convolution (MIOpen) + ONE fused epilogue pass per convolution ------------
# The reference's modules run conv, BatchNorm, ReLU (and the Bottleneck's identity add) as separate passes over the
# feature map (7 elementwise passes per Bottleneck); in eval mode without a graph the BatchNorm is an affine map per
# channel, so the pass after each convolution is csrc/image_epilogue.hip: y = relu(y * scale + shift (+ identity)) in
# place -- 3 passes per Bottleneck -- and the convolution weights are kept in the branch's dtype / channels_last (under
# autocast every call re-cast them).  OPT-IN (OCCF_IMAGE_FUSE=1): r05c measured the R50 branch at 5.3 ms on this route
# against 4.6 ms module by module (MIOpen picks other solvers for the bias-free bf16 convolutions outside autocast);
# training always takes the module route.
_IMAGE_FUSE = os.environ.get("OCCF_IMAGE_FUSE", "0") == "1"
_BN_AFFINE = {}
_CONV_W = {}


def _fused_eval(x, bn):
    return _IMAGE_FUSE and x.is_cuda and not bn.training and not torch.is_grad_enabled() and x.dim() == 4


def _bn_affine(bn):
    from . import fused
    key = (fused.param_version(bn.weight, bn.bias), bn.running_mean._version, bn.running_var._version,
           bn.running_mean.data_ptr())
    hit = _BN_AFFINE.get(id(bn))
    if hit is None or hit[0] != key or hit[3]() is not bn:
        scale = (bn.weight.detach().float() * torch.rsqrt(bn.running_var.float() + bn.eps)).contiguous()
        shift = (bn.bias.detach().float() - bn.running_mean.float() * scale).contiguous()
        hit = _BN_AFFINE[id(bn)] = (key, scale, shift, weakref.ref(bn))
    return hit[1], hit[2]


def _conv_weight(conv, dtype):
    from . import fused
    key = (fused.param_version(conv.weight), dtype)
    hit = _CONV_W.get(id(conv))
    if hit is None or hit[0] != key or hit[2]() is not conv:
        w = conv.weight.detach().to(dtype).contiguous(memory_format=torch.channels_last)
        hit = _CONV_W[id(conv)] = (key, w, weakref.ref(conv))
    return hit[1]


def conv_bn_act(x, conv, bn, relu=True, residual=None):
    """relu?(bn(conv(x)) (+ residual)).  Eval mode on the GPU without a graph: MIOpen convolution + one fused epilogue
    pass (see above); otherwise the reference's module sequence."""
    if not _fused_eval(x, bn):
        y = bn(conv(x))
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    from .ops import get_ops
    scale, shift = _bn_affine(bn)
    if isinstance(conv, ModulatedDeformConv2dPack):
        y = conv(x)                                      # (its own fp32 island; returns the branch's dtype)
    else:
        if torch.is_autocast_enabled():
            x = x.to(torch.get_autocast_dtype("cuda"))
        x = x.contiguous(memory_format=torch.channels_last)
        w = _conv_weight(conv, x.dtype)
        with torch.autocast("cuda", enabled=False):
            if isinstance(conv, nn.ConvTranspose2d):
                y = F.conv_transpose2d(x, w, None, conv.stride, conv.padding, conv.output_padding, conv.groups,
                                       conv.dilation)
            else:
                y = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation, conv.groups)
        if conv.bias is not None:
            shift = shift + conv.bias.detach().float() * scale
    y = y.contiguous(memory_format=torch.channels_last)
    if residual is not None:
        residual = residual.to(y.dtype).contiguous(memory_format=torch.channels_last)
    return get_ops().scale_shift_act(y, scale, shift, residual, relu)


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride=1, downsample=None, style="pytorch", dcn=None):
        super().__init__()
        s1, s2 = (1, stride) if style == "pytorch" else (stride, 1)
        self.conv1 = nn.Conv2d(cin, planes, 1, stride=s1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        if dcn is not None:
            self.conv2 = ModulatedDeformConv2dPack(planes, planes, 3, stride=s2, padding=1,
                                                   deform_groups=dcn.get("deform_groups", 1), bias=False)
        else:
            self.conv2 = nn.Conv2d(planes, planes, 3, stride=s2, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = downsample

    def forward(self, x):
        y = conv_bn_act(x, self.conv1, self.bn1)
        y = conv_bn_act(y, self.conv2, self.bn2)
        ident = x if self.downsample is None else conv_bn_act(x, self.downsample[0], self.downsample[1], relu=False)
        return conv_bn_act(y, self.conv3, self.bn3, residual=ident)


@BACKBONES.register_module()
class ResNet(nn.Module):
    """mmdet 2.14 ResNet-50/101 (Bottleneck, no DCN) with mmdet/torchvision parameter names."""

    arch = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3)}

    def __init__(self, depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=-1, norm_cfg=None,
                 norm_eval=False, style="pytorch", pretrained=None, with_cp=False, dcn=None,
                 stage_with_dcn=None, init_cfg=None, **kwargs):
        super().__init__()
        if dcn is not None and (dcn.get("type") != "DCNv2" or dcn.get("fallback_on_stride", False)):
            raise NotImplementedError("image-backbone dcn: DCNv2 without fallback_on_stride (the R101-DCN config)")
        stage_with_dcn = tuple(stage_with_dcn) if stage_with_dcn is not None else (False,) * num_stages
        self.out_indices = tuple(out_indices)
        self.norm_eval = norm_eval
        self.frozen_stages = frozen_stages
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for i, n in enumerate(self.arch[depth][:num_stages]):
            planes, stride = 64 * 2 ** i, 1 if i == 0 else 2
            blocks = []
            for j in range(n):
                ds = None
                if j == 0 and (stride != 1 or cin != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                blocks.append(_Bottleneck(cin, planes, stride if j == 0 else 1, ds, style,
                                          dcn if dcn is not None and stage_with_dcn[i] else None))
                cin = planes * 4
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        self.num_stages = num_stages

    def train(self, mode=True):
        """mmdet ResNet.train: stem + the first ``frozen_stages`` stages stay in eval mode without gradients;
        ``norm_eval`` keeps every BatchNorm on its running statistics"""
        super().train(mode)
        if self.frozen_stages >= 0:
            for m in [self.conv1, self.bn1] + [getattr(self, f"layer{i}") for i in range(1, self.frozen_stages + 1)]:
                m.eval()
                for p in m.parameters():
                    p.requires_grad = False
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def forward(self, x):
        x = F.max_pool2d(conv_bn_act(x, self.conv1, self.bn1), 3, stride=2, padding=1)
        outs = []
        for i in range(self.num_stages):
            x = getattr(self, f"layer{i + 1}")(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)


@NECKS.register_module()
class SECONDFPN(nn.Module):
    """mmdet3d/models/necks/second_fpn.py:11-91."""

    def __init__(self, in_channels=(128, 128, 256), out_channels=(256, 256, 256), upsample_strides=(1, 2, 4),
                 norm_cfg=None, upsample_cfg=None, conv_cfg=None, use_conv_for_no_stride=False, init_cfg=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="BN", eps=1e-3, momentum=0.01)
        blocks = []
        for cin, cout, s in zip(in_channels, out_channels, upsample_strides):
            if s > 1 or (s == 1 and not use_conv_for_no_stride):
                s = int(s)
                up = nn.ConvTranspose2d(cin, cout, s, stride=s, bias=False)
            else:
                k = int(np.round(1 / s))
                up = nn.Conv2d(cin, cout, k, stride=k, bias=False)
            blocks.append(nn.Sequential(up, nn.BatchNorm2d(cout, eps=norm_cfg.get("eps", 1e-3),
                                                           momentum=norm_cfg.get("momentum", 0.01)),
                                        nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(blocks)

    def forward(self, x):
        ups = [conv_bn_act(x[i], blk[0], blk[1]) for i, blk in enumerate(self.deblocks)]       # blk[2] = nn.ReLU
        return [torch.cat(ups, 1) if len(ups) > 1 else ups[0]]


# ------------------------------------------------------------------ the detector
@DETECTORS.register_module()
class OccupancyFormer(nn.Module):
    STAGES = ("img_encoder", "view_transformer", "bev_encoder", "bev_neck", "mask2former_head")

    def __init__(self, img_backbone=None, img_neck=None, img_view_transformer=None,
                 img_bev_encoder_backbone=None, img_bev_encoder_neck=None, pts_bbox_head=None,
                 train_cfg=None, test_cfg=None, pretrained=None, init_cfg=None, **kwargs):
        super().__init__()
        self.img_backbone = MODELS.build(img_backbone) if img_backbone else None
        self.img_neck = MODELS.build(img_neck) if img_neck else None
        if pts_bbox_head:
            head = dict(pts_bbox_head)
            # mvx_two_stage.py:56-61: the head receives train_cfg.pts / test_cfg.pts
            head["train_cfg"] = train_cfg["pts"] if train_cfg and "pts" in train_cfg else None
            head["test_cfg"] = test_cfg["pts"] if test_cfg and "pts" in test_cfg else None
            self.pts_bbox_head = MODELS.build(head)
        self.img_view_transformer = MODELS.build(img_view_transformer)
        self.img_bev_encoder_backbone = MODELS.build(img_bev_encoder_backbone)
        self.img_bev_encoder_neck = MODELS.build(img_bev_encoder_neck)
        self.train_cfg, self.test_cfg = train_cfg, test_cfg
        # dtype of the 2-D image branch (ResNet / EfficientNet + SECONDFPN on PyTorch-ROCm / MIOpen, SURVEY.md §8f
        # row 3): None = fp32 as the reference runs it; torch.bfloat16 = autocast + channels_last
        self.image_dtype = None
        self.record_time = False
        self.time_stats = collections.defaultdict(list)

    @property
    def with_img_neck(self):
        return self.img_neck is not None

    def _tick(self, name, t0):
        if not self.record_time:
            return t0
        torch.cuda.synchronize()
        t1 = time.time()
        self.time_stats[name].append(t1 - t0)
        return t1

    def image_encoder(self, img):
        # neck features [B, N, C, fH, fW] instead of images [B, N, 3, H, W]: the caller already ran the image branch
        # (a detector without one; or a serving loop that runs ``image_encoder`` of the NEXT frame on a side stream
        # while this frame's 3-D path runs -- bench.py's pipelined from-images record)
        # The signal is explicit (a channel count != 3 would also skip the backbone for 1- or 4-channel
        # images and push a 3-channel feature map through it): this method marks what it returns.
        if self.img_backbone is None or getattr(img, "_occf_neck_features", False):
            return img
        B, N, C, H, W = img.shape
        x = img.view(B * N, C, H, W)
        if self.image_dtype is not None and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
            with torch.autocast("cuda", dtype=self.image_dtype):
                x = self.img_backbone(x)
                if self.with_img_neck:
                    x = self.img_neck(x)
                    if isinstance(x, (list, tuple)):
                        x = x[0]
            x = x.float().contiguous()
        else:
            x = self.img_backbone(x)
            if self.with_img_neck:
                x = self.img_neck(x)
                if isinstance(x, (list, tuple)):
                    x = x[0]
        x = x.view(B, N, *x.shape[1:])
        x._occf_neck_features = True
        return x

    def bev_encoder(self, x):
        t = self._tick("", 0.0) if self.record_time else 0.0
        x = self.img_bev_encoder_backbone(x.float())
        t = self._tick("bev_encoder", t)
        x = self.img_bev_encoder_neck(x)
        self._tick("bev_neck", t)
        return x

    def extract_img_feat(self, img, img_metas=None):
        t = self._tick("", 0.0) if self.record_time else 0.0
        x = self.image_encoder(img[0])
        img_feats = x
        t = self._tick("img_encoder", t)
        rots, trans, intrins, post_rots, post_trans, bda = img[1:7]
        mlp_input = self.img_view_transformer.get_mlp_input(rots, trans, intrins, post_rots, post_trans, bda)
        x, depth = self.img_view_transformer([x, rots, trans, intrins, post_rots, post_trans, bda, mlp_input])
        self._tick("view_transformer", t)
        x = self.bev_encoder(x)
        if not isinstance(x, list):
            x = [x]
        return x, depth, img_feats

    def extract_feat(self, points, img, img_metas):
        voxel_feats, depth, img_feats = self.extract_img_feat(img, img_metas)
        return voxel_feats, img_feats, depth

    def forward(self, return_loss=True, **kwargs):
        """mmdet3d/models/detectors/base.py:46-61."""
        if return_loss:
            return self.forward_train(**kwargs)
        return self.forward_test(**kwargs)

    def forward_train(self, points=None, img_metas=None, img_inputs=None, gt_occ=None, points_occ=None,
                      **kwargs):
        """occupancyformer.py:132-199: depth BCE + the head's Hungarian losses.  In ``train()`` mode every module of
        the path runs as a graph of forward / backward kernel pairs (occformer_amd/autograd.py), so the returned
        losses carry ``grad_fn`` and ``sum(losses).backward()`` produces every parameter gradient -- the reference's
        training step.  In ``eval()`` mode the fused inference kernels run and the losses are plain values."""
        from . import fused
        fused.invalidate_caches()       # a new step: weight layouts / bf16 splits are rebuilt once (fused.py, _EPOCH)
        # ground-truth conversion first: the number of labels present is read back to the host (a synchronisation) --
        # free when ``prefetch_gt`` ran the scan ahead of time on the side stream, cheap while the device queue is still
        # short, and as expensive as the whole forward once extract_feat has been queued
        if "gt_prepared" not in kwargs and hasattr(self.pts_bbox_head, "preprocess_gt"):
            kwargs["gt_prepared"] = self.pts_bbox_head.preprocess_gt(gt_occ, img_metas, scans=self._take_gt_scans(gt_occ),
                                                                     mask_dtype=torch.float32)
        voxel_feats, img_feats, depth = self.extract_feat(points=None, img=img_inputs, img_metas=img_metas)
        losses = {"loss_depth": self.img_view_transformer.get_depth_loss(img_inputs[7], depth)}
        losses.update(self.pts_bbox_head.forward_train(voxel_feats=voxel_feats, img_metas=img_metas, gt_occ=gt_occ,
                                                       points=points_occ, img_feats=img_feats, **kwargs))
        return losses

    # ---- ground-truth label scan one step ahead, on a side stream -------------------------------------------------
    def prefetch_gt(self, gt_occ, ready=None):
        """Start the label scan of a FUTURE training sample (``gt_occ`` [B, X, Y, Z] on the device) on a side HIP
        stream and return immediately.  ``forward_train`` of that sample then finds the per-sample label counts in
        pinned host memory instead of synchronising with everything the main stream still has queued (the previous
        step's backward and optimizer) -- with the scan in flight one step ahead the host issues step n + 1 while the
        device executes step n (r03e: the host needs ~100 ms to issue a step the device runs in ~140 ms; with a
        synchronisation at the start of every step the two serialise wherever the host is the slower one).
        What a prefetching data loader calls right after it has the next batch on the device.
        ``ready``: None = the scan waits (on the device) for the main stream's current position; a ``torch.cuda.Event``
        = it waits for that event; True = ``gt_occ`` is already complete (resident inputs)."""
        if not gt_occ.is_cuda:
            return
        from .training import gt_label_scan
        dev = gt_occ.device
        side = self.__dict__.get("_gt_stream")
        if side is None or side.device != dev:
            side = self.__dict__["_gt_stream"] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        if ready is None:
            ready = torch.cuda.Event()
            ready.record(main)
        if ready is not True:
            side.wait_event(ready)
        nc = self.pts_bbox_head.num_occupancy_classes
        with torch.cuda.stream(side):
            scans = [gt_label_scan(g, nc) for g in gt_occ]
            counts = torch.stack([s[1] for s in scans]).to(torch.int64)
            host = torch.empty(counts.shape, dtype=torch.int64, pin_memory=True)
            host.copy_(counts, non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        for s in scans:
            s[0].record_stream(main)            # allocated on the side stream, consumed on the main stream
        gt_occ.record_stream(side)              # read on the side stream: its block must not be recycled under the scan
        # an entry belongs to the TENSOR OBJECT it was computed from (weak reference), never to its address: the caching
        # allocator hands the next same-shaped ground truth the address (and version 0) of a freed one, and an entry
        # whose sample never reached forward_train would then label another sample
        q = self.__dict__.setdefault("_gt_prefetched", [])
        q[:] = [e for e in q if e[0]() is not None and e[0]() is not gt_occ]
        q.append((weakref.ref(gt_occ), gt_occ._version, [s[0] for s in scans], host, done))
        del q[:-4]

    def _take_gt_scans(self, gt_occ):
        q = self.__dict__.get("_gt_prefetched")
        if not q or not torch.is_tensor(gt_occ):
            return None
        hit = None
        for i, e in enumerate(q):
            if e[0]() is gt_occ:
                hit = q.pop(i)
                break
        q[:] = [e for e in q if e[0]() is not None]          # samples that were dropped before their step
        if hit is None or hit[1] != gt_occ._version:         # (written to since the scan: scan again)
            return None
        _, _, labels, host, done = hit
        done.synchronize()                                   # long complete when the scan ran a step ahead
        torch.cuda.current_stream(gt_occ.device).wait_event(done)
        return [(labels[i], int(host[i])) for i in range(len(labels))]

    def forward_test(self, img_metas=None, img_inputs=None, **kwargs):
        return self.simple_test(img_metas, img_inputs, **kwargs)

    def simple_test(self, img_metas, img=None, rescale=False, points_occ=None, gt_occ=None, points_uv=None):
        voxel_feats, img_feats, depth = self.extract_feat(points=None, img=img, img_metas=img_metas)
        t = self._tick("", 0.0) if self.record_time else 0.0
        output = self.pts_bbox_head.simple_test(voxel_feats=voxel_feats, points=points_occ,
                                                img_metas=img_metas, img_feats=img_feats,
                                                points_uv=points_uv)
        self._tick("mask2former_head", t)
        if output["output_points"] is not None and points_occ is not None:
            output["output_points"] = torch.argmax(output["output_points"][:, 1:], dim=1) + 1
            target = torch.cat(points_occ, dim=0)
            output["evaluation_semantic"] = self.simple_evaluation_semantic(output["output_points"], target)
            output["target_points"] = target
        vox = output["output_voxels"][0]
        occ = tuple(int(v) for v in img_metas[0]["occ_size"])
        if tuple(vox.shape[-3:]) != occ:
            vox = F.interpolate(vox, size=occ, mode="trilinear", align_corners=True)
        output["output_voxels"] = vox
        output["target_voxels"] = gt_occ
        return output

    @staticmethod
    def simple_evaluation_semantic(pred, gt, n=16):
        """P/utils/metric_util.py:8-22 (fast_hist_crop over labels 1..16)."""
        pred = pred.cpu().numpy().astype(np.int64)
        gt = gt.cpu().numpy()[:, 3].astype(np.int64)
        k = (gt >= 0) & (gt < n + 1)      # labels 0..16, row/col 0 cropped below
        hist = np.bincount((n + 1) * gt[k] + pred[k], minlength=(n + 1) ** 2).reshape(n + 1, n + 1)
        return hist[1:, 1:]


@DETECTORS.register_module()
class OccupancyFormer4D(OccupancyFormer):
    """occupancyformer.py:256-313: the two-frame detector.  ``img_inputs[0]`` holds 2 frames per camera
    ([B, 2N, C, H, W], viewed [B, N, 2, ...]: frame index innermost), the calibration tensors 2N entries viewed
    [B, 2, N, ...] (frame index outermost) -- the reference's layouts, kept.  Each frame is lifted on its own with the KEY
    frame's extrinsics in the DepthNet camera vector; the previous frame runs without a graph; the two voxel volumes are
    concatenated along the channels (``img_bev_encoder_backbone.in_channels = 2 * numC_Trans``); depth and image
    features of the key frame are returned."""

    def prepare_voxel_feat(self, img, rot, tran, intrin, post_rot, post_tran, bda, mlp_input):
        x = self.image_encoder(img)
        voxel_feat, depth = self.img_view_transformer([x, rot, tran, intrin, post_rot, post_tran, bda, mlp_input])
        return voxel_feat, depth, x

    def extract_img_feat(self, img, img_metas=None):
        B, N2, C, H, W = img[0].shape
        N = N2 // 2
        frames = img[0].view(B, N, 2, C, H, W).unbind(2)
        rots, trans, intrins, post_rots, post_trans, bda = img[1:7]
        per_frame = [t.view(B, 2, N, *t.shape[2:]).unbind(1) for t in (rots, trans, intrins, post_rots, post_trans)]
        rots, trans, intrins, post_rots, post_trans = per_frame
        voxels, depths, feats = [], [], []
        for f in range(2):
            mlp_input = self.img_view_transformer.get_mlp_input(rots[0], trans[0], intrins[f], post_rots[f],
                                                                post_trans[f], bda)
            args = (frames[f], rots[f], trans[f], intrins[f], post_rots[f], post_trans[f], bda, mlp_input)
            if f == 0:                                      # back-propagation through the key frame only
                v, d, x = self.prepare_voxel_feat(*args)
            else:
                with torch.no_grad():
                    v, d, x = self.prepare_voxel_feat(*args)
            voxels.append(v)
            depths.append(d)
            feats.append(x)
        x = self.bev_encoder(torch.cat(voxels, dim=1))
        if not isinstance(x, list):
            x = [x]
        return x, depths[0], feats[0]
