"""Config dicts with the structure and values of the reference's
projects/configs/occformer_nusc/occformer_nusc_r50_256x704.py (:17-205), generated for a
chosen voxel grid.  ``grid='reference'`` is the shipped 128x128x16 LSS grid (256x256x32
output); ``grid='200'`` is BASELINE.json's 200x200x16 synthetic grid
(point_cloud_range [-50,-50,-5,50,50,3], occ_size [400,400,32]).
The reference's own .py configs load through ``occformer_amd.registry.Config.fromfile``.
"""


def nusc_r50(grid="200", with_image_branch=False, num_queries=100, input_size=(256, 704), focal=557.0):
    if grid == "reference":
        pc_range, occ_size = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0], [256, 256, 32]
    elif grid == "200":
        pc_range, occ_size = [-50.0, -50.0, -5.0, 50.0, 50.0, 3.0], [400, 400, 32]
    else:
        raise KeyError(grid)
    ds = [2, 2, 2]
    vs = [(pc_range[3 + i] - pc_range[i]) / occ_size[i] for i in range(3)]
    data_config = dict(cams=["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_LEFT", "CAM_BACK",
                             "CAM_BACK_RIGHT"], Ncams=6, input_size=tuple(input_size), src_size=(900, 1600))
    grid_config = dict(xbound=[pc_range[0], pc_range[3], vs[0] * ds[0]],
                       ybound=[pc_range[1], pc_range[4], vs[1] * ds[1]],
                       zbound=[pc_range[2], pc_range[5], vs[2] * ds[2]], dbound=[2.0, 58.0, 0.5])
    numC_Trans = 128
    ch = [128, 256, 512, 1024]
    E = 192
    norm_cfg = dict(type="GN", num_groups=32, requires_grad=True)
    num_class = 17
    model = dict(
        type="OccupancyFormer",
        img_view_transformer=dict(type="ViewTransformerLiftSplatShootVoxel", loss_depth_weight=1.0,
                                  grid_config=grid_config, data_config=data_config, numC_Trans=numC_Trans,
                                  vp_megvii=False),
        img_bev_encoder_backbone=dict(type="OccupancyEncoder", num_stage=4, in_channels=numC_Trans,
                                      block_numbers=[2, 2, 2, 2], block_inplanes=ch,
                                      block_strides=[1, 2, 2, 2], out_indices=(0, 1, 2, 3), with_cp=True,
                                      norm_cfg=norm_cfg),
        img_bev_encoder_neck=dict(
            type="MSDeformAttnPixelDecoder3D", strides=[2, 4, 8, 16], in_channels=ch, feat_channels=E,
            out_channels=E, norm_cfg=norm_cfg,
            encoder=dict(type="DetrTransformerEncoder", num_layers=6,
                         transformerlayers=dict(
                             type="BaseTransformerLayer",
                             attn_cfgs=dict(type="MultiScaleDeformableAttention3D", embed_dims=E, num_heads=8,
                                            num_levels=3, num_points=4, im2col_step=64, dropout=0.0,
                                            batch_first=False, norm_cfg=None, init_cfg=None),
                             ffn_cfgs=dict(embed_dims=E), feedforward_channels=E * 4, ffn_dropout=0.0,
                             operation_order=("self_attn", "norm", "ffn", "norm")),
                         init_cfg=None),
            positional_encoding=dict(type="SinePositionalEncoding3D", num_feats=E // 3, normalize=True)),
        pts_bbox_head=dict(
            type="Mask2FormerNuscOccHead", feat_channels=E, out_channels=E, num_queries=num_queries,
            num_occupancy_classes=num_class, pooling_attn_mask=True, sample_weight_gamma=0.25,
            positional_encoding=dict(type="SinePositionalEncoding3D", num_feats=E / 3, normalize=True),
            transformer_decoder=dict(
                type="DetrTransformerDecoder", return_intermediate=True, num_layers=9,
                transformerlayers=dict(
                    type="DetrTransformerDecoderLayer",
                    attn_cfgs=dict(type="MultiheadAttention", embed_dims=E, num_heads=E // 32, attn_drop=0.0,
                                   proj_drop=0.0, dropout_layer=None, batch_first=False),
                    ffn_cfgs=dict(embed_dims=E, num_fcs=2, act_cfg=dict(type="ReLU", inplace=True),
                                  ffn_drop=0.0, dropout_layer=None, add_identity=True),
                    feedforward_channels=E * 8,
                    operation_order=("cross_attn", "norm", "self_attn", "norm", "ffn", "norm")),
                init_cfg=None),
            loss_cls=dict(type="CrossEntropyLoss", use_sigmoid=False, loss_weight=2.0, reduction="mean",
                          class_weight=[1.0] * num_class + [0.1]),
            loss_mask=dict(type="CrossEntropyLoss", use_sigmoid=True, reduction="mean", loss_weight=5.0),
            loss_dice=dict(type="DiceLoss", use_sigmoid=True, activate=True, reduction="mean",
                           naive_dice=True, eps=1.0, loss_weight=5.0),
            point_cloud_range=pc_range),
        # occformer_nusc_r50_256x704.py:191-204
        train_cfg=dict(pts=dict(
            num_points=12544 * 4, oversample_ratio=3.0, importance_sample_ratio=0.75,
            assigner=dict(type="MaskHungarianAssigner", cls_cost=dict(type="ClassificationCost", weight=2.0),
                          mask_cost=dict(type="CrossEntropyLossCost", weight=5.0, use_sigmoid=True),
                          dice_cost=dict(type="DiceCost", weight=5.0, pred_act=True, eps=1.0)),
            sampler=dict(type="MaskPseudoSampler"))),
        test_cfg=dict(pts=dict(semantic_on=True, panoptic_on=False, instance_on=False)),
    )
    if with_image_branch:
        model["img_backbone"] = dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                                     frozen_stages=0, norm_cfg=dict(type="BN", requires_grad=True),
                                     norm_eval=False, style="pytorch")
        model["img_neck"] = dict(type="SECONDFPN", in_channels=[256, 512, 1024, 2048],
                                 upsample_strides=[0.25, 0.5, 1, 2], out_channels=[128, 128, 128, 128])
    meta = dict(pc_range=pc_range, occ_size=occ_size, D=112, C=numC_Trans, groups=32, heads=E // 32,
                fH=input_size[0] // 16, fW=input_size[1] // 16, neck_channels=512, ncams=6,
                grid=tuple(o // d for o, d in zip(occ_size, ds)), focal=focal, input_size=tuple(input_size),
                kitti=False, lidar_points=34720)
    return model, meta


def kitti_effb7(lss_downsample=2):
    """projects/configs/occformer_kitti/occformer_kitti.py:22-205 from the neck features on: one camera,
    384x1280 image, 640 neck channels, 33 camera scalars (4x4 intrinsics / BEV augmentation), 20 classes,
    ``Mask2FormerOccHead``.  ``lss_downsample=2`` is the shipped config (LSS / encoder grid 128x128x16, output
    256x256x32: BASELINE config 0); ``lss_downsample=1`` is the literal reading of BASELINE config 1 (encoder grid
    256x256x32 -- supported by the config, never run by the reference)."""
    model, meta = nusc_r50("reference")
    pc_range, occ_size = [0.0, -25.6, -2.0, 51.2, 25.6, 4.4], [256, 256, 32]
    ds = [int(lss_downsample)] * 3
    vs = [(pc_range[3 + i] - pc_range[i]) / occ_size[i] for i in range(3)]
    num_class = 20
    vt = model["img_view_transformer"]
    vt.update(numC_input=640, cam_channels=33,
              grid_config=dict(xbound=[pc_range[0], pc_range[3], vs[0] * ds[0]],
                               ybound=[pc_range[1], pc_range[4], vs[1] * ds[1]],
                               zbound=[pc_range[2], pc_range[5], vs[2] * ds[2]], dbound=[2.0, 58.0, 0.5]),
              data_config=dict(input_size=(384, 1280), resize=(-0.06, 0.11), rot=(-5.4, 5.4), flip=True,
                               crop_h=(0.0, 0.0), resize_test=0.0))
    head = model["pts_bbox_head"]
    head.update(type="Mask2FormerOccHead", num_occupancy_classes=num_class, point_cloud_range=pc_range)
    head["loss_cls"]["class_weight"] = [1.0] * num_class + [0.1]
    meta.update(pc_range=pc_range, occ_size=occ_size, fH=24, fW=80, neck_channels=640, ncams=1,
                grid=tuple(o // d for o, d in zip(occ_size, ds)), focal=738.0, input_size=(384, 1280), kitti=True,
                lidar_points=0)
    return model, meta


WORKLOADS = {
    # BASELINE.json configs[2] (and [3] under DDP): the grid the metric is quoted on
    "nusc_r50_200": lambda: nusc_r50("200"),
    # the reference's own shipped grid for the same config (occformer_nusc_r50_256x704.py:17-46)
    "nusc_r50_ref128": lambda: nusc_r50("reference"),
    # BASELINE.json configs[0]: the shipped SemanticKITTI config (encoder grid 128x128x16)
    "kitti_effb7_128": lambda: kitti_effb7(2),
    # BASELINE.json configs[1] read literally (encoder grid 256x256x32)
    "kitti_effb7_256lit": lambda: kitti_effb7(1),
    # BASELINE.json configs[4]: R101-DCN 896x1600 (fH x fW = 56 x 100, 3.76 M frustum points), 200-grid
    "nusc_r101": lambda: nusc_r50("200", input_size=(896, 1600), focal=1266.0),
}


def image_branch(name):
    """the 2-D image branch of a workload's reference config (img_backbone, img_neck):
    occformer_nusc_r50_256x704.py:60-74, occformer_nusc_r101_896x1600.py:68-85, occformer_kitti.py:66-80 (without the
    pretrained-checkpoint ``init_cfg``: there is no network for checkpoints, weights are random)"""
    if name.startswith("kitti"):
        return (dict(type="CustomEfficientNet", arch="b7", drop_path_rate=0.2, frozen_stages=0, norm_eval=False,
                     out_indices=(2, 3, 4, 5, 6), with_cp=True),
                dict(type="SECONDFPN", in_channels=[48, 80, 224, 640, 2560], upsample_strides=[0.25, 0.5, 1, 2, 2],
                     out_channels=[128, 128, 128, 128, 128]))
    neck = dict(type="SECONDFPN", in_channels=[256, 512, 1024, 2048], upsample_strides=[0.25, 0.5, 1, 2],
                out_channels=[128, 128, 128, 128])
    if name == "nusc_r101":
        return (dict(type="ResNet", depth=101, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=1,
                     norm_cfg=dict(type="BN2d", requires_grad=False), norm_eval=True, style="caffe", with_cp=True,
                     dcn=dict(type="DCNv2", deform_groups=1, fallback_on_stride=False),
                     stage_with_dcn=(False, False, True, True)), neck)
    return (dict(type="ResNet", depth=50, num_stages=4, out_indices=(0, 1, 2, 3), frozen_stages=0,
                 norm_cfg=dict(type="BN", requires_grad=True), norm_eval=False, style="pytorch"), neck)


def workload(name, with_image_branch=False):
    """(model config, meta) of a named bench / parity workload.  All start at the image-neck features unless
    ``with_image_branch``: then the config carries the reference's img_backbone / img_neck and the sample's first input
    is the raw image batch [B, N, 3, H, W]"""
    cfg, meta = WORKLOADS[name]()
    meta["workload"] = name
    if with_image_branch:
        cfg["img_backbone"], cfg["img_neck"] = image_branch(name)
    return cfg, meta


def oracle_cfg(meta):
    """keyword dict of oracle.occformer_ref.occformer_forward for a workload"""
    return dict(D=meta["D"], C=meta["C"], occ_size=meta["occ_size"], pc_range=meta["pc_range"], groups=32)


def oracle_train_cfg(cfg, meta, class_weight=None):
    """keyword dict of oracle.occformer_train_ref.train_step for a workload (``cfg`` = the model config with
    ``train_cfg.pts``).  SemanticKITTI (``Mask2FormerOccHead``): class-guided sampling weights from the class
    frequencies (mask2former_occ.py:96-100) and the head's class weights (``class_weight``: the built head's)."""
    head = cfg["pts_bbox_head"]
    tc = cfg["train_cfg"]["pts"]
    hd = dict(point_cloud_range=head.get("point_cloud_range"), num_points=tc["num_points"],
              oversample_ratio=tc["oversample_ratio"], importance_sample_ratio=tc["importance_sample_ratio"],
              padding_mode="border", num_classes=head["num_occupancy_classes"],
              class_weight=head["loss_cls"]["class_weight"] if class_weight is None else class_weight)
    if head["type"] == "Mask2FormerOccHead":
        import numpy as np
        from .training import semantic_kitti_class_frequencies
        w = 1.0 / np.asarray(semantic_kitti_class_frequencies, dtype=np.float64)
        hd.update(align_corners=True, sample_weights=(w / w.min()) ** head.get("sample_weight_gamma", 0.25))
    enc = cfg["img_bev_encoder_backbone"]
    return dict(D=meta["D"], C=meta["C"], groups=meta.get("groups", 32), heads=meta.get("heads", 6),
                pd_layers=cfg["img_bev_encoder_neck"]["encoder"]["num_layers"],
                dec_layers=head["transformer_decoder"]["num_layers"], downsample=16,
                block_numbers=tuple(enc.get("block_numbers", (2, 2, 2, 2))),
                dbound=cfg["img_view_transformer"]["grid_config"]["dbound"], head=hd)


def synthetic_sample(meta, device, seed=0):
    """seeded neck features, the 6-camera surround rig (or the one forward camera with 4x4
    intrinsics / BEV augmentation of SemanticKITTI), uniform LiDAR points.
    -> (img_inputs [x, rots, trans, intrins, post_rots, post_trans, bda], img_metas, points or None)"""
    import math

    import torch
    g = torch.Generator().manual_seed(seed)
    B, N = 1, meta["ncams"]
    x = torch.randn(B, N, meta["neck_channels"], meta["fH"], meta["fW"], generator=g)
    yaws = [0.0] if N == 1 else [55.0, 0.0, -55.0, 110.0, 180.0, -110.0][:N]
    rots, trans = [], []
    for yaw in yaws:
        a = math.radians(yaw)
        fwd = torch.tensor([math.cos(a), math.sin(a), 0.0])
        right = torch.tensor([math.sin(a), -math.cos(a), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        rots.append(torch.stack((right, down, fwd), 1))
        trans.append(torch.tensor([1.5 * math.cos(a), 1.5 * math.sin(a), 1.5]))
    rots = torch.stack(rots).unsqueeze(0)
    trans = torch.stack(trans).unsqueeze(0)
    H, W = meta["input_size"]
    # principal point 60 px below the top at 256 rows, scaled with the image height
    cy = 188.0 if meta.get("kitti") else 60.0 * H / 256.0
    K = torch.tensor([[meta["focal"], 0.0, W / 2.0], [0.0, meta["focal"], cy], [0.0, 0.0, 1.0]])
    if meta.get("kitti"):
        K4 = torch.eye(4)
        K4[:3, :3] = K
        K4[:3, 3] = torch.tensor([46.9, 0.2, 0.003])
        intr = K4.view(1, 1, 4, 4).repeat(B, N, 1, 1)
        bda = torch.eye(4).view(1, 4, 4).clone()
        bda[0, :3, 3] = torch.tensor([0.1, -0.2, 0.05])
    else:
        intr = K.view(1, 1, 3, 3).repeat(B, N, 1, 1)
        bda = torch.eye(3).view(1, 3, 3)
    post_rots = torch.eye(3).view(1, 1, 3, 3).repeat(B, N, 1, 1)
    post_trans = torch.zeros(B, N, 3)
    if N > 1:
        # Six physical cameras never share their calibration to the last bit, and the image augmentation draws a
        # resize / crop per camera (loading_nusc_imgs.py:98-174), so intrinsics, post_rots and post_trans differ from
        # camera to camera.  An IDENTICAL column matters numerically: DepthNet normalises the 27 camera scalars with a
        # train-mode BatchNorm1d over the cameras (ViewTransformerLSSBEVDepth.py:453,489), and a constant column there is
        # (x - mean) = rounding noise divided by sqrt(eps) -- with fx = 557 repeated six times that is +-0.02 of pure
        # noise in the layer's output, different on every backend (measured r04e: camera-MLP pre-activations 6e-4 apart
        # between GPU and CPU, their parameters' gradients 3e-2 apart).  The rig below keeps the survey's nominal values
        # and perturbs them per camera the way real calibrations / augmentations do.
        fs = torch.tensor([0.985, 0.993, 1.0, 1.004, 1.011, 0.996])[:N]
        dcx = torch.tensor([-3.0, 2.0, 0.0, 4.0, -1.0, 1.5])[:N]
        dcy = torch.tensor([1.0, -2.0, 0.5, 0.0, 2.0, -1.0])[:N]
        intr[0, :, 0, 0] *= fs
        intr[0, :, 1, 1] *= fs
        intr[0, :, 0, 2] += dcx * (W / 704.0)
        intr[0, :, 1, 2] += dcy * (H / 256.0)
        rs = torch.tensor([1.0, 0.97, 1.03, 0.99, 1.02, 0.96])[:N]          # resize of the augmentation, per camera
        post_rots[0, :, 0, 0] = rs
        post_rots[0, :, 1, 1] = rs
        post_trans[0, :, 0] = torch.tensor([0.0, -6.0, 4.0, -2.0, 8.0, -5.0])[:N] * (W / 704.0)
        post_trans[0, :, 1] = torch.tensor([0.0, 3.0, -4.0, 2.0, -1.0, 5.0])[:N] * (H / 256.0)
    points = None
    if meta.get("lidar_points", 0):
        lo = torch.tensor(meta["pc_range"][:3])
        hi = torch.tensor(meta["pc_range"][3:])
        n = meta["lidar_points"]
        pts = torch.rand(n, 3, generator=g) * (hi - lo) + lo
        points = [torch.cat((pts, torch.zeros(n, 1)), 1).to(device)]
    img_inputs = [t.to(device) for t in (x, rots, trans, intr, post_rots, post_trans, bda)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    return img_inputs, metas, points



def synthetic_targets(meta, device, seed=0):
    """ground truth of one synthetic training sample (occupancyformer.py:132-140 inputs): a blocky semantic label
    volume gt_occ [1, X2, Y2, Z2] (classes 1..16 + 0 = free + 255 = unknown for nuScenes; 0..19 + 255 for KITTI),
    LiDAR points with labels [P, 4] and sparse LiDAR depth maps gt_depths [1, N, H, W] (0 = no return)."""
    import torch
    g = torch.Generator().manual_seed(1000 + seed)
    X2, Y2, Z2 = meta["occ_size"]
    ncls = 20 if meta.get("kitti") else 17
    blk = 16
    lab = torch.randint(0, ncls + 2, (1, X2 // blk + 1, Y2 // blk + 1, max(Z2 // 8, 1)), generator=g)
    lab = torch.where(lab >= ncls, torch.full_like(lab, 255 if not meta.get("kitti") else 0), lab)
    gt_occ = lab.repeat_interleave(blk, 1).repeat_interleave(blk, 2).repeat_interleave(8, 3)[:, :X2, :Y2, :Z2]
    points = None
    if meta.get("lidar_points", 0):
        lo = torch.tensor(meta["pc_range"][:3])
        hi = torch.tensor(meta["pc_range"][3:])
        n = meta["lidar_points"]
        pts = torch.rand(n, 3, generator=g) * (hi - lo) + lo
        pl = torch.randint(1, 17, (n, 1), generator=g).float()
        points = [torch.cat((pts, pl), 1).to(device)]
    H, W = meta["input_size"]
    d = torch.rand(1, meta["ncams"], H, W, generator=g) * 55.0 + 2.0
    d = torch.where(torch.rand(1, meta["ncams"], H, W, generator=g) < 0.03, d, torch.zeros_like(d))
    return gt_occ.contiguous().to(device), points, d.to(device)


def train_cfg_pts():
    """occformer_nusc_r50_256x704.py:191-204 / occformer_kitti.py train_cfg.pts"""
    return dict(num_points=12544 * 4, oversample_ratio=3.0, importance_sample_ratio=0.75,
                assigner=dict(type="MaskHungarianAssigner", cls_cost=dict(type="ClassificationCost", weight=2.0),
                              mask_cost=dict(type="CrossEntropyLossCost", weight=5.0, use_sigmoid=True),
                              dice_cost=dict(type="DiceCost", weight=5.0, pred_act=True, eps=1.0)),
                sampler=dict(type="MaskPseudoSampler"))
