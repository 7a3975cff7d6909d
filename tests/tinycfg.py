"""Small configurations (same structure as projects/configs/occformer_nusc/
occformer_nusc_r50_256x704.py of the reference, shrunk) used by the golden-vector
generator and the parity tests."""
import copy


def tiny_nusc(ncams=3):
    pc_range = [-8.0, -8.0, -2.0, 8.0, 8.0, 2.0]
    occ_size = [32, 32, 16]
    ds = [2, 2, 2]
    vs = [(pc_range[3 + i] - pc_range[i]) / occ_size[i] for i in range(3)]
    grid_config = dict(xbound=[pc_range[0], pc_range[3], vs[0] * ds[0]],
                       ybound=[pc_range[1], pc_range[4], vs[1] * ds[1]],
                       zbound=[pc_range[2], pc_range[5], vs[2] * ds[2]],
                       dbound=[2.0, 10.0, 0.5])
    data_config = dict(Ncams=ncams, input_size=(64, 176), src_size=(900, 1600))
    C = 32
    chans = [32, 64, 128, 256]
    E = 96
    norm_cfg = dict(type='GN', num_groups=8, requires_grad=True)
    model = dict(
        type='OccupancyFormer',
        img_view_transformer=dict(
            type='ViewTransformerLiftSplatShootVoxel', loss_depth_weight=1.0,
            grid_config=grid_config, data_config=data_config, numC_input=32, numC_Trans=C,
            vp_megvii=False),
        img_bev_encoder_backbone=dict(
            type='OccupancyEncoder', num_stage=4, in_channels=C, block_numbers=[2, 2, 2, 2],
            block_inplanes=chans, block_strides=[1, 2, 2, 2], out_indices=(0, 1, 2, 3),
            with_cp=True, norm_cfg=norm_cfg),
        img_bev_encoder_neck=dict(
            type='MSDeformAttnPixelDecoder3D', strides=[2, 4, 8, 16], in_channels=chans,
            feat_channels=E, out_channels=E, norm_cfg=norm_cfg,
            encoder=dict(
                type='DetrTransformerEncoder', num_layers=2,
                transformerlayers=dict(
                    type='BaseTransformerLayer',
                    attn_cfgs=dict(type='MultiScaleDeformableAttention3D', embed_dims=E,
                                   num_heads=8, num_levels=3, num_points=4, im2col_step=64,
                                   dropout=0.0, batch_first=False, norm_cfg=None, init_cfg=None),
                    ffn_cfgs=dict(embed_dims=E), feedforward_channels=E * 4, ffn_dropout=0.0,
                    operation_order=('self_attn', 'norm', 'ffn', 'norm')),
                init_cfg=None),
            positional_encoding=dict(type='SinePositionalEncoding3D', num_feats=E // 3,
                                     normalize=True)),
        pts_bbox_head=dict(
            type='Mask2FormerNuscOccHead', feat_channels=E, out_channels=E, num_queries=20,
            num_occupancy_classes=17, pooling_attn_mask=True, sample_weight_gamma=0.25,
            positional_encoding=dict(type='SinePositionalEncoding3D', num_feats=E / 3,
                                     normalize=True),
            transformer_decoder=dict(
                type='DetrTransformerDecoder', return_intermediate=True, num_layers=6,
                transformerlayers=dict(
                    type='DetrTransformerDecoderLayer',
                    attn_cfgs=dict(type='MultiheadAttention', embed_dims=E, num_heads=E // 32,
                                   attn_drop=0.0, proj_drop=0.0, dropout_layer=None,
                                   batch_first=False),
                    ffn_cfgs=dict(embed_dims=E, num_fcs=2, act_cfg=dict(type='ReLU', inplace=True),
                                  ffn_drop=0.0, dropout_layer=None, add_identity=True),
                    feedforward_channels=E * 8,
                    operation_order=('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm')),
                init_cfg=None),
            loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=2.0,
                          reduction='mean', class_weight=[1.0] * 17 + [0.1]),
            loss_mask=dict(type='CrossEntropyLoss', use_sigmoid=True, reduction='mean',
                           loss_weight=5.0),
            loss_dice=dict(type='DiceLoss', use_sigmoid=True, activate=True, reduction='mean',
                           naive_dice=True, eps=1.0, loss_weight=5.0),
            point_cloud_range=pc_range),
    )
    meta = dict(pc_range=pc_range, occ_size=occ_size, D=16, C=C, groups=8, pd_layers=2,
                dec_layers=6, heads=E // 32, E=E, chans=chans, fH=4, fW=11, focal=140.0,
                input_size=(64, 176), grid=(16, 16, 8))
    return copy.deepcopy(model), meta


def tiny_kitti():
    """SemanticKITTI form of the tiny detector (occformer_kitti.py:22-205): ONE camera with 4x4 intrinsics / BEV
    augmentation (33 camera scalars), ``Mask2FormerOccHead`` with 20 classes and class-guided sampling."""
    model, meta = tiny_nusc(ncams=1)
    model["img_view_transformer"]["cam_channels"] = 33
    head = model["pts_bbox_head"]
    head.update(type="Mask2FormerOccHead", num_occupancy_classes=20)
    head.pop("point_cloud_range", None)
    head["loss_cls"] = dict(head["loss_cls"], class_weight=[1.0] * 20 + [0.1])
    meta = dict(meta, kitti=True)
    return copy.deepcopy(model), meta
