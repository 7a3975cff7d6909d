"""Pins oracle/occformer_train_ref.py against the REFERENCE's own training code and
writes tests/golden/train.npz.

Run in the build container (needs /root/reference):   python tests/golden/make_golden_train.py

The reference modules are imported unmodified through tests/refshim (mmcv/mmdet stand-ins; the mmdet 2.14.0
CrossEntropyLoss / ClassificationCost / weight_reduce_loss are restated there).  Reference and oracle are run
under the same torch.manual_seed: the oracle consumes the global generator in the reference's order, so the
numbers must agree to float rounding.  Inputs are regenerated from tests/paramgen keys, only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import paramgen, refshim, tinycfg          # noqa: E402
from oracle import occformer_train_ref as T           # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def inputs(kind):
    """deterministic synthetic head outputs + ground truth (shared with tests/test_training.py)"""
    B, Q, L = 2, 20, 3
    nc = 17 if kind == "nusc" else 20
    grid, occ = (16, 16, 8), (32, 32, 16)
    cls = [paramgen.tensor(f"{kind}.cls{l}", (B, Q, nc + 1), 7, 1.5) for l in range(L)]
    masks = [paramgen.tensor(f"{kind}.mask{l}", (B, Q) + grid, 7, 2.0) for l in range(L)]
    # blocky label volume with a few classes + void
    lab = (paramgen.uniform(f"{kind}.occ", (B, 8, 8, 4), 7) * 9).long()
    lab = torch.where(lab == 8, torch.full_like(lab, 255), lab * 2 + (1 if kind == "nusc" else 0))
    gt_occ = lab.repeat_interleave(4, 1).repeat_interleave(4, 2).repeat_interleave(4, 3)
    pts = []
    for b in range(B):
        xyz = paramgen.uniform(f"{kind}.pts{b}", (150 + 40 * b, 3), 7) * torch.tensor([17.0, 17.0, 4.4]) - \
            torch.tensor([8.5, 8.5, 2.2])
        pl = (paramgen.uniform(f"{kind}.ptl{b}", (xyz.shape[0], 1), 7) * 16).floor() + 1
        pts.append(torch.cat((xyz, pl), 1))
    return cls, masks, gt_occ, pts


def train_cfg(num_points=256):
    return dict(num_points=num_points, oversample_ratio=3.0, importance_sample_ratio=0.75,
                assigner=dict(type="MaskHungarianAssigner", cls_cost=dict(type="ClassificationCost", weight=2.0),
                              mask_cost=dict(type="CrossEntropyLossCost", weight=5.0, use_sigmoid=True),
                              dice_cost=dict(type="DiceCost", weight=5.0, pred_act=True, eps=1.0)),
                sampler=dict(type="MaskPseudoSampler"))


def oracle_cfg(head_cfg, tc, **extra):
    return dict(point_cloud_range=head_cfg.get("point_cloud_range"), num_points=tc["num_points"],
                oversample_ratio=tc["oversample_ratio"], importance_sample_ratio=tc["importance_sample_ratio"],
                padding_mode="border", num_classes=head_cfg["num_occupancy_classes"],
                class_weight=head_cfg["loss_cls"]["class_weight"], **extra)


def kitti_head_cfg(model):
    h = dict(model["pts_bbox_head"])
    h.update(type="Mask2FormerOccHead", num_occupancy_classes=20)
    h.pop("point_cloud_range", None)
    h["loss_cls"] = dict(h["loss_cls"], class_weight=[1.0] * 20 + [0.1])
    return h


def main():
    refshim.install()
    for n in ("losses.dice_loss", "assigners.match_costs.match_cost", "assigners.mask_hungarian_assigner",
              "samplers.mask_pseudo_sampler", "positional_encodings.positional_encoding"):
        refshim.ref("occformer.mask2former." + n)
    nusc = refshim.ref("occformer.mask2former.mask2former_nusc_occ")
    kitti = refshim.ref("occformer.mask2former.mask2former_occ")
    vtm = refshim.ref("occformer.image2bev.ViewTransformerLSSVoxel")
    model, meta = tinycfg.tiny_nusc()
    from mmcv.utils import ConfigDict
    out, report = {}, {}

    def err(a, b):
        return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())

    # ---------------------------------------------------------------- nuScenes head: loss() (rows 19-21)
    tc = train_cfg()
    hc = ConfigDict(model["pts_bbox_head"])
    args = {k: v for k, v in hc.items() if k != "type"}
    head = nusc.Mask2FormerNuscOccHead(**args, train_cfg=ConfigDict(tc), test_cfg=None)
    cls, masks, gt_occ, pts = inputs("nusc")
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * 2
    gl_ref, gm_ref = head.preprocess_gt(gt_occ, metas)
    gl_o, gm_o = zip(*[T.preprocess_occupancy_gt(g, 17) for g in gt_occ])
    for a, b, c, d in zip(gl_ref, gl_o, gm_ref, gm_o):
        assert torch.equal(a.long(), b) and torch.equal(c, d) and c.dtype == d.dtype
    torch.manual_seed(11)
    ref_losses = head.loss(cls, masks, gl_ref, gm_ref, pts, metas)
    ocfg = oracle_cfg(model["pts_bbox_head"], tc)
    torch.manual_seed(11)
    o_losses = T.head_loss(cls, masks, T.nusc_loss_single, list(gl_o), list(gm_o), pts, cfg=ocfg)
    for k in ref_losses:
        report["nusc." + k] = err(ref_losses[k], o_losses[k])
        out["nusc." + k] = ref_losses[k].detach().numpy()
    # targets of one layer
    torch.manual_seed(12)
    tr = head.get_targets([cls[0][i] for i in range(2)], [masks[0][i] for i in range(2)], gl_ref, gm_ref, pts, metas)
    torch.manual_seed(12)
    to = [T.nusc_get_target_single(cls[0][i], masks[0][i], gl_o[i], gm_o[i], pts[i], ocfg, T.GlobalTorchRNG())
          for i in range(2)]
    for i in range(2):
        assert torch.equal(tr[0][i], to[i]["labels"]), "label targets differ"
        report[f"nusc.mask_weights{i}"] = err(tr[3][i], to[i]["mask_weights"])
        out[f"nusc.labels{i}"] = tr[0][i].numpy()
    # lidarseg training metric
    head.train()
    import unittest.mock as um
    with um.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        miou = head.forward_lidarseg(cls[-1], masks[-1], pts, metas)["point_mean_iou"]
    out["nusc.point_mean_iou"] = miou.numpy()

    # ---------------------------------------------------------------- SemanticKITTI head (rows 18, 20, 21)
    kc = ConfigDict(kitti_head_cfg(model))
    args = {k: v for k, v in kc.items() if k != "type"}
    khead = kitti.Mask2FormerOccHead(**args, train_cfg=ConfigDict(tc), test_cfg=None)
    khead.get_sampling_weights()
    cls, masks, gt_occ, _ = inputs("kitti")
    gl_ref, gm_ref = khead.preprocess_gt(gt_occ, metas)
    gl_o, gm_o = zip(*[T.preprocess_occupancy_gt(g, 20) for g in gt_occ])
    kcfg = oracle_cfg(kitti_head_cfg(model), tc, align_corners=True,
                      sample_weights=T.kitti_sampling_weights(refshim.ref("utils.semkitti").semantic_kitti_class_frequencies, 0.25))
    kcfg["class_weight"] = khead.class_weight
    report["kitti.sample_weights"] = err(khead.sample_weights, kcfg["sample_weights"])
    out["kitti.class_weight"] = np.asarray(khead.class_weight)
    torch.manual_seed(21)
    ref_losses = khead.loss(cls, masks, gl_ref, gm_ref, metas)
    torch.manual_seed(21)
    o_losses = T.head_loss(cls, masks, T.kitti_loss_single, list(gl_o), list(gm_o), cfg=kcfg)
    for k in ref_losses:
        report["kitti." + k] = err(ref_losses[k], o_losses[k])
        out["kitti." + k] = ref_losses[k].detach().numpy()
    # same-resolution branch (torch.gather of the logits): masks at the GT resolution
    big = [torch.nn.functional.interpolate(m, size=(32, 32, 16), mode="trilinear") for m in masks[:1]]
    torch.manual_seed(22)
    ref_same = khead.loss_single(cls[0], big[0], gl_ref, gm_ref, metas)
    torch.manual_seed(22)
    o_same = T.kitti_loss_single(cls[0], big[0], list(gl_o), list(gm_o), kcfg, T.GlobalTorchRNG())
    for i, n in enumerate(("loss_cls", "loss_mask", "loss_dice")):
        report["kitti.same." + n] = err(ref_same[i], o_same[i])
        out["kitti.same." + n] = ref_same[i].detach().numpy()
    # torch.multinomial == exponential race (the restated ATen algorithm)
    w = paramgen.uniform("mn.w", (3, 4000), 9) ** 2
    torch.manual_seed(5)
    a = torch.multinomial(w, 700, replacement=False)
    torch.manual_seed(5)
    b = T.GlobalTorchRNG().multinomial(w, 700)
    assert torch.equal(a, b), "multinomial restatement differs from torch.multinomial"

    # ---------------------------------------------------------------- depth supervision (row 21)
    vt = vtm.ViewTransformerLiftSplatShootVoxel(**{k: v for k, v in model["img_view_transformer"].items() if k != "type"})
    H, W = meta["input_size"]
    gd = paramgen.uniform("depth.gt", (2, 3, H, W), 8) * 14.0
    gd = torch.where(paramgen.uniform("depth.keep", (2, 3, H, W), 8) < 0.03, gd, torch.zeros(()))
    dp = paramgen.uniform("depth.pred", (6, meta["D"], meta["fH"], meta["fW"]), 8).softmax(1)
    ref_d = vt.get_depth_loss(gd, dp)
    gc = model["img_view_transformer"]["grid_config"]
    o_d = T.depth_bce_loss(gd, dp, vt.downsample, gc["dbound"], vt.D, vt.loss_depth_weight)
    report["depth.loss"] = err(ref_d, o_d)
    out["depth.loss"] = ref_d.numpy()

    for k, v in report.items():
        print(f"  {k:28s} {v:.3e}")
    assert max(report.values()) < 2e-5, "oracle deviates from the reference"
    np.savez_compressed(os.path.join(OUT, "train.npz"), **out)
    print("wrote train.npz:", sorted(out))


if __name__ == "__main__":
    main()
