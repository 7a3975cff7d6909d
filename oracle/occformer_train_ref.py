"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's training-only rows.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(occformer_amd/) never does.  Every function cites the reference lines it follows.  All randomness goes
through an ``rng`` object; the default ``GlobalTorchRNG`` draws from torch's global CPU generator with the
same call order, shapes and dtypes as the reference, so under ``torch.manual_seed(s)`` the oracle
reproduces the reference's own numbers (pinned in tests/golden/make_golden.py -> tests/golden/train.npz).

``rng.multinomial`` restates ATen's ``torch.multinomial(w, k, replacement=False)`` (aten/src/ATen/native/
Sampling / TensorFactories: ``q = empty_like(w).exponential_(1); topk(w / q, k)``), which is what lets the
HIP radix-select sampler be compared on identical noise.

P = /root/reference/projects/mmdet3d_plugin/occformer
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy.optimize import linear_sum_assignment


# ----------------------------------------------------------------------------------------- RNG
class GlobalTorchRNG:
    """torch's global CPU generator, consumed exactly like the reference consumes it."""

    def rand(self, *shape):
        return torch.rand(tuple(shape))

    def randperm(self, n):
        return torch.randperm(n)

    def exponential(self, shape, dtype=torch.float32):
        return torch.empty(tuple(shape), dtype=dtype).exponential_(1)

    def multinomial(self, weights, k):
        """== torch.multinomial(weights, k, replacement=False) under the same generator state."""
        q = self.exponential(weights.shape, weights.dtype)
        return torch.topk(weights / q, k, dim=-1)[1]


class RecordingRNG(GlobalTorchRNG):
    """records every draw so that a second consumer (the HIP product path) can replay them"""

    def __init__(self):
        self.tape = []

    def rand(self, *shape):
        t = super().rand(*shape)
        self.tape.append(("rand", t))
        return t

    def randperm(self, n):
        t = super().randperm(n)
        self.tape.append(("randperm", t))
        return t

    def exponential(self, shape, dtype=torch.float32):
        t = super().exponential(shape, dtype)
        self.tape.append(("exponential", t))
        return t


# ----------------------------------------------------------------------------------------- helpers
def point_sample_3d(vol, points, align_corners=False, padding_mode="zeros"):
    """P/mask2former/base/mmdet_utils.py:21-47: grid_sample at points in [0,1] (grid order z,y,x)."""
    n, p = points.shape[:2]
    out = F.grid_sample(vol, (points * 2.0 - 1.0).view(n, p, 1, 1, 3), align_corners=align_corners,
                        padding_mode=padding_mode)
    return out.view(n, vol.shape[1], p)


def unravel_indices(indices, shape):
    """mmdet_utils.py:71-89"""
    out = []
    for d in reversed(shape):
        out.append(indices % d)
        indices = torch.div(indices, d, rounding_mode="floor")
    return torch.stack(out[::-1], -1)


def preprocess_occupancy_gt(gt_occ, num_classes):
    """mmdet_utils.py:426-475 (with_binary_occupancy=False): one 0/1 int64 mask per class present."""
    gt_occ = gt_occ.squeeze(0)
    labels, masks = [], []
    for lab in torch.unique(gt_occ):
        if lab >= num_classes:
            continue
        labels.append(lab)
        masks.append(gt_occ == lab)
    assert len(masks) > 0
    return torch.stack(labels).long(), torch.stack(masks).long()


# ----------------------------------------------------------------------------------------- costs / matching
def classification_cost(cls_pred, gt_labels, weight):
    """mmdet 2.14.0 ClassificationCost: -softmax(cls)[:, gt] * w"""
    return -cls_pred.softmax(-1)[:, gt_labels] * weight


def bce_cost(pred, gt, weight):
    """P/mask2former/assigners/match_costs/match_cost.py:69-128"""
    pred, gt = pred.flatten(1).float(), gt.flatten(1).float()
    n = pred.shape[1]
    pos = F.binary_cross_entropy_with_logits(pred, torch.ones_like(pred), reduction="none")
    neg = F.binary_cross_entropy_with_logits(pred, torch.zeros_like(pred), reduction="none")
    return (pos @ gt.t() + neg @ (1 - gt).t()) / n * weight


def dice_cost(pred, gt, weight, pred_act=True, eps=1.0, naive_dice=True):
    """match_cost.py:9-66"""
    pred = pred.flatten(1)
    if pred_act:
        pred = pred.sigmoid()
    gt = gt.flatten(1).float()
    num = 2 * pred @ gt.t()
    if naive_dice:
        den = pred.sum(-1)[:, None] + gt.sum(-1)[None]
    else:
        den = pred.pow(2).sum(1)[:, None] + gt.pow(2).sum(1)[None]
    return (1 - (num + eps) / (den + eps)) * weight


def hungarian_assign(cls_pred, mask_pred, gt_labels, gt_mask, w_cls=2.0, w_mask=5.0, w_dice=5.0, dice_eps=1.0):
    """P/mask2former/assigners/mask_hungarian_assigner.py:42-126 + samplers/mask_pseudo_sampler.py.
    -> (cost, pos_inds (sorted query ids), pos_assigned_gt_inds)"""
    nq, ngt = mask_pred.shape[0], gt_labels.shape[0]
    if ngt == 0 or nq == 0:
        e = torch.zeros((0,), dtype=torch.long)
        return torch.zeros((nq, ngt)), e, e
    cost = classification_cost(cls_pred, gt_labels, w_cls) + bce_cost(mask_pred, gt_mask, w_mask) + \
        dice_cost(mask_pred, gt_mask, w_dice, eps=dice_eps)
    rows, cols = linear_sum_assignment(cost.detach().cpu())
    gt_inds = torch.zeros((nq,), dtype=torch.long)
    gt_inds[torch.from_numpy(rows)] = torch.from_numpy(cols) + 1
    pos = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1).unique()
    return cost, pos, gt_inds[pos] - 1


# ----------------------------------------------------------------------------------------- losses
def weight_reduce_mean(loss, weight, avg_factor):
    """mmdet 2.14.0 weight_reduce_loss(reduction='mean', avg_factor given)"""
    if weight is not None:
        loss = loss * weight
    return loss.sum() / avg_factor


def ce_loss(cls_scores, labels, label_weights, class_weight, avg_factor, loss_weight=2.0):
    """mmdet CrossEntropyLoss(use_sigmoid=False, class_weight) as called at mask2former_nusc_occ.py:363-368"""
    loss = F.cross_entropy(cls_scores, labels, weight=class_weight, reduction="none")
    return loss_weight * weight_reduce_mean(loss, label_weights.float(), avg_factor)


def bce_point_loss(pred, target, weight, avg_factor, loss_weight=5.0):
    """mmdet CrossEntropyLoss(use_sigmoid=True) on flattened point logits (:411-417 / mask2former_occ.py:437-442)"""
    loss = F.binary_cross_entropy_with_logits(pred, target.float(), reduction="none")
    return loss_weight * weight_reduce_mean(loss, weight, avg_factor)


def dice_point_loss(pred, target, weight, avg_factor, eps=1.0, loss_weight=5.0):
    """P/mask2former/losses/dice_loss.py:8-61 (use_sigmoid, activate, naive_dice=True)"""
    x = pred.sigmoid().flatten(1)
    t = target.flatten(1).float()
    a = (x * t).sum(1)
    d = (2 * a + eps) / (x.sum(1) + t.sum(1) + eps)
    return loss_weight * weight_reduce_mean(1 - d, weight, avg_factor)


# ----------------------------------------------------------------------------------------- nuScenes head
def nusc_get_target_single(cls_score, mask_pred, gt_labels, gt_masks, gt_lidarseg, cfg, rng):
    """P/mask2former/mask2former_nusc_occ.py:196-273"""
    nq, ngt = cls_score.shape[0], gt_labels.shape[0]
    gt_labels = gt_labels.long()
    pcr = torch.tensor(cfg["point_cloud_range"]).type_as(gt_lidarseg)
    coords = (gt_lidarseg[:, :3] - pcr[:3]) / (pcr[3:] - pcr[:3])
    n_lidar = min(cfg["num_points"] // 2, coords.shape[0])
    if n_lidar < coords.shape[0]:
        coords = coords[rng.randperm(coords.shape[0])[:n_lidar]]
    coords = torch.cat((coords, rng.rand(cfg["num_points"] - n_lidar, 3)), 0)[..., [2, 1, 0]]
    pm = cfg.get("padding_mode", "border")
    pred_pts = point_sample_3d(mask_pred.unsqueeze(1), coords.repeat(nq, 1, 1), padding_mode=pm).squeeze(1)
    gt_pts = point_sample_3d(gt_masks.unsqueeze(1).float(), coords.repeat(ngt, 1, 1), padding_mode=pm).squeeze(1)
    cost, pos, pos_gt = hungarian_assign(cls_score, pred_pts, gt_labels, gt_pts)
    labels = gt_labels.new_full((nq,), cfg["num_classes"], dtype=torch.long)
    labels[pos] = gt_labels[pos_gt]
    cw = torch.tensor(cfg["class_weight"]).type_as(cls_score)
    mask_weights = mask_pred.new_zeros((nq,))
    mask_weights[pos] = cw[labels[pos]]
    return dict(labels=labels, label_weights=torch.ones(nq), mask_targets=gt_masks[pos_gt],
                mask_weights=mask_weights, pos_inds=pos, pos_gt=pos_gt, cost=cost, coords=coords)


def nusc_importance_points(mask_pred, gt_lidarseg_list, n_gt_list, cfg, rng):
    """mmdet_utils.py:138-177 (get_nusc_lidarseg_point_coords); mask_pred [n_pos, 1, X, Y, Z]"""
    n_pos = mask_pred.shape[0]
    num_points = cfg["num_points"]
    num_sampled = int(num_points * cfg["oversample_ratio"])
    pcr = torch.tensor(cfg["point_cloud_range"]).type_as(mask_pred)
    rows = []
    for lidar, n_gt in zip(gt_lidarseg_list, n_gt_list):
        c = (lidar[:, :3] - pcr[:3]) / (pcr[3:] - pcr[:3])
        c = torch.cat((c, rng.rand(num_sampled - c.shape[0], 3)), 0)
        rows.extend([c] * n_gt)
    coords = torch.stack(rows, 0)
    logits = point_sample_3d(mask_pred, coords[..., [2, 1, 0]], padding_mode=cfg.get("padding_mode", "border")).squeeze(1)
    n_unc = int(cfg["importance_sample_ratio"] * num_points)
    idx = torch.topk(-logits.abs(), k=n_unc, dim=1)[1]
    coords = torch.gather(coords, 1, idx[..., None].expand(-1, -1, 3))
    if num_points - n_unc > 0:
        coords = torch.cat((coords, rng.rand(n_pos, num_points - n_unc, 3)), 1)
    return coords


def nusc_loss_single(cls_scores, mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, cfg, rng):
    """mask2former_nusc_occ.py:317-424.  -> (loss_cls, loss_mask, loss_dice, debug)"""
    B = cls_scores.shape[0]
    tg = [nusc_get_target_single(cls_scores[i], mask_preds[i], gt_labels_list[i], gt_masks_list[i],
                                 gt_lidarseg_list[i], cfg, rng) for i in range(B)]
    labels = torch.stack([t["labels"] for t in tg]).flatten()
    label_weights = torch.stack([t["label_weights"] for t in tg]).flatten()
    mask_targets = torch.cat([t["mask_targets"] for t in tg], 0)
    mask_weights = torch.stack([t["mask_weights"] for t in tg])
    cw = cls_scores.new_tensor(cfg["class_weight"])
    loss_cls = ce_loss(cls_scores.flatten(0, 1), labels, label_weights, cw, cw[labels].sum())
    sel = mask_weights > 0
    mp = mask_preds[sel]
    mw = mask_weights[sel]
    if mask_targets.shape[0] == 0:
        return loss_cls, mp.sum(), mp.sum(), dict(targets=tg)
    coords = nusc_importance_points(mp.unsqueeze(1), gt_lidarseg_list, [g.shape[0] for g in gt_labels_list], cfg,
                                    rng)[..., [2, 1, 0]]
    pm = cfg.get("padding_mode", "border")
    pp = point_sample_3d(mp.unsqueeze(1), coords, padding_mode=pm).squeeze(1)
    pt = point_sample_3d(mask_targets.unsqueeze(1).float(), coords, padding_mode=pm).squeeze(1)
    total = mw.sum()
    loss_dice = dice_point_loss(pp, pt, mw, total)
    loss_mask = bce_point_loss(pp.reshape(-1), pt.reshape(-1), None, total * cfg["num_points"])
    return loss_cls, loss_mask, loss_dice, dict(targets=tg, coords=coords, point_preds=pp, point_targets=pt)


# ----------------------------------------------------------------------------------------- SemanticKITTI head
def kitti_sampling_weights(class_frequencies, gamma):
    """P/mask2former/mask2former_occ.py:96-100, 158-166 (+ utils/semkitti.py:3-26):
    w = (1/freq) / min(1/freq), raised to gamma"""
    w = 1.0 / np.asarray(class_frequencies, dtype=np.float64)
    w = w / w.min()
    return w ** gamma


def kitti_voxel_weights(gt_labels, gt_masks, sample_weights):
    """mmdet_utils.py:96-98 / 120-122: per-voxel weight = sum_g w[label_g] * mask_g"""
    sw = torch.tensor(sample_weights).to(gt_masks.device)
    return (sw[gt_labels].view(-1, 1, 1, 1) * gt_masks).sum(0).view(-1)


def kitti_sample_valid(num_points, gt_labels, gt_masks, sample_weights, rng):
    """mmdet_utils.py:91-108"""
    w = kitti_voxel_weights(gt_labels, gt_masks, sample_weights)
    idx = rng.multinomial(w, num_points)
    dims = gt_masks.shape[1:]
    coords = unravel_indices(idx, dims).float() / (torch.tensor(dims).type_as(gt_masks).view(1, 1, -1) - 1).float()
    return idx, coords


def kitti_batch_sample_valid(num_points, gt_labels_list, gt_masks_list, sample_weights, rng):
    """mmdet_utils.py:110-136"""
    sw = torch.tensor(sample_weights).float()
    rows = []
    for gl, gm in zip(gt_labels_list, gt_masks_list):
        w = (sw[gl].view(-1, 1, 1, 1) * gm).sum(0).view(-1)
        rows.append(w[None].repeat(gl.shape[0], 1))
    idx = rng.multinomial(torch.cat(rows, 0), num_points)
    dims = gt_masks_list[-1].shape[1:]
    coords = unravel_indices(idx, dims).float() / (torch.tensor(dims).type_as(gt_masks_list[-1]).view(1, 1, -1) - 1).float()
    return idx, coords


def kitti_get_target_single(cls_score, mask_pred, gt_labels, gt_masks, cfg, rng):
    """mask2former_occ.py:224-292"""
    nq, ngt = cls_score.shape[0], gt_labels.shape[0]
    gt_labels = gt_labels.long()
    idx, coords = kitti_sample_valid(cfg["num_points"], gt_labels, gt_masks, cfg["sample_weights"], rng)
    coords = coords[..., [2, 1, 0]]
    pred_pts = point_sample_3d(mask_pred.unsqueeze(1), coords.repeat(nq, 1, 1),
                               align_corners=cfg["align_corners"]).squeeze(1)
    gt_pts = gt_masks.view(ngt, -1)[:, idx]
    cost, pos, pos_gt = hungarian_assign(cls_score, pred_pts, gt_labels, gt_pts)
    labels = gt_labels.new_full((nq,), cfg["num_classes"], dtype=torch.long)
    labels[pos] = gt_labels[pos_gt]
    cw = torch.tensor(cfg["class_weight"]).type_as(cls_score)
    mask_weights = mask_pred.new_zeros((nq,))
    mask_weights[pos] = cw[labels[pos]]
    return dict(labels=labels, label_weights=torch.ones(nq), mask_targets=gt_masks[pos_gt],
                mask_weights=mask_weights, pos_inds=pos, pos_gt=pos_gt, cost=cost, point_indices=idx)


def kitti_uncertain_points(mask_pred, gt_labels_list, gt_masks_list, cfg, rng):
    """mmdet_utils.py:179-246; mask_pred [n_pos, 1, X, Y, Z] at the GT resolution or coarser"""
    n = mask_pred.shape[0]
    num_points = cfg["num_points"]
    num_sampled = int(num_points * cfg["oversample_ratio"])
    idx, coords = kitti_batch_sample_valid(num_sampled, gt_labels_list, gt_masks_list, cfg["sample_weights"], rng)
    if tuple(mask_pred.shape[-3:]) == tuple(gt_masks_list[0].shape[1:]):
        logits = torch.gather(mask_pred.view(n, -1), 1, idx)
    else:
        logits = point_sample_3d(mask_pred, coords[..., [2, 1, 0]], align_corners=True).squeeze(1)
    n_unc = int(cfg["importance_sample_ratio"] * num_points)
    top = torch.topk(-logits.abs(), k=n_unc, dim=1)[1]
    idx = torch.gather(idx, 1, top)
    coords = torch.gather(coords, 1, top[..., None].expand(-1, -1, 3))
    if num_points - n_unc > 0:
        ridx, rcoords = kitti_batch_sample_valid(num_points - n_unc, gt_labels_list, gt_masks_list,
                                                 np.ones_like(cfg["sample_weights"]), rng)
        idx = torch.cat((idx, ridx), 1)
        coords = torch.cat((coords, rcoords), 1)
    return idx, coords


def kitti_loss_single(cls_scores, mask_preds, gt_labels_list, gt_masks_list, cfg, rng):
    """mask2former_occ.py:343-444"""
    B = cls_scores.shape[0]
    tg = [kitti_get_target_single(cls_scores[i], mask_preds[i], gt_labels_list[i], gt_masks_list[i], cfg, rng)
          for i in range(B)]
    labels = torch.stack([t["labels"] for t in tg]).flatten()
    label_weights = torch.stack([t["label_weights"] for t in tg]).flatten()
    mask_targets = torch.cat([t["mask_targets"] for t in tg], 0)
    mask_weights = torch.stack([t["mask_weights"] for t in tg])
    cw = cls_scores.new_tensor(cfg["class_weight"])
    loss_cls = ce_loss(cls_scores.flatten(0, 1), labels, label_weights, cw, cw[labels].sum())
    sel = mask_weights > 0
    mp = mask_preds[sel]
    mw = mask_weights[sel]
    if mask_targets.shape[0] == 0:
        return loss_cls, mp.sum(), mp.sum(), dict(targets=tg)
    idx, coords = kitti_uncertain_points(mp.unsqueeze(1), gt_labels_list, gt_masks_list, cfg, rng)
    pt = torch.gather(mask_targets.view(mask_targets.shape[0], -1), 1, idx)
    pp = point_sample_3d(mp.unsqueeze(1), coords[..., [2, 1, 0]], align_corners=cfg["align_corners"]).squeeze(1)
    loss_dice = dice_point_loss(pp, pt, mw, mw.sum())
    pw = mw.view(-1, 1).repeat(1, cfg["num_points"]).reshape(-1)
    loss_mask = bce_point_loss(pp.reshape(-1), pt.reshape(-1).float(), pw, pw.sum())
    return loss_cls, loss_mask, loss_dice, dict(targets=tg, point_indices=idx, coords=coords, point_preds=pp,
                                                point_targets=pt)


def head_loss(all_cls_scores, all_mask_preds, loss_single, *gt, cfg=None, rng=None):
    """mask2former_nusc_occ.py:275-315: per-decoder-layer loss_single, in layer order; naming of the dict"""
    rng = rng or GlobalTorchRNG()
    per = [loss_single(c, m, *gt, cfg, rng)[:3] for c, m in zip(all_cls_scores, all_mask_preds)]
    out = {"loss_cls": per[-1][0], "loss_mask": per[-1][1], "loss_dice": per[-1][2]}
    for i, (a, b, c) in enumerate(per[:-1]):
        out[f"d{i}.loss_cls"], out[f"d{i}.loss_mask"], out[f"d{i}.loss_dice"] = a, b, c
    return out


# ----------------------------------------------------------------------------------------- depth supervision
def downsampled_gt_depth(gt_depths, downsample, dbound, D):
    """P/image2bev/ViewTransformerLSSVoxel.py:31-51: min non-zero depth per downsample x downsample patch,
    binned to D one-hot classes"""
    B, N, H, W = gt_depths.shape
    g = gt_depths.view(B * N, H // downsample, downsample, W // downsample, downsample, 1)
    g = g.permute(0, 1, 3, 5, 2, 4).contiguous().view(-1, downsample * downsample)
    g = torch.where(g == 0.0, 1e5 * torch.ones_like(g), g).min(-1).values
    g = g.view(B * N, H // downsample, W // downsample)
    g = (g - (dbound[0] - dbound[2] / 2)) / dbound[2]
    vals = g.clone()
    g = torch.where((g < D + 1) & (g >= 0.0), g, torch.zeros_like(g))
    onehot = F.one_hot(g.long(), num_classes=D + 1).view(-1, D + 1)[:, 1:]
    return vals, onehot.float()


def depth_bce_loss(gt_depths, depth_preds, downsample, dbound, D, loss_depth_weight=1.0):
    """ViewTransformerLSSVoxel.py:53-65, 67-75 (bce branch, times loss_depth_weight)"""
    _, lab = downsampled_gt_depth(gt_depths, downsample, dbound, D)
    pred = depth_preds.permute(0, 2, 3, 1).contiguous().view(-1, D)
    fg = lab.max(1).values > 0.0
    loss = F.binary_cross_entropy(pred[fg], lab[fg], reduction="none").sum() / max(1.0, float(fg.sum()))
    return loss_depth_weight * loss


# ----------------------------------------------------------------------------------------- one training step
def train_step(sd, img_feats, cams, gt_depths, gt_occ, points, cfg, rng=None, drop_path=0.2, aspp_drop=0.1,
               depth_aspp_drop=0.5, gates=None):
    """occupancyformer.py:132-199 (forward_train) + loss.backward() on the restated reference path in TRAIN mode
    (oracle.occformer_ref.training_mode): depth BCE + the ten Hungarian prediction-set losses of the nuScenes head,
    then torch.autograd of their sum w.r.t. every trainable parameter.
    cfg: D, C, groups, heads, pd_layers, dec_layers, downsample, dbound, head (= oracle_cfg dict of the head's
    training rows: point_cloud_range, num_points, oversample_ratio, importance_sample_ratio, padding_mode,
    num_classes, class_weight).  A head dict with ``sample_weights`` selects the SemanticKITTI head
    (mask2former_occ.py:343-444: class-guided sampling, class-weighted mask losses, ``align_corners``; no LiDAR
    points).  -> (losses dict, {name: grad})
    ``gates``: an ``occformer_ref.forced_gates`` context -- the head's ReLUs then use the recorded gates of the
    implementation under test (see there)."""
    import contextlib
    from . import occformer_ref as O
    rng = rng or GlobalTorchRNG()
    frozen = ("running_mean", "running_var", "num_batches_tracked", ".frustum", ".dx", ".bx", ".nx",
              "relative_position_index")
    params = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(frozen) else v)
              for k, v in sd.items()}
    g = cfg.get("groups", 32)
    with O.training_mode(rng, drop_path, aspp_drop, depth_aspp_drop), (gates or contextlib.nullcontext()):
        vox, depth = O.view_transformer(params, "img_view_transformer.", img_feats, cams, cfg["D"], cfg["C"])
        enc = O.occupancy_encoder(params, "img_bev_encoder_backbone.", vox, groups=g,
                                  block_numbers=cfg.get("block_numbers", (2, 2, 2, 2)))
        dec = O.pixel_decoder(params, "img_bev_encoder_neck.", enc, groups=g, num_layers=cfg.get("pd_layers", 6))
        cls_list, mask_list = O.mask2former_head(params, "pts_bbox_head.", dec, heads=cfg.get("heads", 6),
                                                 num_layers=cfg.get("dec_layers", 9))
    losses = {"loss_depth": depth_bce_loss(gt_depths, depth, cfg.get("downsample", 16), cfg["dbound"], cfg["D"])}
    gl, gm = zip(*[preprocess_occupancy_gt(o, cfg["head"]["num_classes"]) for o in gt_occ])
    if "sample_weights" in cfg["head"]:
        losses.update(head_loss(cls_list, mask_list, kitti_loss_single, list(gl), list(gm), cfg=cfg["head"], rng=rng))
    else:
        losses.update(head_loss(cls_list, mask_list, nusc_loss_single, list(gl), list(gm), points, cfg=cfg["head"],
                                rng=rng))
    names = [k for k, v in params.items() if torch.is_tensor(v) and v.requires_grad]
    grads = torch.autograd.grad(sum(losses.values()), [params[k] for k in names], allow_unused=True)
    return losses, dict(zip(names, grads))
