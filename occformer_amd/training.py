"""Training-time target assignment and losses of the occupancy heads, forward
values, on the HIP kernels of csrc/sample.hip:

  point sampling         -> occf_point_sample_3d_fwd   (one launch for all queries / GT masks, points shared)
  class-guided sampling  -> occf_sample_wor_fwd        (radix select of the exponential-race keys)
  importance sampling    -> occf_topk_smallest_abs_fwd
  matching costs         -> one fp32 MFMA GEMM [2Q, P] x [P, G] + occf_point_loss_rows_fwd
  point losses           -> occf_point_loss_rows_fwd   (BCE / dice row sums in one pass)

The function names and argument meaning mirror the reference
(P = projects/mmdet3d_plugin/occformer/mask2former):
  P/base/mmdet_utils.py:21-47, 71-136, 138-246, 426-475; P/assigners/mask_hungarian_assigner.py:42-126;
  P/assigners/match_costs/match_cost.py:9-128; P/losses/dice_loss.py:8-61;
  P/mask2former_nusc_occ.py:153-424, 547-587; P/mask2former_occ.py:158-166, 224-444, 525-567.

All randomness comes from an ``rng`` object (``rand``, ``randperm``, ``exponential``) so that a run is
reproducible and testable on injected noise; ``DeviceRNG`` is the default.  The Hungarian step runs on the
device (csrc/assign.hip: shortest augmenting paths, what scipy's linear_sum_assignment implements) -- the
reference's ``cost.cpu()`` + scipy round trip (one host sync per image and decoder layer) is gone.  A cost matrix
with non-finite entries (a diverged step; scipy raises there) leaves GT rows unmatched: that is recorded in a
device flag and raised as ``FloatingPointError`` by ``OccHeadTrainingMixin.loss`` one step later (no host sync in
the step itself).
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import autograd as A
from . import dist_utils
from .ops import get_ops

# P/../utils/semkitti.py:3-26
semantic_kitti_class_frequencies = np.array([
    5.41773033e09, 1.57835390e07, 1.25136000e05, 1.18809000e05, 6.46799000e05, 8.21951000e05, 2.62978000e05,
    2.83696000e05, 2.04750000e05, 6.16887030e07, 4.50296100e06, 4.48836500e07, 2.26992300e06, 5.68402180e07,
    1.57196520e07, 1.58442623e08, 2.06162300e06, 3.69705220e07, 1.15198800e06, 3.34146000e05])


# NOTE on ``x.flip(-1)`` for the (x, y, z) <-> (z, y, x) swaps below: the reference writes ``coords[..., [2, 1, 0]]``;
# indexing with a Python list uploads an index tensor from pageable host memory on every call -- a host
# synchronisation in the middle of the loss loop (found by tests/test_boundary.py under sync-debug mode).
_DEV_CONST = {}


def dev_const(values, device, dtype):
    """a small constant (point-cloud range, grid dimensions, class weights) as a device tensor, uploaded ONCE per
    (values, device, dtype): ``torch.tensor(list, device=...)`` in the loss loop is a pageable host-to-device copy per
    call -- the host blocks on each (90 per training step, r03d)"""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:      # 'cuda' and 'cuda:N' are one device: key on the resolved index
        device = torch.device("cuda", torch.cuda.current_device())
    key = (tuple(float(v) for v in values), str(device), dtype)
    t = _DEV_CONST.get(key)
    if t is None:
        t = _DEV_CONST[key] = torch.tensor([float(v) for v in values], dtype=dtype, device=device)
    return t


class DeviceRNG:
    """noise source on the compute device (a seeded torch.Generator)"""

    def __init__(self, device, seed=0):
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)

    def rand(self, *shape):
        return torch.rand(tuple(shape), device=self.device, generator=self.gen)

    def randperm(self, n):
        return torch.randperm(n, device=self.device, generator=self.gen)

    def exponential(self, shape, dtype=torch.float32):
        return torch.empty(tuple(shape), device=self.device, dtype=torch.float32).exponential_(1, generator=self.gen)


# ------------------------------------------------------------------------------------------ lazy mask logits
class LazyMask:
    """The mask logits of one prediction set in the training step.  ``dense`` [B, Q, X, Y, Z] are the DETACHED logits
    (what target assignment, importance sampling and the attention masks read -- none of them carries a gradient in
    the reference either); the differentiable route to the parameters goes through ``embed`` [B, Q, E] (mask_embed
    output) and ``feat_tok`` [B, V, E] (channels-last mask features) by ``autograd.SampledMaskLogits(Joint)``:
    einsum('bqc,bcxyz->bqxyz') followed by point sampling is linear in both, so the dense [B, Q, X, Y, Z] gradient of
    the reference never has to exist.  The dense logits themselves are LAZY as well: the attention mask comes from the
    fused contraction + pooling kernel, the matching cost from sampled mask features, and only the ~20 matched rows
    of a set (``rows``) -- or, for the last set's lidarseg metric, the whole volume (``dense``) -- are contracted on
    demand (256 MB per set at the 200-grid otherwise).  Indexing by image gives the per-image view."""

    def __init__(self, dense, embed, feat_tok, vol_shape=None, feat_split=None):
        self._dense, self.embed, self.feat_tok = dense, embed, feat_tok
        self.vol_shape = tuple(int(v) for v in (vol_shape if vol_shape is not None else dense.shape[-3:]))
        self.feat_split = feat_split

    @property
    def shape(self):
        return tuple(self.embed.shape[:-1]) + self.vol_shape

    def _contract(self, embed_rows, feat, split):
        """[n, E] x [V, E]^T -> [n, X, Y, Z] detached"""
        out = get_ops().linear(embed_rows.detach().contiguous(), feat.detach(), None, w_split=split, allow_small=False)
        return out.view(embed_rows.shape[0], *self.vol_shape)

    @property
    def dense(self):
        if self._dense is None:
            with torch.no_grad():
                if self.embed.dim() == 2:
                    self._dense = self._contract(self.embed, self.feat_tok, self.feat_split)
                else:
                    self._dense = torch.stack([self[b].dense for b in range(self.embed.shape[0])])
        return self._dense

    def __getitem__(self, b):
        sp = None if self.feat_split is None else (self.feat_split[0][b], self.feat_split[1][b])
        return LazyMask(None if self._dense is None else self._dense[b], self.embed[b], _image(self.feat_tok, b),
                        self.vol_shape, sp)

    def sample_all(self, coords, align_corners=False, padding_mode="zeros"):
        """(one image) logits of all Q queries at ``coords`` [P, 3] -> [Q, P], no gradient: the logits are
        einsum('qc,cxyz->qxyz') and trilinear sampling is linear, so the mask FEATURES are sampled (8 contiguous
        E-float rows per point from the channels-last tokens) and contracted with mask_embed afterwards -- the
        [Q, X, Y, Z] logits would be gathered 8 x Q times per point, 4 scattered bytes each (HBM-bound, 290 us per
        prediction set at the 200-grid)."""
        ops = get_ops()
        f = ops.point_sample_tokens(self.feat_tok.detach(), self.vol_shape, coords.contiguous(), align_corners,
                                    padding_mode)
        return ops.linear(self.embed.detach().contiguous(), f, None, allow_small=False)      # [Q, P]

    def rows(self, idx_per_image):
        """matched rows: idx_per_image[b] = query indices of image b (ascending) -> LazyRows"""
        dense, embed, feat = [], [], []
        for b, i in enumerate(idx_per_image):
            img = self[b]
            e = img.embed[i]
            with torch.no_grad():
                if img._dense is not None:
                    dense.append(img._dense[i])
                elif e.shape[0] == 0:                       # an image without a matched query
                    dense.append(e.new_zeros((0,) + self.vol_shape))
                else:
                    dense.append(img._contract(e, img.feat_tok, img.feat_split))
            embed.append(e)
            feat.append(img.feat_tok)
        return LazyRows(dense, embed, feat)


def _image(t, b):
    """t[b]; for a batch of one as a free view (the backward of an index select writes a zero-filled copy of the whole
    [B, V, E] mask-feature gradient: 491 MB per prediction set at the 200-grid)"""
    return t.squeeze(0) if t.shape[0] == 1 else t[b]


class LazyRows:
    """selected rows of a LazyMask, image by image"""

    def __init__(self, dense, embed, feat_tok):
        self.dense_list, self.embed_list, self.feat_list = dense, embed, feat_tok
        self.dense = dense[0] if len(dense) == 1 else torch.cat(dense, 0)          # [n_pos, X, Y, Z] detached

    @property
    def shape(self):
        return self.dense.shape

    def sample(self, coords, align_corners, padding_mode):
        """coords [n_pos, P, 3] (grid_sample order) -> logits [n_pos, P] with the gradient route"""
        out, r0 = [], 0
        for d, e, f in zip(self.dense_list, self.embed_list, self.feat_list):
            n = d.shape[0]
            if n:
                out.append(A.SampledMaskLogits.apply(d, e, f, coords[r0:r0 + n], align_corners, padding_mode))
            r0 += n
        return torch.cat(out, 0)


def _dense(mp):
    return mp.dense if isinstance(mp, (LazyMask, LazyRows)) else mp.detach()


def _candidate_logits_voxel_major(ops, rows_e, feat, S, G, vol_shape, cand_zyx, pad):
    """logits [S, G, P3] of the S x G matched masks at the S candidate point sets ``cand_zyx`` [S, P3, 3] (no gradient:
    they only rank the candidates).  The channel-major logits [S, G, X, Y, Z] cost a gather of 8 x G scattered 4-byte
    reads per point and set (2.18 ms at the 200-grid, r05n); here the same contraction is written VOXEL-major --
    [V, S x Gp] with Gp = G rounded up to 4 columns, one streaming linear over the mask features -- so that a candidate
    reads 8 contiguous Gp-float rows (0.37 ms for the ten sets + the linear).  None when the geometry does not fit (the
    caller then samples the channel-major volume).

    The ranking logits come from a contraction of their own (freshly split, zero-padded rows), not from ``dense``: in
    the default three-term arithmetic both carry ~2^-16 per product and rank alike (tests/test_training.py compares this
    helper with ``point_sample_3d(dense)``); in the one-term ``bf16`` mode their roundings differ and a near-tie in
    |logit| may be ranked the other way -- a swapped candidate weighs 1 / num_points of one mask loss."""
    if ops.precision == "f32" or G == 0 or feat.dim() != 2 or feat.stride(1) != 1:
        return None
    Gp = (G + 3) // 4 * 4
    N = (S * Gp + 63) // 64 * 64                    # (widths the streaming linear tiles evenly: 64, 128, 192, 256)
    if N > 256:
        return None
    E = rows_e.shape[1]
    w = rows_e.new_zeros((N, E))
    w.view(-1)[: S * Gp * E].view(S, Gp, E)[:, :G].copy_(rows_e.view(S, G, E))
    vm = ops.linear(feat.detach(), w, None, w_split=ops.split_bf16(w), allow_small=False)                 # [V, N]
    out = torch.empty((S, G, cand_zyx.shape[1]), dtype=vm.dtype, device=vm.device)
    for s in range(S):
        lg = ops.point_sample_tokens(vm[:, s * Gp:(s + 1) * Gp], tuple(vol_shape), cand_zyx[s], False, pad)   # [P3, Gp]
        out[s].copy_(lg[:, :G].t())
    return out


def sample_logits(mp, coords, align_corners=False, padding_mode="zeros"):
    """point_sample_3d(mp.unsqueeze(1), coords).squeeze(1) for rows ``mp`` [n, X, Y, Z] (tensor or LazyRows), with
    the backward kernels attached when a gradient is wanted"""
    if isinstance(mp, LazyRows):
        return mp.sample(coords, align_corners, padding_mode)
    if mp.requires_grad and torch.is_grad_enabled():
        return A.PointSample3d.apply(mp.unsqueeze(1), coords, align_corners, padding_mode).squeeze(1)
    return point_sample_3d(mp.unsqueeze(1), coords, align_corners, padding_mode).squeeze(1)


def sample_logits_sets(sets, align_corners=False, padding_mode="zeros"):
    """``sample_logits`` for several prediction sets at once: sets = [(mp_i, coords_i)].  Lazy rows of sets that share
    their mask features (the ten prediction sets of a training step) go through ONE
    ``autograd.SampledMaskLogitsJoint`` per image, so the mask-feature gradient is contracted once instead of once
    per set."""
    lazy = [i for i, (mp, _) in enumerate(sets) if isinstance(mp, LazyRows)]
    out = [None] * len(sets)
    for i, (mp, coords) in enumerate(sets):
        if i not in lazy:
            out[i] = sample_logits(mp, coords, align_corners, padding_mode)
    if not lazy or not torch.is_grad_enabled():
        for i in lazy:
            out[i] = sets[i][0].sample(sets[i][1], align_corners, padding_mode)
        return out
    n_img = len(sets[lazy[0]][0].feat_list)
    parts = {i: [] for i in lazy}
    offs = {i: 0 for i in lazy}
    for b in range(n_img):
        feat = sets[lazy[0]][0].feat_list[b]
        members, vols, embeds, pts = [], [], [], []
        for i in lazy:
            mp, coords = sets[i]
            n = mp.dense_list[b].shape[0]
            if n == 0:
                continue
            f = mp.feat_list[b]
            if f.data_ptr() != feat.data_ptr() or f.shape != feat.shape:
                raise ValueError("sample_logits_sets: the prediction sets must share their mask features")
            members.append(i)
            vols.append(mp.dense_list[b])
            embeds.append(mp.embed_list[b])
            pts.append(coords[offs[i]:offs[i] + n])
            offs[i] += n
        if members:
            res = A.SampledMaskLogitsJoint.apply(feat, align_corners, padding_mode, len(members), *vols, *embeds, *pts)
            for i, r in zip(members, res):
                parts[i].append(r)
    for i in lazy:
        out[i] = torch.cat(parts[i], 0) if len(parts[i]) != 1 else parts[i][0]
    return out


# ------------------------------------------------------------------------------------------ helpers
def point_sample_3d(input, points, align_corners=False, padding_mode="zeros"):
    """mmdet_utils.py:21-47.  input [N, C, X, Y, Z]; points [N, P, 3] (or [1, P, 3] shared) in [0, 1]."""
    return get_ops().point_sample_3d(input.contiguous(), points.contiguous(), align_corners, padding_mode)


class MaskRows:
    """``cat([masks_b[rows_b] for b in images])`` -- the matched ground-truth masks of a prediction set
    (mask2former_nusc_occ.py:253-262) -- without the copy: parts = [(masks_b [R_b, X, Y, Z] float, rows_b int64 [n_b])].
    Materialised, this was a 33 MB row gather plus a 33 MB concatenation per prediction set at the 200-grid (0.24 ms,
    ten times per training step), only to be read once by a point sampler."""

    def __init__(self, parts):
        self.parts = list(parts)

    @property
    def shape(self):
        return (sum(int(r.shape[0]) for _, r in self.parts),) + tuple(self.parts[0][0].shape[1:])

    @staticmethod
    def cat(items):
        return MaskRows([p for it in items for p in it.parts])

    def dense(self):
        return torch.cat([m[r] for m, r in self.parts], 0)

    def sample(self, coords, align_corners=False, padding_mode="zeros"):
        """point_sample_3d(dense().unsqueeze(1).float(), coords).squeeze(1); coords [n, P, 3] in grid_sample order"""
        out, r0 = [], 0
        for m, r in self.parts:
            n = int(r.shape[0])
            out.append(get_ops().point_sample_3d_rows(m.float().contiguous(), r, coords[r0:r0 + n], align_corners,
                                                      padding_mode))
            r0 += n
        return out[0] if len(out) == 1 else torch.cat(out, 0)

    def gather(self, idx):
        """torch.gather(dense().reshape(n, -1), 1, idx); idx [n, K] voxel indices"""
        out, r0 = [], 0
        for m, r in self.parts:
            n = int(r.shape[0])
            out.append(m.reshape(m.shape[0], -1)[r.unsqueeze(1), idx[r0:r0 + n]])
            r0 += n
        return out[0] if len(out) == 1 else torch.cat(out, 0)


def unravel_indices(indices, shape):
    """mmdet_utils.py:71-89"""
    coord = []
    for dim in reversed(shape):
        coord.append(indices % dim)
        indices = torch.div(indices, dim, rounding_mode="floor")
    return torch.stack(coord[::-1], dim=-1)


def gt_label_scan(gt_occ, num_classes):
    """Device-only half of the ground-truth conversion: which labels < num_classes occur in ``gt_occ`` [1, X, Y, Z].
    -> (labels_sorted [num_classes] int64: the present labels in ascending order first, n_present [] int64).
    No data-dependent shape, hence no host synchronisation: it can run on a side stream one step ahead
    (``OccupancyFormer.prefetch_gt``)."""
    # (one comparison per class, not a scatter-add histogram: 640 000 atomics onto 17 counters took 0.93 ms)
    ar = torch.arange(num_classes, device=gt_occ.device)
    eq = gt_occ.reshape(1, -1) == ar.to(gt_occ.dtype).view(-1, 1)                   # [classes, V]
    V = eq.shape[1]
    chunk = 1024
    if V % chunk == 0 and V > chunk:
        # two stages: a row-wise any() over 640 000 columns runs on `classes` workgroups (0.86 ms for 17 rows)
        present = eq.view(num_classes, V // chunk, chunk).any(2).any(1)
    else:
        present = eq.any(1)
    labels_sorted = torch.sort(torch.where(present, ar, torch.full_like(ar, num_classes)))[0]
    return labels_sorted, present.sum()


def preprocess_occupancy_gt(gt_occ, num_classes, img_metas=None, scan=None, n_present=None, mask_dtype=torch.long):
    """mmdet_utils.py:426-475: labels present (< num_classes) and their 0/1 int64 masks.  The reference takes
    ``torch.unique(gt_occ)`` -- sorted values of a data-dependent count, i.e. a host synchronisation; here the count is
    the ONLY thing read back (``n_present``: already on the host when the scan was prefetched)."""
    if scan is None:
        scan = gt_label_scan(gt_occ, num_classes)
    if n_present is None:
        n_present = int(scan[1])                      # host synchronisation (one integer)
    assert n_present > 0
    labels = scan[0][:n_present]
    g = gt_occ.squeeze(0) if gt_occ.dim() == 4 and gt_occ.shape[0] == 1 else gt_occ
    # (``mask_dtype``: the reference's masks are int64; the training step only ever reads them as fp32 -- built directly,
    # the int64 copy and its conversion (0.6 ms at the 200-grid) never exist)
    if mask_dtype.is_floating_point and labels.numel() > 0:
        # one scatter of V ones into a zero-filled [G + 1, V] (row = rank of the voxel's label among the present labels,
        # G = "not a present label"): the broadcast comparison + bool -> float conversion took 0.3 ms at the 200-grid
        flat = g.reshape(-1).to(labels.dtype)
        G = labels.shape[0]
        rank = torch.searchsorted(labels, flat).clamp_(max=G - 1)
        row = torch.where(labels[rank] == flat, rank, torch.full_like(rank, G))
        m = torch.zeros((G + 1, flat.shape[0]), dtype=mask_dtype, device=g.device)
        m.scatter_(0, row.unsqueeze(0), 1.0)
        return labels.long(), m[:G].view(G, *g.shape)
    return labels.long(), (g.unsqueeze(0) == labels.view(-1, *([1] * g.dim()))).to(mask_dtype)


def _voxel_weights(gt_labels, gt_masks, sample_weights):
    sw = torch.as_tensor(np.asarray(sample_weights), dtype=torch.float32, device=gt_masks.device)
    return (sw[gt_labels].view(-1, 1) * gt_masks.reshape(gt_masks.shape[0], -1).float()).sum(0)


def _norm_coords(idx, dims, like):
    return unravel_indices(idx, dims).float() / (dev_const(dims, idx.device, torch.float32).view(1, 1, -1) - 1)


def sample_valid_coords_with_frequencies(num_points, gt_labels, gt_masks, sample_weights, rng):
    """mmdet_utils.py:91-108 -> (indices [num_points], coords [1, num_points, 3])"""
    w = _voxel_weights(gt_labels, gt_masks, sample_weights)
    # (the reference's weights are float64 here, so that is the dtype the noise is drawn in)
    q = rng.exponential((w.numel(),), torch.float64).float().reshape(1, -1).to(w.device)
    idx = get_ops().sample_without_replacement(w[None].contiguous(), q.contiguous(), num_points, exponential=True)[0]
    return idx, _norm_coords(idx[None], gt_masks.shape[1:], gt_masks)


def batch_sample_valid_coords_with_frequencies(num_points, gt_labels_list, gt_masks_list, sample_weights, rng):
    """mmdet_utils.py:110-136 -> (indices [n_total_gt, num_points], coords [n_total_gt, num_points, 3]);
    every GT row of an image shares that image's voxel weights (weights_shared launch, no repeat)"""
    ops = get_ops()
    n_total = sum(int(g.shape[0]) for g in gt_labels_list)
    V = gt_masks_list[0][0].numel()
    q = rng.exponential((n_total, V)).to(gt_masks_list[0].device)
    out, r0 = [], 0
    for gl, gm in zip(gt_labels_list, gt_masks_list):
        n = int(gl.shape[0])
        if n:
            w = _voxel_weights(gl, gm, sample_weights)
            out.append(ops.sample_without_replacement(w[None].contiguous(), q[r0:r0 + n].contiguous(), num_points,
                                                      exponential=True))
        r0 += n
    idx = torch.cat(out, 0)
    return idx, _norm_coords(idx, gt_masks_list[-1].shape[1:], gt_masks_list[-1])


def get_uncertain_point_coords_3d_with_frequency(mask_pred, labels, gt_labels_list, gt_masks_list, sample_weights,
                                                 num_points, oversample_ratio, importance_sample_ratio, rng,
                                                 align_corners=True):
    """mmdet_utils.py:179-246.  mask_pred [n_pos, 1, X, Y, Z]"""
    ops = get_ops()
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    n = mask_pred.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    idx, coords = batch_sample_valid_coords_with_frequencies(num_sampled, gt_labels_list, gt_masks_list,
                                                             sample_weights, rng)
    if tuple(mask_pred.shape[-3:]) == tuple(gt_masks_list[0].shape[1:]):
        logits = torch.gather(mask_pred.reshape(n, -1), 1, idx)
    else:
        logits = point_sample_3d(mask_pred, coords.flip(-1), align_corners=True).squeeze(1)
    n_unc = int(importance_sample_ratio * num_points)
    top = ops.topk_smallest_abs(logits.contiguous(), n_unc)
    idx = torch.gather(idx, 1, top)
    coords = torch.gather(coords, 1, top[..., None].expand(-1, -1, 3))
    if num_points - n_unc > 0:
        ridx, rcoords = batch_sample_valid_coords_with_frequencies(num_points - n_unc, gt_labels_list, gt_masks_list,
                                                                   np.ones_like(np.asarray(sample_weights)), rng)
        idx, coords = torch.cat((idx, ridx), 1), torch.cat((coords, rcoords), 1)
    return idx, coords


def get_nusc_lidarseg_point_coords(mask_pred, gt_lidarseg_list, labels, num_points, oversample_ratio,
                                   importance_sample_ratio, point_cloud_range, rng, padding_mode="border"):
    """mmdet_utils.py:138-177.  mask_pred [n_pos, 1, X, Y, Z] ordered image by image; the GT rows of one image
    share its candidate points, so the logits of an image come from ONE shared-points launch."""
    ops = get_ops()
    assert oversample_ratio >= 1 and 0 <= importance_sample_ratio <= 1
    n_pos = mask_pred.shape[0]
    num_sampled = int(num_points * oversample_ratio)
    pcr = dev_const(point_cloud_range, mask_pred.device, mask_pred.dtype)
    n_unc = int(importance_sample_ratio * num_points)
    out, r0 = [], 0
    for lidar, lab in zip(gt_lidarseg_list, labels):
        n = int(lab.shape[0])
        c = (lidar[:, :3].to(pcr) - pcr[:3]) / (pcr[3:] - pcr[:3])
        c = torch.cat((c, rng.rand(num_sampled - c.shape[0], 3).to(c)), 0)
        if n:
            vol = mask_pred[r0:r0 + n, 0][None]                                     # [1, n, X, Y, Z]
            logits = ops.point_sample_3d(vol.contiguous(), c[None].flip(-1).contiguous(), False, padding_mode)[0]
            top = ops.topk_smallest_abs(logits.contiguous(), n_unc)                  # [n, n_unc]
            out.append(c[top])
        r0 += n
    coords = torch.cat(out, 0)
    if num_points - n_unc > 0:
        coords = torch.cat((coords, rng.rand(n_pos, num_points - n_unc, 3).to(coords)), 1)
    return coords


# ------------------------------------------------------------------------------------------ matching
class MaskHungarianAssigner:
    """mask_hungarian_assigner.py:12-126 with ClassificationCost (mmdet), CrossEntropyLossCost
    (match_cost.py:69-128) and DiceCost (:9-66) evaluated from one GEMM:
        [x ; sigmoid(x)] [2Q, P] @ gt^T [P, G]  ->  x.gt  and  sigmoid(x).gt
        BCE cost  = (sum softplus(x) - x.gt) / P          (pos - neg = -x)
        dice cost = 1 - (2 sigmoid(x).gt + eps) / (sum sigmoid(x) + sum gt + eps)"""

    def __init__(self, cls_cost=None, mask_cost=None, dice_cost=None):
        self.w_cls = float((cls_cost or {}).get("weight", 1.0))
        self.w_mask = float((mask_cost or {}).get("weight", 1.0))
        dc = dice_cost or {}
        self.w_dice = float(dc.get("weight", 1.0))
        self.dice_eps = float(dc.get("eps", 1e-3))
        self.infeasible = None
        if not dc.get("pred_act", False) or not dc.get("naive_dice", True):
            raise NotImplementedError("DiceCost: pred_act=True / naive_dice=True is the configuration built")

    def cost(self, cls_pred, mask_pred, gt_labels, gt_mask):
        ops = get_ops()
        x = mask_pred.flatten(1).float().contiguous()
        g = gt_mask.flatten(1).float().contiguous()
        Q, P = x.shape
        rows = ops.point_loss_rows(x, torch.zeros_like(x))          # [:, 0] = sum softplus(x), [:, 2] = sum sigmoid
        a = torch.cat((x, x.sigmoid()), 0)
        if P % 4:
            a, g2 = F.pad(a, (0, 4 - P % 4)), F.pad(g, (0, 4 - P % 4))
        else:
            g2 = g
        g2 = g2.contiguous()
        # K = the sampled points (12 544): the bf16-split kernel slices K over the chip (4 tiles would otherwise
        # walk it serially); 0/1 targets are exact in bf16, the logits keep 16 mantissa bits
        sp = ops.split_bf16(g2) if ops.precision != "f32" and g2.shape[1] % 32 == 0 else None
        prod = ops.linear(a.contiguous(), g2, allow_small=False, w_split=sp)   # [2Q, G]
        xg, sg = prod[:Q], prod[Q:]
        bce = (rows[:, 0:1] - xg) / P
        dice = 1 - (2 * sg + self.dice_eps) / (rows[:, 2:3] + g.sum(1)[None] + self.dice_eps)
        cls = -cls_pred.softmax(-1)[:, gt_labels]
        return cls * self.w_cls + bce * self.w_mask + dice * self.w_dice

    def assign(self, cls_pred, mask_pred, gt_labels, gt_mask, img_meta=None):
        """-> (assigned_gt_inds [Q] (0 = background, g+1 = matched to GT g), cost).  The assignment runs on the
        device (csrc/assign.hip) -- the reference's ``cost.cpu()`` + scipy round trip, one host sync per
        prediction set, is gone; ``self.last_match`` [G] holds the query of every GT row."""
        Q, G = mask_pred.shape[0], gt_labels.shape[0]
        self.last_match = None
        if G == 0 or Q == 0:
            return torch.zeros((Q,), dtype=torch.long, device=mask_pred.device), mask_pred.new_zeros((Q, G))
        cost = self.cost(cls_pred, mask_pred, gt_labels, gt_mask)
        match, assigned = get_ops().hungarian(cost)
        if G <= Q:
            # every GT row must have found a query; -1 = the row was infeasible (NaN / inf costs: scipy raises
            # "cost matrix is infeasible" there).  Recorded on the device, raised by the head one step later;
            # the index is clamped so that the targets of this (already meaningless) step stay in bounds
            bad = (match < 0).any()
            self.infeasible = bad if self.infeasible is None else (self.infeasible | bad)
            match = match.clamp_min(0)
        self.last_match = match.long()
        return assigned.long(), cost


# ------------------------------------------------------------------------------------------ losses
def cross_entropy_loss(cls_scores, labels, label_weights, class_weight, avg_factor, loss_weight):
    """mmdet 2.14.0 CrossEntropyLoss(use_sigmoid=False, class_weight, reduction='mean') with avg_factor"""
    loss = F.cross_entropy(cls_scores, labels, weight=class_weight, reduction="none")
    return loss_weight * (loss * label_weights.float()).sum() / avg_factor


def point_mask_losses(point_preds, point_targets, mask_weights, num_points, dice_eps, w_mask, w_dice,
                      weight_bce_rows):
    """BCE (mmdet CrossEntropyLoss use_sigmoid) and naive Dice (dice_loss.py:8-61) over sampled points from one
    pass of row sums.  ``weight_bce_rows``: KITTI weights the BCE rows by the class weight
    (mask2former_occ.py:433-442); nuScenes divides by sum(w)*P only (mask2former_nusc_occ.py:411-417)."""
    if point_preds.requires_grad and torch.is_grad_enabled():
        rows = A.PointLossRows.apply(point_preds, point_targets.float())
    else:
        rows = get_ops().point_loss_rows(point_preds.contiguous(), point_targets.float().contiguous())
    # reduce_mean(mask_weights.sum()) across ranks (mask2former_nusc_occ.py:408; mask2former_occ.py:425,437)
    total = dist_utils.reduce_mean(mask_weights.sum().detach()).clamp_min(1e-12)
    d = (2 * rows[:, 1] + dice_eps) / (rows[:, 2] + rows[:, 3] + dice_eps)
    loss_dice = w_dice * ((1 - d) * mask_weights).sum() / total
    if weight_bce_rows:
        loss_mask = w_mask * (rows[:, 0] * mask_weights).sum() / (total * num_points)
    else:
        loss_mask = w_mask * rows[:, 0].sum() / (total * num_points)
    return loss_mask, loss_dice


def fast_hist_crop(output, target, unique_label):
    """P/../utils/metric_util.py: confusion matrix restricted to unique_label + 1"""
    n = int(np.max(unique_label)) + 2
    k = (target >= 0) & (target < n)
    hist = np.bincount(n * target[k].astype(int) + output[k], minlength=n ** 2).reshape(n, n)
    hist = hist[unique_label + 1, :]
    return hist[:, unique_label + 1]


def per_class_iu(hist):
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.diag(hist) / (hist.sum(1) + hist.sum(0) - np.diag(hist))


def fast_hist_crop_device(output, target, num_labels):
    """fast_hist_crop(output, target, arange(num_labels)) on the tensors' own device, int64 counts, with no
    data-dependent shape (so no host synchronisation): the validity mask becomes the increment of a scatter-add
    instead of a boolean selection."""
    n = num_labels + 1                                   # = max(unique_label) + 2
    valid = (target >= 0) & (target < n) & (output >= 0) & (output < n)
    idx = torch.where(valid, n * target + output, torch.zeros_like(target))
    hist = torch.zeros(n * n, dtype=torch.int64, device=target.device)
    hist.scatter_add_(0, idx, valid.long())
    return hist.view(n, n)[1:, 1:]


def mean_iou_device(hist):
    """nanmean(per_class_iu(hist)) in float64 on the device (0/0 -> nan -> ignored, as numpy's nanmean)"""
    h = hist.double()
    d = torch.diagonal(h)
    return torch.nanmean(d / (h.sum(1) + h.sum(0) - d))


# ------------------------------------------------------------------------------------------ head mixin
class OccHeadTrainingMixin:
    """loss / target code shared by the two heads; the head provides num_queries, num_classes, class_weight,
    loss weights and train_cfg fields (set up by ``_init_training``)."""

    def _init_training(self, train_cfg, loss_cls, loss_mask, loss_dice):
        self.train_cfg = train_cfg
        self.class_weight = list((loss_cls or {}).get("class_weight", [1.0] * self.num_classes + [0.1]))
        self.w_cls = float((loss_cls or {}).get("loss_weight", 1.0))
        self.w_mask = float((loss_mask or {}).get("loss_weight", 1.0))
        self.w_dice = float((loss_dice or {}).get("loss_weight", 1.0))
        self.dice_eps = float((loss_dice or {}).get("eps", 1e-3))
        ld, lm = loss_dice or {}, loss_mask or {}
        if not ld.get("naive_dice", True) or not ld.get("activate", True) or not ld.get("use_sigmoid", True) or \
                ld.get("reduction", "mean") != "mean" or not lm.get("use_sigmoid", True) or \
                lm.get("reduction", "mean") != "mean" or (loss_cls or {}).get("use_sigmoid", False):
            raise NotImplementedError("the loss kernels implement the OccFormer configs' options: DiceLoss("
                                      "use_sigmoid, activate, naive_dice, mean), BCE mask loss (use_sigmoid, mean), "
                                      "softmax CE classification loss")
        self.rng = None
        self._infeasible_pending = None
        if train_cfg:
            a = dict(train_cfg["assigner"])
            a.pop("type", None)
            self.assigner = MaskHungarianAssigner(**a)
            self.num_points = train_cfg.get("num_points", 12544)
            self.oversample_ratio = train_cfg.get("oversample_ratio", 3.0)
            self.importance_sample_ratio = train_cfg.get("importance_sample_ratio", 0.75)

    def _rng(self, device):
        if self.rng is not None:
            return self.rng
        from . import noise
        return noise.get_rng(device)

    def _const(self, name, values, device, dtype):
        """small per-head constants (class weights, point-cloud range) as device tensors, uploaded once: a
        ``torch.tensor(list, device=...)`` per prediction set is a pageable host-to-device copy each time (30 per step)"""
        return dev_const(values, device, dtype)

    def preprocess_gt(self, gt_occ, img_metas, scans=None, mask_dtype=torch.long):
        """``scans``: per sample (labels_sorted, n_present on the HOST) from ``gt_label_scan`` run ahead of time"""
        pairs = [preprocess_occupancy_gt(g, self.num_occupancy_classes, scan=None if scans is None else (scans[i][0], None),
                                         n_present=None if scans is None else scans[i][1], mask_dtype=mask_dtype)
                 for i, g in enumerate(gt_occ)]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    def _targets_from_assignment(self, gt_inds, cls_score, mask_pred, gt_labels, gt_masks):
        # samplers/mask_pseudo_sampler.py: positives = matched queries in ascending order
        match = getattr(self.assigner, "last_match", None)
        if match is not None and match.shape[0] <= gt_inds.shape[0]:
            # every GT row is matched (G <= Q): the positives are the sorted matched queries -- the same
            # tensors as nonzero(gt_inds > 0) / gt_inds[pos] - 1, without the device-to-host size query
            pos, pos_gt = torch.sort(match)
        else:
            pos = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1)
            pos_gt = gt_inds[pos] - 1
        labels = gt_labels.new_full((self.num_queries,), self.num_classes, dtype=torch.long)
        labels[pos] = gt_labels[pos_gt]
        cw = self._const("class_weight", self.class_weight, cls_score.device, cls_score.dtype)
        mask_weights = cls_score.new_zeros((self.num_queries,))
        mask_weights[pos] = cw[labels[pos]]
        return labels, torch.ones_like(mask_weights), MaskRows([(gt_masks, pos_gt)]), mask_weights, pos, pos_gt

    def loss(self, all_cls_scores, all_mask_preds, *gt):
        """mask2former_nusc_occ.py:275-315 / mask2former_occ.py:294-341"""
        # the GT masks are sampled as fp32 volumes by every prediction set: convert the int64 masks once per
        # step instead of once per set (17 x 2 M voxels each time)
        gt = list(gt)
        gt[1] = [m.float() for m in gt[1]]
        self._raise_if_infeasible()
        # per set: targets, point coordinates and point targets (no gradient; the noise draws keep the order of the
        # reference's sequential loss_single calls); then the point logits of ALL sets in one joint sampling call;
        # then the loss arithmetic per set
        preps = [self._loss_prepare(c, m, *gt) for c, m in zip(all_cls_scores, all_mask_preds)]
        live = [i for i, p in enumerate(preps) if p["coords"] is not None]
        pps = sample_logits_sets([(preps[i]["mp"], preps[i]["coords"]) for i in live], *self._sample_mode())
        per = [None] * len(preps)
        for i, pp in zip(live, pps):
            per[i] = self._loss_finish(preps[i], pp)
        for i, p in enumerate(preps):
            if per[i] is None:
                per[i] = self._loss_finish(p, None)
        self._post_infeasible_flag()
        out = {"loss_cls": per[-1][0], "loss_mask": per[-1][1], "loss_dice": per[-1][2]}
        for i, (a, b, c) in enumerate(per[:-1]):
            out[f"d{i}.loss_cls"], out[f"d{i}.loss_mask"], out[f"d{i}.loss_dice"] = a, b, c
        return out

    def _post_infeasible_flag(self):
        """copy the assigner's device flag to pinned host memory without blocking; read it at the next step"""
        flag = getattr(self.assigner, "infeasible", None)
        self.assigner.infeasible = None
        if flag is None:
            return
        if flag.is_cuda:
            host = torch.empty((), dtype=torch.bool, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._infeasible_pending = (host, ev)
        else:
            self._infeasible_pending = (flag, None)

    def _raise_if_infeasible(self):
        pend, self._infeasible_pending = getattr(self, "_infeasible_pending", None), None
        if pend is None:
            return
        if pend[1] is not None:
            pend[1].synchronize()
        if bool(pend[0]):
            raise FloatingPointError("Hungarian assignment: non-finite matching costs in the previous training step "
                                     "(scipy.optimize.linear_sum_assignment raises 'cost matrix is infeasible')")

    def _cls_and_select(self, cls_scores, mask_preds, targets):
        labels = torch.stack([t[0] for t in targets]).flatten()
        label_weights = torch.stack([t[1] for t in targets]).flatten()
        mask_targets = MaskRows.cat([t[2] for t in targets])
        mask_weights = torch.stack([t[3] for t in targets])
        cw = self._const("class_weight", self.class_weight, cls_scores.device, cls_scores.dtype)
        loss_cls = cross_entropy_loss(cls_scores.flatten(0, 1), labels, label_weights, cw, cw[labels].sum(),
                                      self.w_cls)
        if min(self.class_weight[:self.num_classes]) > 0:
            # positive class weights: (mask_weights > 0) selects exactly the matched queries, image by image in
            # ascending order = targets[b][4]; gather by index (no boolean-mask size query on the host)
            B, Q = mask_weights.shape
            idx = torch.cat([t[4] + b * Q for b, t in enumerate(targets)])
            if isinstance(mask_preds, LazyMask):
                return loss_cls, mask_preds.rows([t[4] for t in targets]), mask_weights.flatten()[idx], mask_targets
            return loss_cls, mask_preds.flatten(0, 1)[idx], mask_weights.flatten()[idx], mask_targets
        if isinstance(mask_preds, LazyMask):
            raise NotImplementedError("zero class weights with lazy mask logits")
        sel = mask_weights > 0
        return loss_cls, mask_preds[sel], mask_weights[sel], mask_targets


class NuscTrainingMixin(OccHeadTrainingMixin):
    def _get_target_single(self, cls_score, mask_pred, gt_labels, gt_masks, gt_lidarseg, img_metas=None):
        """mask2former_nusc_occ.py:196-273"""
        rng = self._rng(cls_score.device)
        gt_labels = gt_labels.long()
        pcr = self._const("pcr", self.point_cloud_range, cls_score.device, torch.float32)
        coords = (gt_lidarseg[:, :3].float() - pcr[:3]) / (pcr[3:] - pcr[:3])
        n_lidar = min(self.num_points // 2, coords.shape[0])
        if n_lidar < coords.shape[0]:
            coords = coords[rng.randperm(coords.shape[0]).to(coords.device)[:n_lidar]]
        coords = torch.cat((coords, rng.rand(self.num_points - n_lidar, 3).to(coords)), 0).flip(-1)
        lazy = mask_pred if isinstance(mask_pred, LazyMask) else None
        cls_score = cls_score.detach()                                       # targets carry no gradient
        mask_pred = None if lazy is not None else _dense(mask_pred)
        if lazy is not None:
            pred_pts = lazy.sample_all(coords, False, self.padding_mode)
        else:
            pred_pts = point_sample_3d(mask_pred[None], coords[None], padding_mode=self.padding_mode)[0]
        if gt_labels.shape[0]:
            gt_pts = point_sample_3d(gt_masks[None].float(), coords[None], padding_mode=self.padding_mode)[0]
        else:
            gt_pts = pred_pts.new_zeros((0, coords.shape[0]))
        gt_inds, cost = self.assigner.assign(cls_score, pred_pts, gt_labels, gt_pts, img_metas)
        return self._targets_from_assignment(gt_inds, cls_score, mask_pred, gt_labels, gt_masks) + (cost,)

    def _sample_mode(self):
        return False, self.padding_mode

    # ------------------------------------------------------------------ all prediction sets at once
    batched_loss = os.environ.get("OCCF_BATCHED_LOSS", "1") == "1"

    def loss(self, all_cls_scores, all_mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, img_metas=None):
        """mask2former_nusc_occ.py:275-315.  One image per rank (the reference's samples_per_gpu = 1) with every ground-
        truth row matchable: the S prediction sets go through ``_loss_sets`` together; anything else takes the
        set-by-set path of the base class (same values: tests/test_training.py)."""
        if self.batched_loss and len(gt_labels_list) == 1 and all_cls_scores[0].shape[0] == 1 and \
                all(isinstance(m, LazyMask) for m in all_mask_preds) and \
                0 < int(gt_labels_list[0].shape[0]) <= self.num_queries and \
                min(self.class_weight[:self.num_classes]) > 0:
            gt_masks = [m.float() for m in gt_masks_list]
            self._raise_if_infeasible()
            out = self._loss_sets(all_cls_scores, all_mask_preds, gt_labels_list[0].long(), gt_masks[0],
                                  gt_lidarseg_list[0])
            self._post_infeasible_flag()
            return out
        if self.batched_loss and len(gt_labels_list) == 1 and int(gt_labels_list[0].shape[0]) == 0 and \
                dist_utils.world() > 1:
            # a sample without ground-truth rows takes the zero-match exit of the set-by-set path, which (as in the
            # reference, mask2former_nusc_occ.py:381-385) never reaches the normaliser's all-reduce; the other ranks
            # issue ONE all-reduce over the S sets in ``_loss_sets`` -- join it, so that the collectives of the ranks match
            dist_utils.reduce_mean(all_cls_scores[0].new_zeros((len(all_mask_preds),)))
        return super().loss(all_cls_scores, all_mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, img_metas)

    def _loss_sets(self, all_cls_scores, all_mask_preds, gt_labels, gt_masks, lidar):
        """The set-by-set loss (``_get_target_single`` -> ``_cls_and_select`` -> ``get_nusc_lidarseg_point_coords`` ->
        point targets -> point losses, ten times) with the SET as a batch dimension of every kernel and formula: the
        matching / target / sampling arithmetic of one set is ~130 launches of 5-microsecond kernels on 100 x 17
        operands (1 340 launches, 21 ms of a profiled step for the ten sets).  The noise is drawn FIRST, in the
        reference's order (per set: assignment points, oversampled candidates, random points), so the draws are those of
        the sequential formulation; no shape depends on a matching result (every GT row is matched: G <= Q)."""
        ops = get_ops()
        S = len(all_mask_preds)
        dev = all_cls_scores[0].device
        rng = self._rng(dev)
        Q, P, G = self.num_queries, self.num_points, int(gt_labels.shape[0])
        P3 = int(P * self.oversample_ratio)
        n_unc = int(self.importance_sample_ratio * P)
        pad = self.padding_mode
        f32 = torch.float32
        lazies = [m[0] for m in all_mask_preds]
        feat = lazies[0].feat_tok
        vol_shape = lazies[0].vol_shape
        with torch.no_grad():
            pcr = self._const("pcr", self.point_cloud_range, dev, f32)
            lc = (lidar[:, :3].float() - pcr[:3]) / (pcr[3:] - pcr[:3])                      # [L, 3] in [0, 1]
            L = int(lc.shape[0])
            n_lidar = min(P // 2, L)
            # ---- noise, in the order of the sequential formulation
            draws = []
            for _ in range(S):
                perm = rng.randperm(L).to(dev)[:n_lidar] if n_lidar < L else None
                r1 = rng.rand(P - n_lidar, 3).to(lc)
                r2 = rng.rand(P3 - L, 3).to(lc)
                r3 = rng.rand(G, P - n_unc, 3).to(lc) if P - n_unc > 0 else None
                draws.append((perm, r1, r2, r3))
            # ---- matching points: [S, P, 3] in grid_sample order
            mpts = torch.stack([torch.cat((lc if d[0] is None else lc[d[0]], d[1]), 0) for d in draws]).flip(-1)
            mpts = mpts.contiguous()
            fs = ops.point_sample_tokens(feat.detach(), vol_shape, mpts.view(S * P, 3), False, pad)       # [S*P, E]
            Pp = (P + 3) // 4 * 4
            a = torch.empty((S, 2 * Q, Pp), dtype=f32, device=dev) if Pp == P else None      # [x ; sigmoid(x)]
            x = torch.empty((S, Q, P), dtype=f32, device=dev)
            for s in range(S):
                ops.linear(lazies[s].embed.detach().contiguous(), fs[s * P:(s + 1) * P], None, allow_small=False,
                           out=x[s])
            gt_pts = ops.point_sample_3d(gt_masks[None].contiguous(), mpts.view(1, S * P, 3), False, pad)[0]  # [G, S*P]
            g = gt_pts.view(G, S, P).permute(1, 0, 2).contiguous()                                # [S, G, P]
            # ---- matching cost of all sets (MaskHungarianAssigner.cost, batched)
            asg = self.assigner
            rows = ops.point_loss_rows(x.view(S * Q, P), torch.zeros_like(x).view(S * Q, P)).view(S, Q, -1)
            if a is not None:
                a[:, :Q].copy_(x)
                torch.sigmoid(x, out=a[:, Q:])
                g2 = g
            else:
                a = torch.cat((x, x.sigmoid()), 1)                                                # [S, 2Q, P]
                a, g2 = F.pad(a, (0, 4 - P % 4)), F.pad(g, (0, 4 - P % 4))
            a, g2 = a.contiguous(), g2.contiguous()
            sp = None
            if ops.precision != "f32" and g2.shape[-1] % 32 == 0:
                hi, lo = ops.split_bf16(g2.view(S * G, -1))
                sp = (hi.view(S, G, -1), lo.view(S, G, -1))
            prod = torch.empty((S, 2 * Q, G), dtype=f32, device=dev)
            for s in range(S):
                ops.linear(a[s], g2[s], allow_small=False, w_split=None if sp is None else (sp[0][s], sp[1][s]),
                           out=prod[s])
            xg, sg = prod[:, :Q], prod[:, Q:]
            bce = (rows[:, :, 0:1] - xg) / P
            dice = 1 - (2 * sg + asg.dice_eps) / (rows[:, :, 2:3] + g.sum(2)[:, None] + asg.dice_eps)
            cls_all = torch.cat([c.detach() for c in all_cls_scores], 0)                          # [S, Q, C + 1]
            ccost = -cls_all.softmax(-1)[:, :, gt_labels]
            cost = ccost * asg.w_cls + bce * asg.w_mask + dice * asg.w_dice
            match, _ = ops.hungarian(cost)                                                       # [S, G] query of GT g
            bad = (match < 0).any()
            asg.infeasible = bad if asg.infeasible is None else (asg.infeasible | bad)
            pos, pos_gt = torch.sort(match.clamp_min(0).long(), dim=1)                           # ascending queries
            # ---- targets
            cw = self._const("class_weight", self.class_weight, dev, f32)
            lab_pos = gt_labels[pos_gt]                                                           # [S, G]
            labels = torch.full((S, Q), self.num_classes, dtype=torch.long, device=dev).scatter_(1, pos, lab_pos)
            mw = cw[lab_pos]                                                                      # [S, G]
        cls_flat = torch.cat(list(all_cls_scores), 0).flatten(0, 1)                              # [S*Q, C + 1], grad
        ce = F.cross_entropy(cls_flat, labels.flatten(), weight=cw, reduction="none").view(S, Q)
        loss_cls = self.w_cls * ce.sum(1) / cw[labels].sum(1)                                     # (label_weights = 1)
        with torch.no_grad():
            # ---- matched rows of every set: dense logits [S, G, X, Y, Z]
            rows_e = None
            if all(lz._dense is None for lz in lazies):
                # lazy logits: the matched rows of ALL sets from one contraction (the mask features are read once)
                rows_e = torch.cat([lazies[s].embed.detach()[pos[s]] for s in range(S)], 0)            # [S*G, E]
                dense = lazies[0]._contract(rows_e, feat, lazies[0].feat_split).view((S, G) + tuple(vol_shape))
            else:
                dense = torch.empty((S, G) + tuple(vol_shape), dtype=f32, device=dev)
                for s in range(S):
                    lz = lazies[s]
                    if lz._dense is not None:
                        torch.index_select(lz._dense, 0, pos[s], out=dense[s])
                    else:
                        dense[s] = lz._contract(lz.embed[pos[s]], lz.feat_tok, lz.feat_split)
            # ---- importance sampling of the point coordinates (get_nusc_lidarseg_point_coords, batched)
            cand = torch.stack([torch.cat((lc, d[2]), 0) for d in draws])                        # [S, P3, 3]
            logits = None
            if rows_e is not None:
                logits = _candidate_logits_voxel_major(ops, rows_e, feat, S, G, vol_shape, cand.flip(-1).contiguous(), pad)
            if logits is None:
                logits = ops.point_sample_3d(dense, cand.flip(-1).contiguous(), False, pad)       # [S, G, P3]
            top = ops.topk_smallest_abs(logits.view(S * G, P3), n_unc).view(S, G, n_unc)
            coords = torch.gather(cand[:, None].expand(S, G, P3, 3), 2, top[..., None].expand(S, G, n_unc, 3))
            if P - n_unc > 0:
                coords = torch.cat((coords, torch.stack([d[3] for d in draws])), 2)               # [S, G, P, 3]
            coords = coords.flip(-1).contiguous()
            pt = ops.point_sample_3d_rows(gt_masks.contiguous(), pos_gt.reshape(-1).contiguous(),
                                          coords.view(S * G, P, 3), False, pad)                   # [S*G, P]
        # ---- point logits with the gradient route (one joint node), point losses
        mps = [LazyRows([dense[s]], [lazies[s].embed[pos[s]]], [lazies[s].feat_tok]) for s in range(S)]
        pps = sample_logits_sets([(mps[s], coords[s]) for s in range(S)], False, pad)
        pp = torch.cat(pps, 0)                                                                    # [S*G, P]
        if pp.requires_grad and torch.is_grad_enabled():
            prow = A.PointLossRows.apply(pp, pt)
        else:
            prow = ops.point_loss_rows(pp.contiguous(), pt)
        prow = prow.view(S, G, -1)
        total = dist_utils.reduce_mean(mw.sum(1).detach()).clamp_min(1e-12)                       # [S]
        d = (2 * prow[..., 1] + self.dice_eps) / (prow[..., 2] + prow[..., 3] + self.dice_eps)
        loss_dice = self.w_dice * ((1 - d) * mw).sum(1) / total
        loss_mask = self.w_mask * prow[..., 0].sum(1) / (total * P)
        out = {"loss_cls": loss_cls[-1], "loss_mask": loss_mask[-1], "loss_dice": loss_dice[-1]}
        for i in range(S - 1):
            out[f"d{i}.loss_cls"], out[f"d{i}.loss_mask"], out[f"d{i}.loss_dice"] = loss_cls[i], loss_mask[i], loss_dice[i]
        return out

    def _loss_prepare(self, cls_scores, mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, img_metas=None):
        """mask2former_nusc_occ.py:317-400: targets, classification loss, point coordinates, point targets"""
        with torch.no_grad():
            targets = [self._get_target_single(cls_scores[i], mask_preds[i], gt_labels_list[i], gt_masks_list[i],
                                               gt_lidarseg_list[i]) for i in range(cls_scores.shape[0])]
        loss_cls, mp, mw, mask_targets = self._cls_and_select(cls_scores, mask_preds, targets)
        if mask_targets.shape[0] == 0:
            return dict(loss_cls=loss_cls, coords=None, zero=cls_scores.sum() * 0)
        with torch.no_grad():
            mpd = _dense(mp)
            coords = get_nusc_lidarseg_point_coords(mpd.unsqueeze(1), gt_lidarseg_list, gt_labels_list, self.num_points,
                                                    self.oversample_ratio, self.importance_sample_ratio,
                                                    self.point_cloud_range, self._rng(mpd.device),
                                                    padding_mode=self.padding_mode).flip(-1).contiguous()
            pt = mask_targets.sample(coords, False, self.padding_mode)
        return dict(loss_cls=loss_cls, mp=mp, mw=mw, coords=coords, pt=pt)

    def _loss_finish(self, prep, pp):
        """mask2former_nusc_occ.py:400-424"""
        if prep["coords"] is None:
            return prep["loss_cls"], prep["zero"], prep["zero"]
        loss_mask, loss_dice = point_mask_losses(pp, prep["pt"], prep["mw"], self.num_points, self.dice_eps, self.w_mask,
                                                 self.w_dice, weight_bce_rows=False)
        return prep["loss_cls"], loss_mask, loss_dice

    def loss_single(self, cls_scores, mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, img_metas=None):
        """mask2former_nusc_occ.py:317-424"""
        prep = self._loss_prepare(cls_scores, mask_preds, gt_labels_list, gt_masks_list, gt_lidarseg_list, img_metas)
        pp = None if prep["coords"] is None else sample_logits(prep["mp"], prep["coords"], *self._sample_mode())
        return self._loss_finish(prep, pp)

    def lidarseg_metric(self, cls_preds, mask_preds, points, img_metas):
        """training branch of forward_lidarseg (mask2former_nusc_occ.py:526-540): point mIoU, no gradient"""
        probs = self.forward_lidarseg(cls_preds, mask_preds, points, img_metas)
        out = torch.argmax(probs[:, 1:], 1) + 1
        tgt = torch.cat([p[:, -1] for p in points]).long()
        # confusion matrix and mean IoU stay on the device (the reference round-trips through numpy here)
        return {"point_mean_iou": mean_iou_device(fast_hist_crop_device(out, tgt, 16))}

    def forward_train(self, voxel_feats, img_metas, gt_occ, points=None, **kwargs):
        """mask2former_nusc_occ.py:547-587.  In train() mode the head's forward builds the differentiable graph
        (occformer_amd/autograd.py) and the returned losses carry grad_fn; in eval() mode these are loss VALUES."""
        all_cls, all_masks = self(voxel_feats, img_metas)
        # (``gt_prepared``: the detector converts the ground truth BEFORE it queues the view transformer / encoder --
        # torch.unique has a data-dependent output shape, i.e. a host synchronisation that would otherwise wait here
        # for the whole forward)
        gt_labels, gt_masks = kwargs.get("gt_prepared") or self.preprocess_gt(gt_occ, img_metas)
        losses = self.loss(all_cls, all_masks, gt_labels, gt_masks, points, img_metas)
        with torch.no_grad():
            losses.update(self.lidarseg_metric(all_cls[-1].detach(), _dense(all_masks[-1]), points, img_metas))
        return losses


class KittiTrainingMixin(OccHeadTrainingMixin):
    def get_sampling_weights(self):
        """mask2former_occ.py:144-148, 158-166"""
        base = 1 / semantic_kitti_class_frequencies
        base = base / base.min()
        g = self.sample_weight_gamma
        if isinstance(g, (list, tuple)):
            g = np.random.uniform(low=g[0], high=g[1])
        self.sample_weights = base ** g

    def _get_target_single(self, cls_score, mask_pred, gt_labels, gt_masks, img_metas=None):
        """mask2former_occ.py:224-292"""
        gt_labels = gt_labels.long()
        idx, coords = sample_valid_coords_with_frequencies(self.num_points, gt_labels, gt_masks, self.sample_weights,
                                                           self._rng(cls_score.device))
        lazy = mask_pred if isinstance(mask_pred, LazyMask) else None
        cls_score = cls_score.detach()                                       # targets carry no gradient
        mask_pred = None if lazy is not None else _dense(mask_pred)
        if lazy is not None:
            pred_pts = lazy.sample_all(coords[0].flip(-1), self.align_corners, "zeros")
        else:
            pred_pts = point_sample_3d(mask_pred[None], coords.flip(-1), align_corners=self.align_corners)[0]
        gt_pts = gt_masks.reshape(gt_masks.shape[0], -1)[:, idx].float()
        gt_inds, cost = self.assigner.assign(cls_score, pred_pts, gt_labels, gt_pts, img_metas)
        return self._targets_from_assignment(gt_inds, cls_score, mask_pred, gt_labels, gt_masks) + (cost,)

    def _sample_mode(self):
        return self.align_corners, "zeros"

    def _loss_prepare(self, cls_scores, mask_preds, gt_labels_list, gt_masks_list, img_metas=None):
        """mask2former_occ.py:343-420: targets, classification loss, point coordinates, point targets"""
        with torch.no_grad():
            targets = [self._get_target_single(cls_scores[i], mask_preds[i], gt_labels_list[i], gt_masks_list[i])
                       for i in range(cls_scores.shape[0])]
        loss_cls, mp, mw, mask_targets = self._cls_and_select(cls_scores, mask_preds, targets)
        if mask_targets.shape[0] == 0:
            return dict(loss_cls=loss_cls, coords=None, zero=cls_scores.sum() * 0)
        with torch.no_grad():
            mpd = _dense(mp)
            idx, coords = get_uncertain_point_coords_3d_with_frequency(
                mpd.unsqueeze(1), None, gt_labels_list, gt_masks_list, self.sample_weights, self.num_points,
                self.oversample_ratio, self.importance_sample_ratio, self._rng(mpd.device))
            pt = mask_targets.gather(idx).float()
        return dict(loss_cls=loss_cls, mp=mp, mw=mw, coords=coords.flip(-1).contiguous(), pt=pt)

    def _loss_finish(self, prep, pp):
        """mask2former_occ.py:420-444"""
        if prep["coords"] is None:
            return prep["loss_cls"], prep["zero"], prep["zero"]
        loss_mask, loss_dice = point_mask_losses(pp, prep["pt"], prep["mw"], self.num_points, self.dice_eps, self.w_mask,
                                                 self.w_dice, weight_bce_rows=True)
        return prep["loss_cls"], loss_mask, loss_dice

    def loss_single(self, cls_scores, mask_preds, gt_labels_list, gt_masks_list, img_metas=None):
        """mask2former_occ.py:343-444"""
        prep = self._loss_prepare(cls_scores, mask_preds, gt_labels_list, gt_masks_list, img_metas)
        pp = None if prep["coords"] is None else sample_logits(prep["mp"], prep["coords"], *self._sample_mode())
        return self._loss_finish(prep, pp)

    def forward_train(self, voxel_feats, img_metas, gt_occ, **kwargs):
        """mask2former_occ.py:525-567"""
        self.get_sampling_weights()
        all_cls, all_masks = self(voxel_feats, img_metas)
        gt_labels, gt_masks = kwargs.get("gt_prepared") or self.preprocess_gt(gt_occ, img_metas)
        return self.loss(all_cls, all_masks, gt_labels, gt_masks, img_metas)
