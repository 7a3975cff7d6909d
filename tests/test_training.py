"""Training rows: product (HIP kernels) vs the oracle on IDENTICAL injected noise, and the
oracle vs the reference-generated fixture tests/golden/train.npz (tests/golden/make_golden_train.py)."""
import numpy as np
import pytest
import torch

import occformer_amd.ops as ops_mod
from occformer_amd import training as TR
from occformer_amd.registry import HEADS
from oracle import occformer_train_ref as T
from tests import paramgen, tinycfg
from tests.conftest import golden
from tests.golden.make_golden_train import inputs, kitti_head_cfg, oracle_cfg, train_cfg


class ReplayRNG:
    """feeds the product the very draws the oracle consumed (RecordingRNG.tape), in order"""

    def __init__(self, tape, device):
        self.tape, self.i, self.device = tape, 0, device

    def _next(self, kind, numel):
        k, t = self.tape[self.i]
        self.i += 1
        assert k == kind and t.numel() == numel, f"draw {self.i}: product asks {kind}/{numel}, oracle drew {k}/{t.numel()}"
        return t.to(self.device)

    def rand(self, *shape):
        return self._next("rand", int(np.prod(shape))).reshape(shape)

    def randperm(self, n):
        return self._next("randperm", n)

    def exponential(self, shape, dtype=torch.float32):
        return self._next("exponential", int(np.prod(shape))).reshape(tuple(shape)).float()

    def multinomial(self, weights, k):
        """the oracle's formulation (oracle.occformer_train_ref.GlobalTorchRNG.multinomial) on the replayed draw -- used
        when the ORACLE replays a tape the product recorded"""
        q = self.exponential(weights.shape, weights.dtype)
        return torch.topk(weights / q, k, dim=-1)[1]


@pytest.fixture
def bound(be, monkeypatch):
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    return be


def _heads(kind):
    model, meta = tinycfg.tiny_nusc()
    tc = train_cfg()
    if kind == "nusc":
        hc = dict(model["pts_bbox_head"])
        ocfg = oracle_cfg(hc, tc)
    else:
        hc = kitti_head_cfg(model)
        ocfg = oracle_cfg(hc, tc, align_corners=True,
                          sample_weights=T.kitti_sampling_weights(TR.semantic_kitti_class_frequencies, 0.25))
    head = HEADS.build(dict(hc, train_cfg=tc, test_cfg=None))
    if kind == "kitti":
        ocfg["class_weight"] = head.class_weight
    return head, ocfg, meta


# ---------------------------------------------------------------------------------- oracle vs reference fixture
def test_oracle_reproduces_reference_losses():
    g = golden("train")
    for kind, seed, fn in (("nusc", 11, T.nusc_loss_single), ("kitti", 21, T.kitti_loss_single)):
        head, ocfg, meta = _heads(kind)
        cls, masks, gt_occ, pts = inputs(kind)
        gl, gm = zip(*[T.preprocess_occupancy_gt(o, ocfg["num_classes"]) for o in gt_occ])
        gt = (list(gl), list(gm), pts) if kind == "nusc" else (list(gl), list(gm))
        torch.manual_seed(seed)
        out = T.head_loss(cls, masks, fn, *gt, cfg=ocfg)
        for k, v in out.items():
            assert abs(float(v) - float(g[f"{kind}.{k}"])) < 1e-5 * max(1, abs(float(v))), (kind, k)
    assert np.allclose(np.asarray(_heads("kitti")[0].class_weight), np.asarray(g["kitti.class_weight"]))


def test_oracle_multinomial_is_torch_multinomial():
    w = paramgen.uniform("mn.w", (3, 4000), 9) ** 2
    torch.manual_seed(5)
    a = torch.multinomial(w, 700, replacement=False)
    torch.manual_seed(5)
    assert torch.equal(a, T.GlobalTorchRNG().multinomial(w, 700))


# ---------------------------------------------------------------------------------- product vs oracle
def test_match_cost_and_assignment(bound):
    be = bound
    Q, G, P = 20, 6, 250                       # P not a multiple of 4: exercises the GEMM padding
    cls = paramgen.tensor("mc.cls", (Q, 18), 1, 1.5)
    x = paramgen.tensor("mc.x", (Q, P), 1, 2.0)
    gt = (paramgen.uniform("mc.g", (G, P), 1) < 0.3).float()
    labels = torch.tensor([1, 3, 4, 7, 9, 16])
    cost_o, pos_o, posgt_o = T.hungarian_assign(cls, x, labels, gt)
    a = TR.MaskHungarianAssigner(**{k: v for k, v in train_cfg()["assigner"].items() if k != "type"})
    gt_inds, cost = a.assign(*be.to(cls, x, labels, gt))
    assert torch.allclose(cost.cpu(), cost_o, atol=1e-4, rtol=1e-4)
    pos = torch.nonzero(gt_inds.cpu() > 0).squeeze(-1)
    assert torch.equal(pos, pos_o) and torch.equal(gt_inds.cpu()[pos] - 1, posgt_o)
    # no GT -> everything background
    gi, _ = a.assign(*be.to(cls, x, labels[:0], gt[:0]))
    assert int(gi.abs().sum()) == 0


@pytest.mark.parametrize("kind", ["nusc", "kitti"])
def test_head_loss_matches_oracle_on_injected_noise(bound, kind):
    be = bound
    g = golden("train")
    head, ocfg, meta = _heads(kind)
    cls, masks, gt_occ, pts = inputs(kind)
    gl, gm = zip(*[T.preprocess_occupancy_gt(o, ocfg["num_classes"]) for o in gt_occ])
    rec = T.RecordingRNG()
    torch.manual_seed(11 if kind == "nusc" else 21)
    if kind == "nusc":
        ref = T.head_loss(cls, masks, T.nusc_loss_single, list(gl), list(gm), pts, cfg=ocfg, rng=rec)
    else:
        ref = T.head_loss(cls, masks, T.kitti_loss_single, list(gl), list(gm), cfg=ocfg, rng=rec)
    head.rng = ReplayRNG(rec.tape, be.device)
    d = be.device
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * 2
    gl_p, gm_p = head.preprocess_gt(gt_occ.to(d), metas)
    for a, b, c, e in zip(gl_p, gl, gm_p, gm):
        assert torch.equal(a.cpu(), b) and torch.equal(c.cpu(), e)
    cls_d, masks_d = [c.to(d) for c in cls], [m.to(d) for m in masks]
    if kind == "nusc":
        out = head.loss(cls_d, masks_d, gl_p, gm_p, [p.to(d) for p in pts], metas)
    else:
        out = head.loss(cls_d, masks_d, gl_p, gm_p, metas)
    assert head.rng.i == len(rec.tape), "product and oracle consumed different numbers of draws"
    for k, v in ref.items():
        assert abs(float(out[k]) - float(v)) < 1e-3 * max(1.0, abs(float(v))), (k, float(out[k]), float(v))
        # and therefore the reference's own value (the fixture was produced with the same seed)
        assert abs(float(out[k]) - float(g[f"{kind}.{k}"])) < 1e-3 * max(1.0, abs(float(v))), k


@pytest.mark.parametrize("lazy", [False, True])
def test_nusc_loss_of_all_sets_at_once_equals_set_by_set(bound, lazy):
    """NuscTrainingMixin._loss_sets (the prediction set as a batch dimension of every kernel and formula, noise drawn
    up front in the sequential order) against the set-by-set path on the same noise stream: every loss, and the
    gradients w.r.t. the mask features, every set's mask embeddings and class scores"""
    be = bound
    d = be.device
    head, ocfg, meta = _heads("nusc")
    cls, masks, gt_occ, pts = inputs("nusc")
    S = len(cls)
    _, Q, X, Y, Z = masks[0].shape
    E = 32
    feat = paramgen.tensor("bl_feat", (1, X * Y * Z, E), 1).to(d).requires_grad_()
    embeds = [paramgen.tensor(f"bl_e{s}", (1, Q, E), 2 + s, 0.4).to(d).requires_grad_() for s in range(S)]
    cls_d = [c[:1].to(d).clone().requires_grad_() for c in cls]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]

    def run(batched):
        head.batched_loss = batched
        head.rng = TR.DeviceRNG(d, seed=5)
        # lazy: no dense logits (OCCF_LAZY_LOGITS=1 in the head): the matched rows are contracted on demand
        lm = [TR.LazyMask(None if lazy else (e.detach()[0] @ feat.detach()[0].t()).view(1, Q, X, Y, Z).contiguous(), e,
                          feat, (X, Y, Z)) for e in embeds]
        gl_p, gm_p = head.preprocess_gt(gt_occ[:1].to(d), metas)
        out = head.loss(cls_d, lm, gl_p, gm_p, [pts[0].to(d)], metas)
        grads = torch.autograd.grad(sum(out.values()), [feat] + embeds + cls_d)
        return out, grads, int(head.rng.gen.initial_seed())

    out_b, g_b, _ = run(True)
    out_s, g_s, _ = run(False)
    assert set(out_b) == set(out_s) and len(out_b) == 3 * S
    for k in out_s:
        vb, vs = float(out_b[k].detach()), float(out_s[k].detach())
        assert abs(vb - vs) < 2e-6 * max(1.0, abs(vs)), k
    for a, b in zip(g_b, g_s):
        assert float((a - b).norm()) <= 1e-5 * float(b.norm()) + 1e-9


def test_token_bev_slot_and_label_scan_and_mask_dtype():
    """autograd.TokenBevSlot == (tok, tok[..., Z:Z+1, :]) with the slice gradient added in place; the label scan's
    two-stage reduction and the fp32 mask option of preprocess_occupancy_gt against the plain formulations"""
    from occformer_amd import autograd as A
    tok = paramgen.tensor("tbs", (2, 3, 4, 5, 8), 1).requires_grad_()
    w1, w2 = paramgen.tensor("tbs1", (2, 3, 4, 5, 8), 2), paramgen.tensor("tbs2", (2, 3, 4, 1, 8), 3)
    t, slot = A.TokenBevSlot.apply(tok * 1.0, 4)
    ((t * w1).sum() + (slot * w2).sum()).backward()
    g = tok.grad.clone()
    tok.grad = None
    t2 = tok * 1.0
    ((t2 * w1).sum() + (t2[:, :, :, 4:5] * w2).sum()).backward()
    assert torch.equal(g, tok.grad)
    tok.grad = None
    _, slot = A.TokenBevSlot.apply(tok * 1.0, 4)              # only the slot is used
    (slot * w2).sum().backward()
    assert float(tok.grad[:, :, :, :4].abs().max()) == 0.0 and torch.equal(tok.grad[:, :, :, 4:5], w2)
    gen = torch.Generator().manual_seed(3)
    gt = torch.randint(0, 19, (1, 16, 16, 8), generator=gen)          # 2048 voxels: the chunked reduction
    gt[gt == 18] = 255
    gt[gt == 5] = 6                                                    # an absent label
    labels_sorted, n = TR.gt_label_scan(gt, 17)
    want = torch.unique(gt)
    want = want[want < 17]
    assert int(n) == want.numel() and torch.equal(labels_sorted[:int(n)], want)
    lab_l, m_l = TR.preprocess_occupancy_gt(gt, 17)
    lab_f, m_f = TR.preprocess_occupancy_gt(gt, 17, mask_dtype=torch.float32)
    assert m_l.dtype == torch.long and m_f.dtype == torch.float32
    assert torch.equal(lab_l, lab_f) and torch.equal(m_l.float(), m_f)


def test_kitti_same_resolution_branch(bound):
    """mask logits at the GT resolution: the gather branch of get_uncertain_point_coords_3d_with_frequency"""
    be = bound
    g = golden("train")
    head, ocfg, meta = _heads("kitti")
    cls, masks, gt_occ, _ = inputs("kitti")
    big = torch.nn.functional.interpolate(masks[0], size=(32, 32, 16), mode="trilinear")
    gl, gm = zip(*[T.preprocess_occupancy_gt(o, 20) for o in gt_occ])
    rec = T.RecordingRNG()
    torch.manual_seed(22)
    ref = T.kitti_loss_single(cls[0], big, list(gl), list(gm), ocfg, rec)
    head.rng = ReplayRNG(rec.tape, be.device)
    d = be.device
    out = head.loss_single(cls[0].to(d), big.to(d), [x.to(d) for x in gl], [x.to(d) for x in gm])
    for i, n in enumerate(("loss_cls", "loss_mask", "loss_dice")):
        assert abs(float(out[i]) - float(ref[i])) < 1e-3 * max(1.0, abs(float(ref[i])))
        assert abs(float(out[i]) - float(g["kitti.same." + n])) < 1e-3 * max(1.0, abs(float(ref[i])))


def test_nusc_targets_and_lidarseg_metric(bound):
    be = bound
    g = golden("train")
    head, ocfg, meta = _heads("nusc")
    cls, masks, gt_occ, pts = inputs("nusc")
    gl, gm = zip(*[T.preprocess_occupancy_gt(o, 17) for o in gt_occ])
    d = be.device
    for i in range(2):
        rec = T.RecordingRNG()
        torch.manual_seed(12 + 100 * i)
        to = T.nusc_get_target_single(cls[0][i], masks[0][i], gl[i], gm[i], pts[i], ocfg, rec)
        head.rng = ReplayRNG(rec.tape, d)
        labels, lw, mt, mw, pos, pos_gt, cost = head._get_target_single(cls[0][i].to(d), masks[0][i].to(d),
                                                                        gl[i].to(d), gm[i].to(d), pts[i].to(d))
        assert torch.equal(labels.cpu(), to["labels"]) and torch.equal(pos.cpu(), to["pos_inds"])
        assert torch.allclose(mw.cpu(), to["mask_weights"]) and torch.equal(mt.dense().cpu(), to["mask_targets"])
        assert torch.allclose(cost.cpu(), to["cost"], atol=2e-4, rtol=1e-4)
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * 2
    m = head.lidarseg_metric(cls[-1].to(d), masks[-1].to(d), [p.to(d) for p in pts], metas)
    assert abs(float(m["point_mean_iou"]) - float(g["nusc.point_mean_iou"])) < 1e-6


def test_depth_loss_matches_reference_fixture(bound):
    from occformer_amd.registry import MODELS
    be = bound
    g = golden("train")
    model, meta = tinycfg.tiny_nusc()
    vt = MODELS.build(model["img_view_transformer"]).to(be.device)
    H, W = meta["input_size"]
    gd = paramgen.uniform("depth.gt", (2, 3, H, W), 8) * 14.0
    gd = torch.where(paramgen.uniform("depth.keep", (2, 3, H, W), 8) < 0.03, gd, torch.zeros(()))
    dp = paramgen.uniform("depth.pred", (6, meta["D"], meta["fH"], meta["fW"]), 8).softmax(1)
    out = vt.get_depth_loss(gd.to(be.device), dp.to(be.device))
    assert abs(float(out) - float(g["depth.loss"])) < 1e-5


def test_device_confusion_matrix_and_miou_match_numpy():
    """P/utils/metric_util.py semantics (fast_hist_crop / per_class_iu / nanmean) without leaving the device"""
    import numpy as np
    from occformer_amd import training as TR
    g = torch.Generator().manual_seed(11)
    out = torch.randint(1, 17, (5000,), generator=g)
    tgt = torch.randint(-1, 19, (5000,), generator=g)                 # some labels outside 0..16
    tgt[tgt == 7] = 3                                                 # an absent class -> nan IoU, ignored
    out[out == 7] = 2
    ref_hist = TR.fast_hist_crop(out.numpy(), tgt.numpy(), np.arange(16))
    hist = TR.fast_hist_crop_device(out, tgt, 16)
    assert np.array_equal(hist.numpy(), ref_hist)
    ref = np.nanmean(TR.per_class_iu(ref_hist))
    assert abs(float(TR.mean_iou_device(hist)) - ref) < 1e-12


@pytest.mark.parametrize("S,G,dims", [(3, 5, (6, 5, 4)), (10, 6, (5, 4, 8)), (2, 17, (4, 4, 4)), (10, 7, (3, 4, 2))])
def test_candidate_logits_voxel_major(be, S, G, dims):
    """training._candidate_logits_voxel_major (the importance sampling's ranking logits contracted voxel-major and read
    through the channels-last sampler) against the path it replaces -- point_sample_3d of the channel-major logits
    [S, G, X, Y, Z] (mmdet_utils.py:91-246 via mask2former_nusc_occ.py:253-262): G % 4 != 0 (padded columns), S * Gp
    beyond one 64-column block, and S * Gp > 256 where the helper declines (the caller keeps the dense path)."""
    from occformer_amd.training import _candidate_logits_voxel_major
    E, P3 = 32, 50
    X, Y, Z = dims
    V = X * Y * Z
    rows_e = paramgen.tensor("cl_rows", (S * G, E), 1)
    feat = paramgen.tensor("cl_feat", (V, E), 2)
    cand = paramgen.uniform("cl_cand", (S, P3, 3), 3)                     # (x, y, z) in [0, 1]
    dense = (rows_e @ feat.t()).view(S, G, X, Y, Z)
    ops = be.ops
    got = _candidate_logits_voxel_major(ops, be.to(rows_e), be.to(feat), S, G, (X, Y, Z), be.to(cand.flip(-1).contiguous()),
                                        "zeros")
    Gp = (G + 3) // 4 * 4
    if S * Gp > 256:
        assert got is None
        return
    assert got is not None and tuple(got.shape) == (S, G, P3)
    ref = ops.point_sample_3d(be.to(dense), be.to(cand.flip(-1).contiguous()), False, "zeros")
    assert float((got.cpu() - ref.cpu()).abs().max()) <= 2e-4 * float(ref.abs().max())
