"""TEST INFRASTRUCTURE -- CPU restatement of the reference's host-side pipeline / evaluation stages;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.  Pinned against the reference's own
classes imported unmodified (tests/golden/make_golden_pipeline.py -> tests/golden/pipeline.npz).

  create_depth_from_lidar : projects/mmdet3d_plugin/datasets/pipelines/lidar2depth.py:21-41 (project_points), :55-80
  ssc_counts              : projects/mmdet3d_plugin/utils/ssc_metric.py:62-84 (update), :108-175 (the two scores)
"""
import torch


def project_points(points, rots, trans, intrins, post_rots, post_trans):
    """lidar2depth.py:21-41: points [P, 3] -> [P, N, 3] = (u, v, d) per camera"""
    pts = points.view(-1, 1, 3) - trans.view(1, -1, 3)
    pts = rots.inverse().unsqueeze(0) @ pts.unsqueeze(-1)
    if intrins.shape[-1] == 4:
        pts = torch.cat((pts, torch.ones((pts.shape[0], 1, 1, 1))), dim=2)
    pts = (intrins.unsqueeze(0) @ pts).squeeze(-1)
    d = pts[..., 2:3]
    uv = pts[..., :2] / d
    uv = (post_rots[:, :2, :2].unsqueeze(0) @ uv.unsqueeze(-1)).squeeze(-1) + post_trans[..., :2].unsqueeze(0)
    return torch.cat((uv, d), dim=2)


def create_depth_from_lidar(points, rots, trans, intrins, post_rots, post_trans, img_hw):
    """lidar2depth.py:55-80: valid mask on the unrounded pixel, then per image the points sorted by DESCENDING depth
    and written in that order (the last write -- the nearest return -- wins)"""
    H, W = img_hw
    uvd = project_points(points[:, :3].float(), rots, trans, intrins, post_rots, post_trans)
    valid = (uvd[..., 0] >= 0) & (uvd[..., 1] >= 0) & (uvd[..., 0] <= W - 1) & (uvd[..., 1] <= H - 1) & (uvd[..., 2] > 0)
    out = []
    for n in range(rots.shape[0]):
        g = torch.zeros((H, W))
        v = uvd[:, n][valid[:, n]]
        v = v[torch.argsort(v[:, 2], descending=True)]
        g[v[:, 1].round().long(), v[:, 0].round().long()] = v[:, 2]
        out.append(g)
    return torch.stack(out)


def ssc_counts(y_pred, y_true, n_classes, nonempty=None, nonsurface=None):
    """ssc_metric.py:62-84 on COPIES, with the in-place edits of get_score_completion (:111-113: prediction and target
    of the ignored voxels become 0) applied before the second mask is taken, exactly as ``update`` does.
    -> (completion tp, fp, fn, per-class tps, fps, fns) as int64"""
    y_pred, y_true = y_pred.clone(), y_true.clone()
    B = y_pred.shape[0]
    mask = y_true != 255
    if nonempty is not None:
        mask = mask & nonempty
    if nonsurface is not None:
        mask = mask & nonsurface
    # ---- get_score_completion (:108-137)
    y_pred[y_true == 255] = 0
    y_true[y_true == 255] = 0
    t, p, m = y_true.view(B, -1), y_pred.view(B, -1), mask.view(B, -1)
    bt, bp = t > 0, p > 0
    tp = int(((bt & bp) & m).sum())
    fp = int(((~bt & bp) & m).sum())
    fn = int(((bt & ~bp) & m).sum())
    # ---- get_score_semantic_and_completion (:139-175): mask re-taken AFTER the edits
    mask = y_true != 255
    if nonempty is not None:
        mask = mask & nonempty
    m = mask.view(B, -1)
    tps = torch.zeros(n_classes, dtype=torch.int64)
    fps = torch.zeros(n_classes, dtype=torch.int64)
    fns = torch.zeros(n_classes, dtype=torch.int64)
    for j in range(n_classes):
        tps[j] = ((t == j) & (p == j) & m).sum()
        fps[j] = ((t != j) & (p == j) & m).sum()
        fns[j] = ((t == j) & (p != j) & m).sum()
    return tp, fp, fn, tps, fps, fns


def ssc_compute(tp, fp, fn, tps, fps, fns):
    """ssc_metric.py:86-101"""
    tp, fp, fn = float(tp), float(fp), float(fn)
    iou_ssc = tps.double() / (tps.double() + fps.double() + fns.double() + 1e-5)
    return {"precision": tp / (tp + fp), "recall": tp / (tp + fn), "iou": tp / (tp + fp + fn), "iou_ssc": iou_ssc,
            "iou_ssc_mean": float(iou_ssc[1:].mean())}
