"""Golden vectors for the host-side pipeline / evaluation stages, produced by the REFERENCE's own
classes imported unmodified through tests/refshim:
  * CreateDepthFromLiDAR.project_points + the depth-map fill of __call__ (datasets/pipelines/lidar2depth.py:21-80), for
    a nuScenes-like 6-camera rig (3x3 intrinsics) and a SemanticKITTI-like camera (4x4 intrinsics);
  * SSCMetrics.update / compute (utils/ssc_metric.py) with and without the nonempty / nonsurface masks, two updates.
The oracle (oracle/pipeline_ref.py) is checked against them here; inputs are regenerated from seeds by the tests.

    python tests/golden/make_golden_pipeline.py        ->  tests/golden/pipeline.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import paramgen, refshim  # noqa: E402
from oracle import pipeline_ref as PR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def depth_case(kind):
    """seeded LiDAR sweep + camera rig; -> (points [P, 5], rots, trans, intrins, post_rots, post_trans, (H, W))"""
    kitti = kind == "kitti"
    N, (H, W), focal = (1, (48, 160), 90.0) if kitti else (6, (64, 176), 140.0)
    rots, trans, intr, post_rots, post_trans, _ = paramgen.camera_rig(1, N, H, W, focal, seed=41 if kitti else 40,
                                                                      kitti=kitti)
    g = torch.Generator().manual_seed(7 if kitti else 6)
    P = 6000
    r = torch.rand(P, generator=g) * 30.0 + 0.5
    az = torch.rand(P, generator=g) * 2 * np.pi
    el = (torch.rand(P, generator=g) - 0.6) * 0.5
    pts = torch.stack((r * torch.cos(az) * torch.cos(el), r * torch.sin(az) * torch.cos(el), r * torch.sin(el) + 0.5,
                       torch.rand(P, generator=g), torch.zeros(P)), 1)
    pts[:40, :3] = trans[0, 0]                                  # points AT a camera centre: d = 0, u = v = nan
    pts[40:80] = pts[80:120]                                    # duplicates (same pixel, same depth)
    return pts, rots[0], trans[0], intr[0], post_rots[0], post_trans[0], (H, W)


def ssc_case(seed, n_classes, shape=(2, 12, 10, 6)):
    g = torch.Generator().manual_seed(seed)
    y_true = torch.randint(0, n_classes, shape, generator=g)
    y_true[torch.rand(shape, generator=g) < 0.15] = 255
    y_true[torch.rand(shape, generator=g) < 0.4] = 0
    y_pred = torch.where(torch.rand(shape, generator=g) < 0.6, y_true.clamp(max=n_classes - 1),
                         torch.randint(0, n_classes, shape, generator=g))
    nonempty = torch.rand(shape, generator=g) < 0.8
    nonsurface = torch.rand(shape, generator=g) < 0.7
    return y_pred, y_true, nonempty, nonsurface


def main():
    refshim.install()
    l2d = refshim.ref("datasets.pipelines.lidar2depth")
    sscm = refshim.ref("utils.ssc_metric")
    out = {}
    for kind in ("nusc", "kitti"):
        pts, rots, trans, intr, post_rots, post_trans, (H, W) = depth_case(kind)
        t = l2d.CreateDepthFromLiDAR(data_root=None, dataset=kind)
        imgs = torch.zeros(rots.shape[0], 3, H, W)
        proj = t.project_points(pts[:, :3].float(), rots, trans, intr, post_rots, post_trans)
        valid = (proj[..., 0] >= 0) & (proj[..., 1] >= 0) & (proj[..., 0] <= W - 1) & (proj[..., 1] <= H - 1) & \
            (proj[..., 2] > 0)
        # the reference's own __call__ (:43-84), reading the sweep from a file as it does
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            if kind == "kitti":
                d = os.path.join(tmp, "data_velodyne/velodyne/sequences/08/velodyne")
                os.makedirs(d)
                pts[:, :4].numpy().astype(np.float32).tofile(os.path.join(d, "000123.bin"))
                t.data_root = tmp
                results = dict(img_filename=[os.path.join(tmp, "sequences/08/image_2/000123.png")])
            else:
                f = os.path.join(tmp, "sweep.bin")
                pts.numpy().astype(np.float32).tofile(f)
                results = dict(pts_filename=f)
            results["img_inputs"] = (imgs, rots, trans, intr, post_rots, post_trans, torch.zeros(1), torch.zeros(1))
            gts = t(results)["img_inputs"][6]
        o = PR.create_depth_from_lidar(pts, rots, trans, intr, post_rots, post_trans, (H, W))
        assert torch.equal(o, gts), kind
        assert torch.equal(PR.project_points(pts[:, :3].float(), rots, trans, intr, post_rots, post_trans).nan_to_num(-7),
                           proj.nan_to_num(-7))
        out[f"{kind}.gt_depths"] = gts.numpy()
        print(kind, "depth maps:", int((gts > 0).sum()), "pixels hit of", gts.numel(), " valid projections", int(valid.sum()))

    for name, C, names in (("kitti", 20, None), ("nusc", 17, [f"c{i}" for i in range(17)])):
        m = sscm.SSCMetrics(class_names=names)
        tot = None
        for k, masks in enumerate((False, True)):
            y_pred, y_true, ne, ns = ssc_case(50 + k + (10 if name == "nusc" else 0), C)
            a, b = y_pred.clone(), y_true.clone()
            m.update(a, b, ne if masks else None, ns if masks else None)
            c = PR.ssc_counts(y_pred, y_true, C, ne if masks else None, ns if masks else None)
            tot = c if tot is None else tuple(x + y for x, y in zip(tot, c))
        ref = m.compute()
        o = PR.ssc_compute(*tot)
        assert int(m.completion_tp) == tot[0] and int(m.completion_fp) == tot[1] and int(m.completion_fn) == tot[2]
        assert torch.equal(m.tps.long(), tot[3]) and torch.equal(m.fps.long(), tot[4]) and torch.equal(m.fns.long(), tot[5])
        assert abs(ref["iou"] - o["iou"]) < 1e-6 and abs(ref["iou_ssc_mean"] - o["iou_ssc_mean"]) < 1e-6
        out[f"ssc.{name}.completion"] = np.array([int(m.completion_tp), int(m.completion_fp), int(m.completion_fn)])
        out[f"ssc.{name}.tps"], out[f"ssc.{name}.fps"], out[f"ssc.{name}.fns"] = m.tps.numpy(), m.fps.numpy(), m.fns.numpy()
        out[f"ssc.{name}.iou"], out[f"ssc.{name}.iou_ssc_mean"] = np.float64(ref["iou"]), np.float64(ref["iou_ssc_mean"])
        out[f"ssc.{name}.iou_ssc"] = ref["iou_ssc"].numpy()
        out[f"ssc.{name}.precision"], out[f"ssc.{name}.recall"] = ref["precision"].numpy(), ref["recall"].numpy()
        # compute_single (apis/test.py:173): one sample, masks on
        y_pred, y_true, ne, ns = ssc_case(77, C, shape=(1, 12, 10, 6))
        single = sscm.SSCMetrics(class_names=names).compute_single(y_pred.clone(), y_true.clone(), ne, ns)
        for i, v in enumerate(single):
            out[f"ssc.{name}.single{i}"] = np.asarray(v)
        print(name, "SSC: iou", ref["iou"], "mIoU", ref["iou_ssc_mean"])
    np.savez_compressed(os.path.join(OUT, "pipeline.npz"), **out)
    print("wrote pipeline.npz (%.0f KiB)" % (os.path.getsize(os.path.join(OUT, "pipeline.npz")) / 1024))


if __name__ == "__main__":
    main()
