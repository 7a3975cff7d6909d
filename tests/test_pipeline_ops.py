"""Host-side pipeline / evaluation stages on the device (csrc/pipeline.hip, occformer_amd/pipeline.py)
against the reference-generated fixture tests/golden/pipeline.npz (the reference's CreateDepthFromLiDAR.__call__ and
SSCMetrics run unmodified by tests/golden/make_golden_pipeline.py) and against the oracle restatement
(oracle/pipeline_ref.py).  Index / count work: bit-exact."""
import numpy as np
import pytest
import torch

import occformer_amd  # noqa: F401
import occformer_amd.ops as ops_mod
from occformer_amd import pipeline as PL
from oracle import pipeline_ref as PR
from tests.conftest import golden
from tests.golden.make_golden_pipeline import depth_case, ssc_case


@pytest.fixture
def bound(be, monkeypatch):
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    return be


def test_oracle_reproduces_reference_fixture():
    g = golden("pipeline")
    for kind in ("nusc", "kitti"):
        pts, rots, trans, intr, post_rots, post_trans, hw = depth_case(kind)
        assert torch.equal(PR.create_depth_from_lidar(pts, rots, trans, intr, post_rots, post_trans, hw), g[f"{kind}.gt_depths"])


@pytest.mark.parametrize("kind", ["nusc", "kitti"])
def test_create_depth_from_lidar_bit_exact(bound, kind):
    """lidar2depth.py:43-84 through the reference's pipeline contract (results dict in, img_inputs[6] replaced).  The
    camera constants (rots.inverse()) come from the same torch CPU op as the reference's so that the comparison is
    bit-exact on every backend; the case holds points AT a camera centre (d = 0 -> nan pixel), duplicates, points
    behind the cameras and off-image."""
    be = bound
    g = golden("pipeline")
    pts, rots, trans, intr, post_rots, post_trans, (H, W) = depth_case(kind)
    cam, kitti = PL.pack_depth_cameras(rots, trans, intr, post_rots, post_trans)          # host: the reference's inverse
    out = be.ops.lidar_depth(be.to(pts), be.to(cam), rots.shape[0], H, W, kitti)
    ref = g[f"{kind}.gt_depths"]
    assert out.shape == ref.shape
    assert torch.equal(out.cpu(), ref), int((out.cpu() != ref).sum())
    # the registered transform, device-side constants: identical up to the last-bit difference of inverse() per device
    t = PL.PIPELINES._classes["CreateDepthFromLiDAR"](dataset=kind)
    imgs = torch.zeros(rots.shape[0], 3, H, W)
    res = t(dict(points=be.to(pts), img_inputs=(imgs, rots, trans, intr, post_rots, post_trans, torch.zeros(1), None)))
    got = res["img_inputs"][6].cpu()
    assert len(res["img_inputs"]) == 8 and got.shape == ref.shape
    assert int(((got > 0) != (ref > 0)).sum()) <= 4 and float((got - ref).abs().max()) < 1e-4 + 50 * float(((got > 0) != (ref > 0)).any())


def test_create_depth_empty_sweep(bound):
    be = bound
    pts, rots, trans, intr, post_rots, post_trans, (H, W) = depth_case("nusc")
    cam, kitti = PL.pack_depth_cameras(rots, trans, intr, post_rots, post_trans)
    out = be.ops.lidar_depth(be.to(pts[:0].contiguous()), be.to(cam), rots.shape[0], H, W, kitti)
    assert float(out.abs().max()) == 0.0


@pytest.mark.parametrize("name,C", [("kitti", 20), ("nusc", 17)])
def test_ssc_metrics_vs_reference(bound, name, C):
    """two ``update`` calls (without and with the nonempty / nonsurface masks), then ``compute`` -- the counts equal the
    reference's tps / fps / fns / completion state exactly, the scores to float rounding; ``compute_single`` likewise"""
    be = bound
    g = golden("pipeline")
    m = PL.SSCMetrics(class_names=None if name == "kitti" else [f"c{i}" for i in range(C)])
    for k, masks in enumerate((False, True)):
        y_pred, y_true, ne, ns = ssc_case(50 + k + (10 if name == "nusc" else 0), C)
        keep = (y_pred.clone(), y_true.clone())
        m.update(be.to(y_pred), be.to(y_true), be.to(ne) if masks else None, be.to(ns) if masks else None)
        assert torch.equal(y_pred, keep[0]) and torch.equal(y_true, keep[1])
    tp, fp, fn, tps, fps, fns = (t.cpu() for t in m._scores(m.counts, C))
    assert [int(tp), int(fp), int(fn)] == g[f"ssc.{name}.completion"].tolist()
    assert torch.equal(tps.float(), g[f"ssc.{name}.tps"]) and torch.equal(fps.float(), g[f"ssc.{name}.fps"]) and \
        torch.equal(fns.float(), g[f"ssc.{name}.fns"])
    sc = m.compute()
    assert abs(sc["iou"] - g[f"ssc.{name}.iou"]) < 1e-6 and abs(sc["iou_ssc_mean"] - g[f"ssc.{name}.iou_ssc_mean"]) < 1e-6
    assert torch.allclose(sc["iou_ssc"].cpu(), g[f"ssc.{name}.iou_ssc"], atol=1e-6)
    assert torch.allclose(sc["precision"].cpu(), g[f"ssc.{name}.precision"], atol=1e-6)
    assert torch.allclose(sc["recall"].cpu(), g[f"ssc.{name}.recall"], atol=1e-6)
    y_pred, y_true, ne, ns = ssc_case(77, C, shape=(1, 12, 10, 6))
    single = m.compute_single(be.to(y_pred), be.to(y_true), be.to(ne), be.to(ns))
    for i, v in enumerate(single):
        assert np.array_equal(np.asarray(v).reshape(-1), np.asarray(g[f"ssc.{name}.single{i}"]).reshape(-1)), i


def test_ssc_fused_argmax_matches_label_path(bound):
    """apis/test.py:64: y_pred = argmax(output_voxels, 1) -- taken inside the kernel from the class volume (first
    maximum, as torch.argmax) instead of a separate pass over [B, C, X, Y, Z]"""
    be = bound
    C, shape = 17, (2, 9, 8, 5)
    g = torch.Generator().manual_seed(3)
    scores = torch.randn(shape[0], C, *shape[1:], generator=g)
    scores[:, 3] = torch.where(torch.rand(shape, generator=g) < 0.2, scores[:, 5], scores[:, 3])     # exact ties
    _, y_true, ne, ns = ssc_case(91, C, shape=shape)
    a = PL.SSCMetrics(class_names=[str(i) for i in range(C)])
    a.update(be.to(scores.argmax(1)), be.to(y_true), be.to(ne), be.to(ns))
    b = PL.SSCMetrics(class_names=[str(i) for i in range(C)])
    b.update(None, be.to(y_true), be.to(ne), be.to(ns), scores=be.to(scores))
    assert torch.equal(a.counts.cpu(), b.counts.cpu())
    ref = PR.ssc_counts(scores.argmax(1), y_true, C, ne, ns)
    tp, fp, fn, tps, fps, fns = (t.cpu() for t in a._scores(a.counts, C))
    assert (int(tp), int(fp), int(fn)) == ref[:3] and torch.equal(tps, ref[3]) and torch.equal(fps, ref[4]) and \
        torch.equal(fns, ref[5])


def test_image_post_homography_matches_reference_formula():
    """loading_nusc_imgs.py:35-55 restated with the reference's statements"""
    import math
    for resize, crop, flip, rot in ((0.44, (0, 140, 704, 396), False, 0.0), (0.48, (12, 150, 716, 406), True, 3.7)):
        pr, pt = PL.image_post_homography(resize, crop, flip, rot)
        post_rot, post_tran = torch.eye(2), torch.zeros(2)
        post_rot *= resize
        post_tran -= torch.Tensor(crop[:2])
        if flip:
            A = torch.Tensor([[-1, 0], [0, 1]])
            b = torch.Tensor([crop[2] - crop[0], 0])
            post_rot = A.matmul(post_rot)
            post_tran = A.matmul(post_tran) + b
        h = rot / 180 * math.pi
        A = torch.Tensor([[math.cos(h), math.sin(h)], [-math.sin(h), math.cos(h)]])
        b = torch.Tensor([crop[2] - crop[0], crop[3] - crop[1]]) / 2
        b = A.matmul(-b) + b
        assert torch.equal(pr, A.matmul(post_rot)) and torch.equal(pt, A.matmul(post_tran) + b)
