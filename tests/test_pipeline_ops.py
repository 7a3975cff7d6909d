"""Host-side pipeline / evaluation stages on the device (csrc/pipeline.hip, occformer_amd/pipeline.py)
against the reference-generated fixture tests/golden/pipeline.npz (the reference's CreateDepthFromLiDAR.__call__ and
SSCMetrics run unmodified by tests/golden/make_golden_pipeline.py) and against the oracle restatement
(oracle/pipeline_ref.py).  Index / count work: bit-exact."""
import os

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


# ------------------------------------------------------------------ img_inputs producer (loading_nusc_imgs.py:9-193)
def test_image_oracle_is_pillow():
    """the oracle's restatement of Pillow's Image.resize (antialiased bicubic, 22-bit fixed point, uint8 intermediate)
    and Image.rotate (nearest neighbour, 16.16 fixed point) against Pillow itself, bit for bit"""
    Image = pytest.importorskip("PIL.Image")
    from oracle import image_pipeline_ref as IR
    rng = np.random.RandomState(0)
    for (H, W, ow, oh) in [(90, 160, 70, 39), (90, 160, 77, 43), (45, 80, 100, 56), (64, 64, 64, 30), (50, 50, 50, 50)]:
        img = rng.randint(0, 256, (H, W, 3)).astype(np.uint8)
        assert np.array_equal(np.array(Image.fromarray(img).resize((ow, oh))), IR.resize(img, ow, oh)), (H, W, ow, oh)
    for ang in (0.0, 3.3, -5.4, 180.0, 90.0, 45.0, -0.7, 5.399999):
        img = rng.randint(0, 256, (32, 88, 3)).astype(np.uint8)
        assert np.array_equal(np.array(Image.fromarray(img).rotate(ang)), IR.rotate(img, ang)), ang
    img = rng.randint(0, 256, (20, 30, 3)).astype(np.uint8)
    for box in ((3, -4, 25, 12), (-2, 5, 40, 26), (0, 0, 30, 20)):
        assert np.array_equal(np.array(Image.fromarray(img).crop(box)), IR.crop(img, box)), box


def test_image_oracle_reproduces_reference_fixture():
    from oracle import image_pipeline_ref as IR
    from tests.golden.make_golden_image_pipeline import CAMS, DATA_CONFIG, frames
    g = golden("image_pipeline")
    imgs, _, _ = frames()
    for mode in ("train", "train2", "test"):
        np.random.seed(int(g[f"{mode}.seed"]))
        for k, c in enumerate(CAMS):
            rs, dims, crop, flip, rot = IR.sample_augmentation(90, 160, DATA_CONFIG, mode != "test")
            cv = IR.img_transform_core(imgs[c], dims, crop, flip, rot)
            assert np.array_equal(cv, g[f"{mode}.canvas"][k].numpy())
            assert np.array_equal(IR.normalize(cv), g[f"{mode}.imgs"][k].numpy())


@pytest.mark.parametrize("case", [((70, 39), (5, 3, 55, 33), False, 3.3), ((77, 43), (-4, 6, 60, 40), True, -5.4),
                                  ((200, 112), (30, 40, 118, 72), True, 180.0), ((160, 90), (0, 20, 88, 52), False, 0.0),
                                  ((64, 90), (0, 0, 64, 90), False, 45.0)])
def test_image_transform_kernels_vs_oracle(bound, case):
    """resize -> crop -> flip -> rotate -> normalize on the kernels against the oracle: the uint8 frame bit for bit
    (up- and down-scaling, crops that leave the frame, the 0 / 180 degree fast paths), the float image to 1e-6"""
    from oracle import image_pipeline_ref as IR
    dims, crop, flip, rot = case
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, (90, 160, 3)).astype(np.uint8)
    cfg = dict(mean=[103.53, 116.28, 123.675], std=[57.375, 57.12, 58.395], to_rgb=False)
    for norm in (None, cfg):
        x, cv = PL.image_transform(bound.to(torch.from_numpy(img)), dims, crop, flip, rot, norm, want_canvas=True)
        ref_cv = IR.img_transform_core(img, dims, crop, flip, rot)
        assert np.array_equal(cv.cpu().numpy(), ref_cv)
        ref = IR.normalize(ref_cv) if norm is None else IR.normalize(ref_cv, **cfg)
        assert np.allclose(x.cpu().numpy(), ref, rtol=1e-6, atol=1e-6)
    assert np.array_equal(PL.image_resize(bound.to(torch.from_numpy(img)), *dims).cpu().numpy(), IR.resize(img, *dims))


def test_load_multi_view_images_vs_reference_fixture(bound):
    """the registered pipeline stage (same name and results contract as the reference's) on seeded draws against the
    fixture the reference's class produced: canvas bit-exact, images 1e-6, calibration tensors exact"""
    from tests.golden.make_golden_image_pipeline import CAMS, DATA_CONFIG, frames
    g = golden("image_pipeline")
    imgs, l2c, intr = frames()
    assert PL.PIPELINES.get("LoadMultiViewImageFromFiles_OccFormer") is PL.LoadMultiViewImageFromFiles_OccFormer
    for mode in ("train", "train2", "test"):
        t = PL.LoadMultiViewImageFromFiles_OccFormer(DATA_CONFIG, is_train=mode != "test", device=bound.device)
        res = dict(curr=dict(cams={c: dict(img=imgs[c], cam_intrinsic=intr[c]) for c in CAMS}), lidar2cam_dic=l2c)
        np.random.seed(int(g[f"{mode}.seed"]))
        out = t(res)["img_inputs"]
        assert len(out) == 8 and tuple(out[0].shape) == (3, 3, 32, 88) and tuple(out[6].shape) == (3, 1)
        assert np.array_equal(res["canvas"].cpu().numpy(), g[f"{mode}.canvas"].numpy())
        assert torch.allclose(out[0].cpu(), g[f"{mode}.imgs"], rtol=1e-6, atol=1e-6)
        for i, name in ((1, "rots"), (2, "trans"), (3, "intrins"), (4, "post_rots"), (5, "post_trans")):
            assert torch.equal(out[i].cpu(), g[f"{mode}.{name}"]), (mode, name)


# ------------------------------------------------------------------ SemanticKITTI img_inputs (loading_kitti_imgs.py:11-145)
def test_kitti_image_loader_vs_reference_fixture(bound):
    """the monocular producer under the reference's pipeline name against the fixture the reference's class produced:
    canvas bit-exact, image 1e-6, calibration / post-homography tensors exact, the [1, ...] shapes of its eight outputs"""
    from tests.golden.make_golden_occ_loading import KITTI_DATA, kitti_calib, kitti_frame
    g = golden("occ_loading")
    frame = kitti_frame()
    assert np.array_equal(frame, g["kimg.frame"].numpy())
    l2c, intr = kitti_calib()
    assert PL.PIPELINES.get("LoadMultiViewImageFromFiles_SemanticKitti") is PL.LoadMultiViewImageFromFiles_SemanticKitti
    for mode in ("train", "test"):
        t = PL.LoadMultiViewImageFromFiles_SemanticKitti(KITTI_DATA, is_train=mode == "train", device=bound.device)
        res = dict(img_filename=["frame"], img=[frame], cam_intrinsic=[intr], lidar2cam=[l2c])
        np.random.seed(int(g[f"kimg.{mode}.seed"]))
        out = t(res)["img_inputs"]
        assert len(out) == 8 and tuple(out[0].shape) == (1, 3, 48, 160) and tuple(out[6].shape) == (1, 1)
        assert np.array_equal(res["canvas"].cpu().numpy(), g[f"kimg.{mode}.canvas"].numpy())
        assert np.array_equal(res["raw_img"].cpu().numpy(), frame)
        assert torch.allclose(out[0].cpu(), g[f"kimg.{mode}.imgs"], rtol=1e-6, atol=1e-6)
        for i, name in ((1, "rots"), (2, "trans"), (3, "intrins"), (4, "post_rots"), (5, "post_trans"), (7, "cam2lidar")):
            assert torch.equal(out[i].cpu(), g[f"kimg.{mode}.{name}"]), (mode, name)


# ------------------------------------------------------------------ occupancy ground truth (loading_nusc_occ.py, loading_kitti_occ.py)
def test_occ_loading_oracle_is_pillow_and_reference():
    """oracle/occ_loading_ref.py: the label-slice rotation against Pillow itself (fillcolor 255, the 0 / 90 / 180 / 270
    fast paths, square and oblong slices), and the whole restatement against the fixture the reference's loaders produced"""
    Image = pytest.importorskip("PIL.Image")
    from oracle import occ_loading_ref as OR
    from tests.golden.make_golden_occ_loading import (BDA_FLIP, BDA_ROT, GRID, KITTI_RANGE, LEARNING_MAP, RANGE,
                                                      label_volume, sweep)
    rng = np.random.RandomState(1)
    for shape in ((32, 32), (24, 40)):
        lab = rng.randint(0, 20, shape).astype(np.uint8)
        for ang in (0.0, 90.0, 180.0, 270.0, -90.0, 21.0163, -15.56, 45.0, 359.2, 1e-7):
            ref = np.array(Image.fromarray(lab).rotate(ang, resample=Image.Resampling.NEAREST, fillcolor=255))
            assert np.array_equal(OR.rotate_slice(lab, ang), ref), (shape, ang)
    g = golden("occ_loading")
    five, raw = sweep()
    for tag, conf, train in (("flip", BDA_FLIP, True), ("flip2", BDA_FLIP, True), ("rot", BDA_ROT, True), ("test", BDA_FLIP, False)):
        np.random.seed(int(g[f"nusc.{tag}.seed"]))
        mat = np.eye(3, dtype=np.float32)
        if train:
            rot, _, fx, fy, fz = OR.sample_bda(conf)
            mat = OR.bda_matrix(rot, fx, fy, fz)
        assert np.abs(mat - g[f"nusc.{tag}.bda_rot"].numpy()).max() <= (1e-7 if tag == "rot" else 0)
        occ, pocc = OR.nusc_occupancy(five[:, :3], raw, LEARNING_MAP, GRID, RANGE, g[f"nusc.{tag}.bda_rot"].numpy())
        assert np.array_equal(occ, g[f"nusc.{tag}.gt_occ"].numpy()) and np.array_equal(pocc, g[f"nusc.{tag}.points_occ"].numpy())
    vol = label_volume()
    centre = (np.array(KITTI_RANGE[:3]) + np.array(KITTI_RANGE[3:])) / 2
    for tag, conf in (("flip", dict(BDA_FLIP, flip_dz_ratio=0.5)), ("flip2", dict(BDA_FLIP, flip_dz_ratio=0.5)),
                      ("rot", BDA_ROT), ("rot2", BDA_ROT)):
        np.random.seed(int(g[f"kitti.{tag}.seed"]))
        rot, _, fx, fy, fz = OR.sample_bda(conf)
        occ, mat = OR.voxel_transform(vol, rot, fx, fy, fz, center=centre)
        assert np.array_equal(occ, g[f"kitti.{tag}.gt_occ"].numpy())
        assert np.abs(mat - g[f"kitti.{tag}.bda_rot"].numpy()).max() <= (1e-5 if "rot" in tag else 0)


def test_label_volume_rotation_vs_oracle(bound):
    from oracle import occ_loading_ref as OR
    rng = np.random.RandomState(2)
    for shape in ((32, 32, 5), (24, 40, 3)):
        vol = rng.randint(0, 20, shape).astype(np.uint8)
        for ang in (0.0, 90.0, 180.0, 270.0, 21.0163, -15.56, 45.0, -0.3):
            out = PL.rotate_label_volume(bound.to(torch.from_numpy(vol)), ang)
            assert out.dtype == torch.uint8 and np.array_equal(out.cpu().numpy(), OR.custom_rotate_3d(vol, ang)), (shape, ang)


def test_voxelize_point_labels_vs_oracle(bound):
    """majority label per voxel, bit-exact: random sweeps (ties -> the smallest label, points outside the range clipped
    onto the border voxels, label 0 -> 255), an empty sweep, and the reference's uint16 counter wrapping at 65536 points"""
    from oracle import occ_loading_ref as OR
    grid, rng_ = [10, 12, 4], [-5.0, -6.0, -2.0, 5.0, 6.0, 2.0]

    def both(pts, lab):
        out = PL.voxelize_point_labels(bound.to(torch.from_numpy(pts)), bound.to(torch.from_numpy(lab)), grid, rng_, 18, 17)
        ind = np.floor((np.clip(pts.astype(np.float64), rng_[:3], np.array(rng_[3:]) - 1e-5) - np.array(rng_[:3])) / 1.0).astype(np.int64)
        ref = OR.majority_labels(ind, lab, grid, 17).astype(np.int64) if len(pts) else np.full(grid, 17, np.int64)
        ref[ref == 0] = 255
        ref[ref == 17] = 0
        assert out.dtype == torch.int64 and np.array_equal(out.cpu().numpy(), ref)
        return ref

    rs = np.random.RandomState(0)
    for P in (1, 50, 4000):
        pts = rs.uniform(-6.5, 6.5, (P, 3)).astype(np.float32)
        pts[::7] = np.round(pts[::7])                                   # points exactly on voxel faces
        both(pts, rs.randint(0, 17, P).astype(np.int64))
    # a tie: two labels with two points each -> the smaller label
    pts = np.array([[0.5, 0.5, 0.5]] * 4, np.float32)
    assert both(pts, np.array([9, 4, 9, 4], np.int64))[5, 6, 2] == 4
    both(np.zeros((0, 3), np.float32), np.zeros((0,), np.int64))
    n = 65536 + 5
    pts = np.concatenate([np.full((n, 3), 0.25, np.float32), np.full((10, 3), 0.25, np.float32)])
    ref = both(pts, np.concatenate([np.full(n, 3, np.int64), np.full(10, 5, np.int64)]))
    assert ref[5, 6, 2] == 5                                            # 65541 mod 65536 = 5 < 10


def test_occupancy_loaders_vs_reference_fixture(bound, tmp_path):
    """LoadNuscOccupancyAnnotations / LoadSemKittiAnnotation under the reference's pipeline names and results contract,
    seeded draws, against the fixture the reference's classes produced: gt_occ and bda_rot bit-exact for the flip
    augmentations and the test split; with a BEV rotation the rotated points go through a float32 matrix product whose
    last bit is the BLAS's (<= 1e-5 on points_occ, <= 1 % of the voxels may change cell), the rotated KITTI label volume
    is bit-exact again (fixed-point index arithmetic)"""
    import yaml
    from tests.golden.make_golden_occ_loading import (BDA_FLIP, BDA_ROT, GRID, KITTI_RANGE, LEARNING_MAP, RANGE,
                                                      label_volume, sweep)
    g = golden("occ_loading")
    five, raw = sweep()
    assert np.array_equal(five, g["nusc.points"].numpy())
    assert PL.PIPELINES.get("LoadNuscOccupancyAnnotations") is PL.LoadNuscOccupancyAnnotations
    assert PL.PIPELINES.get("LoadSemKittiAnnotation") is PL.LoadSemKittiAnnotation
    meta = tmp_path / "meta.yaml"
    meta.write_text(yaml.safe_dump(dict(learning_map=LEARNING_MAP)))
    five.tofile(str(tmp_path / "sweep.bin"))
    raw.tofile(str(tmp_path / "seg.bin"))
    inputs = lambda: tuple(bound.to(torch.full((1,), float(i))) for i in range(8))    # noqa: E731
    for tag, conf, train in (("flip", BDA_FLIP, True), ("flip2", BDA_FLIP, True), ("rot", BDA_ROT, True), ("test", BDA_FLIP, False)):
        t = PL.LoadNuscOccupancyAnnotations(data_root=str(tmp_path), is_train=train, grid_size=GRID, point_cloud_range=RANGE,
                                            bda_aug_conf=conf, cls_metas=str(meta), device=bound.device)
        np.random.seed(int(g[f"nusc.{tag}.seed"]))
        if tag == "flip2":                                               # the file route, as the reference reads them
            res = t(dict(lidarseg="seg.bin", pts_filename=str(tmp_path / "sweep.bin"), img_inputs=inputs()))
        else:
            res = t(dict(points=five, points_label=raw, img_inputs=inputs()))
        assert len(res["img_inputs"]) == 9 and float(res["img_inputs"][7]) == 6.0 and float(res["img_inputs"][5]) == 5.0
        occ, pocc, bda = res["gt_occ"].cpu(), res["points_occ"].cpu(), res["img_inputs"][6].cpu()
        assert occ.dtype == torch.int64 and tuple(occ.shape) == tuple(GRID) and pocc.dtype == torch.float32
        # host float32, the reference's own ops (sin / cos of a rotation: the host's vector math library, 1e-6)
        assert torch.equal(bda, g[f"nusc.{tag}.bda_rot"]) if tag != "rot" else torch.allclose(bda, g[f"nusc.{tag}.bda_rot"], atol=1e-6), tag
        if tag == "rot":
            assert torch.allclose(pocc, g[f"nusc.{tag}.points_occ"], rtol=1e-5, atol=1e-5)
            assert float((occ != g[f"nusc.{tag}.gt_occ"].long()).float().mean()) <= 0.01
        else:
            assert torch.equal(pocc, g[f"nusc.{tag}.points_occ"]) and torch.equal(occ, g[f"nusc.{tag}.gt_occ"].long()), tag
    t = PL.LoadNuscOccupancyAnnotations(is_test_submit=True, grid_size=GRID, point_cloud_range=RANGE, cls_metas=str(meta),
                                        device=bound.device)
    res = t(dict(points=five, img_inputs=inputs()))
    assert torch.equal(res["img_inputs"][6].cpu(), torch.eye(3)) and "gt_occ" not in res
    assert torch.equal(res["points_occ"].cpu(), torch.cat([torch.from_numpy(five[:, :3]), torch.zeros(len(five), 1)], 1))
    vol = label_volume()
    for tag, conf in (("flip", dict(BDA_FLIP, flip_dz_ratio=0.5)), ("flip2", dict(BDA_FLIP, flip_dz_ratio=0.5)),
                      ("rot", BDA_ROT), ("rot2", BDA_ROT)):
        t = PL.LoadSemKittiAnnotation(conf, is_train=True, point_cloud_range=KITTI_RANGE, device=bound.device)
        np.random.seed(int(g[f"kitti.{tag}.seed"]))
        res = t(dict(gt_occ=vol.copy(), img_inputs=inputs()))
        assert res["gt_occ"].dtype == torch.int64 and torch.equal(res["gt_occ"].cpu(), g[f"kitti.{tag}.gt_occ"].long()), tag
        bda = res["img_inputs"][6].cpu()
        assert torch.equal(bda, g[f"kitti.{tag}.bda_rot"]) if "rot" not in tag else torch.allclose(bda, g[f"kitti.{tag}.bda_rot"], atol=1e-5), tag
    t = PL.LoadSemKittiAnnotation(BDA_FLIP, is_train=False, device=bound.device)
    res = t(dict(gt_occ=vol.copy(), img_inputs=inputs()))
    assert torch.equal(res["gt_occ"].cpu(), torch.from_numpy(vol).long()) and torch.equal(res["img_inputs"][6].cpu(), torch.eye(4))
    res = t(dict(gt_occ=None, img_inputs=inputs()))
    assert res["gt_occ"] is None and torch.equal(res["img_inputs"][6].cpu(), torch.eye(4)) and len(res["img_inputs"]) == 9


def test_config_pipeline_feeds_the_training_step(bound, tmp_path):
    """The reference configs' nuScenes ``train_pipeline`` list (occformer_nusc_r50_256x704.py:223-233: image loader ->
    CreateDepthFromLiDAR -> LoadNuscOccupancyAnnotations -> OccDefaultFormatBundle3D -> Collect3D), built by name from
    PIPELINES and run per sample on the device, collated, and handed to ``OccupancyFormer.forward_train`` in the
    reference's nine-entry ``img_inputs`` layout: every loss equals the oracle's training step on the same tensors
    (<= 1e-3).  The tiny model has no image backbone: a fixed pooling of the loader's images stands in for it on both
    sides -- what is checked is the contract between the producers and the step (calibration order, bda_rot,
    gt_depths, gt_occ, points_occ)."""
    import math

    import yaml
    from occformer_amd import noise
    from occformer_amd.registry import build_model
    from oracle import occformer_train_ref as T
    from tests import paramgen, tinycfg
    from tests.golden.make_golden_occ_loading import LEARNING_MAP
    from tests.golden.make_golden_train import oracle_cfg, train_cfg
    from tests.test_training import ReplayRNG
    be = bound
    cfg, meta = tinycfg.tiny_nusc(ncams=2)
    tc = train_cfg()
    cfg["train_cfg"], cfg["test_cfg"] = dict(pts=tc), None
    model = build_model(cfg)
    sd = paramgen.fill_state_dict(model.state_dict(), 77)
    model.load_state_dict(sd)
    cams = ["CAM_FRONT", "CAM_BACK"]
    data_config = dict(cams=cams, Ncams=2, input_size=meta["input_size"], src_size=(100, 250), resize=(-0.04, 0.06),
                       rot=(-3.0, 3.0), flip=True, crop_h=(0.0, 0.0), resize_test=0.0)
    bda_conf = dict(rot_lim=(0, 0), scale_lim=(0.95, 1.05), flip_dx_ratio=0.5, flip_dy_ratio=0.5, flip_dz_ratio=0.5)
    metas = tmp_path / "nuscenes.yaml"
    metas.write_text(yaml.safe_dump(dict(learning_map=LEARNING_MAP)))
    pipe = PL.Compose([
        dict(type="LoadMultiViewImageFromFiles_OccFormer", is_train=True, data_config=data_config, img_norm_cfg=None),
        dict(type="CreateDepthFromLiDAR", dataset="nusc"),
        dict(type="LoadNuscOccupancyAnnotations", is_train=True, grid_size=meta["occ_size"],
             point_cloud_range=meta["pc_range"], bda_aug_conf=bda_conf, cls_metas=str(metas)),
        dict(type="OccDefaultFormatBundle3D", class_names=None),
        dict(type="Collect3D", keys=["img_inputs", "gt_occ", "points_occ"], meta_keys=["pc_range", "occ_size"]),
    ], device=be.device)
    rs = np.random.RandomState(11)
    yy, xx = np.mgrid[0:100, 0:250]
    samples = []
    for b in range(2):
        frames, l2c, intr = {}, {}, {}
        for i, c in enumerate(cams):
            base = 127 + 80 * np.sin(xx / (8.0 + i + b)) * np.cos(yy / 6.0)
            frames[c] = np.clip(base[..., None] + rs.randint(-30, 30, (100, 250, 3)), 0, 255).astype(np.uint8)
            a = math.pi * i + 0.1 * b
            fwd, right, down = [math.cos(a), math.sin(a), 0.0], [math.sin(a), -math.cos(a), 0.0], [0.0, 0.0, -1.0]
            s2l = np.eye(4)
            s2l[:3, :3] = np.stack([right, down, fwd], 1)
            s2l[:3, 3] = [1.5 * math.cos(a), 1.5 * math.sin(a), 1.0]
            l2c[c] = np.linalg.inv(s2l)
            intr[c] = np.array([[200.0, 0, 125.0], [0, 200.0, 45.0], [0, 0, 1.0]])
        # a blocky labelled scene sampled by a sweep: every 3 m cell one class (the head pads the sweep with random
        # points up to num_points * oversample_ratio = 768, mmdet_utils.py:138-177: it must not be longer)
        P = 1000
        xyz = rs.uniform([-8.5, -8.5, -2.2], [8.5, 8.5, 2.2], (P, 3)).astype(np.float32)
        cell = (np.floor((xyz[:, 0] + 9) / 3).astype(np.int64) * 7 + np.floor((xyz[:, 1] + 9) / 3).astype(np.int64)) % 31 + 1
        keep = (cell % 3) != 0
        five = np.concatenate([xyz[keep], np.zeros((int(keep.sum()), 2), np.float32)], 1)
        np.random.seed(20 + b)
        samples.append(pipe(dict(curr=dict(cams={c: dict(img=frames[c], cam_intrinsic=intr[c]) for c in cams}),
                                 lidar2cam_dic=l2c, points=be.to(torch.from_numpy(five)),
                                 points_label=cell[keep].astype(np.uint8), pc_range=meta["pc_range"],
                                 occ_size=meta["occ_size"], sample_idx=b)))
    batch = PL.collate(samples)
    assert sorted(batch) == ["gt_occ", "img_inputs", "img_metas", "points_occ"] and len(batch["img_inputs"]) == 9
    assert batch["img_metas"] == [dict(pc_range=meta["pc_range"], occ_size=meta["occ_size"])] * 2
    imgs, rots, trans, intrins, post_rots, post_trans, bda, gd, s2s = batch["img_inputs"]
    assert tuple(imgs.shape) == (2, 2, 3, 64, 176) and tuple(bda.shape) == (2, 3, 3) and tuple(gd.shape) == (2, 2, 64, 176)
    assert tuple(batch["gt_occ"].shape) == (2, 32, 32, 16) and int((gd > 0).sum()) > 50
    assert len(set(batch["gt_occ"].unique().tolist()) - {0, 255}) >= 5
    pooled = torch.nn.functional.avg_pool2d(imgs.flatten(0, 1), 16).view(2, 2, 3, meta["fH"], meta["fW"])
    x = torch.cat([pooled * (0.3 + 0.1 * k) for k in range(11)], 2)[:, :, :32].contiguous()
    cpu = lambda t: t.detach().cpu()                                    # noqa: E731
    rec = T.RecordingRNG()
    torch.manual_seed(3)
    ocfg = dict(D=meta["D"], C=meta["C"], groups=meta["groups"], heads=meta["heads"], pd_layers=meta["pd_layers"],
                dec_layers=meta["dec_layers"], downsample=16, dbound=cfg["img_view_transformer"]["grid_config"]["dbound"],
                head=oracle_cfg(cfg["pts_bbox_head"], tc))
    ref_losses, _ = T.train_step(sd, cpu(x), tuple(cpu(t) for t in (rots, trans, intrins, post_rots, post_trans, bda)),
                                 cpu(gd), cpu(batch["gt_occ"]), [cpu(p) for p in batch["points_occ"]], ocfg, rng=rec)
    model = model.to(be.device).train()
    replay = ReplayRNG(rec.tape, be.device)
    noise.set_rng(replay)
    try:
        losses = model(return_loss=True, img_metas=batch["img_metas"], img_inputs=[x] + batch["img_inputs"][1:],
                       gt_occ=batch["gt_occ"], points_occ=batch["points_occ"])
    finally:
        noise.set_rng(None)
    assert replay.i == len(rec.tape)
    assert float(ref_losses["loss_depth"].detach()) > 0
    for k, v in ref_losses.items():
        v = float(v.detach())
        assert abs(float(losses[k].detach()) - v) <= 1e-3 * max(1.0, abs(v)), (k, float(losses[k].detach()), v)


def test_reference_config_pipelines_build_by_name(monkeypatch):
    """every stage of ``train_pipeline`` / ``test_pipeline`` in the reference's shipped nuScenes and SemanticKITTI configs
    resolves in PIPELINES and accepts the config's keyword arguments unchanged (the panoptic config's loader is out of
    scope)"""
    from occformer_amd.registry import Config
    from tests import refshim
    if not refshim.available():
        pytest.skip("needs /root/reference (build container)")
    monkeypatch.chdir(refshim.REFERENCE_ROOT)                            # cls_metas is a path relative to the repo root
    for rel, first, gt in (("occformer_nusc/occformer_nusc_r50_256x704.py", "LoadMultiViewImageFromFiles_OccFormer",
                            "LoadNuscOccupancyAnnotations"),
                           ("occformer_nusc/occformer_nusc_r101_896x1600.py", "LoadMultiViewImageFromFiles_OccFormer",
                            "LoadNuscOccupancyAnnotations"),
                           ("occformer_kitti/occformer_kitti.py", "LoadMultiViewImageFromFiles_SemanticKitti",
                            "LoadSemKittiAnnotation")):
        cfg = Config.fromfile(os.path.join(refshim.REFERENCE_ROOT, "projects/configs", rel))
        for name in ("train_pipeline", "test_pipeline"):
            pipe = PL.Compose(cfg[name], device="cpu")
            kinds = [type(s).__name__ for s in pipe.stages]
            assert kinds[0] == first and gt in kinds and kinds[-2:] == ["OccDefaultFormatBundle3D", "Collect3D"], kinds
            assert pipe.stages[0].is_train == (name == "train_pipeline") and pipe.stages[0].device == "cpu"
