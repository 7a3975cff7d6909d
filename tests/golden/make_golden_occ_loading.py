"""Golden vectors for the occupancy ground-truth loaders and the SemanticKITTI ``img_inputs`` producer, produced by the
REFERENCE's own classes imported unmodified through tests/refshim:
  * ``LoadNuscOccupancyAnnotations`` (projects/mmdet3d_plugin/datasets/pipelines/loading_nusc_occ.py) on a seeded
    synthetic sweep written to temporary ``.bin`` files, train mode (flips; and a second configuration with a BEV rotation
    of the points) and test mode: gt_occ, points_occ, bda_rot;
  * ``LoadSemKittiAnnotation`` (loading_kitti_occ.py) on a seeded label volume, flips only and flips + rotation;
  * ``LoadMultiViewImageFromFiles_SemanticKitti`` (loading_kitti_imgs.py) on one synthetic frame, train and test mode.
Stand-ins, all stated: numba is not installed -> ``numba.jit`` is a pass-through, the reference's nb_process_label loop
runs as plain Python; numpy 2 has no ``np.int`` -> aliased to ``int`` (what it was); ``mmcv.imread`` returns the synthetic
frame and ``normalize_img`` is oracle.image_pipeline_ref.normalize (mmcv is not installed -- as in
make_golden_image_pipeline.py).  The numpy oracle (oracle/occ_loading_ref.py) is checked against the reference here.

    python tests/golden/make_golden_occ_loading.py        ->  tests/golden/occ_loading.npz
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import refshim  # noqa: E402
from oracle import image_pipeline_ref as IR  # noqa: E402
from oracle import occ_loading_ref as OR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
GRID, RANGE = [20, 24, 4], [-10.0, -12.0, -2.0, 10.0, 12.0, 2.0]
BDA_FLIP = dict(rot_lim=(0, 0), scale_lim=(0.95, 1.05), flip_dx_ratio=0.5, flip_dy_ratio=0.5)
BDA_ROT = dict(rot_lim=(-22.5, 22.5), scale_lim=(0.95, 1.05), flip_dx_ratio=0.5, flip_dy_ratio=0.5, flip_dz_ratio=0.5)
KITTI_RANGE = [0, -25.6, -2, 51.2, 25.6, 4.4]
KITTI_DATA = dict(input_size=(48, 160), resize=(-0.06, 0.11), rot=(-5.4, 5.4), flip=True, crop_h=(0.0, 0.0), resize_test=0.0)
LEARNING_MAP = {1: 0, 5: 0, 7: 0, 8: 0, 10: 0, 11: 0, 13: 0, 19: 0, 20: 0, 0: 0, 29: 0, 31: 0, 9: 1, 14: 2, 15: 3, 16: 3,
                17: 4, 18: 5, 21: 6, 2: 7, 3: 7, 4: 7, 6: 7, 12: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13, 27: 14, 28: 15,
                30: 16}


def sweep(seed=3, P=3000):
    """seeded LiDAR sweep: clustered points (several per voxel, mixed labels -> ties and majorities), a few outside the
    range; raw lidarseg labels 0..31"""
    rng = np.random.RandomState(seed)
    centres = rng.uniform([-11, -13, -2.5], [11, 13, 2.5], (150, 3))
    pts = centres[rng.randint(0, 150, P)] + rng.normal(0, 0.35, (P, 3))
    five = np.concatenate([pts, rng.uniform(0, 1, (P, 2))], 1).astype(np.float32)
    lab = rng.randint(0, 32, P).astype(np.uint8)
    lab[rng.rand(P) < 0.5] = rng.choice([9, 17, 24, 30])                  # a dominant class so majorities exist
    return five, lab


def label_volume(seed=4, shape=(32, 32, 8)):
    rng = np.random.RandomState(seed)
    v = rng.randint(0, 20, shape).astype(np.uint8)
    v[rng.rand(*shape) < 0.1] = 255
    return v


def kitti_frame(seed=6):
    rng = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:112, 0:370]
    base = 120 + 80 * np.sin(xx / 9.0) * np.cos(yy / 6.0)
    return np.clip(base[..., None] + rng.randint(-35, 35, (112, 370, 3)), 0, 255).astype(np.uint8)


def kitti_calib():
    l2c = np.array([[0.0, -1.0, 0.0, 0.02], [0.0, 0.0, -1.0, -0.07], [1.0, 0.0, 0.0, -0.3], [0, 0, 0, 1.0]])
    intr = np.array([[200.0, 0, 185.0, 0], [0, 200.0, 56.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])
    return l2c, intr


def placeholder_inputs():
    return tuple(torch.full((1,), float(i)) for i in range(8))


def install_stand_ins():
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = nb
    if not hasattr(np, "int"):
        np.int = int
    refshim.install()


def main():
    install_stand_ins()
    import mmcv
    import yaml
    nusc = refshim.ref("datasets.pipelines.loading_nusc_occ")
    kitti = refshim.ref("datasets.pipelines.loading_kitti_occ")
    kimg = refshim.ref("datasets.pipelines.loading_kitti_imgs")
    out = {}
    five, raw = sweep()
    out["nusc.points"], out["nusc.raw_labels"] = five, raw
    with tempfile.TemporaryDirectory() as tmp:
        five.tofile(os.path.join(tmp, "sweep.bin"))
        raw.tofile(os.path.join(tmp, "seg.bin"))
        with open(os.path.join(tmp, "meta.yaml"), "w") as f:
            yaml.safe_dump(dict(learning_map=LEARNING_MAP), f)
        for tag, conf, train, seed in (("flip", BDA_FLIP, True, 1), ("flip2", BDA_FLIP, True, 2), ("rot", BDA_ROT, True, 8),
                                       ("test", BDA_FLIP, False, 0)):
            t = nusc.LoadNuscOccupancyAnnotations(data_root=tmp, is_train=train, grid_size=GRID, point_cloud_range=RANGE,
                                                  bda_aug_conf=conf, cls_metas=os.path.join(tmp, "meta.yaml"))
            np.random.seed(seed)
            res = t(dict(lidarseg="seg.bin", pts_filename=os.path.join(tmp, "sweep.bin"), img_inputs=placeholder_inputs()))
            bda = res["img_inputs"][6].numpy()
            assert len(res["img_inputs"]) == 9 and float(res["img_inputs"][7]) == 6.0
            # the numpy oracle on the same draws
            np.random.seed(seed)
            if train:
                rot, _, fx, fy, fz = OR.sample_bda(conf)
                mat = OR.bda_matrix(rot, fx, fy, fz)
            else:
                rot, mat = 0.0, np.eye(3, dtype=np.float32)
            assert np.abs(mat - bda).max() <= (0 if rot == 0 else 1e-7), (tag, mat, bda)
            occ, pocc = OR.nusc_occupancy(five[:, :3], raw, LEARNING_MAP, GRID, RANGE, bda)
            assert np.array_equal(occ, res["gt_occ"].numpy()), tag
            assert np.array_equal(pocc, res["points_occ"].numpy()), tag
            out[f"nusc.{tag}.gt_occ"] = res["gt_occ"].numpy().astype(np.uint8)
            out[f"nusc.{tag}.points_occ"], out[f"nusc.{tag}.bda_rot"] = res["points_occ"].numpy(), bda
            out[f"nusc.{tag}.seed"] = np.int64(seed)
            g = res["gt_occ"].numpy()
            print("nusc", tag, "occupied", int((g > 0).sum()), "ignored", int((g == 255).sum()), "bda diag", np.diag(bda))
    vol = label_volume()
    out["kitti.gt_in"] = vol
    for tag, conf, seed in (("flip", dict(BDA_FLIP, flip_dz_ratio=0.5), 3), ("flip2", dict(BDA_FLIP, flip_dz_ratio=0.5), 5),
                            ("rot", BDA_ROT, 4), ("rot2", BDA_ROT, 12)):
        t = kitti.LoadSemKittiAnnotation(conf, is_train=True, point_cloud_range=KITTI_RANGE)
        np.random.seed(seed)
        res = t(dict(gt_occ=vol.copy(), img_inputs=placeholder_inputs()))
        bda = res["img_inputs"][6].numpy()
        np.random.seed(seed)
        rot, _, fx, fy, fz = OR.sample_bda(conf)
        occ, mat = OR.voxel_transform(vol, rot, fx, fy, fz, center=(np.array(KITTI_RANGE[:3]) + np.array(KITTI_RANGE[3:])) / 2)
        assert np.array_equal(occ, res["gt_occ"].numpy()), tag
        assert np.abs(mat - bda).max() <= (0 if rot == 0 else 1e-5), (tag, np.abs(mat - bda).max())
        out[f"kitti.{tag}.gt_occ"], out[f"kitti.{tag}.bda_rot"] = res["gt_occ"].numpy().astype(np.uint8), bda
        out[f"kitti.{tag}.seed"] = np.int64(seed)
        print("kitti", tag, "rot", rot, "flips", fx, fy, fz, "filled", int((res["gt_occ"].numpy() == 255).sum()))
    t = kitti.LoadSemKittiAnnotation(BDA_FLIP, is_train=False)
    res = t(dict(gt_occ=None, img_inputs=placeholder_inputs()))
    assert torch.equal(res["img_inputs"][6], torch.eye(4))
    # the monocular img_inputs producer
    frame = kitti_frame()
    l2c, intr = kitti_calib()
    out["kimg.frame"] = frame
    mmcv.imread = lambda name, flag="color": frame
    for mode, seed in (("train", 7), ("test", 0)):
        t = kimg.LoadMultiViewImageFromFiles_SemanticKitti(KITTI_DATA, is_train=mode == "train")
        t.normalize_img = lambda img, img_norm_cfg=None: torch.from_numpy(IR.normalize(np.array(img)))
        np.random.seed(seed)
        res = dict(img_filename=["frame"], cam_intrinsic=[intr], lidar2cam=[l2c])
        x, rots, trans, intrins, post_rots, post_trans, depth, c2l = t.get_inputs(res)
        assert x.shape == (1, 3, 48, 160) and depth.shape == (1, 1) and res["canvas"].shape == (1, 48, 160, 3)
        for k, v in (("imgs", x), ("rots", rots), ("trans", trans), ("intrins", intrins), ("post_rots", post_rots),
                     ("post_trans", post_trans), ("cam2lidar", c2l)):
            out[f"kimg.{mode}.{k}"] = v.numpy()
        out[f"kimg.{mode}.canvas"], out[f"kimg.{mode}.seed"] = res["canvas"], np.int64(seed)
        print("kimg", mode, "canvas mean", float(res["canvas"].mean()))
    path = os.path.join(OUT, "occ_loading.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
