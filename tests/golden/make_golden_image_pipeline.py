"""Golden vectors for the ``img_inputs`` producer, produced by the REFERENCE's own class
``LoadMultiViewImageFromFiles_OccFormer`` (projects/mmdet3d_plugin/datasets/pipelines/loading_nusc_imgs.py) imported
unmodified through tests/refshim and run on seeded synthetic frames with the real Pillow:
  * ``get_inputs`` in train mode (random resize / crop / flip / rotate per camera, numpy global RNG seeded) and in test
    mode: images, rots, trans, intrins, post_rots, post_trans, the uint8 ``canvas``;
two stand-ins, both stated here: ``mmcv.imread`` returns the synthetic frame registered under the "file name" (JPEG
decoding is I/O), and -- mmcv not being installed -- ``normalize_img`` is oracle.image_pipeline_ref.normalize, the
restatement of mmcv.image.imnormalize (so the float images pin the PIL half and the channel / axis order, not mmcv's
last ulp).  The oracle (oracle/image_pipeline_ref.py) is checked against the canvas bit for bit here.

    python tests/golden/make_golden_image_pipeline.py        ->  tests/golden/image_pipeline.npz
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import refshim  # noqa: E402
from oracle import image_pipeline_ref as IR  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
CAMS = ["CAM_FRONT", "CAM_BACK_LEFT", "CAM_BACK"]
DATA_CONFIG = dict(cams=CAMS, Ncams=3, input_size=(32, 88), src_size=(90, 160), resize=(-0.06, 0.11), rot=(-5.4, 5.4),
                   flip=True, crop_h=(0.0, 0.0), resize_test=0.0)


def frames(seed=0):
    """seeded synthetic camera frames (uint8 [90, 160, 3], smooth + noise so that resampling is exercised) and rig"""
    rng = np.random.RandomState(seed)
    out, l2c, intr = {}, {}, {}
    yy, xx = np.mgrid[0:90, 0:160]
    for i, c in enumerate(CAMS):
        base = 127 + 90 * np.sin(xx / (7.0 + i) + i) * np.cos(yy / (5.0 + 2 * i))
        img = np.clip(base[..., None] + rng.randint(-40, 40, (90, 160, 3)), 0, 255).astype(np.uint8)
        out[c] = img
        a = 0.7 * i
        R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]])
        m = np.eye(4)
        m[:3, :3] = R
        m[:3, 3] = [0.3 * i, -0.2, 1.5]
        l2c[c] = m
        intr[c] = np.array([[140.0 + i, 0, 80.0], [0, 141.0, 45.0], [0, 0, 1]])
    return out, l2c, intr


def results_for(imgs, l2c, intr):
    return dict(curr=dict(cams={c: dict(data_path=c, cam_intrinsic=intr[c]) for c in CAMS}), lidar2cam_dic=l2c)


def main():
    refshim.install()
    import mmcv
    mod = refshim.ref("datasets.pipelines.loading_nusc_imgs")
    imgs, l2c, intr = frames()
    mmcv.imread = lambda name, flag="color": imgs[name]
    out = {}
    for mode, seed in (("train", 5), ("train2", 11), ("test", 0)):
        t = mod.LoadMultiViewImageFromFiles_OccFormer(DATA_CONFIG, is_train=mode != "test")
        t.normalize_img = lambda img, img_norm_cfg=None: torch.from_numpy(IR.normalize(np.array(img)))
        np.random.seed(seed)
        res = results_for(imgs, l2c, intr)
        x, rots, trans, intrins, post_rots, post_trans, gtd, s2s = t.get_inputs(res)
        # the oracle on the same draws
        np.random.seed(seed)
        for k, c in enumerate(CAMS):
            rs, dims, crop, flip, rot = IR.sample_augmentation(90, 160, DATA_CONFIG, mode != "test")
            cv = IR.img_transform_core(imgs[c], dims, crop, flip, rot)
            assert np.array_equal(cv, res["canvas"][k]), (mode, c)
            assert np.array_equal(IR.normalize(cv), x[k].numpy())
        out[f"{mode}.imgs"], out[f"{mode}.canvas"] = x.numpy(), res["canvas"]
        out[f"{mode}.rots"], out[f"{mode}.trans"], out[f"{mode}.intrins"] = rots.numpy(), trans.numpy(), intrins.numpy()
        out[f"{mode}.post_rots"], out[f"{mode}.post_trans"] = post_rots.numpy(), post_trans.numpy()
        out[f"{mode}.seed"] = np.int64(seed)
        print(mode, "canvas mean", float(res["canvas"].mean()), "zero-filled pixels", int((res["canvas"].sum(-1) == 0).sum()))
    np.savez_compressed(os.path.join(OUT, "image_pipeline.npz"), **out)
    print("wrote", os.path.join(OUT, "image_pipeline.npz"), os.path.getsize(os.path.join(OUT, "image_pipeline.npz")), "bytes")


if __name__ == "__main__":
    main()
