"""Time the stages of the nuScenes training data pipeline on the device at the reference's sizes (six 900x1600 frames,
a 34 720-point lidarseg sweep, the 512x512x40 label grid of occformer_nusc_r50_256x704.py): per-stage milliseconds with
HIP events on torch's stream, the frame and point rates, and the number of kernel launches per sample.

    python scripts/pipeline_probe.py [--samples 20] [--small]        (--small: a CPU-sized dry run through the emulation)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import occformer_amd  # noqa: E402,F401
import occformer_amd.ops as ops_mod  # noqa: E402
from occformer_amd import pipeline as PL  # noqa: E402

CAMS = ["CAM_FRONT_LEFT", "CAM_FRONT", "CAM_FRONT_RIGHT", "CAM_BACK_LEFT", "CAM_BACK", "CAM_BACK_RIGHT"]
LEARNING_MAP = {1: 0, 5: 0, 7: 0, 8: 0, 10: 0, 11: 0, 13: 0, 19: 0, 20: 0, 0: 0, 29: 0, 31: 0, 9: 1, 14: 2, 15: 3, 16: 3,
                17: 4, 18: 5, 21: 6, 2: 7, 3: 7, 4: 7, 6: 7, 12: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13, 27: 14, 28: 15,
                30: 16}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=20)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    if args.small:
        from tests.conftest import Backend
        ops_mod._ops = Backend("emu").ops
        dev, src, inp, P, grid = "cpu", (90, 160), (32, 88), 2000, [32, 32, 8]
    else:
        dev, src, inp, P, grid = "cuda", (900, 1600), (256, 704), 34720, [512, 512, 40]
    pc_range = [-51.2, -51.2, -5.0, 51.2, 51.2, 3.0]
    data_config = dict(cams=CAMS, Ncams=6, input_size=inp, src_size=src, resize=(-0.06, 0.11), rot=(-5.4, 5.4), flip=True,
                       crop_h=(0.0, 0.0), resize_test=0.0)
    bda = dict(rot_lim=(0, 0), scale_lim=(0.95, 1.05), flip_dx_ratio=0.5, flip_dy_ratio=0.5, flip_dz_ratio=0.5)
    stages = [PL.LoadMultiViewImageFromFiles_OccFormer(data_config, is_train=True, device=dev),
              PL.CreateDepthFromLiDAR(dataset="nusc"),
              PL.LoadNuscOccupancyAnnotations(is_train=True, grid_size=grid, point_cloud_range=pc_range, bda_aug_conf=bda,
                                              cls_metas=dict(learning_map=LEARNING_MAP), device=dev)]
    rs = np.random.RandomState(0)
    frames = {c: torch.from_numpy(rs.randint(0, 256, src + (3,)).astype(np.uint8)).to(dev) for c in CAMS}
    l2c, intr = {}, {}
    for i, c in enumerate(CAMS):
        a = np.pi / 3 * i
        s2l = np.eye(4)
        s2l[:3, :3] = np.stack([[np.sin(a), -np.cos(a), 0.0], [0.0, 0.0, -1.0], [np.cos(a), np.sin(a), 0.0]], 1)
        s2l[:3, 3] = [1.5 * np.cos(a), 1.5 * np.sin(a), 1.5]
        l2c[c] = np.linalg.inv(s2l)
        intr[c] = np.array([[1260.0 * src[1] / 1600, 0, src[1] / 2], [0, 1260.0 * src[1] / 1600, src[0] / 2], [0, 0, 1.0]])
    pts = torch.from_numpy(np.concatenate([rs.uniform(-50, 50, (P, 2)), rs.uniform(-4, 2, (P, 1)), np.zeros((P, 2))], 1)
                           .astype(np.float32)).to(dev)
    lab = torch.from_numpy(rs.randint(0, 32, P).astype(np.uint8)).to(dev)

    def sample():
        return dict(curr=dict(cams={c: dict(img=frames[c], cam_intrinsic=intr[c]) for c in CAMS}), lidar2cam_dic=l2c,
                    points=pts, points_label=lab)

    gpu = dev == "cuda"
    sync = torch.cuda.synchronize if gpu else (lambda: None)
    for _ in range(3):
        res = sample()
        for st in stages:
            res = st(res)
    sync()
    per = {type(st).__name__: 0.0 for st in stages}
    t0 = time.perf_counter()
    for _ in range(args.samples):
        res = sample()
        for st in stages:
            if gpu:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t = time.perf_counter()
            res = st(res)
            if gpu:
                e1.record()
                e1.synchronize()
                per[type(st).__name__] += e0.elapsed_time(e1)
            else:
                per[type(st).__name__] += 1e3 * (time.perf_counter() - t)
    sync()
    wall = (time.perf_counter() - t0) / args.samples
    print(f"device {dev}: {len(CAMS)} frames {src[0]}x{src[1]} -> {inp[0]}x{inp[1]}, {P} points, label grid {grid}")
    for k, v in per.items():
        print(f"  {k:46s} {v / args.samples:9.3f} ms per sample")
    print(f"  wall (host + device, stages synchronised)      {1e3 * wall:9.3f} ms per sample = {1.0 / wall:.1f} samples/s")
    occ = res["gt_occ"]
    print("  gt_occ", tuple(occ.shape), "occupied voxels", int((occ > 0).sum()), " gt_depths hits", int((res["img_inputs"][7] > 0).sum()))


if __name__ == "__main__":
    main()
