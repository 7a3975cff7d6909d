"""Times the training-only rows at nuScenes R50 sizes on the GPU: head.loss over the 10
prediction sets (100 queries, 128x128x16 mask logits, 256x256x32 GT, ~34k LiDAR points, 50176 sampled points) and,
on a bounded sample (one layer), the CPU oracle beside it.   python scripts/train_probe.py [--layers 10] [--kitti]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd import configs                     # noqa: E402
from occformer_amd.registry import HEADS              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--kitti", action="store_true")
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    model, _ = configs.nusc_r50(grid="reference")
    hc = dict(model["pts_bbox_head"])
    tc = dict(model["train_cfg"]["pts"])
    nc = 17
    if a.kitti:
        hc.update(type="Mask2FormerOccHead", num_occupancy_classes=20)
        hc["loss_cls"] = dict(hc["loss_cls"], class_weight=[1.0] * 20 + [0.1])
        nc = 20
    head = HEADS.build(dict(hc, train_cfg=tc, test_cfg=None)).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    Q, grid, occ = 100, (128, 128, 16), (256, 256, 32)
    cls = [torch.randn((1, Q, nc + 1), device=dev, generator=g) for _ in range(a.layers)]
    masks = [torch.randn((1, Q) + grid, device=dev, generator=g) * 3 for _ in range(a.layers)]
    lab = torch.randint(0, 18 if not a.kitti else 21, (1, 32, 32, 8), device=dev, generator=g)
    lab = torch.where(lab >= nc, torch.full_like(lab, 255), lab)
    gt_occ = lab.repeat_interleave(8, 1).repeat_interleave(8, 2).repeat_interleave(4, 3)
    pcr = torch.tensor(hc.get("point_cloud_range") or [-51.2, -51.2, -5, 51.2, 51.2, 3], device=dev)
    pts = torch.rand((34000, 3), device=dev, generator=g) * (pcr[3:] - pcr[:3]) + pcr[:3]
    pts = [torch.cat((pts, torch.randint(1, 17, (34000, 1), device=dev, generator=g).float()), 1)]
    metas = [dict(occ_size=list(occ), pc_range=pcr.tolist())]
    gl, gm = head.preprocess_gt(gt_occ, metas)
    gt = (gl, gm, pts, metas) if not a.kitti else (gl, gm, metas)

    def step():
        return head.loss(cls, masks, *gt)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    t_gpu = time.perf_counter() - t0
    res = {"what": "head.loss", "head": hc["type"], "layers": a.layers, "n_gt": int(gl[0].shape[0]),
           "gpu_ms": round(t_gpu * 1e3, 2), "gpu_ms_per_layer": round(t_gpu * 1e3 / a.layers, 2),
           "loss_cls": float(out["loss_cls"]), "loss_mask": float(out["loss_mask"]), "loss_dice": float(out["loss_dice"])}
    if not a.no_cpu:
        from oracle import occformer_train_ref as T   # checker / CPU baseline only
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        ocfg = dict(point_cloud_range=pcr.tolist(), num_points=tc["num_points"], oversample_ratio=tc["oversample_ratio"],
                    importance_sample_ratio=tc["importance_sample_ratio"], padding_mode="border", num_classes=nc,
                    class_weight=head.class_weight, align_corners=True)
        if a.kitti:
            ocfg["sample_weights"] = head.sample_weights
        c0, m0 = cls[0].cpu(), masks[0].cpu()
        glc, gmc, ptc = [x.cpu() for x in gl], [x.cpu() for x in gm], [p.cpu() for p in pts]
        t0 = time.perf_counter()
        if a.kitti:
            T.kitti_loss_single(c0, m0, glc, gmc, ocfg, T.GlobalTorchRNG())
        else:
            T.nusc_loss_single(c0, m0, glc, gmc, ptc, ocfg, T.GlobalTorchRNG())
        res["cpu_ms_per_layer"] = round((time.perf_counter() - t0) * 1e3, 1)
        res["cpu_threads"] = torch.get_num_threads()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
