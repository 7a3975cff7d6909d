"""host launch time vs device time of one forward (is the CPU ahead of the GPU?)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from occformer_amd import configs
from occformer_amd.registry import build_model
dev = torch.device("cuda:0"); torch.manual_seed(0)
cfg, meta = configs.nusc_r50("200")
model = build_model(cfg).eval().to(dev)
img_inputs, metas, points = bench.synthetic_sample(meta, dev, seed=0)
def step():
    with torch.no_grad():
        vox, _, _ = model.extract_feat(None, img_inputs, metas)
        return model.pts_bbox_head.simple_test(vox, metas, points=points)
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("host launch %.2f ms, until device idle %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
# per stage host time
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
