"""Does the whole forward capture into a HIP graph, and what does replay save?  (GPU box; diagnostics)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from occformer_amd import configs
from occformer_amd.registry import build_model

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg, meta = configs.nusc_r50("200")
model = build_model(cfg).eval().to(dev)
img_inputs, metas, points = bench.synthetic_sample(meta, dev, seed=0)

def step():
    with torch.no_grad():
        vox, _, _ = model.extract_feat(None, img_inputs, metas)
        return model.pts_bbox_head.simple_test(vox, metas, points=points)

for _ in range(3):
    ref = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(8):
    step()
torch.cuda.synchronize()
print("eager   %.2f ms/step" % ((time.perf_counter() - t0) / 8 * 1e3), flush=True)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
try:
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        g.replay()
    torch.cuda.synchronize()
    print("graph   %.2f ms/step" % ((time.perf_counter() - t0) / 8 * 1e3))
    a, b = out["output_voxels"][0], ref["output_voxels"][0]
    print("max abs diff graph vs eager", float((a - b).abs().max()))
except Exception as e:
    print("capture failed:", type(e).__name__, str(e)[:300])
