"""Is the forward bound by the host (launch issue) or by the GPU?  Per forward: the time the host needs to ISSUE every
launch (returns before the GPU is done) against the time until the GPU is done, per stage of the detector; with
OCCF_DECODER_ROWS=0/1 for the decoder's two launch structures."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occformer_amd  # noqa
from occformer_amd import configs
from occformer_amd.registry import build_model

dev = torch.device("cuda", 0)
torch.manual_seed(0)
cfg, meta = configs.workload("nusc_r50_200")
model = build_model(cfg).to(dev).eval()
img_inputs, metas, points = configs.synthetic_sample(meta, dev, seed=0)


def fwd():
    with torch.no_grad():
        vox, _, _ = model.extract_feat(None, img_inputs, metas)
        t1 = time.perf_counter()
        res = model.pts_bbox_head.simple_test(vox, metas, points=points)
    return t1


for _ in range(3):
    fwd()
torch.cuda.synchronize()
n = 20
iss = head_iss = tot = 0.0
for _ in range(n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t1 = fwd()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    iss += t2 - t0
    head_iss += t2 - t1
    tot += t3 - t0
tag = " ".join(f"{k[5:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OCCF_"))
print(f"[{tag}] per forward: host issue {1e3 * iss / n:.2f} ms (decoder head part {1e3 * head_iss / n:.2f} ms), until the GPU is "
      f"done {1e3 * tot / n:.2f} ms")
