"""micro-driver for the PMC passes: the resample + classify kernel of simple_test at the 200-grid
(mask_pred [1, 100, 200, 200, 16] -> class volume [1, 17, 400, 400, 32]), timed with HIP events"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops()
g = torch.Generator().manual_seed(0)
mp = (torch.randn(1, 100, 200, 200, 16, generator=g) * 2).cuda()
cls = torch.randn(1, 100, 18, generator=g).cuda()
for _ in range(3):
    out = ops.upsample_classify(mp, cls, (400, 400, 32))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out = ops.upsample_classify(mp, cls, (400, 400, 32))
e1.record()
torch.cuda.synchronize()
print("upsample_classify ms", e0.elapsed_time(e1) / 10)
