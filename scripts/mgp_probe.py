"""fused mask contraction + pooling per target level (run with OCCF_MASK_POOL_STREAM=0/1)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
me = torch.randn(1, 100, 192, device=dev); feat = torch.randn(1, 200 * 200 * 16, 192, device=dev)
sp = ops.split_bf16(feat)
def bench(fn, iters=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for target in ((25, 25, 2), (50, 50, 4), (100, 100, 8)):
    print(os.environ.get("OCCF_MASK_POOL_STREAM", "1"), target, "%.1f us" % (1e3 * bench(lambda: ops.mask_gemm_pool(me, sp, (200, 200, 16), target))), flush=True)
