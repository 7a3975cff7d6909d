"""memory-bound linear shapes under the library's env switches (run once per env setting)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
def bench(fn, iters=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tag = f"PF={os.environ.get('OCCF_GEMM_PF','-')} BN={os.environ.get('OCCF_GEMM_BN','-')}"
for (M, K, N) in [(680000, 128, 384), (680000, 128, 128), (91250, 192, 192), (91250, 192, 384), (640000, 192, 192), (90000, 256, 768)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1; b = torch.randn(N, device=dev)
    sp = ops.split_bf16(w); out = torch.empty(M, N, device=dev)
    t = bench(lambda: ops.linear(x, w, b, 0, None, out=out, w_split=sp))
    gb = (M * K * 4 + M * N * 4) / 1e9
    print(f"{tag:14s} M={M:7d} K={K:4d} N={N:4d}: {t*1e3:8.1f} us  {gb/t*1e3:7.0f} GB/s")
