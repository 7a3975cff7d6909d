"""The streaming linear shapes of the path (csrc/gemm_stream.h) against the tile kernel (OCCF_GEMM_STREAM=0) and the
two floors: HBM (rows in + rows out at 6.3 TB/s, the chip's measured float4 copy rate) and the matrix pipe (3 bf16
products per product at 1.25 PF executed).  Buffers rotate through > 1 GB so that no operand is served from the MALL."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops

ops = get_ops()
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(680000, 128, 128, 0), (680000, 128, 384, 0), (680000, 128, 128, 2), (91250, 192, 192, 0), (91250, 192, 384, 0),
          (91250, 192, 768, 1), (90000, 256, 256, 0), (90000, 256, 768, 0), (640000, 224, 192, 0), (640000, 192, 192, 0),
          (640000, 192, 128, 0), (80000, 192, 192, 0), (12500, 512, 512, 0)]


if "quick" in sys.argv:            # (counter passes: three shapes, few iterations)
    SHAPES = [(680000, 128, 128, 0), (680000, 128, 384, 0), (91250, 192, 768, 1)]


def bench(fn, n_buf, iters=6 if "quick" in sys.argv else 24):
    for i in range(3):
        fn(i % n_buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn(i % n_buf)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print(f"{'M':>7s} {'K':>4s} {'N':>4s} act res | tile kernel us | stream us (speed-up) | GB/s | HBM floor us | MFMA floor us | max err vs tile")
for (M, K, N, act) in SHAPES:
    for res in ((False, True) if (M, K, N) in ((680000, 128, 128), (91250, 192, 192)) and act == 0 else (False,)):
        gb = (M * K + M * N * (2 if res else 1)) * 4 / 1e9
        n_buf = max(2, int(1.2 / gb) + 1)
        xs = [torch.randn(M, K, device=dev) for _ in range(n_buf)]
        outs = [torch.empty(M, N, device=dev) for _ in range(n_buf)]
        rs = [torch.randn(M, N, device=dev) for _ in range(n_buf)] if res else None
        w = torch.randn(N, K, device=dev) * K ** -0.5
        b = torch.randn(N, device=dev)
        sp = ops.split_bf16(w)
        fn = lambda i: ops.linear(xs[i], w, b, act, rs[i] if res else None, out=outs[i], w_split=sp)
        os.environ["OCCF_GEMM_STREAM"] = "0"
        t0 = bench(fn, n_buf)
        ref = outs[0].clone()
        os.environ["OCCF_GEMM_STREAM"] = "1"
        n0 = ops.lib.occf_linear_stream_launches()
        t1 = bench(fn, n_buf)
        taken = ops.lib.occf_linear_stream_launches() > n0
        err = float((outs[0] - ref).abs().max())
        hbm = gb / 6.3e3 * 1e6
        mfma = 2.0 * M * K * N * 3 / 1.25e15 * 1e6
        print(f"{M:7d} {K:4d} {N:4d} {act:3d} {int(res):3d} | {t0:8.1f} | {t1:8.1f} ({t0 / t1:4.2f}x{'' if taken else ' NOT TAKEN'}) | "
              f"{gb / t1 * 1e6:6.0f} | {hbm:6.1f} | {mfma:6.1f} | {err:.1e}", flush=True)
        del xs, outs, rs
