"""split-K slice count sweep for the small-M contractions of the bench workload (diagnostics)"""
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
convs = [((6, 16, 44, 1, 512), 512, (3, 3, 1), 1), ((1, 25, 25, 2, 1024), 1024, (3, 3, 3), 1),
         ((1, 50, 50, 4, 512), 1024, (3, 3, 3), 2), ((1, 100, 100, 8, 256), 512, (3, 3, 3), 2),
         ((1, 100, 100, 1, 64), 64, (3, 3, 1), 1), ((1, 50, 50, 1, 128), 128, (3, 3, 1), 1),
         ((1, 25, 25, 1, 256), 256, (3, 3, 1), 1)]
lins = [(1875, 1024, 1024), (1875, 1024, 3072), (4224, 1152, 128), (12500, 512, 512)]
for S in ([int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12]):
    if S: os.environ["OCCF_GEMM_KSPLIT"] = str(S)
    row = []
    for (xs, co, ks, st) in convs:
        x = torch.randn(*xs, device=dev); w = torch.randn(co, ks[0] * ks[1] * ks[2] * xs[-1], device=dev) * 0.02
        sp = ops.split_bf16(w)
        pad = tuple(k // 2 for k in ks)
        t = bench(lambda: ops.conv3d(x, w, ks, st, 1, pad, None, 0, None, w_split=sp))
        row.append(t * 1e3)
    for (M, K, N) in lins:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1
        sp = ops.split_bf16(w)
        t = bench(lambda: ops.linear(x, w, None, 0, None, w_split=sp))
        row.append(t * 1e3)
    print(f"S={S:2d} " + " ".join(f"{v:7.1f}" for v in row), flush=True)
