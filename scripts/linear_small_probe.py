"""Few-tile / long-K and tiny linears of the training step: us per call and error against float64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
for (M, K, N) in [(200, 50176, 17), (100, 50176, 17), (170, 192, 640000), (1875, 1024, 1024), (100, 192, 192)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.02; sp = ops.split_bf16(w)
    f = lambda: ops.linear(x, w, None, w_split=sp)
    ref = x.double() @ w.double().t()
    out = f(); err = ((out.double() - ref).norm() / ref.norm()).item()
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"linear [{M},{K}]x[{N},{K}]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  rel err {err:.2e}")
