"""GPU micro-probe for the memory-bound GEMM shapes (diagnostics, not a test)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops()
dev = torch.device("cuda:0")
torch.manual_seed(0)


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M = 680000
x = torch.randn(M, 128, device=dev)
res = torch.randn(M, 128, device=dev)
for N in (128, 384):
    w = torch.randn(N, 128, device=dev) * 0.1
    b = torch.randn(N, device=dev)
    sp = ops.split_bf16(w)
    out = torch.empty(M, N, device=dev)
    for prec in ("bf16x3", "bf16", "f32"):
        ops.precision = prec
        t = bench(lambda: ops.linear(x, w, b, 0, None, out=out, w_split=sp))
        gb = (M * 128 * 4 + M * N * 4) / 1e9
        print(f"linear M={M} K=128 N={N} {prec:7s}: {t*1e3:8.1f} us  {gb/t*1e3:7.0f} GB/s  {2*M*N*128/t/1e9:7.1f} TF")
    ops.precision = "bf16x3"
    if N == 128:
        t = bench(lambda: ops.linear(x, w, b, 0, res, out=out, w_split=sp))
        print(f"   + residual: {t*1e3:8.1f} us  {(3*M*128*4)/1e9/t*1e3:7.0f} GB/s")
# streaming references
y = torch.empty_like(x)
t = bench(lambda: y.copy_(x))
print(f"torch copy 348MB: {t*1e3:.1f} us  {2*x.numel()*4/1e9/t*1e3:.0f} GB/s")
g = torch.ones(128, device=dev); bb = torch.zeros(128, device=dev)
t = bench(lambda: ops.layernorm(x, g, bb))
print(f"layernorm: {t*1e3:.1f} us  {2*x.numel()*4/1e9/t*1e3:.0f} GB/s")
# K scaling at N=128 (compute share)
for K in (128, 256, 512, 1024):
    xk = torch.randn(200000, K, device=dev)
    wk = torch.randn(128, K, device=dev) * 0.05
    spk = ops.split_bf16(wk)
    ok = torch.empty(200000, 128, device=dev)
    t = bench(lambda: ops.linear(xk, wk, None, 0, None, out=ok, w_split=spk))
    print(f"linear M=200000 K={K} N=128 bf16x3: {t*1e3:8.1f} us  {(200000*K*4+200000*128*4)/1e9/t*1e3:7.0f} GB/s  {2*200000*128*K/t/1e9:7.1f} TF")
