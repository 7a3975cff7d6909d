"""The convolutions of the training step that run on the generic implicit-GEMM kernel (csrc/gemm_bf16.hip CONV == 1:
strided, dilated, 2-D, Z < 4), forward, at the metric's shapes: ms per call and TFLOP/s."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CASES = [((1, 200, 200, 16, 128), 256, (3, 3, 3), 2, 1), ((1, 100, 100, 8, 256), 512, (3, 3, 3), 2, 1),
         ((1, 50, 50, 4, 512), 1024, (3, 3, 3), 2, 1), ((6, 16, 44, 1, 512), 512, (3, 3, 1), 1, 1),
         ((1, 25, 25, 2, 1024), 1024, (3, 3, 3), 1, 1), ((1, 200, 200, 1, 32), 32, (3, 3, 1), 1, 6),
         ((1, 100, 100, 1, 64), 64, (3, 3, 1), 1, 12), ((1, 200, 200, 16, 128), 256, (1, 1, 1), 2, 1)]
for shape, Cout, k, stride, dil in CASES:
    Cin = shape[-1]
    x = torch.randn(*shape, device=dev)
    w = torch.randn(Cout, k[0] * k[1] * k[2] * Cin, device=dev) * 0.02
    sp = ops.split_bf16(w)
    f = lambda: ops.conv3d(x, w, k, stride, dil, None, None, 0, w_split=sp)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"conv x{shape} -> {Cout} k{k} s{stride} d{dil}: {ms:.3f} ms  {ops.last_flops / ms / 1e9:.1f} TF")
