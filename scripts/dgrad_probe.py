"""The strided data gradients of the encoder's downsampling convolutions at the metric's shapes (class-major implicit
GEMM, csrc/gemm_bf16.hip CONV == 2): ms per call and the rate on the FORWARD convolution's multiply-adds."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
CASES = [((1, 100, 100, 8, 256), 128, (3, 3, 3)), ((1, 50, 50, 4, 512), 256, (3, 3, 3)), ((1, 25, 25, 2, 1024), 512, (3, 3, 3)),
         ((1, 100, 100, 8, 256), 128, (1, 1, 1)), ((1, 50, 50, 4, 512), 256, (1, 1, 1))]
for dshape, Cin, k in CASES:
    B, Xo, Yo, Zo, Cout = dshape
    dy = torch.randn(*dshape, device=dev)
    taps = k[0] * k[1] * k[2]
    wt = torch.randn(Cin, taps * Cout, device=dev) * 0.02
    sp = ops.split_bf16(wt)
    in_shape = (B, 2 * Xo, 2 * Yo, 2 * Zo, Cin)
    f = lambda: ops.conv3d_dgrad(dy, sp, in_shape, k, 2, 1)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"dgrad dy{dshape} -> Cin {Cin} k{k} s2: {ms:.3f} ms  {ops.last_flops / ms / 1e9:.1f} TF (forward multiply-adds x 2)")
