"""DepthNet's 3x3 convolution shape (6 cameras x 16 x 44 positions, 512 -> 512: M = 4 224 rows, K = 4 608) on the generic
implicit-GEMM kernel under the split-K / tile-width switches (one process per setting): time per launch from HIP events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(6, 16, 44, 1, 512, device=dev)
w = torch.randn(512, 9 * 512, device=dev) * 0.02
sp = ops.split_bf16(w)
fn = lambda: ops.conv3d(x, w, (3, 3, 1), 1, 1, (1, 1, 0), None, 1, w_split=sp)
out = fn()
for _ in range(3): fn()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): fn()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 20
tag = " ".join(f"{k[5:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OCCF_"))
fl = 2.0 * 6 * 16 * 44 * 512 * 4608
print(f"[{tag}] conv 3x3 [6,16,44,512]->512: {1e3 * t:7.1f} us  {fl / t / 1e9:6.1f} TF algorithmic  checksum {float(out.double().abs().sum()):.6e}", flush=True)
