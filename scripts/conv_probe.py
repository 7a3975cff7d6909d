"""3x3x3 halo convolution at the encoder / neck shapes under the library's env switches (one process per setting):
per-launch time from HIP events, algorithmic TF/s against the bf16 peak, and an output checksum to compare settings."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
def bench(fn, iters=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
tag = " ".join(f"{k[5:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("OCCF_"))
shapes = [(200, 200, 16, 192, 192), (200, 200, 16, 128, 128), (100, 100, 8, 256, 256), (128, 128, 16, 128, 128)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
for (X, Y, Z, Ci, Co) in shapes:
    x = torch.randn(1, X, Y, Z, Ci, device=dev)
    w = torch.randn(Co, 27 * Ci, device=dev) * 0.02
    sp = ops.split_bf16(w)
    out = ops.conv3d(x, w, (3, 3, 3), 1, 1, (1, 1, 1), None, 0, w_split=sp)
    t = bench(lambda: ops.conv3d(x, w, (3, 3, 3), 1, 1, (1, 1, 1), None, 0, w_split=sp))
    fl = 2.0 * 27 * Ci * Co * X * Y * Z
    print(f"[{tag}] conv3 {X}x{Y}x{Z} {Ci}->{Co}: {t:7.3f} ms  {fl / t / 1e9:7.1f} TF/s algorithmic = {fl / t / 1e9 / 2500:.3f} of bf16 peak"
          f"  checksum {float(out.double().sum()):.6e} {float(out.double().abs().sum()):.6e}", flush=True)
