"""The kernels of the convolution family the `roofline` rows quote, at the metric's shapes, a few launches each (for the
rocprofv3 --pmc passes of scripts/gpu_final.sh): forward (Winograd, three-term), data gradient (Winograd, dy in one fp16
piece), weight gradient (G8, two fp16-piece products), at 192 and 128 channels on the 200 x 200 x 16 grid."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for C in (192, 128):
    X, Y, Z = 200, 200, 16
    x = torch.randn(1, X, Y, Z, C, device=dev)
    dy = torch.randn(1, X, Y, Z, C, device=dev) * 1e-4
    w = torch.randn(C, 27 * C, device=dev) * 0.02
    sp = ops.split_bf16(w)
    for _ in range(n):
        ops.conv3d(x, w, (3, 3, 3), 1, 1, (1, 1, 1), None, 0, w_split=sp)
        ops.conv3d(dy, w, (3, 3, 3), 1, 1, (1, 1, 1), None, 0, w_split=sp, act_f16=True)
        ops.conv3d_wgrad(dy, x, (3, 3, 3), 1, 1)
torch.cuda.synchronize()
print("conv family probe done")
