"""G8 weight gradient (two fp16-piece products) at the metric's shapes under the stage geometry switches
OCCF_WG8_KS (16 KS rows per stage) / OCCF_WG8_ST (stages in LDS): ms per call (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
for C in (192, 128):
    X, Y, Z = 200, 200, 16
    x = torch.randn(1, X, Y, Z, C, device=dev)
    dy = torch.randn(1, X, Y, Z, C, device=dev) * 1e-4
    f = lambda: ops.conv3d_wgrad(dy, x, (3, 3, 3), 1, 1)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"KS={os.environ.get('OCCF_WG8_KS', '-')} ST={os.environ.get('OCCF_WG8_ST', '-')} C={C}: {e0.elapsed_time(e1) / 10:.3f} ms (pre-split passes included)")
