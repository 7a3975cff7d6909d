"""DepthNet's 3x3 weight gradient ([6, 16, 44, 1, 512] -> 512) on the register-transposing kernel vs as a [1, 44, 16] volume
on the G8 kernel (ops.wgrad_2d_as_g8): ms per call."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
ops = get_ops(); dev = torch.device("cuda:0"); torch.manual_seed(0)
x = torch.randn(6, 16, 44, 1, 512, device=dev); dy = torch.randn(6, 16, 44, 1, 512, device=dev) * 1e-4
ref = None
for on in (False, True):
    ops.wgrad_2d_as_g8 = on
    f = lambda: ops.conv3d_wgrad(dy, x, (3, 3, 1), 1, 1)
    out = f()[0]
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    if ref is None: ref = out
    print(f"wgrad_2d_as_g8={on}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call; vs the other path rel L2 {float((out - ref).norm() / ref.norm()):.2e}")
