"""msda3d_backward at the metric shape: ms per call under the tile-accumulator / pass-loop switches."""
import torch, sys, os
sys.path.insert(0, ".")
from occformer_amd.ops import get_ops
ops = get_ops()
levels = [(100,100,8),(50,50,4),(25,25,2)]
Nq = sum(x*y*z for x,y,z in levels); H=8; P=4; L=3; E=192
dev="cuda:0"
torch.manual_seed(0)
value = torch.randn(1, Nq, E, device=dev); off = torch.randn(1, Nq, H*L*P*3, device=dev)*0.5; lg = torch.randn(1, Nq, H*L*P, device=dev); dout = torch.randn(1, Nq, E, device=dev)
f = lambda: ops.msda3d_backward(value, off, lg, dout, levels, H, P)
r = f()
for _ in range(3): f()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("msda3d_backward PASSLOOP=%s" % os.environ.get("OCCF_MSDA_PASSLOOP", "1"), round(e0.elapsed_time(e1)/10, 3), "ms")
