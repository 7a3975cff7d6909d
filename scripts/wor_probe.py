import os
import sys
import torch
import ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from occformer_amd.ops import get_ops
from tests import paramgen
ops = get_ops()
for (R, V, k, mode) in [(1, 300, 1, 0), (1, 300, 1, 2), (1, 300, 5, 2), (3, 5000, 700, 2)]:
    w = paramgen.uniform("sw", (1, V), 2) ** 3
    w[:, ::7] = 0.0
    u = paramgen.uniform("su", (R, V), 3).clamp_min(1e-12)
    q = -torch.log(u)
    keys = torch.where(w > 0, w / q.clamp_min(1e-38), torch.zeros(()))
    keys = keys.expand(R, V)
    wd, nd = w.cuda(), (u if mode == 0 else q).cuda().contiguous()
    need = ops.lib.occf_sample_wor_workspace(R, V)
    ws = torch.zeros((need,), device="cuda")
    out = torch.full((R, k), -1, dtype=torch.int64, device="cuda")
    rc = ops.lib.occf_sample_wor_fwd(ctypes.c_void_p(wd.data_ptr()), ctypes.c_void_p(nd.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), R, V, k, int(R > 1), int(mode == 2), ctypes.c_void_p(0))
    torch.cuda.synchronize()
    gk = ws[:R * V].view(R, V).cpu()
    st = ws[R * V + R * 2048: R * V + R * 2048 + R * 4 + R * 2].view(torch.int32).cpu()
    print("case", R, V, k, mode, "rc", rc)
    print("  key max abs diff", float((gk - keys).abs().max()), " rel", float(((gk - keys).abs() / keys.clamp_min(1e-30)).max()))
    kth = torch.topk(keys, k, dim=1)[0][:, -1]
    print("  expected thr bits", [hex(int(x)) for x in kth.view(torch.int32)], " state", [hex(int(x) & 0xffffffff) for x in st])
    ref = torch.topk(keys, k, dim=1)[1]
    for r in range(R):
        a, b = set(out[r].cpu().tolist()), set(ref[r].tolist())
        print("  row", r, "match", a == b, "missing", len(b - a), "extra", len(a - b))
