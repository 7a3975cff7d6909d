"""Where does the msda value-gradient tile kernel's time go?  Ablated builds of csrc/msda3d.hip (textual edits of a COPY,
compiled on the spot; wrong results by construction), the whole backward timed with HIP events at the metric's shape
(91 250 queries, 8 heads x 24 channels, 3 levels x 4 points):

    full          as shipped
    no_atomics    the tile kernel's LDS atomics replaced by nothing (the products are kept alive by a never-taken store)
    plain_store   ... replaced by plain 64-bit LDS stores (same addresses, no read-modify-write)
    no_softmax    the per-query softmax statistics replaced by constants
    no_flush      the tile -> scratch conversion / stores skipped
    no_zero       the tile's zero fill skipped
    no_queries    the query loop skipped (zero fill + flush remain)
    skeleton      no_queries + no_flush + no_zero: launch, geometry and barriers only
    no_tile       the three tile launches skipped altogether (gather pass + the offset / logit gradient pass remain)
    no_gather_p   the offset / logit gradient pass (msda3d_bwd_kernel) skipped

    python scripts/msda_bwd_ablation_probe.py [launches]"""
import ctypes
import os
import shutil
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "occformer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result", "-DOCCF_ABI_HASH=0"]


def edit(src, what):
    def rep(a, b, cnt=1):
        nonlocal src
        assert src.count(a) == cnt, (src.count(a), a)
        src = src.replace(a, b)
    if "no_atomics" in what:
        rep("          for (int c = 0; c < CPL; ++c) atomicAdd(t + c, msda_fx(cs * gch[c]));",
            "          for (int c = 0; c < CPL; ++c) if (cs * gch[c] == 12345.678f) t[c] = 1ull;")
    if "plain_store" in what:
        rep("          for (int c = 0; c < CPL; ++c) atomicAdd(t + c, msda_fx(cs * gch[c]));",
            "          for (int c = 0; c < CPL; ++c) t[c] = msda_fx(cs * gch[c]);")
    if "no_flush" in what:
        rep("    slab[i] = (float)((double)(long long)tile[cell * CHP + (i - cell * CH)] * (double)fx_inv);",
            "    if (fx_inv == 12345.678f) slab[i] = (float)((double)(long long)tile[cell * CHP + (i - cell * CH)] * (double)fx_inv);")
    if "no_zero" in what:
        rep("  for (long i = threadIdx.x; i < ncell * CHP; i += NT) tile[i] = 0ull;", "  if (fx_inv == 12345.678f) for (long i = threadIdx.x; i < ncell * CHP; i += NT) tile[i] = 0ull;")
    if "no_softmax" in what:
        rep("      for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);\n      float sum = 0.f;\n      for (int i = 0; i < LP; ++i) sum += expf(lg[i] - mx);\n      inv = 1.0f / sum;",
            "      mx = lg[0]; inv = 0.08f;")
    if "no_queries" in what:
        rep("  for (int u = slot; u < n_q; u += slots) {", "  for (int u = slot; u < n_q && fx_inv == 12345.678f; u += slots) {")
    if "no_tile" in what:
        rep("        if (tc.cpl == 6)\n          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<6>, grid,", "        if (tc.cpl == 77)\n          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<6>, grid,")
        rep("        else\n          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<3>, grid,", "        else if (tc.cpl == 78)\n          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<3>, grid,")
    return src


VARIANTS = {"full": (), "no_atomics": ("no_atomics",), "no_softmax": ("no_softmax",), "no_flush": ("no_flush",),
            "no_zero": ("no_zero",), "no_queries": ("no_queries",), "skeleton": ("no_queries", "no_flush", "no_zero"),
            "no_tile": ("no_tile",)}


def build(name):
    d = f"/tmp/msda_{name}/a/b"
    shutil.rmtree(f"/tmp/msda_{name}", ignore_errors=True)
    os.makedirs(d)
    shutil.copytree(os.path.join(ROOT, "include"), f"/tmp/msda_{name}/include")
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            shutil.copy(os.path.join(CSRC, f), d)
    p = os.path.join(d, "msda3d.hip")
    open(p, "w").write(edit(open(os.path.join(CSRC, "msda3d.hip")).read(), VARIANTS[name]))
    out = os.path.join(d, "msda.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", d, "-shared", p, "-o", out], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return out


def main():
    a = [v for v in sys.argv[1:] if not v.startswith("--")]
    n = int(a[0]) if a else 10
    if "--build-only" in sys.argv:
        for v in VARIANTS:
            print(v, build(v))
        return
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    vp, ci, cl = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
    levels = [(100, 100, 8), (50, 50, 4), (25, 25, 2)]
    Nq = sum(x * y * z for x, y, z in levels)
    H, P, L, E = 8, 4, 3, 192
    arr = (ctypes.c_int32 * 9)(*[v for s in levels for v in s])
    value = torch.randn(1, Nq, E, device=dev)
    off = torch.randn(1, Nq, H * L * P * 3, device=dev) * 0.5
    lg = torch.randn(1, Nq, H * L * P, device=dev)
    dout = torch.randn(1, Nq, E, device=dev)
    res = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(build(name))
        lib.occf_msda3d_bwd_workspace.restype = cl
        lib.occf_msda3d_bwd_workspace.argtypes = [vp, ci, ci, ci, ci]
        lib.occf_msda3d_bwd.argtypes = [vp] * 8 + [ci] * 7 + [cl] * 4 + [vp, cl, vp]
        need = lib.occf_msda3d_bwd_workspace(ctypes.cast(arr, vp), L, 1, H, E // H)
        ws = torch.empty(max(need, 1), device=dev)
        dv = torch.zeros(1, Nq, E, device=dev)
        doff = torch.empty_like(off)
        dlg = torch.empty_like(lg)
        st = torch.cuda.current_stream().cuda_stream

        def run():
            dv.zero_()
            rc = lib.occf_msda3d_bwd(value.data_ptr(), off.data_ptr(), lg.data_ptr(), dout.data_ptr(), dv.data_ptr(), doff.data_ptr(),
                                     dlg.data_ptr(), ctypes.cast(arr, vp), L, 1, Nq, H, E // H, P, 0, off.stride(1), lg.stride(1),
                                     doff.stride(1), dlg.stride(1), ws.data_ptr(), need, st)
            assert rc == 0, rc
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / n
    for name in VARIANTS:
        print(f"{name:12s} {res[name]:8.3f} ms per backward call ({n} calls)")


if __name__ == "__main__":
    main()
