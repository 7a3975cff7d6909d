"""Where does the G8 weight-gradient kernel's time go?  Ablated builds of csrc/wgrad.hip + wgrad_g8.h (textual edits of a
COPY, compiled on the spot into /tmp; wrong results by construction -- only durations mean something), timed with HIP
events at the metric's shapes, two-product fp16 mode (terms = 2):

    full        as shipped (main kernel + the pre-split passes + the slab reduction)
    no_dma      no LDS-DMA inside the stage loop (the prologue's stages are reused)
    no_barrier  no workgroup barrier per stage
    frag_const  every fragment read of a wave reads one address set
    mfma_only   no_dma + no_barrier + frag_const
    no_epi      the partial-sum stores of the epilogue skipped

    python scripts/wgrad_g8_ablation_probe.py [launches]"""
import ctypes
import os
import shutil
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "occformer_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-I", os.path.join(ROOT, "include"), "-DOCCF_ABI_HASH=0"]


def edit(src, what):
    def rep(a, b):
        nonlocal src
        assert src.count(a) == 1, (src.count(a), a)
        src = src.replace(a, b)
    if "no_dma" in what:
        rep("    if (s + ST - 1 < nsteps) issue(nxt);\n", "    if (s + ST - 1 < nsteps && p.taps == 77) issue(nxt);\n")
    if "no_barrier" in what:
        rep("    __syncthreads();\n    if (s + ST - 1 < nsteps", "    if (p.taps == 77) __syncthreads();\n    if (s + ST - 1 < nsteps")
    if "frag_const" in what:
        rep("      const int row = ks * 2 + lk;\n", "      const int row = lk;\n")
    if "no_epi" in what:
        rep("        o[(long)n * Kt + (long)tap * p.Cin + c] = F16 ? acc[i][j][r] * unscale : acc[i][j][r];",
            "        if (p.taps == 77 || acc[i][j][r] == 12345.678f) o[(long)n * Kt + (long)tap * p.Cin + c] = F16 ? acc[i][j][r] * unscale : acc[i][j][r];")
    return src


VARIANTS = {"full": (), "no_dma": ("no_dma",), "no_barrier": ("no_barrier",), "frag_const": ("frag_const",),
            "mfma_only": ("no_dma", "no_barrier", "frag_const"), "no_epi": ("no_epi",)}


def build(name):
    d = f"/tmp/wg8_{name}/a/b"                  # (the sources include "../../include/occformer_hip.h")
    shutil.rmtree(f"/tmp/wg8_{name}", ignore_errors=True)
    os.makedirs(d)
    shutil.copytree(os.path.join(ROOT, "include"), f"/tmp/wg8_{name}/include")
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            shutil.copy(os.path.join(CSRC, f), d)
    shutil.copy(os.path.join(CSRC, "wgrad.hip"), d)
    p = os.path.join(d, "wgrad_g8.h")
    src = edit(open(p).read(), VARIANTS[name])
    open(p, "w").write(src)
    out = os.path.join(d, "wg8.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", d, "-shared", os.path.join(d, "wgrad.hip"), "-o", out],
                       capture_output=True, text=True)
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
    res = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(build(name))
        lib.occf_conv3d_wgrad_workspace.restype = cl
        lib.occf_conv3d_wgrad_workspace.argtypes = [ci] * 14
        lib.occf_conv3d_wgrad.argtypes = [vp] * 5 + [cl] + [ci] * 14 + [cl] * 4 + [ci, vp]
        for C in (192, 128):
            X, Y, Z = 200, 200, 16
            x = torch.randn(1, X, Y, Z, C, device=dev)
            dy = torch.randn(1, X, Y, Z, C, device=dev) * 1e-4
            dw = torch.empty(C, 27 * C, device=dev)
            need = lib.occf_conv3d_wgrad_workspace(1, X, Y, Z, C, C, 3, 3, 3, 1, 1, 1, 1, 1)
            ws = torch.empty(need, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            s = x.stride()

            def run():
                rc = lib.occf_conv3d_wgrad(dy.data_ptr(), x.data_ptr(), dw.data_ptr(), None, ws.data_ptr(), need, 1, X, Y, Z, C, C,
                                           3, 3, 3, 1, 1, 1, 1, 1, s[0], s[1], s[2], s[3], 2, st)
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
            res[(name, C)] = e0.elapsed_time(e1) / n
    print(f"{'variant':12s} {'192':>9s} {'128':>9s}   (ms per call incl. pre-split passes and slab reduction, {n} calls)")
    for name in VARIANTS:
        print(f"{name:12s} {res[(name, 192)]:9.3f} {res[(name, 128)]:9.3f}")
    for C in (192, 128):
        fl = 2 * 27 * C * C * 640000
        print(f"C = {C}: matrix-pipe floor of two products at 2.5 PF/s: {2 * fl / 2.5e15 * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
