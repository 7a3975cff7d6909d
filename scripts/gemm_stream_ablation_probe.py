"""Where does the weight-resident streaming linear's time go?  Ablated builds of csrc/gemm_bf16.hip + gemm_stream.h
(textual edits of a COPY, compiled on the spot into /tmp; wrong results by construction -- only durations mean
something), timed with HIP events on the training step's streaming shapes:

    full         as shipped
    no_stage     the weight block is not staged into LDS (no LDS-DMA prologue)
    no_rows      no activation-row loads inside the tile loop (the first tile's rows are reused)
    no_store     the epilogue's stores skipped
    lds_const    every weight-fragment read of a wave reads one address set
    no_split     the (hi, lo) split of the rows replaced by two bit casts
    mfma_only    all of the above

    python scripts/gemm_stream_ablation_probe.py [launches]"""
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
    def rep(a, b):
        nonlocal src
        assert src.count(a) == 1, (src.count(a), a)
        src = src.replace(a, b)
    if "no_stage" in what:
        rep("      for (int pc = wave; pc < pieces; pc += GS_NW) {", "      for (int pc = wave; pc < pieces && p.act == 77; pc += GS_NW) {")
    if "no_rows" in what:
        rep("    if (PFK < KS) load_rows(wt, k_pf(), k_end());", "    if (PFK < KS && wt < GS_NW * 100000L) { if (wt == (long)q * GS_NW + wave) load_rows(wt, k_pf(), k_end()); }")
        rep("      load_rows(wt + wt_step < n_wtiles ? wt + wt_step : wt, k_begin(), k_pf());", "      if (p.act == 77) load_rows(wt, k_begin(), k_pf());")
    if "no_store" in what:
        src = src.replace("if (tok_ok) *(float4*)(crow + j * 32 + g * 8) = v;", "if (tok_ok && (p.act == 77 || v.x == 12345.678f)) *(float4*)(crow + j * 32 + g * 8) = v;")
    if "lds_const" in what:
        rep("        const bf16x8 wh = *(const bf16x8*)(Wt + ks * 1024);", "        const bf16x8 wh = *(const bf16x8*)(Wt + (ks & 1) * 1024);")
        rep("          const bf16x8 wl = *(const bf16x8*)(Wt + IMG + ks * 1024);", "          const bf16x8 wl = *(const bf16x8*)(Wt + IMG + (ks & 1) * 1024);")
    if "no_split" in what:
        rep("      occf_bf16_split2(ra[ks].x, ra[ks].y, h[0], l[0]);\n      occf_bf16_split2(ra[ks].z, ra[ks].w, h[1], l[1]);\n"
            "      occf_bf16_split2(rb[ks].x, rb[ks].y, h[2], l[2]);\n      occf_bf16_split2(rb[ks].z, rb[ks].w, h[3], l[3]);",
            "      h[0] = occf_f2u(ra[ks].x); l[0] = occf_f2u(ra[ks].y); h[1] = occf_f2u(ra[ks].z); l[1] = occf_f2u(ra[ks].w);\n"
            "      h[2] = occf_f2u(rb[ks].x); l[2] = occf_f2u(rb[ks].y); h[3] = occf_f2u(rb[ks].z); l[3] = occf_f2u(rb[ks].w);")
    return src


ALL = ("no_stage", "no_rows", "no_store", "lds_const", "no_split")
VARIANTS = {"full": (), **{v: (v,) for v in ALL}, "mfma_only": ALL}
SHAPES = [(91250, 192, 192), (91250, 192, 768), (680000, 128, 128), (80000, 192, 192), (90000, 256, 256)]


def build(name):
    d = f"/tmp/gs_{name}/a/b"
    shutil.rmtree(f"/tmp/gs_{name}", ignore_errors=True)
    os.makedirs(d)
    shutil.copytree(os.path.join(ROOT, "include"), f"/tmp/gs_{name}/include")
    for f in os.listdir(CSRC):
        if f.endswith(".h"):
            shutil.copy(os.path.join(CSRC, f), d)
    shutil.copy(os.path.join(CSRC, "gemm_bf16.hip"), d)
    p = os.path.join(d, "gemm_stream.h")
    src = edit(open(p).read(), VARIANTS[name])
    open(p, "w").write(src)
    out = os.path.join(d, "gs.so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I", d, "-shared", os.path.join(d, "gemm_bf16.hip"), "-o", out],
                       capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    return out


def main():
    a = [v for v in sys.argv[1:] if not v.startswith("--")]
    n = int(a[0]) if a else 20
    only = [v[7:].split(",") for v in sys.argv[1:] if v.startswith("--only=")]
    if only:
        for k in list(VARIANTS):
            if k not in only[0]:
                del VARIANTS[k]
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
        lib.occf_linear_stream_fwd.argtypes = [vp] * 8 + [cl, ci, ci, cl, cl, cl, ci, ci, cl, ci, vp]
        lib.occf_split_bf16.argtypes = [vp, vp, vp, cl, vp]
        for (M, K, N) in SHAPES:
            x = torch.randn(M, K, device=dev)
            w = torch.randn(N, K, device=dev) * 0.05
            b = torch.randn(N, device=dev)
            hi = torch.empty(N * K, dtype=torch.int16, device=dev)
            lo = torch.empty_like(hi)
            out = torch.empty(M, N, device=dev)
            st = torch.cuda.current_stream().cuda_stream
            assert lib.occf_split_bf16(w.data_ptr(), hi.data_ptr(), lo.data_ptr(), N * K, st) == 0

            def run():
                rc = lib.occf_linear_stream_fwd(x.data_ptr(), hi.data_ptr(), lo.data_ptr(), b.data_ptr(), None, out.data_ptr(), None,
                                                None, M, N, K, K, N, 0, 0, 3, 1, 1, st)
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
            res[(name, M, K, N)] = e0.elapsed_time(e1) / n * 1e3
    print(f"{'variant':10s} " + " ".join(f"{f'[{M},{K}]x{N}':>18s}" for (M, K, N) in SHAPES) + f"   (us per launch, {n} launches)")
    for name in VARIANTS:
        print(f"{name:10s} " + " ".join(f"{res[(name, M, K, N)]:18.1f}" for (M, K, N) in SHAPES))
    print(f"{'HBM floor':10s} " + " ".join(f"{4.0 * M * (K + N) / 5.0e6:18.1f}" for (M, K, N) in SHAPES) + "   (rows in + out at 5 TB/s)")
    print(f"{'MFMA floor':10s} " + " ".join(f"{3 * 2.0 * M * K * N / 1.45e9:18.1f}" for (M, K, N) in SHAPES) + "   (three products at 1.45 PF/s)")


if __name__ == "__main__":
    main()
