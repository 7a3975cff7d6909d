"""Where does the Winograd convolution kernel's time go?  Ablated builds of csrc/conv_wino.hip (textual edits of a COPY
of the source, compiled on the spot into /tmp -- the product source and library are untouched; results are wrong by
construction, only the durations mean something), each timed with HIP events at the metric's shapes:

    full        the kernel as shipped
    b_l1        every weight-fragment load of a wave reads ONE address set (L1 hits: no L2 traffic for the weights)
    b_none      no weight-fragment loads inside the tap loop (the prologue's fragments are reused)
    stage_none  no halo staging inside the tap loop (no global loads, no transform / split, no LDS stores)
    a_const     every A-fragment LDS read of a wave reads one address set
    epi_none    the epilogue's stores (and residual) are skipped
    b_stage     b_none + stage_none
    mfma_only   b_none + stage_none + a_const

    python scripts/wino_ablation_probe.py [launches]"""
import ctypes
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "occformer_amd", "csrc", "conv_wino.hip")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result",
         "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "occformer_amd", "csrc"), "-DOCCF_ABI_HASH=0"]


def edit(src, what):
    def rep(a, b, count=1):
        nonlocal src
        assert src.count(a) >= 1, a
        src = src.replace(a, b) if count == 0 else src.replace(a, b, count)
    if "b_l1" in what:
        rep("const long o = ((long)(((c * 4 + wts) * 9 + tap) * 2 + s) * ngrp + jg0) * 512;",
            "const long o = ((long)(wts + 0 * (c + tap + s)) * ngrp + jg0) * 512;")
    if "b_none" in what:
        rep("load_next_b(fh[BLOAD], fl[BLOAD]);                                                                \\",
            "                                                                                                  \\")
        rep("for (int i = 0; i < BD - 1; ++i) load_next_b(fh[i], fl[i]);", "for (int i = 0; i < BD; ++i) load_next_b(fh[i], fl[i]);")
    if "stage_none" in what:
        rep("if (more) load_halo((cc + 1) * 32, 0);", ";")
        rep("if (tap == 3 && more) {", "if (false) {")
        rep("if (tap == 6 && more) store_halo(bufsel ^ 1, 1);", ";")
    if "a_const" in what:
        rep("const unsigned char* ap = H + bufsel * buf_sz + a_base + toff * HROW + s * 32;",
            "const unsigned char* ap = H + a_base + 0 * (bufsel + toff + s);")
    if "epi_none" in what:
        rep("const bool v_ok = x < p.X && y < p.Y;", "const bool v_ok = x < p.X && y < p.Y && p.act == 77;")
        rep("if (p.residual) {\n#pragma unroll\n      for (int q = 0; q < 8; ++q) {\n        const int pos = i * 32 + cw_pos(16 * rh + 2 * q + lk);",
            "if (p.residual && p.act == 77) {\n#pragma unroll\n      for (int q = 0; q < 8; ++q) {\n        const int pos = i * 32 + cw_pos(16 * rh + 2 * q + lk);")
    return src


VARIANTS = {"full": (), "b_l1": ("b_l1",), "b_none": ("b_none",), "stage_none": ("stage_none",), "a_const": ("a_const",),
            "epi_none": ("epi_none",), "b_stage": ("b_none", "stage_none"), "mfma_only": ("b_none", "stage_none", "a_const")}


def build(name):
    out = f"/tmp/cw_{name}.so"
    cp = f"/tmp/cw_{name}.hip"
    open(cp, "w").write(edit(open(SRC).read(), VARIANTS[name]))
    r = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-shared", cp, "-o", out], capture_output=True, text=True)
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
    vp = ctypes.c_void_p
    res = {}
    for name in VARIANTS:
        lib = ctypes.CDLL(build(name))
        lib.occf_conv3x3x3_wino_fwd.argtypes = [vp] * 6 + [ctypes.c_int] * 6 + [ctypes.c_long] * 4 + [ctypes.c_int, vp, vp, vp]
        lib.occf_conv3x3x3_wino_pack.argtypes = [vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
        lib.occf_absmax_flat.argtypes = [vp, ctypes.c_long, vp, vp]
        lib.occf_absmax_slot_words.restype = ctypes.c_long
        for C in (192, 128):
            for f16 in (0, 1):
                X, Y, Z = 200, 200, 16
                x = torch.randn(1, X, Y, Z, C, device=dev) * (1e-4 if f16 else 1.0)
                w = torch.randn(C, 27 * C, device=dev) * 0.02
                fh = torch.empty(36 * C * C, dtype=torch.int16, device=dev)
                fl = torch.empty_like(fh)
                out = torch.empty_like(x)
                slot = torch.zeros(lib.occf_absmax_slot_words(), dtype=torch.int32, device=dev)
                st = torch.cuda.current_stream().cuda_stream
                assert lib.occf_conv3x3x3_wino_pack(w.data_ptr(), fh.data_ptr(), fl.data_ptr(), C, C, f16, st) == 0
                if f16:
                    assert lib.occf_absmax_flat(x.data_ptr(), x.numel(), slot.data_ptr(), st) == 0
                s = x.stride()

                def run():
                    rc = lib.occf_conv3x3x3_wino_fwd(x.data_ptr(), fh.data_ptr(), fl.data_ptr(), None, None, out.data_ptr(), 1, X, Y, Z,
                                                     C, C, s[0], s[1], s[2], s[3], 0, None, slot.data_ptr() if f16 else None, st)
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
                res[(name, C, f16)] = e0.elapsed_time(e1) / n
    print(f"{'variant':12s} {'192 bf16x3':>11s} {'192 f16x2':>11s} {'128 bf16x3':>11s} {'128 f16x2':>11s}   (ms per launch, {n} launches)")
    for name in VARIANTS:
        print(f"{name:12s} " + " ".join(f"{res[(name, C, f)]:11.3f}" for C in (192, 128) for f in (0, 1)))
    # pure-MFMA floor of the shipped arithmetic at the boost clock: 18 / 27 of 2 * 27 * C * C * voxels, x3 / x2 products
    for C in (192, 128):
        fl = 2 * 18 * C * C * 640000
        print(f"C = {C}: matrix-pipe floor at 2.5 PF/s: bf16x3 {3 * fl / 2.5e15 * 1e3:.3f} ms, f16x2 {2 * fl / 2.5e15 * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
