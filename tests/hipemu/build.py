"""TEST-ONLY: compile the kernel sources of occformer_amd/csrc for the HOST with the
fiber emulator (tests/hipemu/hipemu.h) into tests/hipemu/build/libocc_emu.so, so the
CPU test-suite can exercise the kernels' index math against the oracle without a GPU.
The product never loads this library."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "occformer_amd", "csrc")
OUT = os.path.join(HERE, "build")
LIB = os.path.join(OUT, "libocc_emu.so")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
CXX = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-O2", "-std=c++17", "-fPIC", "-DOCCF_EMU", "-ffp-contract=off", "-x", "c++",
         "-Wno-unused-value", "-Wno-unknown-attributes",
         "-I", HERE, "-I", CSRC, "-I", os.path.join(ROOT, "include")]


def available():
    return os.path.exists(CXX)


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(CSRC, f), "rb").read())
    for f in ("hipemu.h", "hipemu.cpp"):
        h.update(open(os.path.join(HERE, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "occformer_hip.h"), "rb").read())
    stamp = os.path.join(OUT, "digest.txt")
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return LIB
    # one builder at a time: the pytest-xdist workers of a CPU run all arrive here with a stale library (they used to
    # compile into the same object files at once, and one of them linked a half-written one)
    import fcntl
    with open(os.path.join(OUT, "build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
            return LIB
        return _build_locked(srcs, h.hexdigest(), stamp)


def _build_locked(srcs, digest, stamp):
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, s.replace(".hip", ".o"))
        objs.append(o)
        from occformer_amd.csrc.build import abi_hash
        procs.append((s, subprocess.Popen([CXX, *FLAGS, f"-DOCCF_ABI_HASH={abi_hash()}", "-c",
                                           os.path.join(CSRC, s), "-o", o],
                                          stderr=subprocess.PIPE, text=True)))
    for s, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"emu compile failed for {s}:\n{err}")
    o = os.path.join(OUT, "hipemu.o")
    subprocess.check_call([CXX, "-O2", "-std=c++17", "-fPIC", "-c", os.path.join(HERE, "hipemu.cpp"), "-o", o])
    subprocess.check_call([CXX, "-shared", "-o", LIB, *objs, o])
    open(stamp, "w").write(digest)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
