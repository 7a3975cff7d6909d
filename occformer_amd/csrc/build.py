"""Build liboccformer_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m occformer_amd.csrc.build [--force]

The library is built IN-TREE (occformer_amd/liboccformer_hip.so) so that it travels to
the GPU box with the repository snapshot.  One translation unit per .hip file, linked
into a single shared object exporting the C ABI declared in include/occformer_hip.h.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "liboccformer_hip.so")
OBJ = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wno-unused-result", "-I", os.path.join(ROOT, "include"), "-I", HERE]


def abi_hash():
    """crc32 of the C header: compiled into the library (occf_abi_hash) and re-checked at load time, so a
    stale .so can never be called through a newer header's prototypes."""
    import zlib
    return zlib.crc32(open(os.path.join(ROOT, "include", "occformer_hip.h"), "rb").read()) & 0x7FFFFFFF


def is_current():
    stamp = os.path.join(OBJ, "digest.txt")
    return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == _digest()


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for f in sorted(os.listdir(HERE)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(HERE, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "occformer_hip.h"), "rb").read())
    # (flags without the checkout's absolute path: the digest identifies the SOURCES -- profiles/*_pmc_traffic.json is
    # stamped with it on one GPU box and checked on another, where the repository lives under a different path)
    h.update(" ".join(f.replace(ROOT, "<root>") for f in FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    stamp = os.path.join(OBJ, "digest.txt")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build {LIB}")

    def compile_one(src):
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, f"-DOCCF_ABI_HASH={abi_hash()}", "-c", os.path.join(HERE, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(compile_one, sources()))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB) from {len(objs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
