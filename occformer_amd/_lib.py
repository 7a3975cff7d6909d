"""Loader for liboccformer_hip.so (the gfx950 C-ABI library, include/occformer_hip.h).

There is deliberately NO fallback: if the library cannot be found/built or a symbol is
missing, importing the ops raises.  ``torch`` must be imported first so that the HIP
runtime already in the process (torch's libamdhip64, SONAME libamdhip64.so.7) is the one
the library binds to -- one runtime, shared streams and allocations.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (must precede the dlopen, see module docstring)

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.path.join(_PKG, "liboccformer_hip.so")
HEADER = os.path.join(_ROOT, "include", "occformer_hip.h")

_CTYPES = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
           "double": ctypes.c_double, "int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64,
           "unsigned": ctypes.c_uint, "uint16_t": ctypes.c_uint16, "uint8_t": ctypes.c_uint8}
_CTYPES["long"] = ctypes.c_long


def parse_header(path=HEADER):
    """Return {name: [(ctype, argname), ...]} for every prototype in the C header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|long)\s+(occf_\w+)\s*\(([^)]*)\)\s*;", src):
        args = [_CTYPES[m.group(1)]]          # slot 0 = return type
        for a in m.group(3).split(","):
            a = " ".join(a.split())
            if a == "void":
                continue
            if "*" in a:
                args.append((ctypes.c_void_p, a.split("*")[-1].strip()))
            else:
                ty, name = a.rsplit(" ", 1)
                args.append((_CTYPES[ty.replace("const ", "").strip()], name))
        protos[m.group(2)] = args
    return protos


def bind(lib_path):
    """dlopen ``lib_path`` and attach argtypes from the header; raises on any missing symbol."""
    lib = ctypes.CDLL(lib_path)
    protos = parse_header()
    for name, args in protos.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = [t for t, _ in args[1:]]
        fn.restype = args[0]
    lib._occf_protos = protos
    from .csrc.build import abi_hash
    if lib.occf_abi_hash() != abi_hash():
        raise RuntimeError(f"{lib_path} was built against a different include/occformer_hip.h "
                           f"(abi {lib.occf_abi_hash()} != header {abi_hash()}); rebuild it: "
                           "python -m occformer_amd.csrc.build")
    return lib


_lib = None


def get():
    global _lib
    if _lib is None:
        from .csrc import build as _b
        if not _b.is_current():
            _b.build(verbose=False)  # sources changed or no library yet; raises if hipcc is unavailable
        _lib = bind(LIB_PATH)
    return _lib
