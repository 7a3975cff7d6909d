// ABI stamp of the shared library (see build.py::abi_hash, occformer_amd/_lib.py::bind).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#ifndef OCCF_ABI_HASH
#error "build with -DOCCF_ABI_HASH=<crc32 of include/occformer_hip.h> (occformer_amd/csrc/build.py)"
#endif

extern "C" int occf_abi_hash(void) { return OCCF_ABI_HASH; }
