#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== uc mfma / lds-valu"; python scripts/uc_probe.py; OCCF_CLASSIFY_MFMA=0 python scripts/uc_probe.py
echo "== tests"; timeout 600 python -m pytest tests/test_attn_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "upsample" 2>&1 | tail -2
