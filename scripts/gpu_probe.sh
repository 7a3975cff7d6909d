#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== frag + A prefetch + 80B rows + lane permutation"; python scripts/bwd_probe.py conv
echo "== tests"; timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "halo or conv" 2>&1 | tail -2
