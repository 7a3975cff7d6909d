#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
echo "== xcd placement"; python scripts/bwd_probe.py wgrad
echo "== round robin"; OCCF_WG_XCD=0 python scripts/bwd_probe.py wgrad
echo "== tests"; timeout 900 python -m pytest tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "wgrad" 2>&1 | tail -2
