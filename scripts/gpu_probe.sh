#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "depthnet or inproj or joint" 2>&1 | tail -3
