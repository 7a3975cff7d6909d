#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02p
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -2 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
