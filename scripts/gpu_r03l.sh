#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03l
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python scripts/graph_train_probe.py 8 2>&1 | grep -v "Warn\|amdgpu.ids" | tail -40 | tee $O/graph_probe.txt
timeout 300 python -m pytest tests/test_image_backbone.py tests/test_bwd_ops.py tests/test_train_ops.py -m gpu -q -x -k "dcnv2 or deform or topk or sample_without or linear_wgrad" -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
