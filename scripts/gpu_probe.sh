#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_train_step.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -v "amdgpu.ids" | tail -60
