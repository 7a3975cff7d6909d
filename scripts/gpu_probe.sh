#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_train_ops.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
