#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python scripts/graph_probe.py 2>&1 | grep -v amdgpu.ids | tail -20
