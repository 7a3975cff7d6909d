#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe26
mkdir -p $O
cd $R
timeout 600 python scripts/aten_profile.py > $O/aten_profile.txt 2>&1; echo rc=$?
grep -v "amdgpu.ids\|Warning\|warn" $O/aten_profile.txt | head -80
