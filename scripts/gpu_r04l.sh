#!/bin/bash
# round 4, visit l: halo conv 128 -> 128 / 256 -> 256 with the 2 x 4 wave layout (OCCF_HALO_WN4)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
for v in 0 1; do OCCF_HALO_WN4=$v timeout 200 python scripts/conv_probe.py; done 2>/dev/null | tee $O/r04l_conv_probe_wn4.txt
