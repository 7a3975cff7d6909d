#!/bin/bash
# round 4, visit a: halo-conv schedule probe + the full GPU suite under OCCF_TEST_POISON=1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
free -g | head -2 > $O/r04a_host.txt; nproc >> $O/r04a_host.txt
for s in 0 1; do OCCF_HALO_SCHED=$s timeout 300 python scripts/conv_probe.py; done > $O/r04a_conv_probe.txt 2>&1
cat $O/r04a_conv_probe.txt
OCCF_TEST_POISON=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 > $O/r04a_pytest_gpu_poison.log
tail -8 $O/r04a_pytest_gpu_poison.log
