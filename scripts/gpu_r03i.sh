#!/bin/bash
# r03i: G8 weight gradient: stages / rows per stage sweep, full-size parity, PMC of the 192 -> 192 launch
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
export TMPDIR=/tmp
( for v in "OCCF_WG8_ST=2" "OCCF_WG8_ST=3" "OCCF_WG8_ST=4" "OCCF_WG8_ST=3 OCCF_WG8_KS=1" "OCCF_WG8_ST=4 OCCF_WG8_KS=1" "OCCF_WG8_ST=2 OCCF_WG8_KS=1" "OCCF_WG8_ST=3 OCCF_WG8_KS=2" "OCCF_WG8_ST=3 OCCF_WG8_S=112" "OCCF_WG8_ST=3 OCCF_WG8_S=56"; do
  echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py wgrad 2>&1 | grep "conv3d_wgrad"
done ) | tee $O/wgrad_probe.txt
for v in "OCCF_WG8_ST=2" "OCCF_WG8_ST=3" "OCCF_WG8_ST=4 OCCF_WG8_KS=1"; do
  echo "-- $v"; env $v timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_bwd_ops.py -m gpu -q -x -k "wgrad" -p no:cacheprovider 2>&1 | tail -2
done | tee $O/pytest.txt
OCCF_WG8_ST=3 bash scripts/pmc_probe.sh r03i/wgrad_g8_192_pmc wgrad_g8 python scripts/bwd_probe.py wgrad192 > $O/pmc.log 2>&1; tail -45 $O/pmc.log
du -sh $O
