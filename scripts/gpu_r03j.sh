#!/bin/bash
# r03j: G8 weight gradient with y-strip slabs: strip width / plane segments sweep, parity, PMC (FETCH_SIZE)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
export TMPDIR=/tmp
( for v in "X=default" "OCCF_WG8_W=200 OCCF_WG8_SEGS=14" "OCCF_WG8_W=8 OCCF_WG8_SEGS=3" "OCCF_WG8_W=10 OCCF_WG8_SEGS=3" "OCCF_WG8_W=13 OCCF_WG8_SEGS=5" "OCCF_WG8_W=17 OCCF_WG8_SEGS=3" "OCCF_WG8_W=25 OCCF_WG8_SEGS=7" "OCCF_WG8_W=25 OCCF_WG8_SEGS=2" "OCCF_WG8_W=40 OCCF_WG8_SEGS=4" "OCCF_WG8_W=10 OCCF_WG8_SEGS=1"; do
  echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py wgrad 2>&1 | grep "conv3d_wgrad"
done ) | tee $O/wgrad_probe.txt
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_bwd_ops.py -m gpu -q -x -k "wgrad" -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest.txt
bash scripts/pmc_probe.sh r03j/wgrad_g8_192_pmc wgrad_g8 python scripts/bwd_probe.py wgrad192 > $O/pmc.log 2>&1; grep -E "FETCH|WRITE_SIZE|GRBM_GUI|MFMA|VALU |SALU|WAIT_INST_ANY|WAVE_CYCLES" $O/pmc.log
du -sh $O
