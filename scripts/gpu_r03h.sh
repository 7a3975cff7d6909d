#!/bin/bash
# r03h: G8 weight-gradient kernel (LDS-DMA staging from pre-transposed bf16 operands): parity + timing vs the old kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bwd_ops.py -m gpu -q -x -k "wgrad" -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_a.log
( for v in "OCCF_WG_G8=0" "OCCF_WG_G8=1" "OCCF_WG_G8=1 OCCF_WG8_S=8" "OCCF_WG_G8=1 OCCF_WG8_S=16" "OCCF_WG_G8=1 OCCF_WG8_S=32" "OCCF_WG_G8=1 OCCF_WG8_S=56" "OCCF_WG_G8=1 OCCF_WG8_S=112"; do
  echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py wgrad 2>&1 | grep " ms"
done ) | tee $O/wgrad_probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/bwd_probe.py wgrad192 > $O/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
find $O/prof -name "*.csv" -size +1M -delete 2>/dev/null
du -sh $O
