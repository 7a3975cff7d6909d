#!/bin/bash
# r03v: dense msda kernel with one chunk of loads in flight: parity + timing
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03v
mkdir -p $O
cd $R
export TMPDIR=/tmp
OCCF_MSDA_DENSE=1 timeout 600 python -m pytest tests/test_bwd_ops.py tests/test_full_size_gpu.py -m gpu -q -x -k "msda" -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
( for v in "OCCF_MSDA_DENSE=0" "OCCF_MSDA_DENSE=1"; do echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py msda 2>&1 | grep " ms"; done ) | tee $O/msda_probe.txt
cd /tmp
OCCF_MSDA_DENSE=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/bwd_probe.py msda > $O/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1; head -9 $O/kernel_stats.txt
find $O/prof -name "*.csv" -size +1M -delete 2>/dev/null
