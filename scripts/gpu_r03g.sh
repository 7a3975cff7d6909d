#!/bin/bash
# r03g: per-step kernel timeline of the training step (kernel trace of the plain loop)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
cd /tmp
export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/train_loop_probe.py 3 3 > $O/trace.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/step_timeline.py $O/trace 120 > $O/step_timeline.txt 2>&1; head -60 $O/step_timeline.txt
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); cut -d, -f8-12 "$f" 2>/dev/null | head -2
gzip -9 "$f"; ls -la $(dirname "$f")
du -sh $O
