#!/bin/bash
# r03d: is the training step host-bound?  GPU idle time from the kernel trace + host profile of the launch side
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
export TMPDIR=/tmp
PROBE_HOSTTIME=1 timeout 300 python scripts/train_loop_probe.py 6 3 2>&1 | grep -v Warn | tail -3 | tee $O/hosttime.txt
PROBE_CPROFILE=1 timeout 300 python scripts/train_loop_probe.py 3 3 > $O/cprofile.txt 2>&1; grep "host time" $O/cprofile.txt
cd /tmp
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/train_loop_probe.py 4 3 > $O/trace.log 2>&1; echo "rocprof rc=$?"; tail -1 $O/trace.log
cd $R
python scripts/gap_analysis.py $O/trace 0.5 > $O/gap_analysis.txt 2>&1; head -45 $O/gap_analysis.txt
find $O/trace -name "*kernel_trace.csv" -delete 2>/dev/null
du -sh $O
