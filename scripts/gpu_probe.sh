#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02n
mkdir -p $O
cd $R
for wl in nusc_r101 kitti_effb7_256lit; do
  echo "== train bench $wl"; timeout 600 python bench.py --mode train --workload $wl --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train_$wl.json 2> $O/err_$wl.txt; echo rc=$?; tail -2 $O/err_$wl.txt
  python -c "
import json
d=json.load(open('$O/bench_train_$wl.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB')})"
done
