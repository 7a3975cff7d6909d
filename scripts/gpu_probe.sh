#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/train_wl
mkdir -p $O
cd $R
for wl in kitti_effb7_128 nusc_r50_ref128; do
echo "== train bench $wl"; timeout 600 python bench.py --mode train --workload $wl --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train_$wl.json 2> $O/err_$wl.txt; echo rc=$?; tail -3 $O/err_$wl.txt
python -c "
import json
d=json.load(open('$O/bench_train_$wl.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB','forward_samples_per_s_same_run')}); print(d['losses'])"
done
