#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe35
mkdir -p $O
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_train_step.py tests/test_training.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
echo "== train bench"; timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/err.txt; echo rc=$?; tail -2 $O/err.txt
python -c "
import json
d=json.load(open('$O/bench_train.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB','forward_samples_per_s_same_run')})
for k,v in list(d['kernels'].items())[:8]: print(k, v['calls'], round(v['total_ms'],2))"
