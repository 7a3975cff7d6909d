#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe16
mkdir -p $O
cd $R
echo "== pytest -m gpu (all)"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo rc=$?; tail -4 $O/pytest_gpu.log
echo "== train bench"; timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/err.txt; echo rc=$?; tail -2 $O/err.txt
python -c "
import json
d=json.load(open('$O/bench_train.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB','forward_samples_per_s_same_run')})
for k,v in list(d['kernels'].items())[:14]: print(k, v['calls'], round(v['total_ms'],2))"
echo "== forward bench --check"; timeout 900 python bench.py --mode forward --steps 10 --warmup 3 --no-cpu-baseline --check > $O/bench_fwd.json 2> $O/err_fwd.txt; echo rc=$?; tail -2 $O/err_fwd.txt
python -c "
import json
d=json.load(open('$O/bench_fwd.json')); print({k:d.get(k) for k in ('value','ms_per_step','check')})"
