#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe38
mkdir -p $O
cd $R
timeout 600 python bench.py --mode forward --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_fwd.json 2> $O/err.txt; echo rc=$?
python -c "
import json
d=json.load(open('$O/bench_fwd.json')); print({k:d.get(k) for k in ('value','ms_per_step')}); print(d['stages_ms'])
for k,v in list(d['kernels'].items())[:6]: print(k, v['calls'], round(v['total_ms'],2))"
timeout 600 python -m pytest tests/test_attn_ops.py tests/test_switches.py -m gpu -q -p no:cacheprovider -k "swin or switch" 2>&1 | tail -1
