#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe39
mkdir -p $O
cd $R
echo "== tests"; timeout 900 python -m pytest tests/test_train_step.py tests/test_training.py -m gpu -q -p no:cacheprovider -s 2>&1 | grep -E "passed|failed|quantiles|whole gradient" | tail -4
for v in 1 0; do
echo "== train bench OCCF_LAZY_LOGITS=$v"; OCCF_LAZY_LOGITS=$v timeout 600 python bench.py --mode train --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_$v.json 2> $O/err_$v.txt; echo rc=$?
python -c "
import json
d=json.load(open('$O/bench_train_$v.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB')})"
done
