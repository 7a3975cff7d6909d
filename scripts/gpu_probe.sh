#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/probe13
mkdir -p $O
cd $R
echo "== dgrad classes"; python scripts/bwd_probe.py dgrad
echo "== tests"; timeout 900 python -m pytest tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "conv" 2>&1 | tail -3
echo "== train bench"; timeout 600 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/err.txt; echo rc=$?; tail -2 $O/err.txt
python -c "
import json
d=json.load(open('$O/bench_train.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB','forward_samples_per_s_same_run')})
for k,v in list(d['kernels'].items())[:12]: print(k, v['calls'], round(v['total_ms'],2))"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1 ; echo "rocprof rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1 ; head -70 $O/kernel_stats.txt
find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
