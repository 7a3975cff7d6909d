#!/bin/bash
# GPU-box visit for the training step: backward-kernel parity tests + train-step gradient test on the real library,
# then the fwd+bwd bench at full size with the per-kernel census.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-train}
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest bwd ops + train step (gpu)"
timeout 900 python -m pytest tests/test_bwd_ops.py tests/test_train_step.py -m gpu -q -s -p no:cacheprovider > $O/pytest_bwd.log 2>&1 ; echo "pytest rc=$?" ; tail -25 $O/pytest_bwd.log
echo "== bench train"
timeout 900 python bench.py --mode train --steps ${STEPS:-3} --warmup 2 ${CPUBASE:---no-cpu-baseline} --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err ; echo "bench rc=$?" ; tail -15 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json"))
    print({k:d[k] for k in ("value","ms_per_step","peak_memory_GiB") if k in d}, "forward", (d.get("forward") or {}).get("value"))
    print(d["roofline"])
    for k,v in list(d["kernels"].items())[:40]: print(k, v)
    print(d.get("losses"))
except Exception as e: print("no json", e)
PY
head -40 $O/shapes_train.txt
if [ "${FWD:-1}" = "1" ]; then
echo "== bench forward"
timeout 600 python bench.py --mode forward --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_fwd.json 2> $O/bench_fwd.err ; echo "bench rc=$?" ; tail -3 $O/bench_fwd.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_fwd.json")); print({k:d[k] for k in ("value","ms_per_step")})
except Exception as e: print("no json", e)
PY
fi
if [ "${PROF:-0}" = "1" ]; then
  echo "== rocprof kernel trace (train)"
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1 ; echo "rocprof rc=$?" ; tail -2 $O/prof.log
  cd $R
  python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1 ; head -60 $O/kernel_stats.txt
  find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
fi
du -sh $O
