#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_boundary.py tests/test_train_step.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_a.log
PROBE_HOSTTIME=1 timeout 300 python scripts/train_loop_probe.py 8 3 2>&1 | grep -v Warn | tail -2 | tee $O/hosttime.txt
PROBE_CPROFILE=1 timeout 300 python scripts/train_loop_probe.py 3 3 > $O/cprofile.txt 2>&1; grep "host time" $O/cprofile.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; echo "rc=$?"; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"][:40], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"])
except Exception as e: print("no json", e)
PY
