#!/bin/bash
# r03f: prepared weights (one launch per step) -- parity tests, step time, per-step kernel timeline from the trace
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_multistep.py tests/test_train_step.py tests/test_boundary.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_a.log
PROBE_HOSTTIME=1 timeout 300 python scripts/train_loop_probe.py 8 3 2>&1 | grep -v Warn | tail -2 | tee $O/hosttime.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err; echo "rc=$?"; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"][:40], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"], d.get("check"))
except Exception as e: print("no json", e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/train_loop_probe.py 3 3 > $O/trace.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/step_timeline.py $O/trace 90 > $O/step_timeline.txt 2>&1; head -45 $O/step_timeline.txt
find $O/trace -name "*.csv" -size +1M -delete 2>/dev/null
du -sh $O
