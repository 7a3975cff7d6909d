#!/bin/bash
# r03p: nuScenes loss over all prediction sets at once: parity tests, bench, phase breakdown of the step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_training.py tests/test_train_step.py tests/test_train_multistep.py tests/test_train_ops.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_a.log
for v in 1; do
OCCF_BATCHED_LOSS=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train_b$v.json 2> $O/bench_train_b$v.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train_b$v.json")); print("batched=$v", {k:d[k] for k in ("value","ms_per_step")}, d.get("check"))
except Exception as e: print("no json", e)
PY
done
PROBE_HOSTTIME=1 timeout 300 python scripts/train_loop_probe.py 8 3 2>&1 | grep -v Warn | tail -2 | tee $O/hosttime.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/train_loop_probe.py 3 3 > $O/trace.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/step_timeline.py $O/trace 40 > $O/step_timeline.txt 2>&1; head -14 $O/step_timeline.txt
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); gzip -9 "$f"
