#!/bin/bash
# r03k: full GPU suite on the G8 wgrad / fork nodes / MaskRows / small-V top-k / small-M wgrad tree, bench, step timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
( timeout 300 python scripts/bwd_probe.py wgrad 2>&1 | grep " ms" ) | tee $O/wgrad_probe.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"][:40], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"], d.get("check"))
    for k,v in list(d["kernels"].items())[:12]: print("  ",k,v)
except Exception as e: print("no json", e)
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python $R/scripts/train_loop_probe.py 3 3 > $O/trace.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/step_timeline.py $O/trace 120 > $O/step_timeline.txt 2>&1; head -70 $O/step_timeline.txt
f=$(find $O/trace -name "*kernel_trace.csv" | head -1); gzip -9 "$f"
du -sh $O
