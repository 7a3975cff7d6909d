#!/bin/bash
# r03n: merged reductions, capped compaction, DCNv2 training test on the GPU; bench + timeline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_image_backbone.py tests/test_bwd_ops.py tests/test_train_ops.py tests/test_train_multistep.py tests/test_train_step.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_a.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"][:40], d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"])
    for k,v in list(d["kernels"].items())[:24]: print("  ",k,v)
except Exception as e: print("no json", e)
PY
