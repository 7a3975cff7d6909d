#!/bin/bash
# r03r: GroupNorm statistics from the convolution epilogue in the training graph: parity + step time
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03r
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_train_step.py tests/test_train_multistep.py tests/test_bwd_ops.py tests/test_training.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_a.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel"][:44], round(d["roofline"]["avg_kernel_ms"],3), round(d["roofline"]["frac"],4))
    for k,v in list(d["kernels"].items())[:30]: print("  ",k,v)
except Exception as e: print("no json", e)
PY
