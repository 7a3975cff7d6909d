#!/bin/bash
# r03y: lazy mask logits in the training graph with the set-batched loss (matched rows of all sets from one contraction)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03y
mkdir -p $O
cd $R
export TMPDIR=/tmp
OCCF_LAZY_LOGITS=1 timeout 900 python -m pytest tests/test_training.py tests/test_train_step.py tests/test_train_multistep.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2 | tee $O/pytest_lazy.txt
OCCF_LAZY_LOGITS=1 timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -x -k "training_step and nusc_r50_200" -p no:cacheprovider 2>&1 | tail -2 | tee -a $O/pytest_lazy.txt
for v in 0 1; do
OCCF_LAZY_LOGITS=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_lazy$v.json 2> $O/bench_lazy$v.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$O/bench_lazy$v.json")); print("lazy=$v", {k:d[k] for k in ("value","ms_per_step","peak_memory_GiB")})
except Exception as e: print("no json", e)
PY
done
