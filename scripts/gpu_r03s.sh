#!/bin/bash
# r03s: which change moved the bench's gradient check (grad_rel_l2 6.4e-4 at r03a -> 1.25e-3 at r03q)?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03s
mkdir -p $O
cd $R
export TMPDIR=/tmp
i=0
for v in "X=default" "OCCF_WG_G8=0" "OCCF_BATCHED_LOSS=0" "OCCF_TRAIN_GN_EPILOGUE=0" "OCCF_TK_SMALL=0"; do
  i=$((i+1))
  env $v timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_$i.json 2> $O/bench_$i.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$i.json")); c=d["check"]; print("$v", round(d["ms_per_step"],1), c["max_rel_loss_diff"], c["grad_rel_l2"], c["per_parameter_rel_l2_quantiles"])
except Exception as e: print("$v", "no json", e)
PY
done
