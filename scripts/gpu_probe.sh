#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02l
mkdir -p $O
cd $R
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -2 $O/pytest_gpu.log
echo "== smoke"; timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -2 $O/smoke.log
echo "== default bench"
SECONDS=0
timeout 1200 python bench.py --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err ; echo "bench rc=$? wall ${SECONDS}s"
python - <<PY
import json
d=json.load(open("$O/bench_train.json"))
print({k:d[k] for k in ("value","ms_per_step","peak_memory_GiB","forward_samples_per_s_same_run")}); print(d["roofline"]); print(d.get("cpu_baseline"))
for k,v in list(d["kernels"].items())[:12]: print(k, v["calls"], round(v["total_ms"],2))
PY
