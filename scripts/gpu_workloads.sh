#!/bin/bash
# One GPU-box visit: full GPU parity suite (incl. the five full-size end-to-end workload tests) and one bench
# line per BASELINE workload with --check (max abs err vs the CPU oracle).  Logs under gpurun_out/<tag>/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-wl}
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; grep -E "^\[|passed|failed|error" $O/pytest_gpu.log | tail -30
for wl in ${WORKLOADS:-nusc_r50_200 nusc_r50_ref128 kitti_effb7_128 kitti_effb7_256lit nusc_r101}; do
  echo "== bench $wl"
  timeout 900 python bench.py --workload $wl --steps ${STEPS:-5} --warmup 2 --check > $O/bench_$wl.json 2> $O/bench_$wl.err ; echo "bench rc=$?" ; tail -2 $O/bench_$wl.err
  python - <<PY
import json
try:
    d=json.load(open("$O/bench_$wl.json"))
    print({k:d[k] for k in ("value","ms_per_step","check") if k in d}, d["config"]["workload"], "cpu", d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("no json", e)
PY
done
du -sh $O
