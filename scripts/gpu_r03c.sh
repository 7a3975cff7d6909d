#!/bin/bash
# r03c: class-major wgrad order, msda gather pass with 12 channels per lane, MFMA cross-attention backward, pipeline ops
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== probes"
( echo "-- default"; timeout 300 python scripts/bwd_probe.py wgrad msda xattn 2>&1 | grep " ms"
  echo "-- OCCF_MSDA_BWD_VEC12=0 OCCF_XATTN_BWD_MFMA=0"; OCCF_MSDA_BWD_VEC12=0 OCCF_XATTN_BWD_MFMA=0 timeout 300 python scripts/bwd_probe.py msda xattn 2>&1 | grep " ms" ) | tee $O/probe.txt
echo "== pytest"
timeout 900 python -m pytest tests/test_bwd_ops.py tests/test_pipeline_ops.py tests/test_train_step.py tests/test_boundary.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_a.log
echo "== bench train (short)"
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err; echo "rc=$?"; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"])
    for k,v in list(d["kernels"].items())[:16]: print("  ",k,v)
except Exception as e: print("no json", e)
PY
echo "== PMC wgrad 192 (traffic only)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python scripts/bwd_probe.py wgrad192 > $O/pmc_$c.log 2>&1 ); echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_traffic.txt 2>&1; head -8 $O/pmc_traffic.txt
find $O -name "*counter_collection.csv" -size +5M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +5M -delete 2>/dev/null
du -sh $O
