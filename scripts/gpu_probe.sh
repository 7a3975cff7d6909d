#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02o
mkdir -p $O
cd $R
export TMPDIR=/tmp
python scripts/bwd_probe.py wgrad 2>&1 | grep -v amdgpu
echo "== tests"; timeout 900 python -m pytest tests/test_bwd_ops.py tests/test_train_step.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1
echo "== PMC passes"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1 ; echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1
cp $O/pmc_traffic.json profiles/r02o_pmc_traffic.json
echo "== default bench"
timeout 1200 python bench.py --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err ; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_train.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["roofline"])
for k,v in list(d["kernels"].items())[:6]: print(k, v["calls"], round(v["total_ms"],2))
PY
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
