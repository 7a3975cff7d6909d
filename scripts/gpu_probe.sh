#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02k
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu (attention / mask head / training files: the sources changed since r02j)"
timeout 900 python -m pytest tests/test_attn_ops.py tests/test_full_size_gpu.py tests/test_train_step.py tests/test_switches.py -m gpu -q -p no:cacheprovider > $O/pytest_gpu_subset.log 2>&1 ; echo "pytest rc=$?" ; tail -2 $O/pytest_gpu_subset.log
echo "== rocprof PMC passes (HBM traffic of the training step)"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1 ; echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1 ; head -4 $O/pmc_summary.txt
cp $O/pmc_traffic.json profiles/r02k_pmc_traffic.json
for wl in nusc_r50_ref128 kitti_effb7_128; do
  echo "== train bench $wl"; timeout 600 python bench.py --mode train --workload $wl --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_train_$wl.json 2> $O/err_$wl.txt; echo rc=$?
  python -c "
import json
d=json.load(open('$O/bench_train_$wl.json')); print({k:d[k] for k in ('value','ms_per_step','peak_memory_GiB')})"
done
echo "== default bench with the traffic summary in place"
timeout 1200 python bench.py --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err ; echo "bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/bench_train.json"))
print({k:d[k] for k in ("value","ms_per_step")}); print(d["roofline"])
PY
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
