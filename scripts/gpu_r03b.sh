#!/bin/bash
# r03b: wgrad retile probes + parity tests + training parity at full size + bench / rocprof / PMC of the wgrad kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== wgrad probes"
for v in "OCCF_WG_MIX=1 OCCF_WG_BUFLOAD=1" "OCCF_WG_MIX=0 OCCF_WG_BUFLOAD=1" "OCCF_WG_MIX=1 OCCF_WG_BUFLOAD=0" "OCCF_WG_MIX=0 OCCF_WG_BUFLOAD=0"; do
  echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py wgrad 2>&1 | grep -v Warn | tail -4
done | tee $O/wgrad_probe.txt
echo "== pytest"
timeout 900 python -m pytest tests/test_bwd_ops.py tests/test_train_multistep.py tests/test_boundary.py -m gpu -q -x -p no:cacheprovider > $O/pytest_a.log 2>&1; echo "rc=$?"; tail -3 $O/pytest_a.log
timeout 1200 python -m pytest tests/test_workloads_gpu.py -m gpu -q -s -k training_step -p no:cacheprovider > $O/pytest_workloads_train.log 2>&1; echo "rc=$?"; grep -E "^\[|passed|failed" $O/pytest_workloads_train.log | cut -c1-600
echo "== bench train (short)"
timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err; echo "rc=$?"
OCCF_DEPTHNET_LIB=1 timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_train_depthnet_lib.json 2> $O/bench_train_dl.err; echo "rc=$?"
python - <<PY
import json
for f in ("bench_train","bench_train_depthnet_lib"):
    try:
        d=json.load(open("$O/%s.json"%f)); print(f, {k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["avg_kernel_ms"], d["roofline"]["frac"], d["forward"]["value"])
        if f=="bench_train":
            for k,v in list(d["kernels"].items())[:30]: print("  ",k,v)
    except Exception as e: print(f, "no json", e)
PY
head -30 $O/shapes_train.txt
echo "== rocprof kernel trace (train)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1; head -50 $O/kernel_stats.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/rocprofv3_kernel_stats.csv 2>/dev/null
find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
echo "== PMC wgrad 192"
bash scripts/pmc_probe.sh r03b/wgrad_192_pmc wgrad_kernel python scripts/bwd_probe.py wgrad192 > $O/pmc.log 2>&1; tail -40 $O/pmc.log
du -sh $O
