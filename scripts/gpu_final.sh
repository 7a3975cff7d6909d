#!/bin/bash
# Final GPU-box visit of a round: all -m gpu tests, smoke, the default bench (training step incl. the CPU baseline),
# the forward bench with --check, rocprofv3 kernel statistics and the FETCH / WRITE counter passes of the training step.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02i}
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -6 $O/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -4 $O/smoke.log
echo "== default bench (training step)"
SECONDS=0
timeout 1200 python bench.py --shape-report $O/shapes_train.txt > $O/bench_train.json 2> $O/bench_train.err ; echo "bench rc=$? wall ${SECONDS}s" ; tail -3 $O/bench_train.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json"))
    print({k:d[k] for k in ("metric","value","ms_per_step","peak_memory_GiB")}, "forward", (d.get("forward") or {}).get("value"))
    print(d["roofline"]); print(d.get("cpu_baseline"))
    for k,v in list(d["kernels"].items())[:14]: print(k, v["calls"], round(v["total_ms"],2))
except Exception as e: print("no json", e)
PY
echo "== forward bench --check"
timeout 1200 python bench.py --mode forward --check --shape-report $O/shapes_fwd.txt > $O/bench_fwd.json 2> $O/bench_fwd.err ; echo "bench rc=$?" ; tail -3 $O/bench_fwd.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_fwd.json"))
    print({k:d.get(k) for k in ("value","ms_per_step","check","stages_ms")}); print(d["roofline"]); print(d.get("cpu_baseline"))
except Exception as e: print("no json", e)
PY
echo "== rocprof kernel trace (training step)"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1 ; echo "rocprof rc=$?" ; tail -2 $O/prof.log
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1 ; head -50 $O/kernel_stats.txt
echo "== rocprof kernel trace (forward)"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > $O/prof_fwd.log 2>&1 ; echo "rocprof fwd rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof_fwd > $O/fwd_kernel_stats.txt 2>&1 ; head -24 $O/fwd_kernel_stats.txt | cut -c1-150
find $O/prof_fwd -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
echo "== rocprof PMC passes (HBM traffic of the halo convolutions on these kernel sources: scripts/conv_probe.py)"
# (round 5: the whole bench under --pmc outgrew any sensible time limit -- it also builds and runs the from-images
# detector now -- so the counters are collected on the roofline kernel's own probe; scripts/gpu_r05.sh stage q)
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/scripts/conv_probe.py 2 > $O/pmc_$c.log 2>&1 ; echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1 ; head -8 $O/pmc_summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $O
