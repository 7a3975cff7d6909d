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
echo "== rocprof PMC passes on the convolution family's own probe (scripts/conv_family_probe.py: forward / data gradient / weight gradient at 192 and 128 channels)"
# (round 5: the whole bench under --pmc outgrew any sensible time limit, so the counters are collected on the roofline
# kernels' own probe; round 6: + the SQ / GRBM sets the roofline argument leans on -- matrix-pipe busy, sustained clock)
cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python $R/scripts/conv_family_probe.py 2 > $O/pmc_$i.log 2>&1 ; echo "pmc pass $i ($set) rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1 ; head -8 $O/pmc_summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
echo "== one rank on RCCL: kernel trace of a DDP step (are the bucket all-reduces under the backward?)"
cd /tmp
OCCF_DIST_AT_WORLD_1=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/prof_rccl -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 $R/bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_rccl.log 2>&1 ; echo "rccl trace rc=$?"
cd $R
python scripts/rccl_overlap.py $O/prof_rccl > $O/rccl_overlap.txt 2>&1 ; tail -12 $O/rccl_overlap.txt | cut -c1-220
find $O/prof_rccl -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $O
