#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel trace.  Everything is logged
# under gpurun_out/ (merged back by gpurun).  Each step has its own timeout.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-run}
mkdir -p $O
cd $R
export TMPDIR=/tmp
echo "== rocminfo" ; (rocminfo | grep -E "Marketing Name|gfx" | head -4) 2>&1
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -15 $O/pytest_gpu.log
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1 ; echo "smoke rc=$?" ; tail -4 $O/smoke.log
echo "== bench"
timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --check --shape-report $O/shapes.txt > $O/bench.json 2> $O/bench.err ; echo "bench rc=$?" ; tail -3 $O/bench.err ; cat $O/bench.json ; head -40 $O/shapes.txt
for pr in ${EXTRA_PREC:-}; do
  timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --precision $pr --check > $O/bench_$pr.json 2> $O/bench_$pr.err ; echo "bench $pr rc=$?" ; cat $O/bench_$pr.json
done
if [ "${TRAIN:-0}" = "1" ]; then
  echo "== training rows probe"
  timeout 600 python scripts/train_probe.py > $O/train_nusc.json 2> $O/train_nusc.err ; echo "train nusc rc=$?" ; tail -2 $O/train_nusc.err ; cat $O/train_nusc.json
  timeout 600 python scripts/train_probe.py --kitti > $O/train_kitti.json 2> $O/train_kitti.err ; echo "train kitti rc=$?" ; tail -2 $O/train_kitti.err ; cat $O/train_kitti.json
fi
echo "== rocprof kernel trace"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1 ; echo "rocprof rc=$?" ; tail -2 $O/prof.log
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1 ; head -${HEADN:-45} $O/kernel_stats.txt
if [ "${PMC:-0}" = "1" ]; then
  echo "== rocprof PMC passes (HBM traffic; separate runs, no tracing domains besides kernel-trace)"
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1 ; echo "pmc $c rc=$?"
  done
  cd $R
  python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1 ; head -30 $O/pmc_summary.txt
  find $O -name "*counter_collection.csv" -size +30M -delete 2>/dev/null
fi
# keep the merged-back artefacts small
find $O/prof -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $O
