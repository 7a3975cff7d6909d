#!/bin/bash
# FETCH / WRITE passes of the training step for the CURRENT kernel sources -> <out>/pmc_traffic.json (stamped with the digest)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$c.log 2>&1 ; echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $O $O/pmc_traffic.json > $O/pmc_summary.txt 2>&1 ; head -6 $O/pmc_summary.txt
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +5M -delete 2>/dev/null
