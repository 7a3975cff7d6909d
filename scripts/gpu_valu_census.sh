#!/bin/bash
# Which kernels of the training step keep the VECTOR pipe busy?  (r06q: the strided data gradient spent ~1 300 VALU
# instructions per thread and k-tile on run-time divisions -- invisible in a kernel-time table.)  Two counter passes
# over ONE training step; scripts/summarize_pmc.py --by-time lists every kernel with its VALU-busy fraction.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06s}
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
# (six SQ counters in one pass: MIOpen's backward-data convolution segfaulted under the profiler, r06s -- two smaller sets)
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  SECONDS=0
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_$i.log 2>&1 ; echo "pmc pass $i ($set) rc=$? ${SECONDS}s"
done
cd $R
python scripts/summarize_pmc.py $O --by-time > $O/valu_census.txt 2>&1 ; head -60 $O/valu_census.txt | cut -c1-200
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
du -sh $O
