#!/bin/bash
# PMC counters of one kernel family: scripts/pmc_probe.sh <out name> <kernel substring> <command...>
# (separate rocprofv3 --pmc passes, kernel trace only -- never combined with the hip / hsa / sys trace domains)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=$1; KSUB=$2; shift 2
O=$R/gpurun_out/$NAME
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_IFETCH" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_$i -- "$@" > $O/pmc_$i.log 2>&1 ); echo "pmc pass $i ($set) rc=$?"
done
cd $R
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0,0]))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "$KSUB" in r["Kernel_Name"]:
            a=agg[r["Kernel_Name"][:60]][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$O/pmc_summary.txt","w") as fo:
    for kn,d in agg.items():
        fo.write(kn+"\n"); print(kn)
        for k,(v,n) in sorted(d.items()):
            line=f"  {k:38s} per launch {v/max(n,1):18.1f}   ({n} launches)"
            print(line); fo.write(line+"\n")
PY
find $O -name "*counter_collection.csv" -size +5M -delete 2>/dev/null
find $O -name "*kernel_trace.csv" -size +5M -delete 2>/dev/null
