#!/bin/bash
# GPU-box visit: full GPU suite, from-images bench, SQ / TA / TCP counters of the resample+classify kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-misc}
mkdir -p $O
cd $R
export TMPDIR=/tmp
if [ "${TESTS:-1}" = "1" ]; then
echo "== pytest -m gpu"
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1 ; echo "pytest rc=$?" ; tail -8 $O/pytest_gpu.log
fi
if [ "${IMAGES:-1}" = "1" ]; then
echo "== bench --from-images (bf16 image branch)"
timeout 900 python bench.py --mode forward --from-images --steps 5 --warmup 3 > $O/bench_from_images.json 2> $O/bench_from_images.err ; echo "rc=$?" ; tail -3 $O/bench_from_images.err
python - <<PY
import json
try:
    d=json.load(open("$O/bench_from_images.json")); print({k:d[k] for k in ("value","ms_per_step","stages_ms")})
except Exception as e: print("no json", e)
PY
fi
if [ "${UC:-1}" = "1" ]; then
echo "== upsample_classify counters"
python scripts/uc_probe.py
cd /tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
grep -c . $O/counters_list.txt
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/uc_pmc_$i -- python $R/scripts/uc_probe.py > $O/uc_pmc_$i.log 2>&1 ; echo "pmc pass $i ($set) rc=$?"
done
cd $R
python - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: [0.0,0])
for f in glob.glob("$O/uc_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "upsample_classify" in r["Kernel_Name"]:
            a=agg[r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
with open("$O/uc_pmc_summary.txt","w") as fo:
    for k,(v,n) in sorted(agg.items()):
        line=f"{k:40s} per launch {v/max(n,1):16.1f}   ({n} launches)"
        print(line); fo.write(line+"\n")
PY
find $O -name "*counter_collection.csv" -size +5M -delete 2>/dev/null
fi
du -sh $O
