#!/bin/bash
# round 4, visit d: kitti_effb7_256lit training parity (first execution anywhere), the training-parity tests under
# HIP_LAUNCH_BLOCKING=1, PMC of the halo conv with / without the explicit k-step pipeline, forward bench with the chained
# MLP at C = 128, kernel statistics of the forward
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_256lit" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04d_kitti256lit_training_parity.log
tail -6 $O/r04d_kitti256lit_training_parity.log | cut -c1-700
HIP_LAUNCH_BLOCKING=1 timeout 1200 python -m pytest tests/test_workloads_gpu.py tests/test_train_multistep.py -m gpu -q -p no:cacheprovider -s -k "not kitti_effb7_256lit" 2>&1 | grep -v "MIOpen(HIP)" > $O/r04d_pytest_train_launch_blocking.log
grep "training step vs oracle\|passed\|failed" $O/r04d_pytest_train_launch_blocking.log | cut -c1-330
for v in 1 2; do OCCF_MLP_CHAIN=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04d_bench_fwd_chain$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04d_bench_fwd_chain$v.json"))
print("OCCF_MLP_CHAIN=$v forward", round(d["value"], 2), "samples/s; mlp_fused", d["kernels"]["mlp_fused"])
PY
done
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $c | cut -c1-8 | tr ' ' '_')
    OCCF_HALO_SCHED=$s timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/r04d_pmc_sched${s}_$tag -- python $R/scripts/conv_probe.py 2 > /dev/null 2>&1
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04d_prof -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python scripts/summarize_prof.py $O/r04d_prof > $O/r04d_fwd_kernel_stats.txt 2>&1
head -40 $O/r04d_fwd_kernel_stats.txt | cut -c1-150
python - <<'PY'
import csv, glob, collections
for s in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/r04d_pmc_sched{s}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3x3x3_halo" not in k: continue
            key = (k[:60], r.get("Grid_Size"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
    for key, d in agg.items():
        print(f"SCHED={s}", key, {c: round(v / n[(key, c)]) for c, v in d.items()})
PY
