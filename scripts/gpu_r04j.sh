#!/bin/bash
# round 4, visit j: every BASELINE workload, training step and forward, on the round-4 kernels; a plain-bf16 line with
# its measured error; DepthNet on MIOpen vs the library's kernels for the SemanticKITTI workloads
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
for w in nusc_r50_200 nusc_r50_ref128 kitti_effb7_128 kitti_effb7_256lit nusc_r101; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/r04j_bench_train_$w.json 2>/dev/null
  timeout 400 python bench.py --workload $w --mode forward --steps 20 --warmup 3 --check > $O/r04j_bench_fwd_$w.json 2>/dev/null
  python - <<PY
import json
t = json.load(open("gpurun_out/r04j_bench_train_$w.json")); f = json.load(open("gpurun_out/r04j_bench_fwd_$w.json"))
print("$w train", round(t["value"], 3), "samples/s", round(t["ms_per_step"], 1), "ms, mem", t["peak_memory_GiB"], "| forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms check", f["check"]["output_voxels_max_abs_err"], "cpu", round(f["cpu_baseline"]["value"], 4))
PY
done
for w in kitti_effb7_128 kitti_effb7_256lit; do
  OCCF_DEPTHNET_LIB=0 timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/r04j_bench_train_${w}_miopen.json 2>/dev/null
  python -c "
import json; t=json.load(open('gpurun_out/r04j_bench_train_${w}_miopen.json')); print('$w DepthNet on MIOpen: train', round(t['value'],3), round(t['ms_per_step'],1))"
done
timeout 400 python bench.py --mode forward --precision bf16 --steps 20 --warmup 3 --check > $O/r04j_bench_fwd_bf16.json 2>/dev/null
timeout 900 python bench.py --precision bf16 > $O/r04j_bench_train_bf16.json 2>/dev/null
python - <<'PY'
import json
f = json.load(open("gpurun_out/r04j_bench_fwd_bf16.json")); t = json.load(open("gpurun_out/r04j_bench_train_bf16.json"))
print("bf16 forward", round(f["value"], 2), "check", f["check"], "roofline", round(f["roofline"]["frac"], 4))
print("bf16 train", round(t["value"], 3), round(t["ms_per_step"], 1), "check", {k: v for k, v in t["check"].items() if k != "what"}, "roofline", round(t["roofline"]["frac"], 4))
PY
