#!/bin/bash
# round 4, visit i: resident MLP kernel with 4 / 8 waves per workgroup (forward bench), training bench with
# OCCF_LAZY_LOGITS=1 and OCCF_DEPTHNET_LIB=1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
for v in 4 8; do OCCF_MLP_RES_WAVES=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04i_bench_fwd_mlpw$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04i_bench_fwd_mlpw$v.json"))
print("OCCF_MLP_RES_WAVES=$v forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; mlp", d["kernels"]["mlp_fused"])
PY
done
for env in "OCCF_LAZY_LOGITS=1" "OCCF_LAZY_LOGITS=1 OCCF_DEPTHNET_LIB=1" "OCCF_LAZY_LOGITS=0"; do
  env $env timeout 600 python bench.py --no-cpu-baseline --steps 15 > $O/r04i_bench_train_tmp.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r04i_bench_train_tmp.json"))
print("$env train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; mem", d["peak_memory_GiB"], "fwd", round(d["forward"]["value"], 2))
PY
done
