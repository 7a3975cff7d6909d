#!/bin/bash
# round 4, visit g: forward bench with / without the weight-resident Swin kernel; the switch test; training parity
# bounds re-check on one workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
for v in 0 1; do OCCF_SWIN_RES=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04g_bench_fwd_swinres$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04g_bench_fwd_swinres$v.json"))
print("OCCF_SWIN_RES=$v forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; swin", d["kernels"]["swin_attention_fused"], "mlp", d["kernels"]["mlp_fused"]["total_ms"])
PY
done
timeout 600 python -m pytest tests/test_attn_ops.py tests/test_switches.py tests/test_gemm_norm_ops.py -m gpu -q -p no:cacheprovider -k "swin or switch or mlp" 2>&1 | tail -3
timeout 300 python bench.py --mode forward --steps 10 --warmup 3 --check > $O/r04g_bench_fwd_check.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04g_bench_fwd_check.json')); print('check', d['check'], d['value'])"
