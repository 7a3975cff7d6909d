#!/bin/bash
# round 4, visit f: training parity of every workload on the per-camera-jittered rig (gate tape incl. DepthNet), forward
# bench with the weight-resident MLP kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04f_pytest_workloads.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r04f_pytest_workloads.log | cut -c1-900
timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --check > $O/r04f_bench_fwd.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04f_bench_fwd.json"))
print("forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; check", d.get("check"), "roofline", round(d["roofline"]["frac"], 4))
for k, v in list(d["kernels"].items())[:12]:
    print(f"  {k:28s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
