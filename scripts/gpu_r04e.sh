#!/bin/bash
# round 4, visit e: the full GPU suite on the current tree (durations of the slowest tests), then the default bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=12 ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04e_pytest_gpu.log
tail -28 $O/r04e_pytest_gpu.log | cut -c1-300
( time timeout 900 python bench.py ) > $O/r04e_bench_train.json 2> $O/r04e_bench_train.err
tail -4 $O/r04e_bench_train.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04e_bench_train.json"))
print("train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", d["roofline"]["kernel"][:60], round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("check", d.get("check"))
print("forward", round(d["forward"]["value"], 2), d["forward"]["roofline"]["frac"], d["forward"].get("check"))
print("cpu", d["cpu_baseline"])
for k, v in list(d["kernels"].items())[:25]:
    print(f"  {k:30s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
