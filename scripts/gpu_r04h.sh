#!/bin/bash
# round 4, visit h: default bench on the per-camera rig; OCCF_LAZY_LOGITS=1 training bench
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
( time timeout 900 python bench.py ) > $O/r04h_bench_train.json 2> $O/r04h_bench_train.err
tail -3 $O/r04h_bench_train.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04h_bench_train.json"))
print("train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", round(d["roofline"]["frac"], 4))
print("check", {k: v for k, v in d.get("check", {}).items() if k != "what"})
print("forward", round(d["forward"]["value"], 2), d["forward"]["roofline"]["frac"], d["forward"].get("check"))
PY
OCCF_LAZY_LOGITS=1 timeout 600 python bench.py --no-cpu-baseline > $O/r04h_bench_train_lazy.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04h_bench_train_lazy.json"))
print("LAZY train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; mem", d["peak_memory_GiB"])
for k, v in list(d["kernels"].items())[:40]:
    if k in ("linear", "mask_pool", "mask_gemm_pool", "point_sample_3d", "point_sample_3d_rows"): print(f"  {k:28s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04h_bench_train.json"))
print("default mem", d["peak_memory_GiB"])
for k, v in list(d["kernels"].items())[:40]:
    if k in ("linear", "mask_pool", "mask_gemm_pool", "point_sample_3d", "point_sample_3d_rows"): print(f"  {k:28s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
