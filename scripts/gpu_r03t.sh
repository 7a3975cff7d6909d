#!/bin/bash
# r03t: dense matrix-core value gradient of the coarsest msda level: parity + timing; img_inputs producer on the GPU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03t
mkdir -p $O
cd $R
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bwd_ops.py tests/test_full_size_gpu.py tests/test_pipeline_ops.py -m gpu -q -x -k "msda or image or load_multi" -p no:cacheprovider 2>&1 | tail -3 | tee $O/pytest.txt
( for v in "OCCF_MSDA_DENSE=0" "OCCF_MSDA_DENSE=1"; do echo "-- $v"; env $v timeout 300 python scripts/bwd_probe.py msda 2>&1 | grep " ms"; done ) | tee $O/msda_probe.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/scripts/bwd_probe.py msda > $O/prof.log 2>&1; echo "rocprof rc=$?"
cd $R
python scripts/summarize_prof.py $O/prof > $O/kernel_stats.txt 2>&1; head -12 $O/kernel_stats.txt
find $O/prof -name "*.csv" -size +1M -delete 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err; echo "bench rc=$?"
python - <<PY
import json
try:
    d=json.load(open("$O/bench_train.json")); print({k:d[k] for k in ("value","ms_per_step")}); print(d["kernels"].get("msda3d_backward"))
except Exception as e: print("no json", e)
PY
