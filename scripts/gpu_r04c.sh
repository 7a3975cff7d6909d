#!/bin/bash
# round 4, visit c: forward bench with the spill-free fused Swin kernel (+ per-shape table, kernel stats), then the
# full GPU suite under OCCF_TEST_POISON=1 with the complete failure output
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
timeout 600 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline --shape-report $O/r04c_fwd_shapes.txt > $O/r04c_bench_fwd.json 2> $O/r04c_bench_fwd.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04c_bench_fwd.json"))
print("forward", d["value"], "samples/s", d["ms_per_step"], "ms; roofline", d["roofline"]["frac"], d["roofline"]["avg_kernel_ms"])
for k, v in list(d["kernels"].items())[:16]:
    print(f"  {k:28s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
head -30 $O/r04c_fwd_shapes.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/r04c_prof -o fwd -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cd $R
python scripts/summarize_prof.py $O/r04c_prof > $O/r04c_fwd_kernel_stats.txt 2>&1 || ls -R $O/r04c_prof | head
head -30 $O/r04c_fwd_kernel_stats.txt
OCCF_TEST_POISON=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "MIOpen(HIP)" > $O/r04c_pytest_gpu_poison.log
tail -5 $O/r04c_pytest_gpu_poison.log
