#!/bin/bash
# Every BASELINE workload on the current tree: training step and forward, from the neck features (bench.py sub-records off).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06wl}
mkdir -p $O
cd $R
for w in nusc_r50_200 nusc_r50_ref128 kitti_effb7_128 kitti_effb7_256lit nusc_r101; do
  timeout 600 python bench.py --workload $w --no-cpu-baseline --steps 10 --warmup 3 > $O/train_$w.json 2> $O/train_$w.err ; echo "$w train rc=$?"
  timeout 600 python bench.py --workload $w --mode forward --no-cpu-baseline --steps 20 --warmup 3 > $O/fwd_$w.json 2> $O/fwd_$w.err ; echo "$w fwd rc=$?"
done
python - <<PY
import json
for w in "nusc_r50_200 nusc_r50_ref128 kitti_effb7_128 kitti_effb7_256lit nusc_r101".split():
    row = [w]
    for m in ("train", "fwd"):
        try:
            d = json.load(open("$O/%s_%s.json" % (m, w)))
            row.append("%.2f samples/s (%.1f ms)" % (d["value"], d["ms_per_step"]))
        except Exception as e:
            row.append("failed: %r" % (e,))
    print(" | ".join(row))
PY
