"""Do the gradient all-reduces run UNDER the backward?  From a rocprofv3 --kernel-trace of a DDP training step
(one rank on RCCL: OCCF_DIST_AT_WORLD_1=1 torchrun --nproc-per-node 1 bench.py ...): every RCCL kernel with its start /
end, and how much of its lifetime other (compute) kernels were running on the device -- timestamps, not inference.

    python scripts/rccl_overlap.py <rocprof output dir>"""
import csv
import glob
import os
import sys

root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
if not rows:
    sys.exit("no kernel trace found")
is_cc = lambda n: any(t in n.lower() for t in ("nccl", "rccl"))
cc = [r for r in rows if is_cc(r[2])]
comp = [r for r in rows if not is_cc(r[2])]
t0 = rows[0][0]
print(f"{len(rows)} kernel launches, {len(cc)} of them RCCL; trace span {(rows[-1][1] - t0) / 1e6:.2f} ms")
# the last training step of the trace: from the last optimizer kernel backwards is fragile; report per RCCL kernel instead
import bisect
starts = [c[0] for c in comp]
tot_cc = tot_ov = 0
lines = []
for s, e, n, q, st in cc:
    i = bisect.bisect_left(starts, s) - 64
    ov = 0
    names = []
    for cs, ce, cn, cq, cst in comp[max(i, 0):]:
        if cs >= e:
            break
        o = min(e, ce) - max(s, cs)
        if o > 0:
            ov += o
            if len(names) < 3:
                names.append(cn.split("(")[0][:40])
    tot_cc += e - s
    tot_ov += min(ov, e - s)
    lines.append(f"  +{(s - t0) / 1e6:9.3f} ms  {(e - s) / 1e3:9.1f} us  queue {q:>3s}  compute kernels running during {100.0 * min(ov, e - s) / max(e - s, 1):5.1f} % of it"
                 f"  {n[:48]:48s} | beside: {', '.join(names)}")
for l in lines[-40:]:
    print(l)
if tot_cc:
    print(f"RCCL kernel time {tot_cc / 1e6:.3f} ms in the trace, {100.0 * tot_ov / tot_cc:.1f} % of it with compute kernels running on the device at the same time")
