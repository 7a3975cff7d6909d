"""Where one training step spends its GPU time, from a `rocprofv3 --kernel-trace` CSV of scripts/train_loop_probe.py.
    python scripts/step_timeline.py <dir with *kernel_trace.csv> [top]
The probe brackets every timed step with a marker launch (a fill of a float64 tensor: the only FillFunctor<double> in the
process).  Between two markers: total busy / idle time, and per kernel name the launches, summed duration and share, with
the kernels grouped into OWN (this library), ATEN (at::native / rocclr copies / fills), LIB (MIOpen / rocBLAS / hipBLASLt)."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
marks = [i for i, r in enumerate(rows) if "FillFunctor<double>" in r[2]]
if len(marks) < 2:
    raise SystemExit("no step markers in the trace (%d kernels)" % len(rows))


def group(n):
    if n.startswith("void at::") or n.startswith("at::") or "rocclr" in n or "at::native" in n or "at_cuda_detail" in n:
        return "ATEN"
    if n.startswith("Cijk_") or "igemm" in n or "naive_conv" in n or "miopen" in n.lower() or "SubTensorOp" in n \
            or "batched_transpose" in n or "gridwise" in n or "ck::" in n:
        return "LIB"
    return "OWN"


steps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a + 1:b]
    if len(seg) < 10:
        continue
    t0, t1 = rows[a][1], rows[b][0]
    busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    steps.append((t1 - t0, busy, seg))
print("steps in trace: %d" % len(steps))
for i, (w, b, seg) in enumerate(steps):
    print("  step %d: wall %.2f ms, busy %.2f ms, idle %.2f ms, %d launches" % (i, w / 1e6, b / 1e6, (w - b) / 1e6, len(seg)))
w, b, seg = steps[-1]
agg = defaultdict(lambda: [0, 0])
grp = defaultdict(lambda: [0, 0])
for s, e, n in seg:
    k = n.split("(")[0][:100]
    agg[k][0] += 1
    agg[k][1] += e - s
    g = group(n)
    grp[g][0] += 1
    grp[g][1] += e - s
tot = sum(v[1] for v in agg.values())
print("last step: sum of kernel durations %.2f ms over %d launches" % (tot / 1e6, len(seg)))
for g, (c, t) in sorted(grp.items(), key=lambda kv: -kv[1][1]):
    print("  %-5s %6d launches %9.3f ms  %5.1f %%" % (g, c, t / 1e6, 100.0 * t / tot))
print("%-100s %6s %9s %8s %6s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print("%-100s %6d %9.3f %8.1f %6.2f  %s" % (k, c, t / 1e6, t / c / 1e3, 100.0 * t / tot, group(k)))
