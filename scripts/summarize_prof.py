"""Summarise rocprofv3 --kernel-trace --stats CSV output: per-kernel calls / total / average."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)
if files:
    rows = list(csv.DictReader(open(files[0])))
    print(f"# {files[0]}")
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:60]:
        name = r.get("Name", r.get("KernelName", "?"))[:90]
        print(f"{name:90s} {r.get('Calls', '?'):>6s} {float(r.get('TotalDurationNs', 0)) / 1e6:10.3f} "
              f"{float(r.get('AverageNs', 0)) / 1e3:10.2f} {r.get('Percentage', '?'):>6s}")
else:
    traces = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    agg = defaultdict(lambda: [0, 0.0])
    for f in traces:
        for r in csv.DictReader(open(f)):
            dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += dur
    tot = sum(v[1] for v in agg.values()) or 1.0
    print(f"{'kernel':90s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print(f"{k[:90]:90s} {n:6d} {t / 1e6:10.3f} {t / n / 1e3:10.2f} {100 * t / tot:6.2f}")
