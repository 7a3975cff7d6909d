"""GPU busy / idle time from a rocprofv3 --kernel-trace CSV: how much of a training step the GPU waits for the host.
    python scripts/gap_analysis.py <dir with *kernel_trace.csv> [skip_first_fraction]
Kernels are sorted by start; overlapping intervals merged; every idle gap is attributed to the kernel that ENDS it
(the launch the host was late with).  The first part of the trace (warm-up) is skipped."""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if not rows:
    raise SystemExit("no kernel trace")
t_begin, t_end = rows[0][0], max(r[1] for r in rows)
cut = t_begin + int((t_end - t_begin) * skip)
rows = [r for r in rows if r[0] >= cut]
busy = 0
idle = 0
gaps = defaultdict(lambda: [0, 0])
small = defaultdict(lambda: [0, 0])
cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, n in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        g = s - cur_e
        idle += g
        key = n.split("(")[0][:70]
        gaps[key][0] += 1
        gaps[key][1] += g
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    if e - s < 5000:
        k2 = n.split("(")[0][:70]
        small[k2][0] += 1
        small[k2][1] += e - s
busy += cur_e - cur_s
tot = busy + idle
print(f"window {tot / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({100 * busy / tot:.1f} %), idle {idle / 1e6:.2f} ms "
      f"({100 * idle / tot:.1f} %), {len(rows)} kernels, {len(rows) / (tot / 1e6):.1f} kernels per ms")
print("idle time by the kernel that ended the gap:")
for k, (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"  {g / 1e6:8.3f} ms  {c:6d} gaps  avg {g / c / 1e3:7.1f} us  {k}")
print("kernels shorter than 5 us (count, total ms):")
for k, (c, t) in sorted(small.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {c:6d}  {t / 1e6:8.3f} ms  {k}")
