"""Per-kernel counters from separate rocprofv3 --pmc passes under <root>/pmc_*/ (one counter set per pass, kernel trace
only): FETCH_SIZE / WRITE_SIZE -> HBM traffic (KiB units; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the
bytes of a wide coalesced stream, so the read side is doubled), and -- when the passes exist -- SQ_VALU_MFMA_BUSY_CYCLES,
SQ_INSTS_MFMA, SQ_INSTS_VALU, SQ_WAIT_INST_ANY, SQ_WAVE_CYCLES, GRBM_GUI_ACTIVE -> matrix-pipe busy fraction
(busy cycles / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)), sustained clock (GRBM_GUI_ACTIVE / 8 / kernel duration), the
share of wave cycles spent waiting on a counter, VALU instructions per MFMA.

    python scripts/summarize_pmc.py <root> [out.json]        (out.json is stamped with the kernel-source digest)"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]
val = defaultdict(lambda: defaultdict(float))        # (kernel, grid) -> counter -> sum
cnt = defaultdict(lambda: defaultdict(int))          # (kernel, grid) -> counter -> launches
dur = defaultdict(lambda: [0.0, 0])                  # (kernel, grid) -> [sum ns, launches]  (GRBM pass, else any)
for d in sorted(glob.glob(os.path.join(root, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    seen = set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Kernel_Name"], int(float(r.get("Grid_Size", 0) or 0)))
            c = r["Counter_Name"]
            val[key][c] += float(r["Counter_Value"])
            cnt[key][c] += 1
            did = (f, r.get("Dispatch_Id"))
            if c == "GRBM_GUI_ACTIVE" and did not in seen and r.get("Start_Timestamp") and r.get("End_Timestamp"):
                seen.add(did)
                dur[key][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                dur[key][1] += 1
    if any("GRBM_GUI_ACTIVE" in v for v in val.values()) and not any(v[1] for v in dur.values()):
        # (no timestamps in the counter file: the kernel trace of the same pass)
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                key = (r["Kernel_Name"], int(float(r.get("Grid_Size", 0) or 0)))
                if "GRBM_GUI_ACTIVE" in val.get(key, {}):
                    dur[key][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    dur[key][1] += 1


def per_launch(key, c):
    return val[key][c] / cnt[key][c] if cnt[key].get(c) else None


rows = []
for key in val:
    k, g = key
    fe, wr = per_launch(key, "FETCH_SIZE"), per_launch(key, "WRITE_SIZE")
    row = dict(kernel=k, grid=g, calls=max(cnt[key].values()), fetch_bytes=2 * (fe or 0.0) * 1024, write_bytes=(wr or 0.0) * 1024)
    gui, busy, mf, va = (per_launch(key, c) for c in ("GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_INSTS_VALU"))
    wa, wc = per_launch(key, "SQ_WAIT_INST_ANY"), per_launch(key, "SQ_WAVE_CYCLES")
    if gui and busy:
        row["mfma_busy_frac"] = busy / 1024.0 / (gui / 8.0)
    if gui and dur[key][1]:
        row["clock_ghz"] = gui / 8.0 / (dur[key][0] / dur[key][1])
        row["duration_us_under_pmc"] = dur[key][0] / dur[key][1] / 1e3
    if wa and wc:
        row["wait_inst_any_frac"] = wa / wc
    if mf and va:
        row["valu_per_mfma"] = va / mf
        row["mfma_per_launch"] = mf
    if gui and va:
        # a wave64 VALU instruction occupies its SIMD's vector pipe for 4 cycles (MFMAs are counted in SQ_INSTS_VALU too:
        # they issue in one pass and run on the matrix pipe, so they are taken out first)
        row["valu_busy_frac"] = (va - (mf or 0.0)) * 4.0 / 1024.0 / (gui / 8.0)
    rows.append(row)
BY_TIME = "--by-time" in sys.argv
if BY_TIME:
    sys.argv.remove("--by-time")
    rows.sort(key=lambda r: -r.get("duration_us_under_pmc", 0.0) * r["calls"])
else:
    rows.sort(key=lambda r: -(r["fetch_bytes"] + r["write_bytes"]) * r["calls"])
print(f"{'kernel':72s} {'grid':>9s} {'calls':>5s} {'fetch_x2 MB':>12s} {'write MB':>9s} {'mfma busy':>9s} {'GHz':>5s} {'wait':>5s} {'valu/mfma':>9s} {'valu busy':>9s} {'us':>8s}")
fmt = lambda v, f: (f % v) if v is not None else "-"
for r in rows[:(120 if BY_TIME else 40)]:
    print(f"{r['kernel'][:72]:72s} {r['grid']:9d} {r['calls']:5d} {r['fetch_bytes'] / 1e6:12.2f} {r['write_bytes'] / 1e6:9.2f} "
          f"{fmt(r.get('mfma_busy_frac'), '%.3f'):>9s} {fmt(r.get('clock_ghz'), '%.2f'):>5s} {fmt(r.get('wait_inst_any_frac'), '%.2f'):>5s} "
          f"{fmt(r.get('valu_per_mfma'), '%.2f'):>9s} {fmt(r.get('valu_busy_frac'), '%.3f'):>9s} {fmt(r.get('duration_us_under_pmc'), '%.1f'):>8s}")
if len(sys.argv) > 2:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from occformer_amd.csrc.build import _digest
    json.dump(dict(source_digest=_digest(),
                   note="rocprofv3 --pmc, one counter set per pass, per-launch averages; fetch_bytes = 2 x FETCH_SIZE KiB x 1024 "
                        "(MI355X_MICROARCH.md gfx950 correction); mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / "
                        "(GRBM_GUI_ACTIVE / 8 XCDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration",
                   kernels=rows[:200]), open(sys.argv[2], "w"), indent=1)
