"""Fill the @@...@@ placeholders of DESIGN.md / README.md / profiles/README.md from a final visit's files
(gpurun_out/<tag>/ written by scripts/gpu_final.sh) and copy the evidence to profiles/r06/<tag>_*.

    python scripts/fill_final_docs.py <tag>"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles", "r06")
for f in ("bench_train.json", "bench_fwd.json", "kernel_stats.txt", "fwd_kernel_stats.txt", "pmc_traffic.json", "pmc_summary.txt",
          "pytest_gpu.log", "smoke.log", "rccl_overlap.txt", "shapes_train.txt", "shapes_fwd.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
d = json.load(open(os.path.join(src, "bench_train.json")))
c = d["check"]
q = c["per_parameter_rel_l2_quantiles"]
fw, fi, ti = d["forward"], d["forward_from_images"], d["train_from_images"]
r = d["roofline"]
log = open(os.path.join(src, "pytest_gpu.log")).read()
m = re.search(r"(\d+) passed, (\d+) skipped", log)
failed = re.search(r"(\d+) failed", log)
pytest_s = f"{m.group(1)} passed, {m.group(2)} skipped" + (f", {failed.group(1)} FAILED" if failed else "")
smoke = open(os.path.join(src, "smoke.log")).read()
ms = re.search(r"whole-gradient rel L2 ([0-9.e+-]+)", smoke)
e = lambda v: f"{v:.1e}".replace("e-0", "e-")
short = lambda k: k.split("  [")[0]
fam = "; ".join(f"`{short(f['kernel'])}` {f['frac']:.3f} ({f['mfma_products_per_algorithmic_product']:g} products)" for f in r["family"][:7])
traffic = r.get("traffic")
roof = (f"{r['frac']:.3f} (`{short(r['kernel'])}`, {r['avg_kernel_ms']:.2f} ms, {r['mfma_products_per_algorithmic_product']:g} products per product"
        + (f", matrix pipe {100 * r['mfma_busy_frac']:.0f} % busy at {r['clock_ghz']:.2f} GHz" if r.get("mfma_busy_frac") else "")
        + (f", traffic {traffic / 1e9:.2f} GB per launch for {r['algorithmic_bytes_per_launch'] / 1e9:.2f} GB algorithmic" if traffic else "") + ")")
rep = {
    "@@FINALTAG@@": tag,
    "@@TRAIN2@@": f"{d['value']:.2f} samples/s ({d['ms_per_step']:.1f} ms per step)",
    "@@TRAIN2MS@@": f"{d['ms_per_step']:.0f}",
    "@@CPU2@@": f"{d['cpu_baseline']['value']:.4f}",
    "@@GRAD2@@": e(c["grad_rel_l2"]), "@@UNGATED2@@": e(c["grad_rel_l2_ungated"]), "@@P902@@": e(q["90%"]), "@@WORST2@@": e(q["100%"]),
    "@@FWD2@@": f"{fw['value']:.1f}", "@@FWDERR2@@": e(fw["check"]["output_voxels_max_abs_err"]),
    "@@FWDIMG2@@": f"{fi['value']:.1f}", "@@FWDPIPE2@@": f"{fi['pipelined']['value']:.1f}", "@@TRAINIMG2@@": f"{ti['value']:.2f}",
    "@@ROOF2@@": roof, "@@FAMILY2@@": fam, "@@PYTEST2@@": pytest_s, "@@SMOKE2@@": (ms.group(1) if ms else "ok"),
}
rep["@@FINALROW@@"] = (f"`pytest -m gpu` **{pytest_s}**, smoke (whole gradient {rep['@@SMOKE2@@']}), default bench **{rep['@@TRAIN2@@']}** with "
                       f"`check` {rep['@@GRAD2@@']} gated / {rep['@@UNGATED2@@']} ungated, worst parameter {rep['@@WORST2@@']}; forward {rep['@@FWD2@@']}, "
                       f"from images {rep['@@FWDIMG2@@']} sequential / {rep['@@FWDPIPE2@@']} pipelined, `train_from_images` {rep['@@TRAINIMG2@@']}, "
                       f"`cpu_baseline` {rep['@@CPU2@@']}; forward bench `--check`, rocprofv3 kernel statistics of both, the counter passes of the "
                       f"convolution family stamped with the final sources' digest, the one-rank RCCL trace")
for p in ("DESIGN.md", "README.md", os.path.join("profiles", "README.md")):
    fp = os.path.join(ROOT, p)
    s = open(fp).read()
    for k, v in rep.items():
        s = s.replace(k, v)
    left = re.findall(r"@@[A-Z0-9]+@@", s)
    open(fp, "w").write(s)
    print(p, "left:", left)
for k, v in rep.items():
    print(k, v[:160])
