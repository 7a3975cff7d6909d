#!/bin/bash
# The GPU visits of round 4, one stage per visit:  bash scripts/gpu_r04.sh <stage>   (results under gpurun_out/, the
# files worth keeping are copied to profiles/r04/ and indexed in profiles/README.md).  scripts/gpu_final.sh is the
# end-of-round visit (suite, smoke, bench, rocprof statistics, PMC traffic stamp).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
case "${1:-}" in
a)  # round 4, visit a: halo-conv schedule probe + the full GPU suite under OCCF_TEST_POISON=1
free -g | head -2 > $O/r04a_host.txt; nproc >> $O/r04a_host.txt
for s in 0 1; do OCCF_HALO_SCHED=$s timeout 300 python scripts/conv_probe.py; done > $O/r04a_conv_probe.txt 2>&1
cat $O/r04a_conv_probe.txt
OCCF_TEST_POISON=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -25 > $O/r04a_pytest_gpu_poison.log
tail -8 $O/r04a_pytest_gpu_poison.log
;;
b)  # round 4, visit b: the kitti_effb7_128 training-parity test alone, with and without OCCF_TEST_POISON (full output)
OCCF_TEST_POISON=1 timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_128" > $O/r04b_kitti128_poison.log 2>&1
tail -60 $O/r04b_kitti128_poison.log | cut -c1-400
timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_128" > $O/r04b_kitti128_plain.log 2>&1
tail -12 $O/r04b_kitti128_plain.log | cut -c1-600
;;
c)  # round 4, visit c: forward bench with the spill-free fused Swin kernel (+ per-shape table, kernel stats), then the full GPU suite under OCCF_TEST_POISON=1 with the complete failure output
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
python scripts/summarize_prof.py $O/r04c_prof > $O/r04c_fwd_kernel_stats.txt 2>&1 || ls -R $O/r04c_prof | head
head -30 $O/r04c_fwd_kernel_stats.txt
OCCF_TEST_POISON=1 timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -v "MIOpen(HIP)" > $O/r04c_pytest_gpu_poison.log
tail -5 $O/r04c_pytest_gpu_poison.log
;;
d)  # round 4, visit d: kitti_effb7_256lit training parity (first execution anywhere), the training-parity tests under HIP_LAUNCH_BLOCKING=1, PMC of the halo conv with / without the explicit k-step pipeline, forward bench with the chained MLP at C = 128, kernel statistics of the forward
( time timeout 1500 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -x -s -k "training_step and kitti_effb7_256lit" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04d_kitti256lit_training_parity.log
tail -6 $O/r04d_kitti256lit_training_parity.log | cut -c1-700
HIP_LAUNCH_BLOCKING=1 timeout 1200 python -m pytest tests/test_workloads_gpu.py tests/test_train_multistep.py -m gpu -q -p no:cacheprovider -s -k "not kitti_effb7_256lit" 2>&1 | grep -v "MIOpen(HIP)" > $O/r04d_pytest_train_launch_blocking.log
grep "training step vs oracle\|passed\|failed" $O/r04d_pytest_train_launch_blocking.log | cut -c1-330
for v in 1 2; do OCCF_MLP_CHAIN=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04d_bench_fwd_chain$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04d_bench_fwd_chain$v.json"))
print("OCCF_MLP_CHAIN=$v forward", round(d["value"], 2), "samples/s; mlp_fused", d["kernels"]["mlp_fused"])
PY
done
cd /tmp && export TMPDIR=/tmp
for s in 0 1; do
  for c in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
    tag=$(echo $c | cut -c1-8 | tr ' ' '_')
    OCCF_HALO_SCHED=$s timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/r04d_pmc_sched${s}_$tag -- python $R/scripts/conv_probe.py 2 > /dev/null 2>&1
  done
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04d_prof -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python scripts/summarize_prof.py $O/r04d_prof > $O/r04d_fwd_kernel_stats.txt 2>&1
head -40 $O/r04d_fwd_kernel_stats.txt | cut -c1-150
python - <<'PY'
import csv, glob, collections
for s in (0, 1):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(f"gpurun_out/r04d_pmc_sched{s}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "conv3x3x3_halo" not in k: continue
            key = (k[:60], r.get("Grid_Size"))
            agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
    for key, d in agg.items():
        print(f"SCHED={s}", key, {c: round(v / n[(key, c)]) for c, v in d.items()})
PY
;;
e)  # round 4, visit e: the full GPU suite on the current tree (durations of the slowest tests), then the default bench
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=12 ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04e_pytest_gpu.log
tail -28 $O/r04e_pytest_gpu.log | cut -c1-300
( time timeout 900 python bench.py ) > $O/r04e_bench_train.json 2> $O/r04e_bench_train.err
tail -4 $O/r04e_bench_train.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04e_bench_train.json"))
print("train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", d["roofline"]["kernel"][:60], round(d["roofline"]["frac"], 4), "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_source"])
print("check", d.get("check"))
print("forward", round(d["forward"]["value"], 2), d["forward"]["roofline"]["frac"], d["forward"].get("check"))
print("cpu", d["cpu_baseline"])
for k, v in list(d["kernels"].items())[:25]:
    print(f"  {k:30s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
;;
f)  # round 4, visit f: training parity of every workload on the per-camera-jittered rig (gate tape incl. DepthNet), forward bench with the weight-resident MLP kernel
( time timeout 1500 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04f_pytest_workloads.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r04f_pytest_workloads.log | cut -c1-900
timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --check > $O/r04f_bench_fwd.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04f_bench_fwd.json"))
print("forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; check", d.get("check"), "roofline", round(d["roofline"]["frac"], 4))
for k, v in list(d["kernels"].items())[:12]:
    print(f"  {k:28s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
;;
g)  # round 4, visit g: forward bench with / without the weight-resident Swin kernel; the switch test; training parity bounds re-check on one workload
for v in 0 1; do OCCF_SWIN_RES=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04g_bench_fwd_swinres$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04g_bench_fwd_swinres$v.json"))
print("OCCF_SWIN_RES=$v forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; swin", d["kernels"]["swin_attention_fused"], "mlp", d["kernels"]["mlp_fused"]["total_ms"])
PY
done
timeout 600 python -m pytest tests/test_attn_ops.py tests/test_switches.py tests/test_gemm_norm_ops.py -m gpu -q -p no:cacheprovider -k "swin or switch or mlp" 2>&1 | tail -3
timeout 300 python bench.py --mode forward --steps 10 --warmup 3 --check > $O/r04g_bench_fwd_check.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r04g_bench_fwd_check.json')); print('check', d['check'], d['value'])"
;;
h)  # round 4, visit h: default bench on the per-camera rig; OCCF_LAZY_LOGITS=1 training bench
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
;;
i)  # round 4, visit i: resident MLP kernel with 4 / 8 waves per workgroup (forward bench), training bench with OCCF_LAZY_LOGITS=1 and OCCF_DEPTHNET_LIB=1
for v in 4 8; do OCCF_MLP_RES_WAVES=$v timeout 300 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $O/r04i_bench_fwd_mlpw$v.json 2>/dev/null; python - <<PY
import json
d = json.load(open("gpurun_out/r04i_bench_fwd_mlpw$v.json"))
print("OCCF_MLP_RES_WAVES=$v forward", round(d["value"], 2), "samples/s", round(d["ms_per_step"], 2), "ms; mlp", d["kernels"]["mlp_fused"])
PY
done
for env in "OCCF_LAZY_LOGITS=1" "OCCF_LAZY_LOGITS=1 OCCF_DEPTHNET_LIB=1" "OCCF_LAZY_LOGITS=0"; do
  env $env timeout 600 python bench.py --no-cpu-baseline --steps 15 > $O/r04i_bench_train_tmp.json 2>/dev/null
  python - <<PY
import json
d = json.load(open("gpurun_out/r04i_bench_train_tmp.json"))
print("$env train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; mem", d["peak_memory_GiB"], "fwd", round(d["forward"]["value"], 2))
PY
done
;;
j)  # round 4, visit j: every BASELINE workload, training step and forward, on the round-4 kernels; a plain-bf16 line with its measured error; DepthNet on MIOpen vs the library's kernels for the SemanticKITTI workloads
for w in nusc_r50_200 nusc_r50_ref128 kitti_effb7_128 kitti_effb7_256lit nusc_r101; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/r04j_bench_train_$w.json 2>/dev/null
  timeout 400 python bench.py --workload $w --mode forward --steps 20 --warmup 3 --check > $O/r04j_bench_fwd_$w.json 2>/dev/null
  python - <<PY
import json
t = json.load(open("gpurun_out/r04j_bench_train_$w.json")); f = json.load(open("gpurun_out/r04j_bench_fwd_$w.json"))
print("$w train", round(t["value"], 3), "samples/s", round(t["ms_per_step"], 1), "ms, mem", t["peak_memory_GiB"], "| forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms check", f["check"]["output_voxels_max_abs_err"], "cpu", round(f["cpu_baseline"]["value"], 4))
PY
done
for w in kitti_effb7_128 kitti_effb7_256lit; do
  OCCF_DEPTHNET_LIB=0 timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline > $O/r04j_bench_train_${w}_miopen.json 2>/dev/null
  python -c "
import json; t=json.load(open('gpurun_out/r04j_bench_train_${w}_miopen.json')); print('$w DepthNet on MIOpen: train', round(t['value'],3), round(t['ms_per_step'],1))"
done
timeout 400 python bench.py --mode forward --precision bf16 --steps 20 --warmup 3 --check > $O/r04j_bench_fwd_bf16.json 2>/dev/null
timeout 900 python bench.py --precision bf16 > $O/r04j_bench_train_bf16.json 2>/dev/null
python - <<'PY'
import json
f = json.load(open("gpurun_out/r04j_bench_fwd_bf16.json")); t = json.load(open("gpurun_out/r04j_bench_train_bf16.json"))
print("bf16 forward", round(f["value"], 2), "check", f["check"], "roofline", round(f["roofline"]["frac"], 4))
print("bf16 train", round(t["value"], 3), round(t["ms_per_step"], 1), "check", {k: v for k, v in t["check"].items() if k != "what"}, "roofline", round(t["roofline"]["frac"], 4))
PY
;;
k)  # round 4, visit k: msda3d value-gradient tiles -- LDS budget / thread-count sweep (scripts/bwd_probe.py msda)
for kb in 124 140 156; do for th in 1024 768; do
  echo -n "OCCF_MSDA_LDS_KB=$kb OCCF_MSDA_TILE_THREADS=$th: "
  OCCF_MSDA_LDS_KB=$kb OCCF_MSDA_TILE_THREADS=$th timeout 120 python scripts/bwd_probe.py msda 2>/dev/null | tail -1
done; done | tee $O/r04k_msda_lds_sweep.txt
cd /tmp && export TMPDIR=/tmp
for kb in 124 156; do
OCCF_MSDA_LDS_KB=$kb timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r04k_prof$kb -- python $R/scripts/bwd_probe.py msda > /dev/null 2>&1
python $R/scripts/summarize_prof.py $R/$O/r04k_prof$kb | grep -i "msda\|kernel " | head -8 | cut -c1-160
done
;;
l)  # round 4, visit l: halo conv 128 -> 128 / 256 -> 256 with the 2 x 4 wave layout (OCCF_HALO_WN4)
for v in 0 1; do OCCF_HALO_WN4=$v timeout 200 python scripts/conv_probe.py; done 2>/dev/null | tee $O/r04l_conv_probe_wn4.txt
;;
m)  # round 4, visit m: msda value-gradient tiles with 3 / 6 channels per lane (scripts/bwd_probe.py msda + the full-size test)
for v in 3 6 12; do echo -n "OCCF_MSDA_CPL=$v: "; OCCF_MSDA_CPL=$v timeout 120 python scripts/bwd_probe.py msda 2>/dev/null | tail -1; done | tee $O/r04m_msda_cpl.txt
timeout 600 python -m pytest tests/test_full_size_gpu.py tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "msda" 2>&1 | tail -3
;;
n)  # round 4, visit n: the five training-parity tests with their figures printed, on the final tree (DepthNet's training convolutions on the library's kernels where the shape rule says so)
( time timeout 1500 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s -k training_step ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r04n_pytest_workloads_train.log
grep "training step vs oracle\|passed\|failed\|^real" $O/r04n_pytest_workloads_train.log | cut -c1-900
;;
*) echo "usage: $0 <stage a..n>"; exit 2;;
esac
