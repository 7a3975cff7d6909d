#!/bin/bash
# The GPU visits of round 5, one stage per visit:  bash scripts/gpu_r05.sh <stage>   (results under gpurun_out/, the
# files worth keeping are copied to profiles/r05/ and indexed in profiles/README.md).  scripts/gpu_final.sh is the
# end-of-round visit (suite, smoke, bench, rocprof statistics, PMC traffic stamp).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
case "${1:-}" in
a)  # round 5, visit a: the gradient gates on the full gate tape (tiny configs, multistep, smoke at 1e-3), the ungated figure, default bench with forward_from_images, from-images training step + the other backbones, one-rank RCCL run
( time timeout 900 python -m pytest tests/test_train_step.py tests/test_train_multistep.py tests/test_switches.py tests/test_image_backbone.py -m gpu -q -p no:cacheprovider -s -k "training_step or three_fused or training_switches or dcnv2" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r05a_pytest_gates.log
grep -i "gated differently\|whole gradient\|passed\|failed\|error\|^real" $O/r05a_pytest_gates.log | cut -c1-400
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | tee $O/r05a_smoke.log
( time timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s -k "training_step and nusc_r50_200" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r05a_workload_nusc200.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r05a_workload_nusc200.log | cut -c1-1200
( time timeout 900 python bench.py --shape-report $O/r05a_shapes_train.txt ) > $O/r05a_bench_train.json 2> $O/r05a_bench_train.err; echo "bench rc=$?"
tail -3 $O/r05a_bench_train.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05a_bench_train.json"))
print("train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", round(d["roofline"]["frac"], 4))
print("check", {k: v for k, v in d.get("check", {}).items() if k != "what"})
print("forward", round(d["forward"]["value"], 2), d["forward"]["roofline"]["frac"], d["forward"].get("check"))
print("forward_from_images", d.get("forward_from_images"))
print("cpu", d["cpu_baseline"])
for k, v in list(d["kernels"].items())[:30]:
    print(f"  {k:30s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
head -40 $O/r05a_shapes_train.txt
timeout 600 python bench.py --from-images --steps 10 --warmup 3 > $O/r05a_bench_train_from_images.json 2> $O/r05a_bench_train_from_images.err; echo "rc=$?"; tail -2 $O/r05a_bench_train_from_images.err
for w in kitti_effb7_128 nusc_r101 kitti_effb7_256lit; do
  timeout 600 python bench.py --workload $w --mode forward --from-images --steps 20 --warmup 3 > $O/r05a_bench_fwd_from_images_$w.json 2> $O/r05a_bench_fwd_from_images_$w.err; echo "$w rc=$?"; tail -2 $O/r05a_bench_fwd_from_images_$w.err
done
python - <<'PY'
import json
for f in ("train_from_images", "fwd_from_images_kitti_effb7_128", "fwd_from_images_nusc_r101", "fwd_from_images_kitti_effb7_256lit"):
    try:
        d = json.load(open(f"gpurun_out/r05a_bench_{f}.json"))
        print(f, round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms", d.get("stages_ms"), "mem", d["peak_memory_GiB"])
    except Exception as e:
        print(f, "no json", e)
PY
OCCF_DIST_AT_WORLD_1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --sync-bn > $O/r05a_bench_rccl_world1.json 2> $O/r05a_bench_rccl_world1.err; echo "rccl world-1 rc=$?"
tail -3 $O/r05a_bench_rccl_world1.err
python -c "
import json; d=json.load(open('gpurun_out/r05a_bench_rccl_world1.json')); print('rccl world 1:', round(d['value'],3), 'samples/s', round(d['ms_per_step'],2), 'ms', d['config']['parallelism'])"
;;
b)  # round 5, visit b: the streaming linear kernel (shape table vs the tile kernel and the floors, its GPU test, end-to-end benches), the precision table
timeout 600 python scripts/stream_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $O/r05b_stream_probe.txt
timeout 600 python -m pytest tests/test_gemm_norm_ops.py -m gpu -q -p no:cacheprovider -k "streaming or linear" 2>&1 | tail -3
for v in 0 1; do
  OCCF_GEMM_STREAM=$v timeout 400 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline --shape-report $O/r05b_shapes_fwd_stream$v.txt > $O/r05b_bench_fwd_stream$v.json 2>/dev/null
  OCCF_GEMM_STREAM=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r05b_shapes_train_stream$v.txt > $O/r05b_bench_train_stream$v.json 2>/dev/null
  python - <<PY
import json
f = json.load(open("gpurun_out/r05b_bench_fwd_stream$v.json")); t = json.load(open("gpurun_out/r05b_bench_train_stream$v.json"))
print("OCCF_GEMM_STREAM=$v forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms, linear", f["kernels"]["linear"]["total_ms"], "| train", round(t["value"], 3), round(t["ms_per_step"], 2), "ms, linear", t["kernels"]["linear"]["total_ms"], "| train-bench forward", round(t["forward"]["value"], 2))
PY
done
( time timeout 1500 python scripts/precision_probe.py ) 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)" > $O/r05b_precision_probe.txt
tail -12 $O/r05b_precision_probe.txt | cut -c1-400
;;
c)  # round 5, visit c: streaming linear with LDS-DMA staging + next-tile prefetch; fused inference route of the image branch (from-images forward with / without); 2-term fp16 in the weight gradients only
timeout 600 python scripts/stream_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $O/r05c_stream_probe.txt
timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_image_backbone.py -m gpu -q -p no:cacheprovider -k "streaming or linear or image or dcn or resnet or scale_shift" 2>&1 | grep -v "MIOpen(HIP)" | tail -8
for v in 0 1; do
  OCCF_IMAGE_FUSE=$v timeout 400 python bench.py --mode forward --from-images --steps 20 --warmup 3 > $O/r05c_bench_fwd_from_images_fuse$v.json 2>/dev/null
  python - <<PY
import json
f = json.load(open("gpurun_out/r05c_bench_fwd_from_images_fuse$v.json"))
print("OCCF_IMAGE_FUSE=$v forward from images", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms", f["stages_ms"])
PY
done
timeout 400 python bench.py --workload nusc_r101 --mode forward --from-images --steps 20 --warmup 3 > $O/r05c_bench_fwd_from_images_nusc_r101.json 2>/dev/null
python -c "
import json; f=json.load(open('gpurun_out/r05c_bench_fwd_from_images_nusc_r101.json')); print('nusc_r101 from images', round(f['value'],2), round(f['ms_per_step'],2), f['stages_ms'])"
timeout 400 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline --shape-report $O/r05c_shapes_fwd.txt > $O/r05c_bench_fwd.json 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r05c_shapes_train.txt > $O/r05c_bench_train.json 2>/dev/null
python - <<PY
import json
f = json.load(open("gpurun_out/r05c_bench_fwd.json")); t = json.load(open("gpurun_out/r05c_bench_train.json"))
print("forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms, linear", f["kernels"]["linear"]["total_ms"], "| train", round(t["value"], 3), round(t["ms_per_step"], 2), "ms, linear", t["kernels"]["linear"]["total_ms"], "| train-bench forward", round(t["forward"]["value"], 2), "from images", round(t["forward_from_images"]["value"], 2))
PY
( time timeout 900 python scripts/precision_probe.py wg11 ) 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)" > $O/r05c_precision_probe_wg11.txt
tail -6 $O/r05c_precision_probe_wg11.txt | cut -c1-400
;;
d)  # round 5, visit d: access-pattern micro-probe (row-per-lane vs coalesced), kernel statistics of the from-images forward with / without the fused image route
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/row_access_probe.hip -o /tmp/rap 2>/dev/null && /tmp/rap | tee $O/r05d_row_access_probe.txt
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  OCCF_IMAGE_FUSE=$v timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r05d_prof_img$v -- python $R/bench.py --mode forward --from-images --steps 10 --warmup 3 > /dev/null 2>&1
  python $R/scripts/summarize_prof.py $R/$O/r05d_prof_img$v > $R/$O/r05d_fwd_from_images_fuse${v}_kernel_stats.txt 2>&1
  echo "== OCCF_IMAGE_FUSE=$v"; grep -i "miopen\|conv\|batch_norm\|elementwise\|scale_shift\|nchw\|nhwc\|transpose\|copy\|Cijk\|igemm\|naive" $R/$O/r05d_fwd_from_images_fuse${v}_kernel_stats.txt | head -40 | cut -c1-170
done
cd $R
find $O -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
;;
e)  # round 5, visit e: streaming linear with whole-line stores (token-row accumulator layout); image epilogue without the 64-bit modulo
timeout 600 python scripts/stream_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $O/r05e_stream_probe.txt
timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_image_backbone.py -m gpu -q -p no:cacheprovider -k "streaming or linear or resnet or scale_shift" 2>&1 | grep -v "MIOpen(HIP)" | tail -3
for v in 0 1; do
  OCCF_IMAGE_FUSE=$v timeout 400 python bench.py --mode forward --from-images --steps 20 --warmup 3 > $O/r05e_bench_fwd_from_images_fuse$v.json 2>/dev/null
  python - <<PY
import json
f = json.load(open("gpurun_out/r05e_bench_fwd_from_images_fuse$v.json"))
print("OCCF_IMAGE_FUSE=$v forward from images", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms", f["stages_ms"])
PY
done
timeout 400 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline --shape-report $O/r05e_shapes_fwd.txt > $O/r05e_bench_fwd.json 2>/dev/null
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r05e_shapes_train.txt > $O/r05e_bench_train.json 2>/dev/null
python - <<PY
import json
f = json.load(open("gpurun_out/r05e_bench_fwd.json")); t = json.load(open("gpurun_out/r05e_bench_train.json"))
print("forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms, linear", f["kernels"]["linear"]["total_ms"], "| train", round(t["value"], 3), round(t["ms_per_step"], 2), "ms, linear", t["kernels"]["linear"]["total_ms"], "| train-bench forward", round(t["forward"]["value"], 2), "from images", round(t["forward_from_images"]["value"], 2))
PY
;;
f)  # round 5, visit f: access-pattern probe with the dword whole-line stores and the LDS-transposed float4 stores
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/row_access_probe.hip -o /tmp/rap 2>/dev/null && /tmp/rap | tee $O/r05f_row_access_probe.txt
;;
g)  # round 5, visit g: the Swin block's DropPath branches on the streaming kernel's epilogues (training step with / without), their GPU test, full-size training parity with them
timeout 600 python -m pytest tests/test_bwd_ops.py tests/test_gemm_norm_ops.py -m gpu -q -p no:cacheprovider -k "swin_block_fused or streaming" 2>&1 | tail -3
for v in 0 1; do
  OCCF_TRAIN_SWIN_FUSE=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05g_bench_train_swinfuse$v.json 2>/dev/null
  python - <<PY
import json
t = json.load(open("gpurun_out/r05g_bench_train_swinfuse$v.json"))
k = t["kernels"]
print("OCCF_TRAIN_SWIN_FUSE=$v train", round(t["value"], 3), "samples/s", round(t["ms_per_step"], 2), "ms | linear", k["linear"]["total_ms"], "linear_stream", k.get("linear_stream", {}).get("total_ms"), "droppath", k["droppath"]["total_ms"], "act_fwd", k.get("act_forward", {}).get("total_ms"), "act_bwd", k["act_backward"]["total_ms"], "mem", t["peak_memory_GiB"])
PY
done
( time timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s -k "training_step and (nusc_r50_200 or kitti_effb7_128)" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r05g_workloads_train.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r05g_workloads_train.log | cut -c1-1300
;;
h)  # round 5, visit h: default bench (pipelined from-images record, 32-bit index arithmetic in droppath / GroupNorm backward), GN / droppath GPU tests
timeout 600 python -m pytest tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "groupnorm or droppath" 2>&1 | tail -2
( time timeout 900 python bench.py --shape-report $O/r05h_shapes_train.txt ) > $O/r05h_bench_train.json 2> $O/r05h_bench_train.err; echo "bench rc=$?"
tail -3 $O/r05h_bench_train.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05h_bench_train.json"))
print("train", round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", round(d["roofline"]["frac"], 4))
print("check", {k: v for k, v in d.get("check", {}).items() if k not in ("what", "per_parameter_rel_l2_quantiles", "per_parameter_rel_l2_quantiles_ungated")})
print("forward", round(d["forward"]["value"], 2), d["forward"]["roofline"]["frac"], d["forward"].get("check"))
print("forward_from_images", {k: v for k, v in d["forward_from_images"].items() if k != "metric"})
for k, v in list(d["kernels"].items())[:24]:
    print(f"  {k:30s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
;;
i)  # round 5, visit i: the streaming normalisation / elementwise passes at full resolution with the bytes they move (GroupNorm backward apply: thread-per-float4 vs row-looping form)
timeout 600 python scripts/bwd_probe.py gn 2>&1 | grep -v "amdgpu.ids" | tee $O/r05i_elementwise_probe.txt
timeout 600 python -m pytest tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "groupnorm" 2>&1 | tail -2
;;
j)  # round 5, visit j: halo conv on half-size tiles (two workgroups per CU) vs the 256-voxel tiles: probe, GPU tests, forward bench
for v in 0 1; do OCCF_HALO_SMALL=$v timeout 300 python scripts/conv_probe.py 2>/dev/null; done | tee $O/r05j_conv_probe_small.txt
timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "halo or conv or groupnorm_stats" 2>&1 | tail -3
for v in 0 1; do
  OCCF_HALO_SMALL=$v timeout 400 python bench.py --mode forward --steps 20 --warmup 3 --check > $O/r05j_bench_fwd_small$v.json 2>/dev/null
  python - <<PY
import json
f = json.load(open("gpurun_out/r05j_bench_fwd_small$v.json"))
print("OCCF_HALO_SMALL=$v forward", round(f["value"], 2), "samples/s", round(f["ms_per_step"], 2), "ms, conv3d", f["kernels"]["conv3d"]["total_ms"], "roofline", round(f["roofline"]["frac"], 4), f["roofline"]["kernel"][:60], "check", f["check"])
PY
done
;;
k)  # round 5, visit k: the full GPU suite + smoke on the current tree
( time timeout 1700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=10 ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r05k_pytest_gpu.log
tail -22 $O/r05k_pytest_gpu.log | cut -c1-300
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | tee $O/r05k_smoke.log
;;
l)  # round 5, visit l: counters of the streaming linear against the tile kernel (SQ wait / issue / MFMA counters, FETCH_SIZE, WRITE_SIZE: separate passes)
cd /tmp && export TMPDIR=/tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $c | cut -c1-8 | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$O/r05l_pmc_$tag -- python $R/scripts/stream_probe.py quick > /dev/null 2>&1
done
cd $R
python - <<'PY' | tee gpurun_out/r05l_stream_pmc.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/r05l_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "gemm_stream_kernel" not in k and "gemm_bf16_kernel" not in k: continue
        key = (k[:48], r.get("Grid_Size"))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
print("per launch; FETCH_SIZE / WRITE_SIZE in KiB (HBM bytes: FETCH_SIZE x 2 x 1024 -- the gfx950 correction of MI355X_MICROARCH.md, as scripts/summarize_pmc.py -- and WRITE_SIZE x 1024)")
for key, d in sorted(agg.items()):
    print(key, {c: round(v / n[(key, c)]) for c, v in sorted(d.items())})
PY
find $O -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
;;
m)  # round 5, visit m: the tiny gradient gates with 2 048 sampling points, three repetitions; kitti_effb7_128 full-size training parity; smoke
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_train_multistep.py tests/test_train_step.py -m gpu -q -p no:cacheprovider -s -k "three_fused or training_step" 2>&1 | grep -i "whole gradient\|passed\|failed" | cut -c1-160
done | tee $O/r05m_tiny_gates_x3.log
timeout 300 python __graft_entry__.py --smoke 2>&1 | grep -v "MIOpen(HIP)" | tail -3 | tee $O/r05m_smoke.log
timeout 900 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s -k "training_step and kitti_effb7_128" 2>&1 | grep "training step vs oracle\|passed\|failed" | cut -c1-700 | tee $O/r05m_kitti128.log
;;
n)  # round 5, visit n: the head loss's importance sampling, channel-major vs voxel-major logits
timeout 300 python scripts/bwd_probe.py psample 2>&1 | grep -v "amdgpu.ids" | tee $O/r05n_psample_probe.txt
;;
o)  # round 5, visit o: head loss with the voxel-major candidate logits: training bench + full-size parity of two workloads
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r05o_bench_train.json 2>/dev/null
python - <<'PY'
import json
t = json.load(open("gpurun_out/r05o_bench_train.json"))
k = t["kernels"]
print("train", round(t["value"], 3), "samples/s", round(t["ms_per_step"], 2), "ms | point_sample_3d", k["point_sample_3d"], "point_sample_tokens", k.get("point_sample_tokens"), "linear", k["linear"]["total_ms"], "groupnorm_backward", k["groupnorm_backward"]["total_ms"])
PY
( time timeout 900 python -m pytest tests/test_workloads_gpu.py tests/test_training.py -m gpu -q -p no:cacheprovider -s -k "(training_step and (nusc_r50_200 or nusc_r101)) or test_training" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r05o_workloads_train.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r05o_workloads_train.log | cut -c1-600
;;
q)  # round 5, visit q: FETCH_SIZE / WRITE_SIZE of the roofline kernel (halo conv 192 -> 192 at 200x200x16) on the final kernel sources, from scripts/conv_probe.py (the bench under --pmc outgrew its time limit in r05z: it now also builds and runs the from-images detector)
Q=$R/$O/r05zz
mkdir -p $Q
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 170 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $Q/pmc_$c -- python $R/scripts/conv_probe.py 2 > $Q/pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
cd $R
python scripts/summarize_pmc.py $Q $Q/pmc_traffic.json > $Q/pmc_summary.txt 2>&1; head -8 $Q/pmc_summary.txt | cut -c1-160
find $Q -name "*counter_collection.csv" -size +20M -delete 2>/dev/null
timeout 200 python -m pytest tests/test_gemm_norm_ops.py tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -k "halo or groupnorm_stats_from or (end_to_end and nusc_r50_200)" 2>&1 | tail -2
timeout 200 python bench.py --mode forward --steps 20 --warmup 3 --no-cpu-baseline > $Q/bench_fwd.json 2>/dev/null
python -c "
import json; f=json.load(open('$Q/bench_fwd.json')); print('forward', round(f['value'],2), 'samples/s', round(f['ms_per_step'],2), 'ms; roofline', round(f['roofline']['frac'],4), 'traffic', f['roofline']['traffic'], f['roofline']['traffic_source'], f['roofline']['kernel'][:52])"
;;
s)  # round 5, visit s: msda value-gradient tiles with / without the gather pass's per-sample records
for v in 0 1; do echo -n "OCCF_MSDA_RECORDS=$v: "; OCCF_MSDA_RECORDS=$v timeout 120 python scripts/bwd_probe.py msda 2>/dev/null | tail -1; done | tee $O/r05s_msda_records.txt
timeout 300 python -m pytest tests/test_bwd_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "msda" 2>&1 | tail -2
;;
*) echo "usage: $0 <stage>"; exit 2;;
esac
