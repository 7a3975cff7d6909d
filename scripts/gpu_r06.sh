#!/bin/bash
# The GPU visits of round 6, one stage per visit:  bash scripts/gpu_r06.sh <stage>   (results under gpurun_out/, the
# files worth keeping are copied to profiles/r06/ and indexed in profiles/README.md).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
O=gpurun_out
mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
export TMPDIR=/tmp
brief() {  # brief <json> : the figures of one bench line
python - "$1" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
except Exception as e:
    print(sys.argv[1], "no json", e); sys.exit(0)
print(sys.argv[1].split("/")[-1], round(d["value"], 3), "samples/s", round(d["ms_per_step"], 2), "ms; roofline", round(d["roofline"]["frac"], 4))
if d.get("check"): print("  check", {k: v for k, v in d["check"].items() if k not in ("what", "forced_relu_gates")})
for k in ("forward", "forward_from_images", "train_from_images"):
    if d.get(k): print(" ", k, {a: d[k].get(a) for a in ("value", "ms_per_step", "stages_ms", "check", "error", "pipelined") if d[k].get(a) is not None})
for k, v in list(d["kernels"].items())[:22]:
    print(f"    {k:30s} {v['calls']:4d} {v['total_ms']:8.3f} ms")
PY
}
case "${1:-}" in
a)  # the two-product fp16 weight gradients (tests, same-visit pair), the per-family precision table
( time timeout 900 python -m pytest tests/test_bwd_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "wgrad" ) 2>&1 | grep -v "MIOpen(HIP)" | tail -6 | tee $O/r06a_pytest_wgrad.log
for v in 0 1; do
  OCCF_WGRAD_F16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06a_shapes_train_wgf16_$v.txt > $O/r06a_bench_train_wgf16_$v.json 2> $O/r06a_bench_train_wgf16_$v.err; echo "wgf16=$v rc=$?"
  brief $O/r06a_bench_train_wgf16_$v.json
done
grep "wgrad" $O/r06a_shapes_train_wgf16_0.txt | head -12; echo; grep "wgrad" $O/r06a_shapes_train_wgf16_1.txt | head -12
( time timeout 2400 python scripts/precision_probe.py None wg3 wgx cf11 cd11 cfd11 lin11 ) 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)\|UserWarning\|_grad_figures\|Consider using" > $O/r06a_precision_probe.txt
tail -14 $O/r06a_precision_probe.txt | cut -c1-400
;;
b)  # Winograd F(2,3)-along-x halo convolution: tests, probe pair, end-to-end pairs; the probe's forward column (buffers restored)
( time timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_bwd_ops.py tests/test_full_size_gpu.py -m gpu -q -p no:cacheprovider -k "wino or conv_epilogue or wgrad or conv3d" ) 2>&1 | grep -v "MIOpen(HIP)" | tail -6 | tee $O/r06b_pytest_wino.log
for v in 0 1; do OCCF_WINO=$v timeout 300 python scripts/conv_probe.py 2>&1 | grep -v "amdgpu.ids"; done | tee $O/r06b_conv_probe_wino.txt
for v in 0 1; do
  OCCF_WINO=$v timeout 400 python bench.py --mode forward --check --steps 20 --warmup 3 --shape-report $O/r06b_shapes_fwd_wino$v.txt > $O/r06b_bench_fwd_wino$v.json 2> $O/r06b_bench_fwd_wino$v.err; echo "fwd wino=$v rc=$?"
  brief $O/r06b_bench_fwd_wino$v.json
  python -c "
import json; d=json.load(open('$O/r06b_bench_fwd_wino$v.json')); print('  check', d.get('check'), 'stages', d.get('stages_ms'))"
  OCCF_WINO=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06b_shapes_train_wino$v.txt > $O/r06b_bench_train_wino$v.json 2> $O/r06b_bench_train_wino$v.err; echo "train wino=$v rc=$?"
  brief $O/r06b_bench_train_wino$v.json
done
( time timeout 1800 python scripts/precision_probe.py None cf11 cd11 lin11 ) 2>&1 | grep -v "amdgpu.ids\|MIOpen(HIP)\|UserWarning\|_grad_figures\|Consider using" > $O/r06b_precision_probe.txt
tail -8 $O/r06b_precision_probe.txt | cut -c1-400
;;
c)  # data gradients on the two-product Winograd kernel: tests, same-visit pair, the default bench with its check, the training-parity gates
( time timeout 900 python -m pytest tests/test_gemm_norm_ops.py tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "wino or wgrad" ) 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | tee $O/r06c_pytest_wino_f16.log
for v in 0 1; do
  OCCF_DGRAD_F16=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06c_shapes_train_dgf16_$v.txt > $O/r06c_bench_train_dgf16_$v.json 2> $O/r06c_bench_train_dgf16_$v.err; echo "train dgrad_f16=$v rc=$?"
  brief $O/r06c_bench_train_dgf16_$v.json
done
( time timeout 1500 python -m pytest tests/test_train_step.py tests/test_train_multistep.py -m gpu -q -p no:cacheprovider -s -k "training_step or three_fused" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r06c_pytest_gates.log
grep -i "gated differently\|whole gradient\|passed\|failed\|error\|^real" $O/r06c_pytest_gates.log | cut -c1-400
( time timeout 900 python bench.py --shape-report $O/r06c_shapes_train.txt ) > $O/r06c_bench_train.json 2> $O/r06c_bench_train.err; echo "bench rc=$?"
tail -3 $O/r06c_bench_train.err
brief $O/r06c_bench_train.json
;;
d)  # decoder rows kernels: tests, forward pair, forward kernel statistics
( time timeout 900 python -m pytest tests/test_attn_ops.py tests/test_switches.py tests/test_modules_golden.py -m gpu -q -p no:cacheprovider -k "decoder_rows or every_switch or head" ) 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | tee $O/r06d_pytest_decoder_rows.log
for v in 0 1; do
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --check --steps 30 --warmup 3 --no-cpu-baseline > $O/r06d_bench_fwd_rows$v.json 2> $O/r06d_bench_fwd_rows$v.err; echo "fwd rows=$v rc=$?"
  brief $O/r06d_bench_fwd_rows$v.json
  python -c "
import json; d=json.load(open('$O/r06d_bench_fwd_rows$v.json')); print('  stages', d.get('stages_ms'))"
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --from-images --steps 30 --warmup 3 > $O/r06d_bench_fwd_from_images_rows$v.json 2> $O/r06d_bench_fwd_from_images_rows$v.err; echo "fwd from images rows=$v rc=$?"
  python -c "
import json; d=json.load(open('$O/r06d_bench_fwd_from_images_rows$v.json')); print('  from images', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms', d.get('stages_ms'))"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06d_prof_fwd -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/r06d_prof_fwd.log 2>&1 ; echo "rocprof fwd rc=$?"
cd $R
python scripts/summarize_prof.py $O/r06d_prof_fwd > $O/r06d_fwd_kernel_stats.txt 2>&1 ; head -40 $O/r06d_fwd_kernel_stats.txt | cut -c1-150
find $O/r06d_prof_fwd -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
;;
e)  # decoder rows with the fragment ring; reproducibility of the training step; linear_wgrad shapes
( time timeout 900 python -m pytest tests/test_attn_ops.py tests/test_bwd_ops.py tests/test_lss_ops.py tests/test_image_backbone.py -m gpu -q -p no:cacheprovider -k "decoder_rows or point_sample or lift_splat or deform or dcn" ) 2>&1 | grep -v "MIOpen(HIP)" | tail -4 | tee $O/r06e_pytest.log
for v in 0 1; do
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --check --steps 30 --warmup 3 --no-cpu-baseline > $O/r06e_bench_fwd_rows$v.json 2> $O/r06e_bench_fwd_rows$v.err; echo "fwd rows=$v rc=$?"
  python -c "
import json; d=json.load(open('$O/r06e_bench_fwd_rows$v.json')); print('  fwd', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms', d.get('stages_ms'), {k: (v['calls'], v['total_ms']) for k, v in d['kernels'].items() if 'decoder' in k or k in ('linear', 'layernorm')})"
done
( time timeout 600 python -m pytest tests/test_train_step.py -m gpu -q -p no:cacheprovider -s -k "reproducible" ) 2>&1 | grep -v "MIOpen(HIP)" | grep "reproducibility\|passed\|failed\|Error" | cut -c1-1500 | tee $O/r06e_reproducible.log
OCCF_DETERMINISTIC=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06e_bench_train_det.json 2> $O/r06e_bench_train_det.err; echo "train det rc=$?"
brief $O/r06e_bench_train_det.json | head -3
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06e_shapes_train.txt > $O/r06e_bench_train.json 2> $O/r06e_bench_train.err; echo "train rc=$?"
brief $O/r06e_bench_train.json | head -3
grep "linear_wgrad" $O/r06e_shapes_train.txt | head -24
;;
f)  # decoder rows on one XCD with cached launch arguments: host issue time vs GPU time, forward pairs; reproducibility
for v in 0 1; do OCCF_DECODER_ROWS=$v timeout 300 python scripts/issue_probe.py 2>&1 | grep "per forward"; done | tee $O/r06f_issue_probe.txt
for v in 0 1; do
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --check --steps 30 --warmup 3 --no-cpu-baseline > $O/r06f_bench_fwd_rows$v.json 2> $O/r06f_bench_fwd_rows$v.err; echo "fwd rows=$v rc=$?"
  python -c "
import json; d=json.load(open('$O/r06f_bench_fwd_rows$v.json')); print('  fwd', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms', d.get('check'), d.get('stages_ms'), {k: (v['calls'], v['total_ms']) for k, v in d['kernels'].items() if 'decoder' in k or k in ('linear', 'layernorm')})"
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --from-images --steps 30 --warmup 3 > $O/r06f_bench_fwd_from_images_rows$v.json 2> $O/r06f_bench_fwd_from_images_rows$v.err
  python -c "
import json; d=json.load(open('$O/r06f_bench_fwd_from_images_rows$v.json')); print('  from images', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms', d.get('stages_ms'))"
done
( time timeout 600 python -m pytest tests/test_train_step.py tests/test_attn_ops.py -m gpu -q -p no:cacheprovider -s -k "reproducible or decoder_rows" ) 2>&1 | grep -v "MIOpen(HIP)" | grep "reproducibility\|passed\|failed\|Error" | cut -c1-1500 | tee $O/r06f_reproducible.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06f_shapes_train.txt > $O/r06f_bench_train.json 2> $O/r06f_bench_train.err; echo "train rc=$?"
brief $O/r06f_bench_train.json | head -12
;;
g)  # (also run as visit h with the branch-free ring cycle) decoder rows with the flat weight ring + L2 touch: forward pairs and per-kernel statistics of both launch structures
timeout 300 python -m pytest tests/test_attn_ops.py -m gpu -q -p no:cacheprovider -k "decoder_rows" 2>&1 | tail -2
for v in 0 1; do
  OCCF_DECODER_ROWS=$v timeout 400 python bench.py --mode forward --check --steps 30 --warmup 3 --no-cpu-baseline > $O/r06g_bench_fwd_rows$v.json 2> $O/r06g_bench_fwd_rows$v.err; echo "fwd rows=$v rc=$?"
  python -c "
import json; d=json.load(open('$O/r06g_bench_fwd_rows$v.json')); print('  fwd', round(d['value'],2), 'samples/s', round(d['ms_per_step'],2), 'ms', d.get('check'), d.get('stages_ms'), {k: (v['calls'], v['total_ms']) for k, v in d['kernels'].items() if 'decoder' in k or k in ('linear', 'layernorm')})"
done
cd /tmp
for v in 0 1; do
  OCCF_DECODER_ROWS=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/r06g_prof_fwd_rows$v -- python $R/bench.py --mode forward --steps 10 --warmup 2 --no-cpu-baseline > $R/$O/r06g_prof_fwd_rows$v.log 2>&1 ; echo "rocprof fwd rows=$v rc=$?"
  ( cd $R; python scripts/summarize_prof.py $O/r06g_prof_fwd_rows$v > $O/r06g_fwd_kernel_stats_rows$v.txt 2>&1 )
  find $R/$O/r06g_prof_fwd_rows$v -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
done
cd $R
python - <<'PY'
import re
def load(p):
    d = {}
    for line in open(p):
        m = re.match(r"(.{1,92}?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$", line.rstrip())
        if m:
            d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
a, b = load("gpurun_out/r06g_fwd_kernel_stats_rows0.txt"), load("gpurun_out/r06g_fwd_kernel_stats_rows1.txt")
print("total ms (13 forwards incl. warmup... same count both):", round(sum(v[1] for v in a.values()), 2), round(sum(v[1] for v in b.values()), 2))
diff = sorted(((b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k, a.get(k, (0, 0)), b.get(k, (0, 0))) for k in set(a) | set(b)), key=lambda t: -abs(t[0]))
for dlt, k, x, y in diff[:14]:
    print(f"{dlt:+9.3f} ms  {k[:70]:70s} rows0 {x}  rows1 {y}")
PY
;;
j)  # reproducibility gate, counters of the msda value-gradient tiles, the conv family's counters through the new summarizer, the default bench with its check
( time timeout 600 python -m pytest tests/test_train_step.py -m gpu -q -p no:cacheprovider -s -k "reproducible" ) 2>&1 | grep -v "MIOpen(HIP)" | grep "reproducibility\|passed\|failed\|Error" | cut -c1-1200 | tee $O/r06j_reproducible.log
bash scripts/pmc_probe.sh r06j_msda_pmc msda3d_bwd python $R/scripts/bwd_probe.py msda > $O/r06j_msda_pmc.log 2>&1; tail -60 $O/r06j_msda_pmc.log | grep -v "^pmc pass" | cut -c1-160
P=$R/$O/r06j_conv_pmc; mkdir -p $P; cd /tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $P/pmc_$i -- python $R/scripts/conv_family_probe.py 2 > $P/pmc_$i.log 2>&1 ; echo "pmc pass $i ($set) rc=$?"
done
cd $R
python scripts/summarize_pmc.py $P $P/pmc_traffic.json > $O/r06j_conv_pmc_summary.txt 2>&1 ; head -14 $O/r06j_conv_pmc_summary.txt | cut -c1-190
cp $P/pmc_traffic.json $O/r06j_pmc_traffic.json
find $P -name "*.csv" -size +2M -delete 2>/dev/null
( time timeout 900 python bench.py --shape-report $O/r06j_shapes_train.txt ) > $O/r06j_bench_train.json 2> $O/r06j_bench_train.err; echo "bench rc=$?"
tail -3 $O/r06j_bench_train.err
brief $O/r06j_bench_train.json | head -16
python -c "
import json; d=json.load(open('$O/r06j_bench_train.json')); print(json.dumps(d['roofline'], indent=0)[:3000])"
;;
k)  # the BEV ASPP's maps on the gate tape (level heavy+bev): full-size training parity + the default bench's check
( time timeout 1200 python -m pytest tests/test_workloads_gpu.py -m gpu -q -p no:cacheprovider -s -k "training_step and (nusc_r50_200 or kitti_effb7_128)" ) 2>&1 | grep -v "MIOpen(HIP)" > $O/r06k_workloads_train.log
grep "training step vs oracle\|passed\|failed\|^real\|Error" $O/r06k_workloads_train.log | cut -c1-1500
( time timeout 900 python bench.py --shape-report $O/r06k_shapes_train.txt ) > $O/r06k_bench_train.json 2> $O/r06k_bench_train.err; echo "bench rc=$?"
tail -3 $O/r06k_bench_train.err
brief $O/r06k_bench_train.json | head -8
;;
l)  # the whole GPU suite on the current tree + the one-pass small-M weight gradient + the msda tile padding pair
for v in 0 1; do OCCF_MSDA_PAD=$v timeout 300 python scripts/bwd_probe.py msda 2>&1 | grep "msda3d_backward" | sed "s/^/OCCF_MSDA_PAD=$v: /"; done | tee $O/r06l_msda_pad.txt
( time timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider ) > $O/r06l_pytest_gpu.log 2>&1; echo "pytest rc=$?"; grep -v "MIOpen(HIP)" $O/r06l_pytest_gpu.log | tail -12 | cut -c1-300
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --shape-report $O/r06l_shapes_train.txt > $O/r06l_bench_train.json 2> $O/r06l_bench_train.err; echo "train rc=$?"
brief $O/r06l_bench_train.json | head -30
grep "linear_wgrad x(100" $O/r06l_shapes_train.txt
;;
m)  # msda tile padding with the scaled LDS budget; staggered staging taps of the Winograd kernel
for v in 0 1; do OCCF_MSDA_PAD=$v timeout 300 python scripts/bwd_probe.py msda 2>&1 | grep "msda3d_backward" | sed "s/^/OCCF_MSDA_PAD=$v: /"; done | tee $O/r06m_msda_pad.txt
for v in 0 1; do OCCF_WINO_STAGGER=$v timeout 300 python scripts/conv_probe.py 2>&1 | grep -v "amdgpu.ids"; done | tee $O/r06m_conv_probe_stagger.txt
timeout 600 python -m pytest tests/test_gemm_norm_ops.py tests/test_bwd_ops.py -m gpu -q -p no:cacheprovider -k "wino or msda" 2>&1 | tail -2
for v in 0 1; do
  OCCF_WINO_STAGGER=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06m_bench_train_stagger$v.json 2> $O/r06m_bench_train_stagger$v.err; echo "train stagger=$v rc=$?"
  brief $O/r06m_bench_train_stagger$v.json | head -9
done
;;
n)  # Winograd kernel with the weight fragments two k-steps ahead (128-channel variant); msda padding off again
timeout 300 python scripts/conv_probe.py 2>&1 | grep -v "amdgpu.ids" | tee $O/r06n_conv_probe_bd3.txt
timeout 600 python -m pytest tests/test_gemm_norm_ops.py -m gpu -q -p no:cacheprovider -k "wino or conv_epilogue" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/r06n_bench_train.json 2> $O/r06n_bench_train.err; echo "train rc=$?"
brief $O/r06n_bench_train.json | head -12
;;
o)  # DepthNet's 3x3 convolution on the generic kernel: split-K / tile-width sweep
timeout 120 python scripts/depthnet_conv_probe.py 2>&1 | grep "conv 3x3" | tee $O/r06o_depthnet_conv_probe.txt
for k in 1 2 3 4 5 6 8; do OCCF_GEMM_KSPLIT=$k timeout 120 python scripts/depthnet_conv_probe.py 2>&1 | grep "conv 3x3"; done | tee -a $O/r06o_depthnet_conv_probe.txt
for k in 1 2 3 4; do OCCF_GEMM_BN=64 OCCF_GEMM_KSPLIT=$k timeout 120 python scripts/depthnet_conv_probe.py 2>&1 | grep "conv 3x3"; done | tee -a $O/r06o_depthnet_conv_probe.txt
;;
p)  # DDP bucket timeline on one rank (nccl); DepthNet conv sweep
OCCF_DIST_AT_WORLD_1=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29543 scripts/ddp_bucket_timeline.py 2>&1 | grep -v "MIOpen\|amdgpu.ids\|Warning\|warn" | tail -14 | tee $O/r06p_ddp_bucket_timeline.txt
bash scripts/gpu_r06.sh o
;;
*) echo "unknown stage"; exit 2;;
esac
