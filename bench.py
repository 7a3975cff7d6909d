#!/usr/bin/env python
"""bench.py -- the OccFormer hot path on MI355X.

One "step" (default, ``--mode train`` = BASELINE.json's fwd+bwd metric) = one TRAINING step of the hot path over one
synthetic nuScenes sample (6 cameras): image-neck features [1,6,512,16,44] + camera calibration + sparse LiDAR depth
+ occupancy ground truth [200,200,16] + 34 720 labelled LiDAR points -> ``OccupancyFormer.forward_train`` (LSS voxel
pooling -> dual-path 3-D encoder -> 3-D deformable pixel decoder -> Mask2Former occupancy decoder -> Hungarian targets,
point-sampled losses, depth loss) -> ``sum(losses).backward()`` on the library's backward kernels -> (DDP gradient
all-reduce over RCCL for N > 1) -> grad-clip -> fused AdamW, on BASELINE.json's 200x200x16 grid.  ``--mode forward``
times the inference hot path (`simple_test`: occupancy volume [1,17,400,400,32] + lidarseg points).  Weights are
random (seeded), data synthetic.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant hand-written kernel, HIP-event
timed on the launching stream; `traffic` from the committed rocprofv3 --pmc summary of the same kernel sources) and
`cpu_baseline` (the CPU oracle = restated reference path, timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA


def synthetic_sample(meta, device, seed=0):
    from occformer_amd import configs
    return configs.synthetic_sample(meta, device, seed)


class KernelCensus:
    """Times every C-ABI op call of ONE step with HIP events recorded on the launching
    (current) stream and counts its algorithmic bytes = bytes of every tensor argument and
    result, each counted once (what an ideal kernel must move)."""

    def __init__(self, ops, only=None):
        self.ops = ops
        self.only = only
        self.records = {}
        self.shapes = {}
        self._orig = {}

    def __enter__(self):
        names = [n for n in dir(type(self.ops)) if not n.startswith("_") and callable(getattr(self.ops, n))]
        if self.only:
            names = [n for n in names if n in self.only]
        for name in names:
            fn = getattr(self.ops, name)
            self._orig[name] = fn

            def wrapped(*a, _fn=fn, _name=name, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.ops.last_flops = 0
                e0.record()
                out = _fn(*a, **kw)
                e1.record()
                outs = out if isinstance(out, (tuple, list)) else (out,)
                tensors = [t for t in list(a) + list(kw.values()) + list(outs) if torch.is_tensor(t)]
                nbytes = sum(t.numel() * t.element_size() for t in {t.data_ptr(): t for t in tensors}.values())
                self.records.setdefault(_name, []).append((e0, e1, nbytes, self.ops.last_flops))
                if _name in ("linear", "conv3d", "conv3d_wgrad", "conv3d_dgrad", "linear_wgrad", "linear_stream"):
                    second = a[1][0] if isinstance(a[1], (tuple, list)) and torch.is_tensor(a[1][0]) else a[1]
                    shp = (tuple(a[0].shape), tuple(second.shape) if torch.is_tensor(second) else tuple(a[2]))
                    # (the data gradients of the stride-1 3^3 convolutions ask for the two-product kernel: their own row)
                    tag = _name + ("[dy in one fp16 piece]" if _name == "conv3d" and kw.get("act_f16") and
                                   getattr(self.ops, "use_wino", False) else "")
                    self.shapes.setdefault((tag, shp), []).append((e0, e1, self.ops.last_flops))
                return out

            setattr(self.ops, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, fn in self._orig.items():
            setattr(self.ops, name, fn)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = [r[0].elapsed_time(r[1]) for r in recs]
            out[name] = dict(calls=len(recs), total_ms=sum(ms), avg_ms=sum(ms) / len(ms),
                             bytes_per_call=sum(r[2] for r in recs) / len(recs),
                             flops_per_call=sum(r[3] for r in recs) / len(recs))
        return out


def shape_report(census, path):
    rows = []
    for (name, shp), recs in census.shapes.items():
        ms = sum(r[0].elapsed_time(r[1]) for r in recs)
        fl = sum(r[2] for r in recs)
        rows.append((ms, name, shp, len(recs), fl / (ms * 1e-3) / 1e12))
    rows.sort(reverse=True)
    with open(path, "w") as f:
        for ms, name, shp, n, tf in rows:
            f.write(f"{ms:8.3f} ms  {n:3d}x  {tf:7.1f} TF  {name} x{shp[0]} w{shp[1]}\n")


def pmc_traffic(kernel_substr, grid_size, pick="calls"):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 --pmc summary
    (profiles/*_pmc_traffic.json, written by scripts/summarize_pmc.py; FETCH_SIZE already x2-corrected for
    gfx950 per MI355X_MICROARCH.md).  None when no summary is present."""
    import glob
    # (by name, newest run last: a fresh checkout gives every file the same mtime)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "**", "*_pmc_traffic.json"), recursive=True), key=os.path.basename)
    if not files:
        return None, None
    # a counter summary is only valid for the kernels it was collected on: scripts/summarize_pmc.py stamps the
    # digest of the kernel sources; the summary of THESE sources is quoted, any other is refused (traffic = null)
    doc = None
    try:
        from occformer_amd.csrc.build import _digest
        dig = _digest()
        for f in reversed(files):
            cand = json.load(open(f))
            if cand.get("source_digest") == dig:
                doc, files = cand, [f]
                break
        if doc is None:
            return None, os.path.basename(files[-1]) + " (stale: kernel sources changed since)"
    except Exception:
        doc = json.load(open(files[-1]))
    rows = doc["kernels"]
    best = [r for r in rows if kernel_substr in r["kernel"] and (grid_size is None or r["grid"] == grid_size)]
    if not best:
        return None, None
    # pick = "bytes": the launch shape that moves the most (the dominant launch of a kernel used at many shapes)
    r = max(best, key=(lambda r: r["fetch_bytes"] + r["write_bytes"]) if pick == "bytes" else (lambda r: r["calls"]))
    _PMC_EXTRA.clear()
    _PMC_EXTRA.update({k: r[k] for k in ("mfma_busy_frac", "clock_ghz", "wait_inst_any_frac", "valu_per_mfma") if k in r})
    return r["fetch_bytes"] + r["write_bytes"], os.path.basename(files[-1])


_PMC_EXTRA = {}          # SQ / GRBM counters of the kernel pmc_traffic() last returned (scripts/summarize_pmc.py)


def _conv_entry(name, shp, ms, flops, prec):
    """one (op, shape) row of the convolution family: kernel name, algorithmic rate, matrix-core products per
    algorithmic product, PMC traffic of that kernel on these sources (or None)"""
    from occformer_amd.ops import get_ops
    ops = get_ops()
    avg_ms = sum(ms) / len(ms)
    achieved = flops / (avg_ms * 1e-3) / 1e12
    terms = 3 if prec == "bf16x3" else 1
    traffic, src, nbytes, extra = None, None, None, {}
    products = float({"f32": 1, "bf16x3": 3, "bf16": 1}[prec])
    base = name.split("[")[0]
    if base == "conv3d":
        B, X, Y, Z, Cin = shp[0]
        Cout = shp[1][0]
        halo = shp[1][1] == 27 * Cin and Cin % 32 == 0 and prec != "f32"
        tz = 16 if Z >= 16 else Z
        wino = halo and prec == "bf16x3" and getattr(ops, "use_wino", False) and Cout % 64 == 0 and tz in (4, 8, 16) and \
            Z % tz == 0 and os.environ.get("OCCF_WINO", "1") != "0"
        tn = 2 if Cout % 128 == 0 else 3 if Cout % 192 == 0 else 1
        nbytes = 4 * (B * X * Y * Z * (Cin + Cout)) + 4 * Cout * shp[1][1]
        if wino:
            # csrc/conv_wino.hip: Winograd F(2, 3) along x -- 18 instead of 27 multiply-adds per output voxel
            f16 = name != base
            w1 = f16 and getattr(ops, "dgrad_f16_single", False)       # the filters too as one fp16 piece: ONE product
            kname = f"conv3x3x3_wino_kernel<{tn}, {'true' if f16 else 'false'}, {'true' if w1 else 'false'}>"
            products = (1 if w1 else 2 if f16 else 3) * 2.0 / 3.0
            grid = B * ((X + 1) // 2) * ((Y + 64 // tz - 1) // (64 // tz)) * (Z // tz) * (Cout // (64 * tn)) * 512
            traffic, src = pmc_traffic(kname, grid)
            extra = dict(_PMC_EXTRA) if traffic is not None else {}
        elif halo:
            frag = "true" if get_halo_frag() else "false"
            sch = (1 if os.environ.get("OCCF_HALO_SCHED", "1") != "0" else 0) if get_halo_frag() else 0
            small = get_halo_frag() and os.environ.get("OCCF_HALO_SMALL", "0") not in ("0", "")
            kname = f"conv3x3x3_halo_kernel<{tn}, {terms}, {frag}, {sch}, {2 if small else 4}>"
            ty = (64 if small else 128) // tz
            bn = 64 * tn
            grid = B * ((X + 1) // 2) * ((Y + ty - 1) // ty) * (Z // tz) * ((Cout + bn - 1) // bn) * (256 if small else 512)
            traffic, src = pmc_traffic(kname, grid)
            extra = dict(_PMC_EXTRA) if traffic is not None else {}
        else:
            kname = "gemm_bf16_kernel<CONV>"
    elif base == "conv3d_wgrad":
        B, X, Y, Z, Cout = shp[0]
        Cin = shp[1][-1]
        presplit = taps_hint(shp, flops) >= 9 and Cin % 8 == 0 and Cout % 8 == 0
        f16 = prec == "bf16x3" and getattr(ops, "wgrad_f16", False) and presplit
        kname = f"wgrad_kernel<{2 if f16 else terms}, {'true' if presplit else 'false'}, ...>"
        taps = flops // max(2 * B * X * Y * Z * Cout * Cin, 1)
        same_grid = list(shp[1][:4]) == [B, X, Y, Z]
        if taps == 27 and same_grid and Z in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0 and terms == 3 \
                and os.environ.get("OCCF_WG_G8", "1") != "0":
            # csrc/wgrad_g8.h: tiles of 192 / 128 / 64 channels, LDS-DMA staging (the largest tile class of the launch)
            tile = lambda c: 3 if c % 192 == 0 or (c % 128 == 64 and c >= 192) else 2 if c % 128 == 0 else 1
            ti, tc = tile(Cout), tile(Cin)
            x1 = f16 and getattr(ops, "wgrad_f16_single", False)
            kname = (f"wgrad_g8_kernel<{ti}, {tc}, 2, 2, true, true>" if x1 else
                     f"wgrad_g8_kernel<{ti}, {tc}, {1 if ti + tc >= 5 else 2}, 2, {'true' if f16 else 'false'}, false>")
            if x1:
                products = 1.0           # dy and x each ONE fp16 piece (csrc/wgrad_g8.h X1; precision_probe.py wg1c)
        if (taps == 9 and same_grid and Z == 1 and X in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0 and f16
                and getattr(ops, "wgrad_2d_as_g8", False) and getattr(ops, "wgrad_f16_single", False)):
            # ops.conv3d_wgrad: a 2-D 3x3 weight gradient with a power-of-two first extent runs as a [1, Y, X] volume on G8
            tile = lambda c: 3 if c % 192 == 0 or (c % 128 == 64 and c >= 192) else 2 if c % 128 == 0 else 1
            kname = f"wgrad_g8_kernel<{tile(Cout)}, {tile(Cin)}, 2, 2, true, true> (+ two transposes)"
            products = 1.0
        if f16 and products != 1.0:
            products = 2.0
        nbytes = 4 * (B * X * Y * Z * Cout) + 4 * int(torch.tensor(shp[1]).prod()) + 4 * Cout * taps * Cin
        traffic, src = pmc_traffic(kname.split(", ...")[0], None, pick="bytes")
        extra = dict(_PMC_EXTRA) if traffic is not None else {}
    else:
        kname = "gemm_bf16_kernel<CONV, transposed loader>"
    return dict(kernel=f"{kname}  [{name} {list(shp[0])} {list(shp[1])}]", avg_kernel_ms=avg_ms, achieved=achieved,
                launches_timed=len(ms), total_ms=sum(ms), algorithmic_flops_per_launch=flops,
                mfma_products_per_algorithmic_product=round(products, 3), traffic=traffic, traffic_source=src,
                algorithmic_bytes_per_launch=nbytes, **extra)


def roofline(timed_census, kernels, prec, steps):
    """The dominant kernel = the (op, shape) with the largest total time among the convolution launches (forward,
    data gradient, weight gradient) of the timed region (HIP events on the launching stream); priced on ALGORITHMIC
    flops against the dense MFMA peak.  ``family``: the other large rows of the same census, each with the matrix-core
    products it spends per algorithmic product (3-term split x Winograd's 2/3; two fp16-piece products in the data and
    weight gradients) -- the executed-MFMA utilisation of a row is ``frac`` x that number."""
    torch.cuda.synchronize()
    peak = F32_MFMA_PEAK_TF if prec == "f32" else BF16_MFMA_PEAK_TF
    rows = []
    for (name, shp), recs in timed_census.shapes.items():
        if not name.startswith("conv3d"):
            continue
        ms = [r[0].elapsed_time(r[1]) for r in recs]
        rows.append((sum(ms), name, shp, ms, recs[0][2]))
    if not rows:
        return None
    rows.sort(key=lambda r: -r[0])
    fam = []
    for tot, name, shp, ms, flops in rows[:8]:
        e = _conv_entry(name, shp, ms, flops, prec)
        e["frac"] = e["achieved"] / peak
        e["launches_per_step"] = len(ms) // max(steps, 1)
        fam.append(e)
    # the dominant KERNEL = the kernel name with the largest total time over all its launch shapes (among the rows above);
    # the record quotes its heaviest shape
    per_kernel = {}
    for e in fam:
        k = e["kernel"].split("  [")[0]
        per_kernel[k] = per_kernel.get(k, 0.0) + e["total_ms"]
    dom = max(per_kernel, key=per_kernel.get)
    top = max((e for e in fam if e["kernel"].split("  [")[0] == dom), key=lambda e: e["total_ms"])
    out = {"bound": "mfma", "kernel": top["kernel"], "achieved": top["achieved"], "peak": peak, "unit": "TFLOP/s",
           "frac": top["frac"], "traffic": top["traffic"], "traffic_source": top["traffic_source"]}
    out.update({k: v for k, v in top.items() if k not in out and k != "total_ms"})
    out["kernel_total_ms_in_timed_region"] = round(per_kernel[dom], 3)
    out["family"] = [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()
                      if k in ("kernel", "avg_kernel_ms", "achieved", "frac", "launches_per_step",
                               "mfma_products_per_algorithmic_product", "traffic", "mfma_busy_frac", "clock_ghz")}
                     for e in fam]
    return out


def get_halo_frag():
    from occformer_amd.ops import get_ops
    return bool(getattr(get_ops(), "halo_frag", False))


def taps_hint(shp, flops):
    B, X, Y, Z, Cout = shp[0]
    Cin = shp[1][-1]
    return flops // max(2 * B * X * Y * Z * Cout * Cin, 1)


def cpu_baseline(model, meta, img_inputs, points):
    """The oracle (CPU fp32 restatement of the reference path, pinned against the reference's
    own Python) on the host cores: one sample of the same workload, forward."""
    from oracle import occformer_ref as O
    from occformer_amd import configs
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    x = img_inputs[0].cpu()
    cams = tuple(t.cpu() for t in img_inputs[1:7])
    cfg = configs.oracle_cfg(meta)
    cores = min(os.cpu_count() or 1, 32)     # torch CPU ops stop scaling (and oversubscribe) beyond this
    torch.set_num_threads(cores)
    t0 = time.perf_counter()
    with torch.no_grad():
        res = O.occformer_forward(sd, x, cams, cfg, None if points is None else [p.cpu() for p in points])
    dt = time.perf_counter() - t0
    return dict(value=1.0 / dt, unit="samples/s", cores=cores, kind="port",
                sample="1 sample of the same workload (oracle/occformer_ref.occformer_forward, fp32, "
                       f"torch CPU, {cores} threads): {dt:.1f} s"), res


def cpu_baseline_train(model, cfg, meta, img_inputs, targets, tape, gates):
    """The oracle's training step (train-mode forward + torch.autograd backward of the summed losses,
    oracle/occformer_train_ref.train_step) on the host cores: one sample of the same workload, on the noise draws the
    GPU step just consumed (``tape``, replayed in call order).  Two passes: (1) the oracle as it is -- the TIMED
    ``cpu_baseline`` and the UNGATED side of ``check``; (2) with the heavy ReLU gates the GPU step used (``gates``:
    oracle.occformer_ref.forced_gates -- both sides differentiate the same piecewise-linear function): the gated side."""
    from oracle import occformer_ref as O
    from oracle import occformer_train_ref as T
    from occformer_amd import configs
    gt_occ, points, gt_depths = targets
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    ocfg = configs.oracle_train_cfg(cfg, meta, class_weight=model.pts_bbox_head.class_weight)
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    oargs = (sd, img_inputs[0].cpu(), tuple(t.cpu() for t in img_inputs[1:7]), gt_depths.cpu(), gt_occ.cpu(),
             None if points is None else [p.cpu() for p in points], ocfg)
    replay = _Replay(tape, torch.device("cpu"))
    t0 = time.perf_counter()
    losses_u, grads_u = T.train_step(*oargs, rng=replay)
    dt = time.perf_counter() - t0
    if replay.i != len(tape):
        raise RuntimeError("the oracle consumed a different number of noise draws than the GPU step")
    replay = _Replay(tape, torch.device("cpu"))
    forced = O.forced_gates(gates, level="heavy+bev")
    losses, grads = T.train_step(*oargs, rng=replay, gates=forced)
    if replay.i != len(tape) or forced.i != len(gates):
        raise RuntimeError("the oracle consumed a different number of noise draws / heavy ReLUs than the GPU step")
    return dict(value=1.0 / dt, unit="samples/s", cores=cores, kind="port",
                sample="1 training step (fwd + bwd) of the same workload on the CPU oracle, on the weights the timed "
                       "GPU steps left behind "
                       f"(oracle/occformer_train_ref.train_step, fp32, torch CPU autograd, {cores} threads): {dt:.1f} s"), \
        {k: float(v) for k, v in losses.items()}, grads, forced, ({k: float(v) for k, v in losses_u.items()}, grads_u)


class _Replay:
    """feeds the GPU step the very noise draws the CPU oracle consumed (same protocol as tests/test_training.ReplayRNG)"""

    def __init__(self, tape, device):
        self.tape, self.i, self.device = tape, 0, device

    def _next(self, kind, numel):
        k, t = self.tape[self.i]
        self.i += 1
        if k != kind or t.numel() != numel:
            raise RuntimeError(f"noise draw {self.i}: the GPU step asks {kind}/{numel}, the oracle drew {k}/{t.numel()}")
        return t.to(self.device)

    def rand(self, *shape):
        import math
        return self._next("rand", math.prod(shape)).reshape(shape)

    def randperm(self, n):
        return self._next("randperm", n)

    def exponential(self, shape, dtype=torch.float32):
        import math
        return self._next("exponential", math.prod(shape)).reshape(tuple(shape)).float()

    def multinomial(self, weights, k):
        """the oracle's statement of torch.multinomial(.., replacement=False) on the replayed exponential draw"""
        q = self.exponential(weights.shape, weights.dtype)
        return torch.topk(weights / q, k, dim=-1)[1]


def train_check_gpu_step(model, net_kwargs, device):
    """ONE more forward_train + backward on the GPU at the (post-training) weights, with its noise draws and its
    heavy ReLU gates (noise.relu_gate) taped for the oracle -> (losses, tape, gates); the gradients stay in ``param.grad``"""
    from occformer_amd import noise
    from occformer_amd.training import DeviceRNG
    rec = noise.RecordedRNG(DeviceRNG(device, 1234))
    noise.set_rng(rec)
    gates = noise.record_gates("heavy+bev")
    try:
        for p in model.parameters():
            p.grad = None
        losses = model(return_loss=True, **net_kwargs)
        total = sum(v for k, v in losses.items() if "loss" in k)
        total.backward()
    finally:
        noise.set_rng(None)
        noise.record_gates(False)
    torch.cuda.synchronize()
    return {k: float(v.detach()) for k, v in losses.items()}, rec.tape, gates


MAX_REL_Z = 5e-4        # a forced gate may differ from the oracle's own only where |z| <= this x the tensor's RMS


# per-parameter bounds of a FULL-SIZE training-step comparison, shared by this file's `check` and
# tests/test_workloads_gpu.py (one filter, one set of bounds: VERDICT r5 weak #1)
# (round 6: the worst parameter back at 6e-3 -- it had been loosened to 2e-2 in round 5 for "a dilated BEV-ASPP convolution
# of a coarse stage" that moved between 1e-3 and 1e-2 from visit to visit; that tail was ReLU units of the BEV ASPP's small
# maps decided differently within rounding, and with those maps on the gate tape (level "heavy+bev") the worst parameter
# measures 3.3e-4 ... 1.1e-3: profiles/r06/r06k_*)
PER_PARAMETER_BOUNDS = {"90%": 1e-3, "99%": 3e-3, "100%": 6e-3}
WHOLE_GRADIENT_BOUND = 1e-3


def grad_figures(named, gl, cpu_losses, cpu_grads):
    """-> (worst relative loss difference, whole-gradient relative L2, per-parameter quantiles, [(rel L2, name)] sorted).
    Per-parameter figures over the parameters that carry gradient: squared norm >= 1e-8 of the whole vector's."""
    worst = max(abs(float(gl[k]) - float(v)) / max(1.0, abs(float(v))) for k, v in cpu_losses.items())
    num = den = 0.0
    per = []
    for k, g in cpu_grads.items():
        if g is None or named[k].grad is None:
            continue
        d = float((named[k].grad.detach().cpu() - g).norm()) ** 2
        n = float(g.norm()) ** 2
        num, den = num + d, den + n
        per.append(((d / max(n, 1e-30)) ** 0.5, k, n))
    per = sorted((e, k) for e, k, n in per if n >= 1e-8 * den)
    q = lambda f: per[int(f * (len(per) - 1))][0] if per else None
    return worst, (num / max(den, 1e-30)) ** 0.5, {"50%": q(0.5), "90%": q(0.9), "99%": q(0.99), "100%": q(1.0)}, per


def _grad_figures(model, gl, cpu_losses, cpu_grads):
    worst, whole, quant, per = grad_figures(dict(model.named_parameters()), gl, cpu_losses, cpu_grads)
    return worst, whole, quant, len(per)


def bounds_violated(whole, quant):
    """the parity bounds a `check` is held to (None = met)"""
    bad = [f"whole gradient {whole:.2e} > {WHOLE_GRADIENT_BOUND:.0e}"] if whole > WHOLE_GRADIENT_BOUND else []
    bad += [f"{k} of the parameters {quant[k]:.2e} > {b:.0e}" for k, b in PER_PARAMETER_BOUNDS.items()
            if quant.get(k) is not None and quant[k] > b]
    return "; ".join(bad) or None


def train_check(model, gl, cpu_losses, cpu_grads, forced, n_draws, ungated):
    """`losses` (GPU) and `cpu_losses` are the same quantity (same weights, inputs, noise); `check` says how far apart
    they and the gradients are -- with the heavy ReLU gates of the GPU step forced into the oracle (the figure the
    parity gate is held to) AND with the oracle on its own gates (``*_ungated``: what the forcing buys on this box)"""
    named = dict(model.named_parameters())
    worst, whole, quant, per = grad_figures(named, gl, cpu_losses, cpu_grads)
    n = len(per)
    worst_u, whole_u, quant_u, _ = grad_figures(named, gl, *ungated)
    return dict(max_rel_loss_diff=worst, grad_rel_l2=whole, per_parameter_rel_l2_quantiles=quant,
                worst_parameters=[{"name": k, "rel_l2": e} for e, k in per[-3:]],
                bounds={"whole_gradient": WHOLE_GRADIENT_BOUND, "per_parameter": PER_PARAMETER_BOUNDS,
                        "violated": bounds_violated(whole, quant)},
                max_rel_loss_diff_ungated=worst_u, grad_rel_l2_ungated=whole_u,
                per_parameter_rel_l2_quantiles_ungated=quant_u,
                parameters_compared=n, noise_draws_replayed=n_draws,
                forced_relu_gates={"units": forced.units, "gated_differently": forced.flipped,
                                 "largest_abs_preactivation_among_them": forced.max_abs_z,
                                 "largest_preactivation_over_tensor_rms": forced.max_rel_z,
                                 "bound_on_that": MAX_REL_Z},
                what="GPU forward_train + backward vs the CPU oracle's train_step: same weights (after the timed "
                     "optimizer steps), same inputs, the GPU step's noise tape replayed by the oracle.  Gated figures: "
                     "the ReLU gates of the decoder head's MLPs, of DepthNet and of the BEV ASPP's image-level vector and "
                     "the decoder's boolean attention masks taken from the GPU step (decisions whose pre-activations / "
                     "pooled logits straddle zero within rounding: counted above; bench.py exits non-zero when one of "
                     "them is further from zero than the bound).  "
                     "*_ungated: the oracle on its own gates.  Losses relative to max(1, |loss|), gradients as "
                     "relative L2 of the whole vector / per parameter (parameters whose squared gradient norm is >= 1e-8 "
                     "of the whole vector's -- the filter and the bounds of tests/test_workloads_gpu.py; bench.py exits "
                     "non-zero when `bounds.violated` is set)")


WORKLOAD_DESC = {
    "nusc_r50_200": "nuScenes R50 256x704, 200x200x16 voxels",
    "nusc_r50_ref128": "nuScenes R50 256x704, 128x128x16 voxels (the reference's shipped grid)",
    "kitti_effb7_128": "SemanticKITTI EfficientNetB7 mono 384x1280, 128x128x16 voxels (output 256x256x32)",
    "kitti_effb7_256lit": "SemanticKITTI EfficientNetB7 mono 384x1280, 256x256x32 voxels (lss_downsample 1)",
    "nusc_r101": "nuScenes R101-DCN 896x1600, 200x200x16 voxels",
}


def raw_images(meta, device, seed=0):
    g = torch.Generator().manual_seed(7 + seed)
    return torch.randn(1, meta["ncams"], 3, *meta["input_size"], generator=g).to(device)


def set_image_dtype(model, image_dtype):
    if image_dtype == "bf16":
        model.image_dtype = torch.bfloat16
        model.img_backbone.to(memory_format=torch.channels_last)
        model.img_neck.to(memory_format=torch.channels_last)


def forward_from_images(workload, device, image_dtype, steps):
    """the forward as the reference's own fps harness times it (tools/analysis_tools/benchmark.py:69-94:
    ``model(return_loss=False, **data)`` from the raw images): a second detector WITH the workload's image branch,
    random weights, ``steps`` timed inference steps + one step with the stage timers"""
    from occformer_amd import configs
    from occformer_amd.registry import build_model
    cfg, meta = configs.workload(workload, with_image_branch=True)
    torch.manual_seed(1)
    m = build_model(cfg).to(device).eval()
    set_image_dtype(m, image_dtype)
    img_inputs, metas, points = synthetic_sample(meta, device, seed=0)
    img_inputs[0] = raw_images(meta, device)

    def step():
        with torch.no_grad():
            vox, _, _ = m.extract_feat(None, img_inputs, metas)
            t = m._tick("", 0.0)
            res = m.pts_bbox_head.simple_test(vox, metas, points=points)
            m._tick("mask2former_head", t)
            return res
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    m.record_time = True
    m.time_stats.clear()
    step()
    m.record_time = False
    stages = {k: round(1e3 * sum(v) / len(v), 3) for k, v in m.time_stats.items() if k}

    # the same K frames as a SERVING loop would run them: the image branch of frame t + 1 on a side HIP stream while the
    # 3-D path of frame t runs on the main stream (two streams, one event per frame; the per-frame latency is the
    # sequential one above, the throughput is what a stream of frames sees).  Reported NEXT TO the sequential figure,
    # which is the one the reference's harness measures.
    if workload == "nusc_r101":
        # (its DCNv2 layers run on the library's kernels and share the op layer's split-K workspace with the main stream:
        # not made stream-safe, so no pipelined figure for this workload)
        del m
        torch.cuda.empty_cache()
        return {"metric": "samples/sec forward FROM THE RAW IMAGES", "value": steps / dt, "unit": "samples/s",
                "steps": steps, "warmup": 3, "ms_per_step": 1e3 * dt / steps,
                "input": [1, meta["ncams"], 3, *meta["input_size"]], "stages_ms": stages, "pipelined": None}
    side = torch.cuda.Stream(device=device)
    main = torch.cuda.current_stream(device)

    def encode_next():
        side.wait_stream(main)                       # (the frame buffer is resident; keeps allocator reuse ordered)
        with torch.cuda.stream(side), torch.no_grad():
            f = m.image_encoder(img_inputs[0])
            ev = torch.cuda.Event()
            ev.record(side)
        return f, ev

    def step_pipelined(feat_ev):
        f, ev = feat_ev
        nxt = encode_next()                          # frame t + 1: issued first, runs beside this frame's 3-D path
        main.wait_event(ev)
        f.record_stream(main)
        with torch.no_grad():
            vox, _, _ = m.extract_feat(None, [f] + list(img_inputs[1:]), metas)
            m.pts_bbox_head.simple_test(vox, metas, points=points)
        return nxt
    fe = encode_next()
    for _ in range(3):
        fe = step_pipelined(fe)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fe = step_pipelined(fe)
    torch.cuda.synchronize()
    dtp = time.perf_counter() - t0
    del m, fe
    torch.cuda.empty_cache()
    return {"metric": "samples/sec forward FROM THE RAW IMAGES (img_backbone + img_neck on PyTorch-ROCm / MIOpen, "
                      f"{image_dtype}, inside the timed region -> inference hot path) -- north_star target >= 30",
            "value": steps / dt, "unit": "samples/s", "steps": steps, "warmup": 3, "ms_per_step": 1e3 * dt / steps,
            "input": [1, meta["ncams"], 3, *meta["input_size"]], "stages_ms": stages,
            "pipelined": {"value": steps / dtp, "unit": "samples/s", "ms_per_step": 1e3 * dtp / steps,
                          "what": "the same frames as a serving loop: image branch of frame t + 1 on a side HIP stream "
                                  "beside the 3-D path of frame t (throughput; the sequential `value` above is what "
                                  "tools/analysis_tools/benchmark.py measures)"}}


def train_from_images(workload, device, image_dtype, steps, kitti_train_cfg=None):
    """the reference's WHOLE training step (occupancyformer.py:132-199: img_backbone + img_neck inside forward_train,
    their gradients and optimizer state included) as a secondary record: a second detector with the workload's image
    branch, ``steps`` timed steps of forward_train + backward + grad-clip + fused AdamW"""
    from occformer_amd import configs
    from occformer_amd.registry import build_model
    cfg, meta = configs.workload(workload, with_image_branch=True)
    if meta.get("kitti"):
        cfg["train_cfg"] = dict(pts=configs.train_cfg_pts())
    torch.manual_seed(2)
    torch.cuda.reset_peak_memory_stats()
    m = build_model(cfg).to(device).train()
    set_image_dtype(m, image_dtype)
    img_inputs, metas, _ = synthetic_sample(meta, device, seed=0)
    img_inputs[0] = raw_images(meta, device)
    gt_occ, gt_points, gt_depths = configs.synthetic_targets(meta, device, seed=0)
    kw = dict(img_metas=metas, img_inputs=list(img_inputs) + [gt_depths], gt_occ=gt_occ, points_occ=gt_points)
    params = [p for p in m.parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
    max_norm = 20.0 if meta.get("kitti") else 5.0

    def step():
        opt.zero_grad(set_to_none=True)
        m.prefetch_gt(gt_occ, ready=True)
        losses = m(return_loss=True, **kw)
        sum(v for k, v in losses.items() if "loss" in k).backward()
        torch.nn.utils.clip_grad_norm_(params, max_norm)
        opt.step()
        return losses
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rec = {"metric": "samples/sec fwd+bwd FROM THE RAW IMAGES (img_backbone + img_neck on PyTorch-ROCm / MIOpen, "
                     f"{image_dtype}, trained inside the step: occupancyformer.py:132-199)",
           "value": steps / dt, "unit": "samples/s", "steps": steps, "warmup": 3, "ms_per_step": 1e3 * dt / steps,
           "input": [1, meta["ncams"], 3, *meta["input_size"]],
           "parameters_M": round(sum(p.numel() for p in params) / 1e6, 1),
           "peak_memory_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
           "finite_losses": bool(all(torch.isfinite(v.detach()).all() for v in last.values()))}
    del m, opt, params, last
    torch.cuda.empty_cache()
    return rec


def launch_ranks(n, argv=None):
    """``python bench.py --gpus N`` without a launcher around it: re-run this very command line as N ranks of ONE node
    under ``torch.distributed.run`` (one process per GPU, LOCAL_RANK -> device, rendezvous on 127.0.0.1; the
    reference's counterpart is tools/dist_train.sh:9-19) and hand its exit code back.  Rank 0 of that job prints the one
    JSON line with ``n_gpus`` = N."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + \
          list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this host driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    return subprocess.call(cmd, env=env)


def dry_run_rank(args, rank, world):
    """``--dry-run``: the launcher / rendezvous / timing / reporting skeleton of the bench on CPU ranks over gloo with a
    mock step (tests/test_boundary.py runs it at N = 2: no GPU, no library) -- everything around the step is the code
    the real run goes through: barrier, K timed steps, barrier, max over ranks, rank 0 prints the one line."""
    from occformer_amd import dist_utils
    dist = dist_utils.init("gloo") if world > 1 else None
    w = torch.eye(64)
    step = lambda: (w @ w).sum().item()
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    dt = dist_utils.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        emit(json.dumps({"metric": "dry run (mock step on CPU ranks, gloo)", "value": world * args.steps / dt,
                          "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
                          "dry_run": True}))
    if dist is not None:
        dist.destroy_process_group()


_REAL_STDOUT = None


def emit(line):
    sys.stdout.flush()
    if _REAL_STDOUT is None:
        print(line, flush=True)
    else:
        os.write(_REAL_STDOUT, (line + "\n").encode())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", default="train", choices=["train", "forward"],
                    help="train (default) = BASELINE.json's metric: forward + backward + gradient all-reduce + clip + "
                         "AdamW step of OccupancyFormer.forward_train; forward = the inference hot path (simple_test)")
    ap.add_argument("--workload", default="nusc_r50_200",
                    choices=["nusc_r50_200", "nusc_r50_ref128", "kitti_effb7_128", "kitti_effb7_256lit", "nusc_r101"],
                    help="BASELINE.json configs: [2]/[3] nusc_r50_200 (the metric's grid; default), the reference's "
                         "own 128-grid, [0] kitti_effb7_128, [1] kitti_effb7_256lit, [4] nusc_r101")
    ap.add_argument("--grid", default=None, choices=["200", "reference"], help="(legacy) nusc_r50 grid")
    ap.add_argument("--precision", default=None, choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the dense contractions (default: the library default, bf16x3)")
    ap.add_argument("--from-images", action="store_true",
                    help="start at the raw images [1, N, 3, H, W]: the workload's img_backbone + img_neck "
                         "(PyTorch-ROCm / MIOpen, --image-dtype) run inside the timed region, forward or training step "
                         "(stage img_encoder in forward mode)")
    ap.add_argument("--image-dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="train mode, N > 1: the reference's sync_bn = True (tools/train.py:221-223) -- DepthNet's BatchNorms "
                         "normalise with all ranks' statistics (one small all-gather per layer); default: per-rank "
                         "statistics, the gradient all-reduce stays the only collective (north_star)")
    ap.add_argument("--shape-report", default=None, help="write a per-shape GEMM/conv timing table here")
    ap.add_argument("--check", action="store_true", help="also report max abs err vs the oracle output")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    # the contract is ONE line on stdout: everything else this process (or a library under it: RCCL prints a version
    # banner to stdout when the communicator comes up, r05a) writes to fd 1 goes to stderr; the JSON line is written to
    # the real stdout at the end
    global _REAL_STDOUT
    if _REAL_STDOUT is None and not ("WORLD_SIZE" not in os.environ and args.gpus > 1):
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under a launcher: become one (N ranks of this node, one per GPU)
        if not args.dry_run and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py --gpus {args.gpus}: this node shows {torch.cuda.device_count()} GPU(s)")
        raise SystemExit(launch_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); reporting n_gpus = {world}",
              file=sys.stderr)
    if args.dry_run:
        return dry_run_rank(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU implementation")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from occformer_amd import dist_utils
    # under a launcher the process group is brought up even for ONE rank (and the step then goes through the DDP wrap,
    # its bucket hooks and -- with OCCF_DIST_AT_WORLD_1=1 -- every collective of the path on RCCL): what a 1-GPU box can
    # verify of the N > 1 path.  A plain ``python bench.py`` (the N = 1 record) stays free of it.
    dist = dist_utils.init("nccl", device) if world > 1 or "WORLD_SIZE" in os.environ else None

    import occformer_amd
    from occformer_amd import configs
    from occformer_amd.ops import get_ops
    from occformer_amd.registry import build_model

    if args.precision:
        get_ops().precision = args.precision
    prec = get_ops().precision
    torch.manual_seed(0)
    if args.grid:
        args.workload = {"200": "nusc_r50_200", "reference": "nusc_r50_ref128"}[args.grid]
    cfg, meta = configs.workload(args.workload)
    train = args.mode == "train"
    if args.from_images:
        # the reference's own entry: model(**data) from the raw images (tools/analysis_tools/benchmark.py:69-94 for the
        # forward, occupancyformer.py:132-199 for the training step) -- the workload's img_backbone / img_neck
        # (PyTorch-ROCm / MIOpen; DCNv2 of the R101 on csrc/dcn.hip) run inside the timed region
        cfg, meta = configs.workload(args.workload, with_image_branch=True)
        args.no_cpu_baseline = True            # the oracle starts at the neck features
    if train and meta.get("kitti"):
        cfg["train_cfg"] = dict(pts=configs.train_cfg_pts())
    model = build_model(cfg).to(device)
    if args.sync_bn and train and dist is not None:
        dist_utils.convert_sync_batchnorm(model)
    img_inputs, metas, points = synthetic_sample(meta, device, seed=rank)
    if args.from_images:
        img_inputs[0] = raw_images(meta, device, rank)
        set_image_dtype(model, args.image_dtype)
    targets = configs.synthetic_targets(meta, device, seed=rank) if train else None

    def step_forward():
        with torch.no_grad():
            vox, img_feats, depth = model.extract_feat(None, img_inputs, metas)
            t = model._tick("", 0.0)
            res = model.pts_bbox_head.simple_test(vox, metas, points=points)
            model._tick("mask2former_head", t)
            return res

    census_ops = ("conv3d",)
    if train:
        # the reference's training step (P/occformer/apis/mmdet_train.py:72-80, occformer_nusc_r50_256x704.py:284-301):
        # DDP (gradient all-reduce over RCCL, overlapped with backward), grad-clip 5 (nuScenes) / 20 (KITTI), AdamW
        model.train()
        net = model
        if dist is not None:
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                            gradient_as_bucket_view=True, bucket_cap_mb=64)
        params = [p for p in model.parameters() if p.requires_grad]
        try:
            opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
        except Exception:
            opt = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, foreach=True)
        max_norm = 20.0 if meta.get("kitti") else 5.0
        gt_occ, gt_points, gt_depths = targets
        train_inputs = list(img_inputs) + [gt_depths]
        census_ops = ("conv3d", "conv3d_wgrad", "conv3d_dgrad")

        net_kwargs = dict(img_metas=metas, img_inputs=train_inputs, gt_occ=gt_occ, points_occ=gt_points)

        def step_train():
            opt.zero_grad(set_to_none=True)
            # the label scan of this sample's ground truth runs on a side stream (inputs are resident: ready=True), so
            # that forward_train does not synchronise with the previous step's still queued backward / optimizer
            model.prefetch_gt(gt_occ, ready=True)
            losses = net(return_loss=True, **net_kwargs)
            total = sum(v for k, v in losses.items() if "loss" in k)
            total.backward()
            torch.nn.utils.clip_grad_norm_(params, max_norm)
            opt.step()
            return losses
        step = step_train
    else:
        model.eval()
        step = step_forward

    for i in range(args.warmup):
        step()
        if train and i == 0:
            missing = [n for n, p in model.named_parameters() if p.requires_grad and p.grad is None]
            if missing:
                raise SystemExit(f"parameters without a gradient after backward (DDP would stall): {missing[:8]}")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    # the timed region; the convolution launches (the dominant kernels) carry HIP events on the launching
    # stream so that `roofline` is measured over exactly these K steps
    with KernelCensus(get_ops(), only=census_ops) as timed_census:
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = time.perf_counter() - t0
    dt = dist_utils.max_over_ranks(dt, device)
    mem_gb = torch.cuda.max_memory_allocated() / 2 ** 30

    # stage timers (the reference's own stage names) + per-kernel census on ONE more step
    model.record_time = not train
    model.time_stats.clear()
    with KernelCensus(get_ops()) as census:
        res_gpu = step()
    model.record_time = False
    kernels = census.summary()
    if args.shape_report and rank == 0:
        shape_report(census, args.shape_report)
    stages = {k: round(1e3 * sum(v) / len(v), 3) for k, v in model.time_stats.items() if k}

    # the north-star FORWARD figure from the same process (N = 1): >= 20 timed steps of the inference hot path with its
    # own roofline (the halo convolution) -- and, below, its own check against the oracle's forward
    forward_rec = None
    if train and rank == 0 and world == 1:
        model.eval()
        for _ in range(3):
            step_forward()
        torch.cuda.synchronize()
        fsteps = max(20, args.steps)
        with KernelCensus(get_ops(), only=("conv3d",)) as fcensus:
            t1 = time.perf_counter()
            for _ in range(fsteps):
                res_fwd = step_forward()
            torch.cuda.synchronize()
            fdt = time.perf_counter() - t1
        forward_rec = {"metric": "samples/sec forward (inference hot path, simple_test) -- north_star target >= 30",
                       "value": fsteps / fdt, "unit": "samples/s", "steps": fsteps, "warmup": 3,
                       "ms_per_step": 1e3 * fdt / fsteps, "roofline": roofline(fcensus, None, prec, fsteps)}
        model.train()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    roof = roofline(timed_census, kernels, prec, args.steps)
    gate_failure = None
    what = "fwd+bwd" if train else "forward"
    out = {
        "metric": f"samples/sec ({meta['ncams']}-cam frame) {what}, {WORKLOAD_DESC[args.workload]}, "
                  + ("training step of the hot path from %s (forward_train + backward + gradient all-reduce + "
                     "grad-clip + AdamW)" % ("the raw images (img_backbone + img_neck on PyTorch-ROCm / MIOpen, %s)"
                                             % args.image_dtype if args.from_images else "image-neck features")
                     if train else
                     ("raw images -> img_backbone + img_neck (PyTorch-ROCm / MIOpen, %s) -> " % args.image_dtype
                      if args.from_images else "") +
                     "hot path (LSS voxel pooling -> dual-path encoder -> pixel decoder -> occupancy decoder)"),
        "value": world * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16x3": "f32 (contractions as 3-term bf16 split on the bf16 matrix cores, fp32 "
                  "accumulate)", "bf16": "bf16 products, fp32 accumulate"}[prec], "data": "synthetic",
        "config": {"workload": f"{args.workload}_{'train_step' if train else 'forward'}_from_" +
                               (f"images_{args.image_dtype}_image_branch" if args.from_images else "neck_features"),
                   "grid": list(meta["grid"]), "input_size": list(meta["input_size"]), "global_batch": world,
                   "parallelism": f"dp{world} " + (
                       "(DDP: one RCCL gradient all-reduce per step, bucketed, overlapped with backward; BatchNorm on %s)"
                       % ("all ranks' statistics (sync_bn)" if args.sync_bn and world > 1 else "per-rank batch statistics")
                       if train else "(independent samples, no data-path collective)")},
        "roofline": roof,
        "kernels": {k: {"calls": v["calls"], "total_ms": round(v["total_ms"], 3), "avg_ms": round(v["avg_ms"], 4),
                        "GBps": round(v["bytes_per_call"] / (v["avg_ms"] * 1e-3) / 1e9, 1),
                        "TFLOPs": round(v["flops_per_call"] / (v["avg_ms"] * 1e-3) / 1e12, 2)}
                    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["total_ms"])},
        "stages_ms": stages, "peak_memory_GiB": round(mem_gb, 2),
    }
    if train:
        out["losses"] = {k: round(float(v.detach()), 5) for k, v in res_gpu.items()}
        out["forward"] = forward_rec
        if forward_rec is not None and not args.from_images:
            # (a secondary record: whatever goes wrong in the MIOpen image branch must not take the metric's line with it)
            try:
                out["forward_from_images"] = forward_from_images(args.workload, device, args.image_dtype,
                                                                 max(20, args.steps))
            except Exception as e:            # noqa: BLE001
                out["forward_from_images"] = {"error": f"{type(e).__name__}: {e}"[:500]}
            try:
                out["train_from_images"] = train_from_images(args.workload, device, args.image_dtype, max(20, args.steps))
            except Exception as e:            # noqa: BLE001
                out["train_from_images"] = {"error": f"{type(e).__name__}: {e}"[:500]}
    if world == 1 and not args.no_cpu_baseline:
        if train:
            if forward_rec is not None:
                model.eval()
                fbase, res_cpu = cpu_baseline(model, meta, img_inputs, points)
                forward_rec["cpu_baseline"] = fbase
                res_fwd = step_forward()                   # on the same (post-training) weights the oracle just used
                forward_rec["check"] = {
                    "output_voxels_max_abs_err": float((res_fwd["output_voxels"][0].cpu() - res_cpu["output_voxels"]).abs().max()),
                    "output_points_max_abs_err": None if res_cpu["output_points"] is None else float(
                        (res_fwd["output_points"].cpu() - res_cpu["output_points"]).abs().max())}
                del res_cpu
                model.train()
            cpu_leg = None
            # `losses` is re-taken on a taped step (its noise is what the oracle replays) so that it is comparable
            # with `cpu_losses`
            out["losses_last_timed_step"] = out["losses"]
            gl, tape, gates = train_check_gpu_step(model, net_kwargs, device)
            out["losses"] = {k: round(v, 5) for k, v in gl.items()}
            try:
                cpu_leg = cpu_baseline_train(model, cfg, meta, img_inputs, targets, tape, gates)
            except (MemoryError, RuntimeError) as e:       # host RAM: fall back to the forward leg
                # (torch reports a failed host allocation as a RuntimeError from DefaultCPUAllocator; anything else
                # is a real error and propagates)
                if isinstance(e, RuntimeError) and not any(m in str(e) for m in ("not enough memory", "DefaultCPUAllocator")):
                    raise
                base, _ = cpu_baseline(model, meta, img_inputs, points)
                base["sample"] = "FORWARD ONLY (the CPU training step did not fit: %s); " % type(e).__name__ + base["sample"]
                out["cpu_baseline"] = base
            if cpu_leg is not None:
                out["cpu_baseline"], out["cpu_losses"], cpu_grads, forced, ungated = cpu_leg
                out["check"] = train_check(model, gl, out["cpu_losses"], cpu_grads, forced, len(tape), ungated)
                if out["check"]["bounds"]["violated"]:
                    gate_failure = "bench.py: check outside the parity bounds: " + out["check"]["bounds"]["violated"]
                if forced.max_rel_z > MAX_REL_Z or forced.flipped > 1e-4 * forced.units:
                    gate_failure = (f"bench.py: {forced.flipped} of {forced.units} forced ReLU gates differ from the "
                                    f"oracle's own, the furthest at |z| = {forced.max_rel_z:.1e} of its tensor's RMS "
                                    f"(bound {MAX_REL_Z:.0e}, at most 1e-4 of the units): the forcing is not legitimate")
        else:
            base, res_cpu = cpu_baseline(model, meta, img_inputs, points)
            out["cpu_baseline"] = base
            if args.check:
                a, b = res_gpu["output_voxels"][0].cpu(), res_cpu["output_voxels"]
                out["check"] = {"output_voxels_max_abs_err": float((a - b).abs().max()),
                                "output_points_max_abs_err": None if res_cpu["output_points"] is None else float(
                                    (res_gpu["output_points"].cpu() - res_cpu["output_points"]).abs().max())}
    emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if gate_failure:
        raise SystemExit(gate_failure)


if __name__ == "__main__":
    main()
