"""The training LOOP of the path, not just its first step: ``torch.optim.AdamW(fused=True)`` updates
the parameters through ``_fused_adamw_`` WITHOUT bumping ``param._version`` on this torch build, so every value derived
from a weight (bf16 hi/lo split, tap-major / transposed / flipped layouts: occformer_amd/fused.py, autograd.py) must be
invalidated by something else -- the process-wide epoch of ``fused.invalidate_caches`` (optimizer post-step hook +
``forward_train`` entry).  Reference loop: projects/mmdet3d_plugin/occformer/apis/mmdet_train.py:72-80,
projects/configs/occformer_nusc/occformer_nusc_r50_256x704.py:284-301 (AdamW + grad-clip).

  * ``test_fused_adamw_invalidates_weight_caches``: the cache primitive itself, every optimizer flavour, and a
    ``p.data`` write + ``invalidate_caches()``;
  * ``test_linear_after_fused_adamw_steps``: one kernel pair (``A.linear``) stepped 3 times, forward vs ``F.linear``
    on the CURRENT weights after every step (was 1.5e-5 / 3.68 / 7.3 before the fix);
  * ``test_three_fused_adamw_steps_then_oracle``: 3 steps of AdamW(fused) + clip on the (shrunk) detector, then one
    more ``forward_train`` + backward on the UPDATED weights vs the CPU oracle on replayed noise: every loss <= 1e-3,
    whole gradient vector <= 1e-3 relative L2 (measured before the fix: 0.63 relative on ``loss_mask``)."""
import pytest
import torch
import torch.nn.functional as F

import occformer_amd  # noqa: F401
import occformer_amd.ops as ops_mod
from occformer_amd import autograd as A
from occformer_amd import fused, noise
from occformer_amd.registry import build_model
from occformer_amd.training import DeviceRNG
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


def _optimizers(params):
    yield "adamw_fused", torch.optim.AdamW(params, lr=0.05, fused=True)
    yield "adamw_foreach", torch.optim.AdamW(params, lr=0.05, foreach=True)
    yield "sgd", torch.optim.SGD(params, lr=0.05)


def test_fused_adamw_invalidates_weight_caches():
    for name, _ in _optimizers([torch.nn.Parameter(torch.zeros(1))]):
        p = torch.nn.Parameter(torch.arange(12.0).view(3, 4))
        opt = dict(_optimizers([p]))[name]
        cache = {}
        v0 = fused._versioned(cache, p, lambda: p.detach().clone())
        assert fused._versioned(cache, p, lambda: None) is v0               # a hit while nothing changed
        p.grad = torch.ones_like(p)
        opt.step()
        v1 = fused._versioned(cache, p, lambda: p.detach().clone())
        assert v1 is not v0 and torch.equal(v1, p.detach()), name
    # out-of-band write: .data bypasses the version counter and no optimizer is involved
    p = torch.nn.Parameter(torch.ones(4))
    cache = {}
    fused._versioned(cache, p, lambda: p.detach().clone())
    p.data.mul_(2.0)
    fused.invalidate_caches()
    assert torch.equal(fused._versioned(cache, p, lambda: p.detach().clone()), p.detach())


def test_linear_after_fused_adamw_steps(bound):
    d = bound.device
    lin = torch.nn.Linear(64, 96).to(d)
    x = paramgen.tensor("ms_lin_x", (256, 64), 3).to(d).requires_grad_(True)
    opt = torch.optim.AdamW(lin.parameters(), lr=0.05, fused=True)
    for step in range(4):
        y = A.linear(x, lin)
        ref = F.linear(x.detach(), lin.weight.detach(), lin.bias.detach())
        err = float((y.detach() - ref).abs().max())
        assert err < 2e-4, (step, err)
        opt.zero_grad(set_to_none=True)
        (y * y).mean().backward()
        gx = x.grad.clone()
        x.grad = None
        # the data gradient reads W^T (another cached layout): check it against autograd of the plain formulation
        xr = x.detach().clone().requires_grad_(True)
        (F.linear(xr, lin.weight.detach(), lin.bias.detach()) ** 2).mean().backward()
        assert float((gx - xr.grad).abs().max()) < 2e-4 * max(1.0, float(xr.grad.abs().max())), step
        opt.step()


def _small():
    cfg, meta = tinycfg.tiny_nusc(ncams=2)
    cfg["pts_bbox_head"]["transformer_decoder"]["num_layers"] = 3
    cfg["img_bev_encoder_backbone"]["block_numbers"] = [1, 1, 1, 1]
    cfg["img_bev_encoder_neck"]["encoder"]["num_layers"] = 1
    tc = train_cfg(num_points=2048)        # (see tests/test_train_step._setup: a swapped sampling point weighs 1 / num_points)
    cfg["train_cfg"] = dict(pts=tc)
    cfg["test_cfg"] = None
    meta = dict(meta, pd_layers=1, dec_layers=3, block_numbers=(1, 1, 1, 1))
    return cfg, meta, tc


def small_sample(meta, r=0):
    B, N = 1, 2
    cams = paramgen.camera_rig(B, N, *meta["input_size"], meta["focal"], seed=30 + r)
    x = paramgen.tensor(f"ms_x{r}", (B, N, 32, meta["fH"], meta["fW"]), 5)
    _, _, gt_occ, pts = inputs("nusc")
    H, W = meta["input_size"]
    gd = paramgen.uniform(f"ms_d{r}", (B, N, H, W), 5) * 12.0
    gd = torch.where(paramgen.uniform(f"ms_k{r}", (B, N, H, W), 6) < 0.05, gd, torch.zeros_like(gd))
    return cams, x, gt_occ[r:r + 1], [pts[r]], gd


def oracle_step(cfg, meta, tc, sd, cams, x, gt_occ, pts, gd, rng, **kw):
    ocfg = dict(D=meta["D"], C=meta["C"], groups=meta["groups"], heads=meta["heads"], pd_layers=meta["pd_layers"],
                dec_layers=meta["dec_layers"], downsample=16, block_numbers=meta["block_numbers"],
                dbound=cfg["img_view_transformer"]["grid_config"]["dbound"], head=oracle_cfg(cfg["pts_bbox_head"], tc))
    return T.train_step(sd, x, cams, gd, gt_occ, pts, ocfg, rng=rng, **kw)


def test_three_fused_adamw_steps_then_oracle(bound):
    """On the GPU the three optimizer steps run REPRODUCIBLY (``ops.deterministic`` + the switches of
    test_training_step_is_reproducible): with float-atomic scatter sums in their backward the weights the 4th step
    starts from differed in their low bits from run to run, and one such problem instance in ~20 sits next to a discrete
    decision of the matcher / samplers the tape does not cover -- whole gradient 4.8e-3 instead of the usual 4e-5 ... 4e-4
    (profiles/r06/r06ag_multistep_*.txt: the same rate with the round's arithmetic cuts switched off).  A fixed instance
    gives a fixed verdict."""
    be = bound
    d = be.device
    det = _Deterministic(be) if torch.device(d).type == "cuda" else None
    try:
        _three_steps_then_oracle(be, d)
    finally:
        if det is not None:
            det.restore()


class _Deterministic:
    def __init__(self, be):
        from occformer_amd import view_transformer
        self.vt, self.ops = view_transformer, be.ops
        self.saved = (view_transformer._DEPTHNET_LIB, torch.backends.cudnn.deterministic, be.ops.deterministic)
        view_transformer._DEPTHNET_LIB = "1"
        torch.backends.cudnn.deterministic = True
        be.ops.deterministic = True

    def restore(self):
        self.vt._DEPTHNET_LIB, torch.backends.cudnn.deterministic, self.ops.deterministic = self.saved


def _three_steps_then_oracle(be, d):
    cfg, meta, tc = _small()
    model = build_model(cfg)
    model.load_state_dict(paramgen.fill_state_dict(model.state_dict(), 77))
    model = model.to(d).train()
    cams, x, gt_occ, pts, gd = small_sample(meta)
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    img_inputs = [t.to(d) for t in (x, *cams)] + [gd.to(d)]
    kw = dict(img_metas=metas, img_inputs=img_inputs, gt_occ=gt_occ.to(d), points_occ=[p.to(d) for p in pts])
    params = [p for p in model.parameters() if p.requires_grad]
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    opt = torch.optim.AdamW(params, lr=2e-3, weight_decay=0.01, fused=True)      # the bench's optimizer, a large lr
    noise.set_rng(DeviceRNG(d, seed=5))
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        losses = model(return_loss=True, **kw)
        sum(v for k, v in losses.items() if "loss" in k).backward()
        torch.nn.utils.clip_grad_norm_(params, 5.0)
        opt.step()
    moved = sorted(float((p.detach() - before[k]).norm() / before[k].norm().clamp_min(1e-9))
                   for k, p in model.named_parameters() if p.requires_grad and p.dim() > 1)
    assert moved[len(moved) // 2] > 5e-3, "the optimizer steps are too small to expose stale weight layouts"

    # the 4th forward / backward on the UPDATED weights against the oracle on identical noise: the product runs on its
    # own taped draws with every ReLU gate of the path recorded, the oracle replays the draws and differentiates with
    # those gates (oracle.occformer_ref.forced_gates, level "all": in this one-sample configuration ONE unit of a
    # GroupNorm + ReLU map gated differently on a 1e-6 difference of the device's ATen / MIOpen ops weighed 7.4e-3 of
    # the gradient -- r03/r04 ran this gate at 3e-2 on the GPU for that reason)
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    rec = noise.RecordedRNG(DeviceRNG(d, seed=6))
    noise.set_rng(rec)
    gates = noise.record_gates("all")
    opt.zero_grad(set_to_none=True)
    try:
        losses = model(return_loss=True, **kw)
    finally:
        noise.record_gates(False)
    forced = O.forced_gates(gates, level="all")
    replay = ReplayRNG(rec.tape, torch.device("cpu"))
    ref_losses, ref_grads = oracle_step(cfg, meta, tc, sd, cams, x, gt_occ, pts, gd, replay, gates=forced)
    assert replay.i == len(rec.tape) and forced.i == len(gates)
    print(f"ReLU gates (every ReLU of the path): {forced.flipped} of {forced.units} gated differently, largest |z| "
          f"among them {forced.max_abs_z:.1e} ({forced.max_rel_z:.1e} of the tensor's RMS)")
    assert forced.max_rel_z <= 1e-3 and forced.flipped <= max(8, 1e-4 * forced.units), (forced.flipped, forced.units)
    worst = max(abs(float(losses[k].detach()) - float(v.detach())) / max(1.0, abs(float(v.detach()))) for k, v in ref_losses.items())
    print("after 3 fused AdamW steps: worst relative loss difference vs the oracle", worst)
    assert worst <= TOL, {k: (float(losses[k].detach()), float(v.detach())) for k, v in ref_losses.items()}
    sum(v for k, v in losses.items() if "loss" in k).backward()
    named = dict(model.named_parameters())
    num = sum(float((named[k].grad.cpu() - g).norm() ** 2) for k, g in ref_grads.items() if g is not None)
    den = sum(float(g.norm() ** 2) for g in ref_grads.values() if g is not None)
    print("whole gradient vector: relative L2 error", (num / den) ** 0.5)
    # (stale weights showed as 0.63 on the LOSSES; emulation measured 2.9e-5 before the tape)
    assert (num / den) ** 0.5 < TOL


def test_prepared_layouts_match_the_slow_derivations(bound):
    """csrc/prep.hip (one launch over a descriptor table) against the per-parameter derivations it replaces, for every
    layout kind and weight rank, before and after a fused AdamW step (the table must be re-run, not re-built)."""
    d = bound.device
    ops = bound.ops
    ws = [torch.nn.Parameter(paramgen.tensor("prep_w%d" % i, s, 3).to(d))
          for i, s in enumerate([(24, 16), (16, 8, 3, 3, 3), (8, 12, 3, 3, 1), (10, 6, 1, 1, 1), (6, 4, 3, 3)])]
    opt = torch.optim.AdamW(ws, lr=0.1, fused=True)

    def slow(w, kind):
        w = w.detach()
        w5 = w.reshape(*w.shape, *([1] * (5 - w.dim()))) if w.dim() > 2 else w.reshape(*w.shape, 1, 1, 1)
        cout, cin = w5.shape[:2]
        if kind == "split":
            return w.reshape(cout, -1)
        if kind == "tap":
            return w5.permute(0, 2, 3, 4, 1).reshape(cout, -1)
        if kind == "wt":
            return w.reshape(cout, -1).t()
        if kind == "dg":
            return w5.permute(1, 2, 3, 4, 0).reshape(cin, -1)
        return w5.flip(2, 3, 4).permute(1, 2, 3, 4, 0).reshape(cin, -1)

    for step in range(3):
        seen = 0
        for w in ws:
            for kind in fused._WeightPrep.KINDS:
                hit = fused.prepared(w, kind)
                taps = w[0, 0].numel() if w.dim() > 2 else 1
                if kind == "wt" and taps != 1:
                    assert hit is None
                    continue
                if hit is None:          # odd inner dimension: stays on the per-parameter caches
                    continue
                seen += 1
                want = slow(w, kind).contiguous()
                f32, (hi, lo) = hit
                if kind != "split":
                    assert torch.equal(f32, want), (step, tuple(w.shape), kind)
                rhi, rlo = ops.split_bf16(want)
                assert torch.equal(hi, rhi) and torch.equal(lo, rlo), (step, tuple(w.shape), kind)
        assert seen >= 18
        for w in ws:
            w.grad = torch.ones_like(w) * (step + 1)
        opt.step()
