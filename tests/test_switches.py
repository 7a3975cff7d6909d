"""Every OCCF_* switch of the host layer (ops.py reads them at construction) selects a different kernel path for the
same math: the tiny forward must give the same outputs under each value, so the non-default paths cannot rot
unnoticed.  Runs on the host emulation (CPU) and on the real library (-m gpu)."""
import pytest
import torch

import occformer_amd  # noqa: F401
import occformer_amd.ops as ops_mod
from occformer_amd.registry import MODELS
from tests import paramgen, tinycfg

SWITCHES = [
    {},                                        # defaults
    {"use_halo_conv": False},                  # OCCF_HALO_CONV=0
    {"use_fused_swin": False},                 # OCCF_FUSED_SWIN=0
    {"use_fused_mlp": False},                  # OCCF_FUSED_MLP=0
    {"use_fused_mask_pool": False},            # OCCF_FUSED_MASK_POOL=0
    {"precision": "f32"},                      # OCCF_PRECISION=f32
    {"halo_frag": False},                      # OCCF_HALO_FRAG=0: weight slabs through LDS instead of global fragments
    {"swin_frag": False},                      # OCCF_SWIN_FRAG=0: row-major weights in the fused Swin kernel
    {"use_wino": False},                       # OCCF_WINO=0: the direct 27-tap halo kernel instead of Winograd F(2, 3)
    {"use_decoder_rows": False},               # OCCF_DECODER_ROWS=0: the decoder's per-query chain as one launch per op
]


def _forward(dev):
    cfg, meta = tinycfg.tiny_nusc()
    cfg["pts_bbox_head"].update(train_cfg=None, test_cfg=None)
    mods = {}
    for i, key in enumerate(("img_view_transformer", "img_bev_encoder_backbone", "img_bev_encoder_neck",
                             "pts_bbox_head")):
        m = MODELS.build(cfg[key])
        m.load_state_dict(paramgen.fill_state_dict(m.state_dict(), 10 + i))
        mods[key] = m.eval().to(dev)
    cams = [c.to(dev) for c in paramgen.camera_rig(1, 3, *meta["input_size"], meta["focal"], seed=3)]
    x = paramgen.tensor("sw_x", (1, 3, 32, meta["fH"], meta["fW"]), 3).to(dev)
    lo, hi = torch.tensor(meta["pc_range"][:3]), torch.tensor(meta["pc_range"][3:])
    pts = [(paramgen.uniform("sw_pts", (200, 3), 3) * (hi - lo) + lo).to(dev)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    with torch.no_grad():
        vt = mods["img_view_transformer"]
        vox, _ = vt([x, *cams, vt.get_mlp_input(*cams)])
        feats = mods["img_bev_encoder_neck"](mods["img_bev_encoder_backbone"](vox))
        res = mods["pts_bbox_head"].simple_test(feats, metas, points=pts)
    return res["output_voxels"][0].cpu(), res["output_points"].cpu()


def test_every_switch_gives_the_same_forward(be, monkeypatch):
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    saved = {k: getattr(be.ops, k) for s in SWITCHES for k in s}
    # (the host emulation is ~100x slower than the chip: the CPU run covers the three paths with their own kernels,
    # the GPU run all of them)
    todo = SWITCHES if be.kind == "hip" else [SWITCHES[0], SWITCHES[1], SWITCHES[2], SWITCHES[5], SWITCHES[6],
                                              SWITCHES[8], SWITCHES[9]]
    try:
        ref = None
        for sw in todo:
            for k, v in saved.items():
                setattr(be.ops, k, v)
            for k, v in sw.items():
                setattr(be.ops, k, v)
            out = _forward(be.device)
            if ref is None:
                ref = out
            else:
                assert float((out[0] - ref[0]).abs().max()) < 1e-3, sw
                assert float((out[1] - ref[1]).abs().max()) < 1e-3, sw
    finally:
        for k, v in saved.items():
            setattr(be.ops, k, v)


@pytest.mark.parametrize("switch", ["lazy_logits_off", "depthnet_lib_off", "depthnet_lib_on"])
def test_training_switches_give_the_same_step(be, monkeypatch, switch):
    """The two switches of the TRAINING graph whose defaults changed in round 4: ``OCCF_LAZY_LOGITS`` (default
    1: dense mask logits contracted only for the matched rows) and ``OCCF_DEPTHNET_LIB`` (default "auto": DepthNet's
    training convolutions on the library's kernels from 4 096 rows up, GPU tensors only).  The non-default sides --
    dense logits, DepthNet forced to ATen / forced to the library -- must give the same losses and gradients on the
    same noise, so that neither fallback rots behind its default."""
    from occformer_amd import noise, view_transformer
    from occformer_amd.training import DeviceRNG
    from tests.test_train_step import _setup
    if be.kind == "emu" and switch == "depthnet_lib_off":
        pytest.skip('on CPU tensors "auto" already is ATen: the pair is the GPU leg\'s')
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    cfg, meta, tc, model, sd, cams, x, gt_occ, pts, gd = _setup(B=1, N=2)
    d = be.device
    model = model.to(d).train()
    kw = dict(img_metas=[dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])],
              img_inputs=[t[:1].to(d) for t in (x, *cams)] + [gd[:1].to(d)], gt_occ=gt_occ[:1].to(d),
              points_occ=[pts[0].to(d)])
    if torch.device(d).type == "cuda":
        # both steps REPRODUCIBLE (fixed-point scatter sums, deterministic library algorithms): with float atomics in the
        # backward the pair's difference moved from run to run -- 1e-3 ... 4e-3 against the 3e-3 below, one failure in four
        # visits of round 6 (r06fin2) -- so the verdict was a coin with a small head.  Now it is one fixed number.
        monkeypatch.setattr(be.ops, "deterministic", True)
        monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)

    def step():
        for p in model.parameters():
            p.grad = None
        noise.set_rng(DeviceRNG(d, 11))
        try:
            losses = model(return_loss=True, **kw)
            sum(v for k, v in losses.items() if "loss" in k).backward()
        finally:
            noise.set_rng(None)
        return ({k: float(v.detach()) for k, v in losses.items()},
                {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None})

    ref_l, ref_g = step()
    if switch == "lazy_logits_off":
        assert model.pts_bbox_head.lazy_train_logits, "the default is the lazy contraction"
        monkeypatch.setattr(type(model.pts_bbox_head), "lazy_train_logits", False)
    else:
        monkeypatch.setattr(view_transformer, "_DEPTHNET_LIB", "0" if switch == "depthnet_lib_off" else "1")
    l, g = step()
    for k, v in ref_l.items():
        assert abs(l[k] - v) <= 1e-4 * max(1.0, abs(v)), (switch, k, l[k], v)
    assert g.keys() == ref_g.keys()
    num = sum(float((g[k] - v).norm() ** 2) for k, v in ref_g.items())
    den = sum(float(v.norm() ** 2) for v in ref_g.values())
    # (the library's kernels and ATen sum in different orders; a flipped ReLU gate of this tiny model weighs ~1e-3)
    print(f"{switch}: whole gradient of the pair, relative L2 {(num / den) ** 0.5:.3e}")
    assert (num / den) ** 0.5 < 3e-3, (switch, (num / den) ** 0.5)
