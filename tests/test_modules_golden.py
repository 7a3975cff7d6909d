"""Product modules (built from the reference-style config dicts through the registry)
against the golden vectors the REFERENCE produced (tests/golden/make_golden.py).  The
parameter checksum proves that the product's state-dict keys/shapes equal the
reference's: both sides are filled by tests/paramgen.py keyed on
the parameter names.

CPU: the HIP kernels run through the host emulation of their sources (test-only);
-m gpu: the real gfx950 library on cuda:0."""
import pytest
import torch

import occformer_amd
from occformer_amd import ops as ops_mod
from occformer_amd.registry import MODELS
from tests import paramgen, tinycfg
from tests.conftest import golden

TOL = dict(atol=1e-3, rtol=1e-3)      # north_star: <= 1e-3 fp32 vs the reference path


@pytest.fixture
def bound(be, monkeypatch):
    """route the product modules' get_ops() to the backend under test"""
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    return be


def _build(cfg, seed, be, expect_checksum):
    m = MODELS.build(cfg)
    sd = paramgen.fill_state_dict(m.state_dict(), seed)
    assert abs(paramgen.checksum(sd) - expect_checksum) < 1e-6 * expect_checksum, \
        "state-dict keys/shapes differ from the reference module"
    m.load_state_dict(sd)
    return m.eval().to(be.device)


def _close(a, b, what):
    a = a.detach().cpu()
    err = float((a - b).abs().max())
    assert torch.allclose(a, b, **TOL), f"{what}: max abs err {err:.3e} (ref max {float(b.abs().max()):.3e})"


@torch.no_grad()
def test_view_transformer(bound):
    g = golden("view_transformer")
    model, meta = tinycfg.tiny_nusc()
    vt = _build(model["img_view_transformer"], g["seed"], bound, g["param_checksum"])
    cams = bound.to(g["rots"], g["trans"], g["intrins"], g["post_rots"], g["post_trans"], g["bda"])
    mlp = vt.get_mlp_input(*cams)
    _close(mlp, g["mlp_input"], "mlp_input")
    vox, depth = vt([bound.to(g["x"]), *cams, mlp])
    _close(depth, g["depth"], "depth")
    _close(vox, g["voxel"], "voxel")


@torch.no_grad()
def test_encoder(bound):
    g = golden("encoder")
    model, meta = tinycfg.tiny_nusc()
    enc = _build(model["img_bev_encoder_backbone"], g["seed"], bound, g["param_checksum"])
    outs = enc(bound.to(g["x"]))
    for i, o in enumerate(outs):
        _close(o, g[f"out{i}"], f"encoder out{i}")
    # odd sizes + shifted windows + padding
    blk = enc.layers[0][1]
    _close(blk(bound.to(g["blk_in"])), g["blk_out"], "shifted block on 10x9x3")


@torch.no_grad()
def test_pixel_decoder(bound):
    g, ge = golden("pixel_decoder"), golden("encoder")
    model, meta = tinycfg.tiny_nusc()
    pd = _build(model["img_bev_encoder_neck"], g["seed"], bound, g["param_checksum"])
    outs = pd([bound.to(ge[f"out{i}"]) for i in range(4)])
    for i, o in enumerate(outs):
        _close(o, g[f"out{i}"], f"pixel decoder out{i}")


@torch.no_grad()
def test_head(bound):
    g, gp = golden("head"), golden("pixel_decoder")
    model, meta = tinycfg.tiny_nusc()
    head = _build(dict(model["pts_bbox_head"], train_cfg=None, test_cfg=None), g["seed"], bound,
                  g["param_checksum"])
    feats = [bound.to(gp[f"out{i}"]) for i in range(4)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"]) for _ in range(2)]
    cls_list, mask_list = head(feats, metas)
    _close(cls_list[0], g["cls0"], "cls0")
    _close(mask_list[0], g["mask0"], "mask0")
    _close(cls_list[-1], g["cls_last"], "cls_last")
    _close(mask_list[-1], g["mask_last"], "mask_last")
    res = head.simple_test(feats, metas, points=[bound.to(g["pts0"]), bound.to(g["pts1"])])
    _close(res["output_voxels"][0], g["output_voxels"], "output_voxels")
    _close(res["output_points"], g["output_points"], "output_points")


@torch.no_grad()
def test_head_stacked_kv_projection(bound, monkeypatch):
    """batch 1: the cross-attention key / value projections of a level's layers run as one stacked GEMM; the
    predictions must equal the per-layer projections (OCCF_STACK_KV=0) to GEMM rounding"""
    g, gp = golden("head"), golden("pixel_decoder")
    model, meta = tinycfg.tiny_nusc()
    head = _build(dict(model["pts_bbox_head"], train_cfg=None, test_cfg=None), g["seed"], bound,
                  g["param_checksum"])
    feats = [bound.to(gp[f"out{i}"])[:1].contiguous() for i in range(4)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"])]
    monkeypatch.setenv("OCCF_STACK_KV", "1")
    if head._project_level_tokens(*[[f.flatten(2).transpose(1, 2).contiguous() for f in feats[1:]]] * 2) is None:
        pytest.skip("stacked projection not applicable in this precision mode / geometry")
    cls_a, mask_a = head(feats, metas)
    monkeypatch.setenv("OCCF_STACK_KV", "0")
    cls_b, mask_b = head(feats, metas)
    for a, b in zip(cls_a + mask_a, cls_b + mask_b):
        assert torch.allclose(a.cpu(), b.cpu(), atol=2e-4, rtol=1e-4), float((a.cpu() - b.cpu()).abs().max())
    # and both agree with the reference's batch-2 golden output on sample 0
    _close(mask_a[-1], g["mask_last"][:1], "mask_last (stacked, batch 1)")


@torch.no_grad()
def test_view_transformer_kitti(bound):
    """SemanticKITTI form (BASELINE configs 0-1): one camera, 4x4 intrinsics / BEV augmentation, 33 camera
    scalars -- against the reference's own module (tests/golden/make_golden_kitti.py)"""
    g = golden("view_transformer_kitti")
    model, meta = tinycfg.tiny_nusc(ncams=1)
    vt = _build(dict(model["img_view_transformer"], cam_channels=33), g["seed"], bound, g["param_checksum"])
    cams = bound.to(g["rots"], g["trans"], g["intrins"], g["post_rots"], g["post_trans"], g["bda"])
    mlp = vt.get_mlp_input(*cams)
    assert mlp.shape[-1] == 33
    _close(mlp, g["mlp_input"], "mlp_input")
    vox, depth = vt([bound.to(g["x"]), *cams, mlp])
    _close(depth, g["depth"], "depth")
    _close(vox, g["voxel"], "voxel")


@torch.no_grad()
def test_head_kitti(bound):
    """Mask2FormerOccHead.simple_test (SemanticKITTI: 20 classes, no LiDAR branch) against the reference's own
    module on the pixel-decoder vectors (tests/golden/make_golden_kitti.py)"""
    from tests.golden.make_golden_train import kitti_head_cfg
    g, gp = golden("head_kitti"), golden("pixel_decoder")
    model, meta = tinycfg.tiny_nusc()
    head = _build(dict(kitti_head_cfg(model), train_cfg=None, test_cfg=None), g["seed"], bound, g["param_checksum"])
    feats = [bound.to(gp[f"out{i}"]) for i in range(4)]
    metas = [dict(occ_size=meta["occ_size"], pc_range=meta["pc_range"]) for _ in range(2)]
    cls_list, mask_list = head(feats, metas)
    _close(cls_list[-1], g["cls_last"], "cls_last")
    _close(mask_list[-1], g["mask_last"], "mask_last")
    res = head.simple_test(feats, metas)
    assert res["output_points"] is None
    _close(res["output_voxels"][0], g["output_voxels"], "output_voxels")
