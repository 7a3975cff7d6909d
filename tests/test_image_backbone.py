"""Image-branch glue of the SemanticKITTI configs: ``CustomEfficientNet`` on plain PyTorch,
against vectors the REFERENCE's module produced (tests/golden/make_golden_image.py)."""
import os

import numpy as np
import pytest
import torch

from occformer_amd.registry import MODELS, Config, build_model
from tests import paramgen, refshim
from tests.conftest import GOLDEN


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLDEN, "efficientnet.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@pytest.mark.parametrize("arch,out_indices", [("b0", (2, 3, 4, 5, 6)), ("b2", (1, 3, 6))])
@torch.no_grad()
def test_efficientnet_matches_reference_vectors(gold, arch, out_indices):
    m = MODELS.build(dict(type="CustomEfficientNet", arch=arch, drop_path_rate=0.2, out_indices=out_indices,
                          frozen_stages=0, norm_eval=False, with_cp=True))
    sd = paramgen.fill_state_dict(m.state_dict(), int(gold[f"{arch}.seed"]))
    want = float(gold[f"{arch}.param_checksum"])
    assert abs(paramgen.checksum(sd) - want) < 1e-6 * want, "state-dict keys / shapes differ from the reference module"
    m.load_state_dict(sd)
    m.eval()
    outs = m(torch.from_numpy(gold[f"{arch}.x"]))
    assert len(outs) == len(out_indices)
    for i, o in zip(out_indices, outs):
        ref = torch.from_numpy(gold[f"{arch}.out{i}"])
        assert o.shape == ref.shape
        assert torch.allclose(o, ref, atol=2e-5, rtol=1e-4), (arch, i, float((o - ref).abs().max()))


def test_efficientnet_b7_checkpoint_keys(gold):
    """the b7 of occformer_kitti.py: every key (and size) of the reference module's state dict, i.e. of the
    released mmcls checkpoint after its 'backbone.' prefix is stripped"""
    m = MODELS.build(dict(type="CustomEfficientNet", arch="b7", drop_path_rate=0.2, out_indices=(2, 3, 4, 5, 6),
                          with_cp=True))
    sd = m.state_dict()
    assert sorted(sd) == [str(k) for k in gold["b7.keys"]]
    assert [sd[k].numel() for k in sorted(sd)] == [int(n) for n in gold["b7.numel"]]
    x = torch.zeros(1, 3, 64, 96)
    with torch.no_grad():
        chans = [o.shape[1] for o in m.eval()(x)]
    assert chans == [48, 80, 224, 640, 2560]            # img_neck.in_channels of the config


@pytest.mark.skipif(not refshim.available(), reason="needs /root/reference (build container)")
def test_kitti_reference_config_loads_unchanged():
    cfg = Config.fromfile(os.path.join(refshim.REFERENCE_ROOT, "projects/configs/occformer_kitti/occformer_kitti.py"))
    assert cfg.model.img_backbone.type == "CustomEfficientNet" and cfg.model.pts_bbox_head.type == "Mask2FormerOccHead"
    m = build_model(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    names = set(m.state_dict())
    for prefix in ("img_backbone.layers.0.conv.weight", "img_neck.deblocks.0.0.weight",
                   "img_view_transformer.depth_net.reduce_conv.0.weight", "img_bev_encoder_backbone.layers.0.0.",
                   "img_bev_encoder_neck.", "pts_bbox_head.transformer_decoder.layers.0.attentions.0.attn.in_proj_weight"):
        assert any(n.startswith(prefix) for n in names), prefix
    assert m.pts_bbox_head.num_occupancy_classes == 20 and m.img_view_transformer.cam_channels == 33


# ------------------------------------------------------------------ DCNv2 of the R101-DCN image backbone
def _dcn(cin=6, cout=5, stride=1, dg=1):
    from occformer_amd.detector import ModulatedDeformConv2dPack
    m = ModulatedDeformConv2dPack(cin, cout, 3, stride=stride, padding=1, deform_groups=dg)
    with torch.no_grad():
        m.weight.copy_(paramgen.tensor("dcn.w", m.weight.shape, 1, 0.3))
    return m


@torch.no_grad()
def test_dcnv2_product_has_no_cpu_path():
    """without the oracle's stand-in (tests/conftest.py installs it) a CPU tensor raises: the grid_sample formulation
    that used to sit in the module as a silent second backend lives in oracle/dcn_ref.py"""
    from occformer_amd.detector import ModulatedDeformConv2dPack
    m = _dcn()
    prev, ModulatedDeformConv2dPack.cpu_reference = ModulatedDeformConv2dPack.cpu_reference, None
    try:
        with pytest.raises(RuntimeError, match="no CPU path"):
            m(paramgen.tensor("dcn.x", (1, 6, 5, 5), 2))
    finally:
        ModulatedDeformConv2dPack.cpu_reference = prev


@torch.no_grad()
def test_dcnv2_zero_init_is_half_a_convolution():
    """freshly built (conv_offset = 0): no displacement, modulation sigmoid(0) = 0.5"""
    for stride in (1, 2):
        m = _dcn(stride=stride)
        x = paramgen.tensor("dcn.x", (2, 6, 9, 11), 2)
        ref = 0.5 * torch.nn.functional.conv2d(x, m.weight, None, stride, 1)
        assert torch.allclose(m(x), ref, atol=1e-5, rtol=1e-5)


@torch.no_grad()
def test_dcnv2_against_direct_bilinear_loops():
    """random offsets / modulation: every output element recomputed with explicit bilinear taps (zero outside)"""
    import math
    m = _dcn(cin=4, cout=3, stride=2, dg=2)
    m.conv_offset.weight.copy_(paramgen.tensor("dcn.ow", m.conv_offset.weight.shape, 3, 0.15))
    m.conv_offset.bias.copy_(paramgen.tensor("dcn.ob", m.conv_offset.bias.shape, 3, 0.8))
    x = paramgen.tensor("dcn.x2", (1, 4, 6, 7), 4)
    out = m(x)
    raw = m.conv_offset(x)
    o1, o2, logit = torch.chunk(raw, 3, 1)
    off, mask = torch.cat((o1, o2), 1), torch.sigmoid(logit)
    B, C, H, W = x.shape
    Ho, Wo = out.shape[-2:]

    def tap(c, y, xx):
        y0, x0 = math.floor(y), math.floor(xx)
        v = 0.0
        for yy, wy in ((y0, 1 - (y - y0)), (y0 + 1, y - y0)):
            for xc, wx in ((x0, 1 - (xx - x0)), (x0 + 1, xx - x0)):
                if 0 <= yy < H and 0 <= xc < W:
                    v += wy * wx * float(x[0, c, yy, xc])
        return v

    ref = torch.zeros_like(out)
    for oy in range(Ho):
        for ox in range(Wo):
            for t in range(9):
                ky, kx = divmod(t, 3)
                for c in range(C):
                    g = c // (C // 2)
                    dy = float(off[0, g * 18 + 2 * t, oy, ox])
                    dx = float(off[0, g * 18 + 2 * t + 1, oy, ox])
                    v = tap(c, oy * 2 - 1 + ky + dy, ox * 2 - 1 + kx + dx) * float(mask[0, g * 9 + t, oy, ox])
                    ref[0, :, oy, ox] += m.weight[:, c, ky, kx] * v
    assert torch.allclose(out, ref, atol=1e-5, rtol=1e-4), float((out - ref).abs().max())


@pytest.mark.skipif(not refshim.available(), reason="needs /root/reference (build container)")
def test_r101_dcn_reference_config_loads_unchanged():
    cfg = Config.fromfile(os.path.join(refshim.REFERENCE_ROOT,
                                       "projects/configs/occformer_nusc/occformer_nusc_r101_896x1600.py"))
    assert cfg.model.img_backbone.dcn.type == "DCNv2"
    m = build_model(cfg.model, train_cfg=cfg.get("train_cfg"), test_cfg=cfg.get("test_cfg"))
    names = set(m.state_dict())
    # DCN in stages 3 and 4 only, with mmdet's parameter names
    assert "img_backbone.layer3.0.conv2.conv_offset.weight" in names and "img_backbone.layer4.2.conv2.weight" in names
    assert "img_backbone.layer2.0.conv2.conv_offset.weight" not in names
    assert m.img_backbone.layer3[0].conv2.conv_offset.out_channels == 27 and len(m.img_backbone.layer3) == 23


@pytest.mark.parametrize("stride,dg", [(1, 1), (2, 2)])
def test_dcnv2_hip_im2col_vs_grid_sample_formulation(be, stride, dg):
    """DCNv2 on the kernels (csrc/dcn.hip modulated im2col + GEMM) against the module's own differentiable
    grid_sample statement, which test_dcnv2_against_direct_bilinear_loops pins to explicit bilinear loops"""
    m = _dcn(cin=16, cout=8, stride=stride, dg=dg)
    with torch.no_grad():
        m.conv_offset.weight.copy_(paramgen.tensor("dcn2.ow", m.conv_offset.weight.shape, 3, 0.1))
        m.conv_offset.bias.copy_(paramgen.tensor("dcn2.ob", m.conv_offset.bias.shape, 3, 0.8))
        x = paramgen.tensor("dcn2.x", (2, 16, 9, 11), 4)
        ref = m(x)                                                     # CPU tensors: grid_sample path
        o1, o2, logit = torch.chunk(m.conv_offset(x), 3, dim=1)
        off, mask = torch.cat((o1, o2), 1).contiguous(), torch.sigmoid(logit).contiguous()
        col = be.ops.deform_im2col(be.to(x.permute(0, 2, 3, 1).contiguous()), be.to(off), 3, stride, 1, 1, 1, dg,
                                   mask=be.to(mask))
        w2 = m.weight.permute(0, 2, 3, 1).reshape(8, -1).contiguous()
        out = be.ops.linear(col.flatten(1), be.to(w2)).cpu()
    Ho, Wo = ref.shape[-2:]
    assert torch.allclose(out.view(2, Ho, Wo, 8).permute(0, 3, 1, 2), ref, atol=1e-4, rtol=1e-4)


@pytest.mark.parametrize("stride,dg", [(1, 1), (2, 2)])
def test_dcnv2_training_step_on_the_kernels(be, stride, dg, monkeypatch):
    """DCNv2 forward + backward as ONE autograd node on csrc/dcn.hip (modulated im2col, occf_modulated_deform_col2im:
    dx, doffset, dmask) and the split-bf16 contractions, against torch autograd through the module's grid_sample
    statement: output and the gradients of the input, the weight and conv_offset (through offsets AND modulation)"""
    import occformer_amd.ops as ops_mod
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    m = _dcn(cin=16, cout=8, stride=stride, dg=dg)
    with torch.no_grad():
        m.conv_offset.weight.copy_(paramgen.tensor("dcn4.ow", m.conv_offset.weight.shape, 3, 0.1))
        m.conv_offset.bias.copy_(paramgen.tensor("dcn4.ob", m.conv_offset.bias.shape, 3, 0.8))
    x0 = paramgen.tensor("dcn4.x", (2, 16, 9, 11), 4)
    dy = None
    res = []
    for hipmode in (False, True):
        mm = m if not hipmode else m.to(be.device)
        mm.force_hip = hipmode
        for p in mm.parameters():
            p.grad = None
        x = x0.clone().to(be.device if hipmode else "cpu").requires_grad_()
        y = mm(x)
        if dy is None:
            dy = paramgen.tensor("dcn4.dy", tuple(y.shape), 5)
        y.backward(dy.to(y.device))
        # (clones: Module.to() moves .grad tensors in place)
        res.append([t.detach().cpu().clone() for t in (y, x.grad, mm.weight.grad, mm.conv_offset.weight.grad,
                                                       mm.conv_offset.bias.grad)])
    for a, b, name in zip(res[1], res[0], ("y", "dx", "dweight", "dconv_offset.weight", "dconv_offset.bias")):
        assert float((a - b).norm() / b.norm().clamp_min(1e-12)) < 2e-4, name


@pytest.mark.gpu
def test_dcnv2_module_on_gpu_uses_the_kernel(hip):
    """ModulatedDeformConv2dPack on a GPU tensor (R101 stage-3 width) == its CPU grid_sample statement"""
    m = _dcn(cin=256, cout=256, stride=1, dg=1)
    with torch.no_grad():
        m.conv_offset.weight.copy_(paramgen.tensor("dcn3.ow", m.conv_offset.weight.shape, 3, 0.02))
        m.conv_offset.bias.copy_(paramgen.tensor("dcn3.ob", m.conv_offset.bias.shape, 3, 0.8))
        x = paramgen.tensor("dcn3.x", (2, 256, 14, 25), 4)
        ref = m(x)
        calls = []
        orig = hip.ops.deform_im2col
        hip.ops.deform_im2col = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            out = m.to(hip.device)(x.to(hip.device)).cpu()
            # the bf16 image branch (bench.py --image-dtype bf16: autocast) crosses the layer as an fp32 island on the
            # SAME kernels (it used to fall to a grid_sample composition)
            x16 = x.to(hip.device).bfloat16()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out16 = m(x16)
            # the island includes the offset convolution and the sigmoid (under autocast they returned
            # bf16-rounded sampling positions): the layer on the bf16-ROUNDED input in plain fp32 is the same
            # computation up to the one rounding of the result
            ref16 = m(x16.float()).cpu()
        finally:
            hip.ops.deform_im2col = orig
    assert len(calls) == 3 and out16.dtype == torch.bfloat16
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-4
    assert float((out16.float().cpu() - ref).abs().max() / ref.abs().max()) < 3e-2
    assert float((out16.float().cpu() - ref16).abs().max() / ref16.abs().max()) < 5e-3     # 2^-8 of the result's rounding


@pytest.mark.gpu
def test_image_branch_gpu_vs_cpu(hip):
    """R50 + SECONDFPN (the nuScenes image branch, PyTorch-ROCm / MIOpen glue outside the hand-written path) on the
    GPU: fp32 equals the CPU run of the same module; the bf16 channels_last mode of `bench.py --from-images` stays
    within bf16 accuracy.  (First use compiles / looks up the MIOpen solvers: the test warms them itself.)"""
    from occformer_amd import configs
    from occformer_amd.registry import build_model
    torch.manual_seed(0)
    cfg, meta = configs.nusc_r50("reference", with_image_branch=True)
    model = build_model(cfg).eval()
    img = paramgen.tensor("imgbr.x", (1, 2, 3, 128, 352), 1)
    with torch.no_grad():
        ref = model.image_encoder(img)
        model = model.to(hip.device)
        out = model.image_encoder(img.to(hip.device)).cpu()
        assert float((out - ref).abs().max() / ref.abs().max()) < 1e-3
        model.image_dtype = torch.bfloat16
        model.img_backbone.to(memory_format=torch.channels_last)
        model.img_neck.to(memory_format=torch.channels_last)
        out16 = model.image_encoder(img.to(hip.device)).cpu()
        assert float((out16 - ref).abs().max() / ref.abs().max()) < 5e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("with_res,relu", [(True, True), (False, True), (False, False)])
@pytest.mark.parametrize("shape", [(3, 24, 5, 7), (2, 64, 23, 9)])      # (a thread walks the channel groups / stays on one)
def test_scale_shift_act_epilogue(be, dtype, with_res, relu, shape):
    """csrc/image_epilogue.hip: eval BatchNorm + identity add + ReLU of the image branch as one in-place pass over a
    channels_last feature map (fp32 and bf16), against the module sequence it replaces"""
    C = shape[1]
    y = paramgen.tensor("ie.y", shape, 1, 1.5).to(dtype).contiguous(memory_format=torch.channels_last)
    r = paramgen.tensor("ie.r", shape, 2).to(dtype).contiguous(memory_format=torch.channels_last)
    scale = 1 + 0.3 * paramgen.tensor("ie.s", (C,), 3)
    shift = 0.2 * paramgen.tensor("ie.b", (C,), 4)
    ref = y.float() * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if with_res:
        ref = ref + r.float()
    if relu:
        ref = torch.relu(ref)
    yd = be.to(y)
    out = be.ops.scale_shift_act(yd, be.to(scale), be.to(shift), be.to(r) if with_res else None, relu)
    assert out.data_ptr() == yd.data_ptr() and out.dtype == dtype
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert float((out.float().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())


@pytest.mark.gpu
def test_resnet_fused_inference_route_equals_the_module_route(hip, monkeypatch):
    """ResNet-50 + SECONDFPN in eval mode on the GPU: the fused route (MIOpen convolution + csrc/image_epilogue.hip, the
    default) against the module-by-module route (OCCF_IMAGE_FUSE=0), fp32 and the bf16 channels_last mode"""
    import occformer_amd.detector as D
    from occformer_amd.registry import MODELS
    from occformer_amd import configs
    bb_cfg, neck_cfg = configs.image_branch("nusc_r50_200")
    torch.manual_seed(0)
    bb, neck = MODELS.build(bb_cfg).to(hip.device).eval(), MODELS.build(neck_cfg).to(hip.device).eval()
    for m in list(bb.modules()) + list(neck.modules()):
        if isinstance(m, torch.nn.BatchNorm2d):              # non-trivial running statistics
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 64, 96, device=hip.device)
    calls = []
    orig = hip.ops.scale_shift_act
    monkeypatch.setattr(hip.ops, "scale_shift_act", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
    with torch.no_grad():
        monkeypatch.setattr(D, "_IMAGE_FUSE", False)
        ref = neck(bb(x))[0]
        assert not calls
        monkeypatch.setattr(D, "_IMAGE_FUSE", True)
        out = neck(bb(x))[0]
        assert len(calls) == 53 + 4                          # 53 backbone BatchNorms (stem, 16 x 3, 4 downsample) + 4 deblocks
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out16 = neck(bb(x.contiguous(memory_format=torch.channels_last)))[0]
    assert out.shape == ref.shape
    assert float((out - ref).abs().max() / ref.abs().max()) < 1e-4
    assert float((out16.float() - ref).abs().max() / ref.abs().max()) < 6e-2
