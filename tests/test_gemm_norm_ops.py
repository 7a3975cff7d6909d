"""MFMA GEMM / implicit-GEMM conv, GroupNorm, LayerNorm and dual-path fusion kernels against
plain PyTorch fp32 (the floating-point reference of the same op) and the oracle."""
import pytest
import torch
import torch.nn.functional as F

from tests import paramgen

TOL = dict(atol=3e-5, rtol=1e-4)


def conv_weight_tapmajor(w):
    """[Cout, Cin, kX, kY, kZ] -> [Cout, kX*kY*kZ*Cin]"""
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous()


@pytest.mark.parametrize("M,N,K,act,use_bias,use_res", [
    (300, 96, 64, 0, True, False),          # 64x64 tile, ragged M and N
    (130, 130, 32, 1, True, True),          # asymmetric-tail N > 128
    (1, 18, 192, 0, True, False),           # single row (decoder-style)
    (257, 64, 48, 2, False, True),
])
def test_linear(be, M, N, K, act, use_bias, use_res):
    x = paramgen.tensor("x", (M, K), 1)
    w = paramgen.tensor("w", (N, K), 2, K ** -0.5)           # asymmetric: catches transposes
    b = paramgen.tensor("b", (N,), 3) if use_bias else None
    r = paramgen.tensor("r", (M, N), 4) if use_res else None
    ref = F.linear(x, w, b)
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    if use_res:
        ref = ref + r
    out = be.ops.linear(be.to(x), be.to(w), be.to(b) if use_bias else None, act,
                        be.to(r) if use_res else None).cpu()
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())


def test_linear_identity_asymmetric(be):
    """A = I with an asymmetric W detects a row/col swap in the MFMA C layout."""
    K = 64
    x = torch.eye(K)
    w = torch.arange(96 * K, dtype=torch.float32).view(96, K) / 100.0
    out = be.ops.linear(be.to(x), be.to(w), allow_small=False).cpu()
    assert torch.equal(out, w.t())
    assert torch.equal(be.ops.linear(be.to(x), be.to(w)).cpu(), w.t())       # tiny-problem kernel


def test_linear_strided_rows(be):
    x = paramgen.tensor("xs", (70, 3, 32), 5)
    w = paramgen.tensor("ws", (40, 32), 5)
    xv = x[:, 1]                                             # row stride 96
    out = be.ops.linear(be.to(x)[:, 1], be.to(w)).cpu()
    assert torch.allclose(out, F.linear(xv, w), **TOL)


@pytest.mark.parametrize("shape,cin,cout,k,stride,dil", [
    ((1, 6, 5, 4), 16, 32, 3, 1, 1),
    ((2, 8, 8, 4), 32, 48, 3, 2, 1),
    ((1, 7, 6, 3), 16, 16, 1, 2, 1),
    ((1, 9, 9, 1), 16, 32, 3, 1, 2),      # 2-D dilated (ASPP style): kZ = 1
])
def test_conv3d(be, shape, cin, cout, k, stride, dil):
    B, X, Y, Z = shape
    x = paramgen.tensor("cx", (B, cin, X, Y, Z), 1)
    kz = 1 if Z == 1 else k
    w = paramgen.tensor("cw", (cout, cin, k, k, kz), 2, (cin * k * k * kz) ** -0.5)
    b = paramgen.tensor("cb", (cout,), 3)
    pad = (dil * (k - 1) // 2, dil * (k - 1) // 2, dil * (kz - 1) // 2)
    ref = F.conv3d(x, w, b, stride=stride, padding=pad, dilation=dil)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous()
    out = be.ops.conv3d(be.to(x_cl), be.to(conv_weight_tapmajor(w)), (k, k, kz), stride, dil, pad,
                        bias=be.to(b)).cpu()
    out = out.permute(0, 4, 1, 2, 3)
    assert out.shape == ref.shape
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())


def test_conv3d_strided_input_view(be):
    """input read through strides: the Z height slices of a [B,X,Y,Z+1,C] token buffer"""
    B, X, Y, Z, C = 1, 5, 4, 3, 16
    tok = paramgen.tensor("tokv", (B, X, Y, Z + 1, C), 7)
    w = paramgen.tensor("cw2", (32, C, 3, 3, 3), 7, 0.05)
    view = tok[:, :, :, :Z]
    ref = F.conv3d(view.permute(0, 4, 1, 2, 3), w, padding=1)
    out = be.ops.conv3d(be.to(tok)[:, :, :, :Z], be.to(conv_weight_tapmajor(w)), (3, 3, 3)).cpu()
    assert torch.allclose(out.permute(0, 4, 1, 2, 3), ref, **TOL)


@pytest.mark.parametrize("stride,dil,B", [(2, 1, 1), (1, 2, 2), (2, 1, 2)])
def test_conv3d_generic_loader_on_a_view(be, monkeypatch, stride, dil, B):
    """the implicit-GEMM kernel's bounded-buffer loader (csrc/gemm_bf16.hip CONV = 1: byte offset = row offset + tap offset,
    padding taps = an out-of-range offset) on an input that is a strided view -- the Z height slices AND a channel block of
    a wider token buffer, two batches: the buffer's extent comes from the strides, border rows start at negative offsets"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    X, Y, Z, C = 14, 10, 4, 32                       # (>= 64 output rows at stride 2: the matrix-core path)
    tok = paramgen.tensor("tokg", (B, X, Y, Z + 1, C + 32), 9)
    w = paramgen.tensor("cwg", (64, C, 3, 3, 3), 9, (27 * C) ** -0.5)
    view = tok[:, :, :, :Z, 32:]
    ref = F.conv3d(view.permute(0, 4, 1, 2, 3).double(), w.double(), stride=stride, padding=dil, dilation=dil).float()
    wt = conv_weight_tapmajor(w)
    out = be.ops.conv3d(be.to(tok)[:, :, :, :Z, 32:], be.to(wt), (3, 3, 3), stride, dil,
                        w_split=be.ops.split_bf16(be.to(wt))).cpu()
    err = float((out.permute(0, 4, 1, 2, 3) - ref).abs().max() / ref.abs().max())
    assert err < 3e-5, err


@pytest.mark.parametrize("C,G,shape", [(32, 8, (2, 5, 4, 3)), (192, 32, (1, 6, 6, 2)), (48, 8, (1, 40, 30, 2))])
def test_groupnorm(be, C, G, shape):
    B, X, Y, Z = shape
    x = paramgen.tensor("gx", (B, C, X, Y, Z), 1, 2.0) + 0.5
    gamma = 1 + 0.2 * paramgen.tensor("gg", (C,), 2)
    beta = 0.1 * paramgen.tensor("gb", (C,), 3)
    res = paramgen.tensor("gr", (B, C, X, Y, Z), 4)
    ref = F.group_norm(x, G, gamma, beta, 1e-5)
    x_cl = x.permute(0, 2, 3, 4, 1).contiguous()
    stats = be.ops.groupnorm_stats(be.to(x_cl), G)
    out = be.ops.groupnorm_apply(be.to(x_cl), stats, be.to(gamma), be.to(beta), G).cpu()
    assert torch.allclose(out.permute(0, 4, 1, 2, 3), ref, **TOL)
    # ReLU + token buffer (z-mean slot) + residual variants
    tok = be.ops.groupnorm_apply(be.to(x_cl), stats, be.to(gamma), be.to(beta), G, relu=True, tokens=True).cpu()
    r = F.relu(ref).permute(0, 2, 3, 4, 1)
    assert torch.allclose(tok[..., :Z, :], r, **TOL) and torch.allclose(tok[..., Z, :], r.mean(3), **TOL)
    outr = be.ops.groupnorm_apply(be.to(x_cl), stats, be.to(gamma), be.to(beta), G,
                                  residual=be.to(res.permute(0, 2, 3, 4, 1).contiguous())).cpu()
    assert torch.allclose(outr.permute(0, 4, 1, 2, 3), ref + res, **TOL)


@pytest.mark.parametrize("C", [32, 128, 192, 256, 1024])
def test_layernorm(be, C):
    x = paramgen.tensor("lx", (37, C), 1, 3.0) + 1.0
    g = 1 + 0.2 * paramgen.tensor("lg", (C,), 2)
    b = 0.1 * paramgen.tensor("lb", (C,), 3)
    out = be.ops.layernorm(*be.to(x, g, b)).cpu()
    assert torch.allclose(out, F.layer_norm(x, (C,), g, b, 1e-5), **TOL)


def test_dualpath_combine(be):
    B, X, Y, Z, C = 2, 3, 4, 5, 64
    tok = paramgen.tensor("ct", (B, X, Y, Z + 1, C), 1)
    bev = paramgen.tensor("cb", (B, X, Y, C), 2)
    w = paramgen.tensor("cw", (C,), 3, 0.3)
    bias = torch.tensor([0.2])
    ident = paramgen.tensor("ci", (B, X, Y, Z, C), 4)
    sl = tok[..., :Z, :]
    ref = sl + torch.sigmoid((sl * w).sum(-1, keepdim=True) + bias) * bev.unsqueeze(3) + ident
    out = be.ops.dualpath_combine(*be.to(tok, bev, w, bias, ident)).cpu()
    assert torch.allclose(out, ref, **TOL)


@pytest.mark.parametrize("lo,hi", [((4, 4, 2), (8, 8, 4)), ((3, 5, 2), (7, 9, 5))])
def test_upsample_add(be, lo, hi):
    B, C = 2, 16
    coarse = paramgen.tensor("uc", (B, C, *lo), 1)
    lat = paramgen.tensor("ul", (B, C, *hi), 2)
    ref = lat + F.interpolate(coarse, size=hi, mode="trilinear", align_corners=False)
    out = be.ops.upsample_add(be.to(coarse.permute(0, 2, 3, 4, 1).contiguous()),
                              be.to(lat.permute(0, 2, 3, 4, 1).contiguous())).cpu()
    assert torch.allclose(out.permute(0, 4, 1, 2, 3), ref, **TOL)


def test_linear_few_tiles_long_k(be, monkeypatch):
    """two output tiles, 256 k-tiles (the matching costs' [200, 50 176] x [17, 50 176] in small): the split-K cost model may
    take up to 128 slices when a handful of tiles would otherwise leave the chip empty (csrc/gemm_bf16.hip
    occf_pick_ksplit); the slab workspace the host sizes must match what the launch picks, and the fixed-order slab
    reduction must still give the fp32-class result"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    M, N, K = 200, 17, 8192
    assert be.ops.lib.occf_gemm_bf16_workspace(M, N, K) > 16 * M * N, "more than 16 slices expected for this shape"
    x = paramgen.tensor("fx", (M, K), 1)
    w = paramgen.tensor("fw", (N, K), 2, K ** -0.5)
    b = paramgen.tensor("fb", (N,), 3)
    ref = F.linear(x.double(), w.double(), b.double()).float()
    wd = be.to(w)
    out = be.ops.linear(be.to(x), wd, be.to(b), w_split=be.ops.split_bf16(wd)).cpu()
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-5


# ---------------------------------------------------------------- split-bf16 matrix-core path
@pytest.mark.parametrize("M,N,K,act", [(300, 96, 64, 0), (130, 256, 96, 2), (100, 1000, 192, 0), (257, 192, 32, 1)])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 2e-2)])
def test_linear_bf16_modes(be, monkeypatch, M, N, K, act, prec, tol):
    monkeypatch.setattr(be.ops, "precision", prec)
    x = paramgen.tensor("x", (M, K), 1)
    w = paramgen.tensor("w", (N, K), 2, K ** -0.5)
    b = paramgen.tensor("b", (N,), 3)
    r = paramgen.tensor("r", (M, N), 4)
    ref = F.linear(x.double(), w.double(), b.double())
    ref = F.relu(ref) if act == 1 else (F.gelu(ref) if act == 2 else ref)
    ref = (ref + r.double()).float()
    out = be.ops.linear(be.to(x), be.to(w), be.to(b), act, be.to(r), w_split=be.ops.split_bf16(be.to(w))).cpu()
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < tol, err


def test_split_bf16_exactness(be):
    w = paramgen.tensor("sw", (37, 65), 9, 3.0)
    hi, lo = be.ops.split_bf16(be.to(w))
    up = lambda t: (t.cpu().to(torch.int32) << 16).view(torch.float32)
    rec = up(hi) + up(lo)
    assert float(((rec - w).abs() / w.abs().clamp_min(1e-30)).max()) < 2.0 ** -15
    assert torch.equal(up(hi), w.to(torch.bfloat16).float())          # hi is exactly RNE bf16


@pytest.mark.parametrize("shape,cin,cout,k,stride", [((1, 6, 5, 4), 32, 64, 3, 1), ((2, 8, 8, 4), 64, 96, 3, 2),
                                                     ((1, 7, 6, 3), 32, 32, 1, 2)])
def test_conv3d_bf16x3(be, monkeypatch, shape, cin, cout, k, stride):
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    B, X, Y, Z = shape
    x = paramgen.tensor("cx", (B, cin, X, Y, Z), 1)
    w = paramgen.tensor("cw", (cout, cin, k, k, k), 2, (cin * k ** 3) ** -0.5)
    pad = (k // 2,) * 3
    ref = F.conv3d(x.double(), w.double(), stride=stride, padding=pad).float()
    wt = conv_weight_tapmajor(w)
    out = be.ops.conv3d(be.to(x.permute(0, 2, 3, 4, 1).contiguous()), be.to(wt), (k, k, k), stride, 1, pad,
                        w_split=be.ops.split_bf16(be.to(wt))).cpu().permute(0, 4, 1, 2, 3)
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < 2e-5, err


def test_splitk_small_m_long_k(be, monkeypatch):
    """few output tiles + long K (coarse encoder stages) -> split-K slabs + ordered reduce"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    M, N, K = 70, 128, 2048
    assert be.ops.lib.occf_gemm_bf16_workspace(M, N, K) > 0
    x = paramgen.tensor("skx", (M, K), 1)
    w = paramgen.tensor("skw", (N, K), 2, K ** -0.5)
    b = paramgen.tensor("skb", (N,), 3)
    r = paramgen.tensor("skr", (M, N), 4)
    ref = (F.relu(F.linear(x.double(), w.double(), b.double())) + r.double()).float()
    out = be.ops.linear(be.to(x), be.to(w), be.to(b), 1, be.to(r), w_split=be.ops.split_bf16(be.to(w))).cpu()
    assert float((out - ref).abs().max() / ref.abs().max()) < 2e-5
    # conv with Cin=64, 27 taps, tiny volume
    xc = paramgen.tensor("skc", (1, 64, 4, 4, 2), 5)
    wc = paramgen.tensor("skcw", (64, 64, 3, 3, 3), 6, (64 * 27) ** -0.5)
    assert be.ops.lib.occf_gemm_bf16_workspace(32, 64, 27 * 64) > 0
    refc = F.conv3d(xc.double(), wc.double(), padding=1).float()
    wt = conv_weight_tapmajor(wc)
    outc = be.ops.conv3d(be.to(xc.permute(0, 2, 3, 4, 1).contiguous()), be.to(wt), (3, 3, 3),
                         w_split=be.ops.split_bf16(be.to(wt))).cpu().permute(0, 4, 1, 2, 3)
    assert float((outc - refc).abs().max() / refc.abs().max()) < 2e-5


@pytest.mark.parametrize("shape,cin,cout", [((1, 5, 9, 16), 32, 64), ((2, 4, 20, 8), 64, 128), ((1, 3, 34, 4), 32, 192),
                                            ((1, 2, 8, 32), 32, 64)])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 2e-2)])
# weights in fragment order from global memory, on half-size tiles (two workgroups per CU: OCCF_HALO_SMALL=1) and on the
# 256-voxel tiles (the default); weight slabs through LDS (256-voxel tiles only)
@pytest.mark.parametrize("frag,small", [(True, "1"), (True, "0"), (False, "1")])
def test_conv3x3x3_halo(be, monkeypatch, shape, cin, cout, prec, tol, frag, small):
    """LDS-halo conv kernel (odd X, ragged Y tiles, Z = 4/8/16/32) vs fp64 conv3d; with bias/ReLU/residual"""
    monkeypatch.setenv("OCCF_HALO_SMALL", small)
    monkeypatch.setattr(be.ops, "precision", prec)
    monkeypatch.setattr(be.ops, "use_halo_conv", True)
    monkeypatch.setattr(be.ops, "use_wino", False)          # (the direct kernel; test_conv3x3x3_wino is the default route)
    monkeypatch.setattr(be.ops, "halo_frag", frag)
    B, X, Y, Z = shape
    x = paramgen.tensor("hx", (B, cin, X, Y, Z), 1)
    w = paramgen.tensor("hw", (cout, cin, 3, 3, 3), 2, (cin * 27) ** -0.5)
    b = paramgen.tensor("hb", (cout,), 3)
    r = paramgen.tensor("hr", (B, X, Y, Z, cout), 4)
    ref = (F.relu(F.conv3d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 4, 1) + r.double()).float()
    wt = conv_weight_tapmajor(w)
    calls = []
    orig = be.ops.lib.occf_conv3x3x3_halo_fwd
    out = be.ops.conv3d(be.to(x.permute(0, 2, 3, 4, 1).contiguous()), be.to(wt), (3, 3, 3), bias=be.to(b), act=1,
                        residual=be.to(r), w_split=be.ops.split_bf16(be.to(wt))).cpu()
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < tol, err
    # the halo kernel (not the generic fallback) must have accepted these shapes
    xs = be.to(x.permute(0, 2, 3, 4, 1).contiguous())
    sp = be.ops.split_bf16(be.to(wt))
    o2 = torch.empty_like(out).to(be.device)
    import ctypes
    rc = orig(ctypes.c_void_p(xs.data_ptr()), ctypes.c_void_p(sp[0].data_ptr()), ctypes.c_void_p(sp[1].data_ptr()),
              None, None, ctypes.c_void_p(o2.data_ptr()), B, X, Y, Z, cin, cout, xs.stride(0), xs.stride(1),
              xs.stride(2), xs.stride(3), 0, 3 if prec == "bf16x3" else 1, None, None, None, None)
    assert rc == 0


@pytest.mark.parametrize("shape,cin,cout", [((1, 5, 9, 16), 32, 64), ((2, 4, 20, 8), 64, 128), ((1, 3, 34, 4), 32, 192),
                                            ((1, 2, 8, 32), 32, 64), ((1, 6, 5, 16), 96, 256), ((1, 1, 12, 8), 32, 64)])
@pytest.mark.parametrize("epilogue", ["plain", "bias_relu_residual", "bias_gelu", "f16_plain", "f16_residual", "f16x1_plain",
                                      "f16x1_residual"])
def test_conv3x3x3_wino(be, monkeypatch, shape, cin, cout, epilogue):
    """csrc/conv_wino.hip -- Winograd F(2, 3) along x over the LDS halo tile -- vs fp64 conv3d: odd X (the second output
    of the last pair masked), X = 1, ragged Y tiles, Z = 4 / 8 / 16 / 32 (two z tiles), two batches, 1 / 3 chunks of
    input channels (the double-buffered staging), one / two / three 32-column tiles per wave and two N tiles per
    position tile (256 channels); bias / ReLU / GELU / residual epilogues"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    monkeypatch.setattr(be.ops, "use_halo_conv", True)
    monkeypatch.setattr(be.ops, "use_wino", True)
    # f16_*: the two-product form (``act_f16``: x as ONE fp16 piece after its power-of-two scale, filters fp16 (hi, lo)) --
    # what the data gradients run on; x is scaled far below the fp16 range on purpose.  2^-12 per element: bound 3e-4
    # f16x1_*: the ONE-product form of it (the transformed filters too as one fp16 piece: the W1 kernel): bound 6e-4
    f16 = epilogue.startswith("f16")
    x1 = epilogue.startswith("f16x1")
    monkeypatch.setattr(be.ops, "dgrad_f16_single", x1)
    epilogue = {"f16_plain": "plain", "f16_residual": "bias_relu_residual", "f16x1_plain": "plain",
                "f16x1_residual": "bias_relu_residual"}.get(epilogue, epilogue)
    B, X, Y, Z = shape
    x = paramgen.tensor("wx", (B, cin, X, Y, Z), 1) * (3e-6 if f16 else 1.0)
    w = paramgen.tensor("ww", (cout, cin, 3, 3, 3), 2, (cin * 27) ** -0.5)
    b = paramgen.tensor("wb", (cout,), 3) * (1e-6 if f16 else 1.0) if epilogue != "plain" else None
    r = paramgen.tensor("wr", (B, X, Y, Z, cout), 4) * (1e-6 if f16 else 1.0) if epilogue == "bias_relu_residual" else None
    y = F.conv3d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    if epilogue == "bias_relu_residual":
        y = F.relu(y)
    elif epilogue == "bias_gelu":
        y = F.gelu(y)
    ref = y.permute(0, 2, 3, 4, 1)
    if r is not None:
        ref = ref + r.double()
    ref = ref.float()
    wt = conv_weight_tapmajor(w)
    calls = []
    orig = be.ops.lib.occf_conv3x3x3_wino_fwd

    def counted(*a):
        rc = orig(*a)
        calls.append(rc)
        return rc
    monkeypatch.setattr(be.ops.lib, "occf_conv3x3x3_wino_fwd", counted)
    out = be.ops.conv3d(be.to(x.permute(0, 2, 3, 4, 1).contiguous()), be.to(wt), (3, 3, 3),
                        bias=None if b is None else be.to(b), act={"plain": 0, "bias_relu_residual": 1, "bias_gelu": 2}[epilogue],
                        residual=None if r is None else be.to(r), w_split=be.ops.split_bf16(be.to(wt)),
                        act_f16=f16).cpu()
    assert calls == [0], "the Winograd kernel did not take this shape"
    err = float((out - ref).abs().max() / ref.abs().max())
    assert err < (6e-4 if x1 else 3e-4 if f16 else 3e-5), err


@pytest.mark.parametrize("C,H,act,ln_mode,M", [(128, 128, 2, 1, 150), (192, 768, 1, 2, 70), (256, 256, 2, 1, 64),
                                               (128, 256, 1, 0, 33), (128, 128, 1, 2, 900)])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 3e-5), ("bf16", 3e-2)])
def test_mlp_fused(be, monkeypatch, C, H, act, ln_mode, M, prec, tol):
    monkeypatch.setattr(be.ops, "precision", prec)
    # C = H = 128 runs the weight-resident persistent kernel (csrc/mlp_chain.hip): two workgroups walk the 8 token
    # tiles of the M = 900 case (4 iterations each, the last tile ragged, next-tile prefetch on every iteration)
    monkeypatch.setenv("OCCF_MLP_RES_WGS", "2")
    monkeypatch.setenv("OCCF_MLP_RES_WAVES", "4" if M == 150 else "8")      # both workgroup shapes of that kernel
    x = paramgen.tensor("mx", (M, C), 1, 1.5) + 0.3
    w1 = paramgen.tensor("mw1", (H, C), 2, C ** -0.5)
    b1 = paramgen.tensor("mb1", (H,), 3, 0.2)
    w2 = paramgen.tensor("mw2", (C, H), 4, H ** -0.5)
    b2 = paramgen.tensor("mb2", (C,), 5, 0.2)
    g = 1 + 0.2 * paramgen.tensor("mg", (C,), 6)
    bt = 0.1 * paramgen.tensor("mbt", (C,), 7)
    xd = x.double()
    h = F.layer_norm(xd, (C,), g.double(), bt.double(), 1e-5) if ln_mode == 1 else xd
    h = F.linear(h, w1.double(), b1.double())
    h = F.gelu(h) if act == 2 else F.relu(h)
    ref = xd + F.linear(h, w2.double(), b2.double())
    if ln_mode == 2:
        ref = F.layer_norm(ref, (C,), g.double(), bt.double(), 1e-5)
    out = be.ops.mlp_fused(be.to(x), be.to(g), be.to(bt), be.ops.split_bf16(be.to(w1)), be.to(b1),
                           be.ops.split_bf16(be.to(w2)), be.to(b2), act, ln_mode).cpu()
    err = float((out - ref.float()).abs().max() / ref.abs().max())
    assert err < tol, err


@pytest.mark.parametrize("M,K,N,act,bias,res", [(300, 128, 128, 0, True, False), (300, 128, 384, 2, True, True),
                                               (257, 192, 192, 1, False, True), (97, 224, 192, 0, True, False),
                                               (300, 256, 256, 0, True, True), (65, 64, 96, 1, True, False),
                                               (4500, 192, 768, 0, True, False)])
@pytest.mark.parametrize("prec,tol", [("bf16x3", 2e-5), ("bf16", 2e-2)])
def test_linear_streaming_kernel(be, monkeypatch, M, K, N, act, bias, res, prec, tol):
    """csrc/gemm_stream.h: the weight-resident persistent linear of the streaming shapes (here forced on from 64 rows
    and capped at 16 workgroups so that every stream walks several ragged token tiles; N = 384 / 768 split into N blocks
    that share their rows, K = 224 is the pixel decoder's concatenated input, strided rows = a column block of a wider
    tensor).  Against float64; and the launch counter proves the kernel -- not the tile kernel -- ran."""
    monkeypatch.setattr(be.ops, "precision", prec)
    monkeypatch.setenv("OCCF_GEMM_STREAM", "64")
    monkeypatch.setenv("OCCF_GEMM_STREAM_WGS", "16")
    if be.kind == "emu" and M > 1000 and prec == "bf16":
        pytest.skip("the large case once on the emulation")
    wide = paramgen.tensor("gs.x", (M, K + 32), 1, 1.2) + 0.2
    x = wide[:, 16:16 + K]                                         # row stride K + 32
    w = paramgen.tensor("gs.w", (N, K), 2, K ** -0.5)
    b = paramgen.tensor("gs.b", (N,), 3, 0.3) if bias else None
    r = paramgen.tensor("gs.r", (M, N), 4) if res else None
    ref = F.linear(x.double(), w.double(), None if b is None else b.double())
    ref = F.gelu(ref) if act == 2 else F.relu(ref) if act == 1 else ref
    if r is not None:
        ref = ref + r.double()
    lib = be.ops.lib
    n0 = lib.occf_linear_stream_launches()
    xd = be.to(wide)[:, 16:16 + K]
    wd = be.to(w)
    out = be.ops.linear(xd, wd, None if b is None else be.to(b), act, None if r is None else be.to(r),
                        w_split=be.ops.split_bf16(wd), allow_small=False)
    assert lib.occf_linear_stream_launches() == n0 + 1, "the streaming kernel did not take this shape"
    err = float((out.cpu() - ref.float()).abs().max() / ref.abs().max())
    assert err < tol, err
    # the tile kernel on the same problem (OCCF_GEMM_STREAM=0): same result to rounding
    monkeypatch.setenv("OCCF_GEMM_STREAM", "0")
    out0 = be.ops.linear(xd, wd, None if b is None else be.to(b), act, None if r is None else be.to(r),
                         w_split=be.ops.split_bf16(wd), allow_small=False)
    assert lib.occf_linear_stream_launches() == n0 + 1
    assert float((out0.cpu() - out.cpu()).abs().max() / ref.abs().max()) < 2 * tol


@pytest.mark.parametrize("M,K,N", [(300, 128, 128), (170, 192, 384)])
def test_linear_stream_training_epilogues(be, monkeypatch, M, K, N):
    """occf_linear_stream_fwd: the three epilogues the Swin block's training graph uses (csrc/gemm_stream.h) against
    their fp64 formulas -- (a) GELU with the pre-activation as a second output, (b) identity + DropPath scale per token
    slice x (x W^T + b), (c) (x W^T + b) x GELU'(aux); and None outside the envelope (the caller's unfused fallback)"""
    monkeypatch.setenv("OCCF_GEMM_STREAM", "64")
    monkeypatch.setenv("OCCF_GEMM_STREAM_WGS", "16")
    ops = be.ops
    if ops.precision == "f32":
        pytest.skip("the streaming kernel computes in the bf16 split modes")
    x = paramgen.tensor("se.x", (M, K), 1)
    w = paramgen.tensor("se.w", (N, K), 2, K ** -0.5)
    b = paramgen.tensor("se.b", (N,), 3, 0.3)
    r = paramgen.tensor("se.r", (M, N), 4)
    xd, wd, bd, rd = be.to(x, w, b, r)
    sp = ops.split_bf16(wd)
    z_ref = F.linear(x.double(), w.double(), b.double())
    # (a)
    f, z = ops.linear_stream(xd, sp, bd, 2, pre_out=True)
    assert float((z.cpu() - z_ref.float()).abs().max()) < 3e-5 * float(z_ref.abs().max())
    assert float((f.cpu() - F.gelu(z_ref).float()).abs().max()) < 3e-5 * float(z_ref.abs().max())
    # (b) rows ((b XY + xy) S + s) with B = 1: slice s = row % S
    S = 5 if M % 5 == 0 else 2
    XY = M // S
    scale = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25][:S])
    out = ops.linear_stream(xd, sp, bd, 0, residual=rd, row_scale=be.to(scale), XY=XY, S=S)
    ref = r.double() + scale.double()[torch.arange(M) % S][:, None] * z_ref
    assert float((out.cpu() - ref.float()).abs().max()) < 3e-5 * float(ref.abs().max())
    # (b') two samples: sample index (row / (XY S)) * S + row % S, a scale vector of B * S entries; and a
    # row count that is not whole samples is refused instead of reading past the vector
    if M % (2 * S) == 0:
        XY2 = M // (2 * S)
        scale2 = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 2.0, 0.0, 0.5, 1.0, 0.25][:2 * S])
        out = ops.linear_stream(xd, sp, bd, 0, residual=rd, row_scale=be.to(scale2), XY=XY2, S=S)
        rows = torch.arange(M)
        ref = r.double() + scale2.double()[(rows // (XY2 * S)) * S + rows % S][:, None] * z_ref
        assert float((out.cpu() - ref.float()).abs().max()) < 3e-5 * float(ref.abs().max())
    with pytest.raises(Exception):
        ops.linear_stream(xd, sp, bd, 0, residual=rd, row_scale=be.to(scale), XY=XY + 1, S=S)
    # (c)
    aux = paramgen.tensor("se.a", (M, N), 5, 1.5)
    a64 = aux.double()
    gelu_grad = 0.5 * (1 + torch.erf(a64 / 2 ** 0.5)) + a64 * torch.exp(-0.5 * a64 * a64) / (2 * torch.pi) ** 0.5
    out = ops.linear_stream(xd, sp, bd, 3, aux=be.to(aux))
    ref = z_ref * gelu_grad
    assert float((out.cpu() - ref.float()).abs().max()) < 3e-5 * float(ref.abs().max())
    # outside the envelope: K = 320
    x2 = be.to(paramgen.tensor("se.x2", (M, 320), 6))
    w2 = be.to(paramgen.tensor("se.w2", (N, 320), 7))
    assert ops.linear_stream(x2, ops.split_bf16(w2), None, 0) is None


def test_linear_head_major_output(be):
    """value projection written directly as [B, heads, Nq, dh] (what msda3d gathers from)"""
    B, Nq, E, H = 2, 150, 96, 8
    x = paramgen.tensor("hm.x", (B, Nq, E), 1)
    w = paramgen.tensor("hm.w", (E, E), 1, E ** -0.5)
    b = paramgen.tensor("hm.b", (E,), 1)
    ops = be.ops
    if not ops.head_major_supported(B * Nq, E, E, E // H):
        pytest.skip("precision mode without the bf16 GEMM")
    xd, wd, bd = be.to(x, w, b)
    out = ops.linear(xd, wd, bd, w_split=ops.split_bf16(wd), head_major=(Nq, E // H)).cpu()
    ref = torch.nn.functional.linear(x, w, b).view(B, Nq, H, E // H).permute(0, 2, 1, 3)
    assert out.shape == ref.shape and torch.allclose(out, ref, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("kind", ["wino", "halo", "halo256", "strided", "linear", "linear192"])
def test_groupnorm_stats_from_conv_epilogue(be, kind, monkeypatch):
    """the GroupNorm statistics emitted by the conv / GEMM epilogues equal groupnorm_stats of the output"""
    ops = be.ops
    monkeypatch.setattr(ops, "use_wino", kind == "wino")
    if kind == "wino":                   # csrc/conv_wino.hip: tiles of one x-pair x 64 positions (odd X: a masked plane)
        kind = "halo"
        monkeypatch.setenv("OCCF_HALO_SMALL", "0")
    elif kind == "halo256":                # the halo kernel's 256-voxel tiles (the default); "halo": the half-size tiles
        monkeypatch.setenv("OCCF_HALO_SMALL", "0")
        kind = "halo"
    elif kind == "halo":
        monkeypatch.setenv("OCCF_HALO_SMALL", "1")
    # the epilogue path belongs to launches without K slices (large M); this small case would be sliced
    monkeypatch.setenv("OCCF_GEMM_KSPLIT", "1")
    if ops.precision == "f32":
        pytest.skip("epilogue statistics live in the bf16 GEMM kernels")
    G = 8
    if kind == "linear192":          # 6 channels per group straddle the 64-wide tiles; ragged last row tile
        G = 32
        B, V, cin, cout = 1, 200, 32, 192
        x = paramgen.tensor("gx", (B, V, cin), 1)
        w = paramgen.tensor("gw", (cout, cin), 2, cin ** -0.5)
        xd, wd = be.to(x, w)
        y = ops.linear(xd, wd, None, w_split=ops.split_bf16(wd), gn=(G, 1e-5, V))
    elif kind == "linear":
        B, V, cin, cout = 2, 256, 32, 64
        x = paramgen.tensor("gx", (B, V, cin), 1)
        w = paramgen.tensor("gw", (cout, cin), 2, cin ** -0.5)
        xd, wd = be.to(x, w)
        y = ops.linear(xd, wd, None, w_split=ops.split_bf16(wd), gn=(G, 1e-5, V))
    else:
        B, cin, cout = 2, 32, 64
        X, Y, Z = (5, 16, 8) if ops.use_wino else (4, 16, 8) if kind == "halo" else (8, 16, 8)
        x = paramgen.tensor("gx", (B, X, Y, Z, cin), 1)
        w = paramgen.tensor("gw", (cout, cin, 3, 3, 3), 2, (cin * 27) ** -0.5)
        wt = conv_weight_tapmajor(w)
        xd, wd = be.to(x, wt)
        y = ops.conv3d(xd, wd, (3, 3, 3), stride=1 if kind == "halo" else 2, pad=(1, 1, 1),
                       w_split=ops.split_bf16(wd), gn=(G, 1e-5))
    st = ops.last_gn_stats
    assert st is not None, "the epilogue path was not taken"
    ref = ops.groupnorm_stats(y.contiguous(), G, 1e-5)
    assert torch.allclose(st.cpu(), ref.cpu(), rtol=2e-5, atol=2e-6)
