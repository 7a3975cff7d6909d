"""Backward kernels (csrc/bwd_elem.hip, wgrad.hip, attn_bwd.hip, msda3d.hip, dcn.hip) through the C ABI against
torch.autograd of the plain fp32 PyTorch formulation of the same op (the reference's training step is ATen autograd
through exactly these ops).  'emu' = host emulation of the same kernel sources (CPU), 'hip' = the real library."""
import pytest
import torch
import torch.nn.functional as F

from tests import paramgen


def _rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def _t(key, shape, seed=0, scale=1.0):
    return paramgen.tensor(key, shape, seed, scale)


def test_colsum(be):
    for M, N in ((1000, 192), (37, 18), (5000, 1536)):
        x = _t("cs", (M, N), M)
        out = be.ops.colsum(be.to(x)).cpu()
        assert _rel(out, x.sum(0)) < 1e-5


@pytest.mark.parametrize("C", [128, 192, 96, 1024])
def test_layernorm_backward(be, C):
    M = 333
    x = _t("ln_x", (M, C), C).requires_grad_()
    g = (_t("ln_g", (C,), C) * 0.2 + 1).requires_grad_()
    b = _t("ln_b", (C,), C + 1).requires_grad_()
    dy = _t("ln_dy", (M, C), C + 2)
    F.layer_norm(x, (C,), g, b, 1e-5).backward(dy)
    dx, dg, db = be.ops.layernorm_backward(*be.to(x.detach(), g.detach(), dy), 1e-5)
    assert _rel(dx.cpu(), x.grad) < 1e-4 and _rel(dg.cpu(), g.grad) < 1e-4 and _rel(db.cpu(), b.grad) < 1e-4
    # the residual connection's gradient added in the same pass
    add = _t("ln_add", (M, C), C + 3)
    dx2, dg2, db2 = be.ops.layernorm_backward(*be.to(x.detach(), g.detach(), dy), 1e-5, addend=be.to(add))
    assert torch.equal(dx2.cpu(), dx.cpu() + add) and torch.equal(dg2.cpu(), dg.cpu()) and torch.equal(db2.cpu(), db.cpu())


def test_fork_nodes_fuse_the_residual_gradient(be, monkeypatch):
    """LayerNormFork / conv_fork (autograd.py): x -> x + f(LN(x)) and x -> conv(x) (+) x with ONE autograd node seeing
    both gradients of x -- same values as the two-consumer graph, whose sum autograd computes with a separate add"""
    import occformer_amd.ops as ops_mod
    from occformer_amd import autograd as A
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    d = be.device
    ln = torch.nn.LayerNorm(128).to(d)
    lin = torch.nn.Linear(128, 128).to(d)
    x = _t("fk_x", (300, 128), 1).to(d).requires_grad_()
    w = _t("fk_w", (300, 128), 2).to(d)

    def run(fork):
        for p in (*ln.parameters(), *lin.parameters()):
            p.grad = None
        x.grad = None
        h = x * 1.0
        if fork:
            r, n = A.layernorm_fork(h, ln)
        else:
            r, n = h, A.layernorm(h, ln)
        y = A.linear(n, lin, residual=r)
        (y * w).sum().backward()
        return [x.grad.clone()] + [p.grad.clone() for p in (*ln.parameters(), *lin.parameters())]
    for a, b in zip(run(True), run(False)):
        assert _rel(a.cpu(), b.cpu()) < 1e-5
    # only the normalised branch used / only the residual used
    h = x * 1.0
    r, n = A.layernorm_fork(h, ln)
    x.grad = None
    (n * w).sum().backward()
    ga = x.grad.clone()
    x.grad = None
    (A.layernorm(x * 1.0, ln) * w).sum().backward()
    assert _rel(ga.cpu(), x.grad.cpu()) < 1e-5

    conv = torch.nn.Conv3d(64, 64, 3, padding=1, bias=False).to(d)
    xc = _t("fk_xc", (1, 5, 6, 8, 64), 3).to(d).requires_grad_()
    wc = _t("fk_wc", (1, 5, 6, 8, 64), 4).to(d)

    def runc(fork):
        conv.weight.grad = None
        xc.grad = None
        h = xc * 1.0
        if fork:
            r, y, _ = A.conv_fork(h, conv)
        else:
            r, y = h, A.conv(h, conv)
        ((y + r * 0.5) * wc).sum().backward()
        return xc.grad.clone(), conv.weight.grad.clone()
    for a, b in zip(runc(True), runc(False)):
        assert _rel(a.cpu(), b.cpu()) < 1e-5


@pytest.mark.parametrize("relu,tokens,res", [(True, True, False), (False, False, False), (True, False, True),
                                             (False, False, True)])
@pytest.mark.parametrize("apply_form", ["0", "2"])       # thread-per-float4 / row-walking apply pass (2 = forced at any size)
def test_groupnorm_backward(be, monkeypatch, relu, tokens, res, apply_form):
    monkeypatch.setenv("OCCF_GNB_APPLY_ROWS", apply_form)
    B, X, Y, Z, C, G = 2, 5, 6, 4, 64, 8
    x = _t("gn_x", (B, X, Y, Z, C), 3).requires_grad_()
    g = (_t("gn_g", (C,), 4) * 0.2 + 1).requires_grad_()
    b = (_t("gn_b", (C,), 5) * 0.3).requires_grad_()
    r = _t("gn_r", (B, X, Y, Z, C), 6).requires_grad_() if res else None
    y = F.group_norm(x.permute(0, 4, 1, 2, 3), G, g, b, 1e-5).permute(0, 2, 3, 4, 1)
    if relu:
        y = F.relu(y)
    if res:
        y = y + r
    if tokens:
        y = torch.cat((y, y.mean(3, keepdim=True)), 3)
    dy = _t("gn_dy", tuple(y.shape), 7)
    y.backward(dy)
    xd, gd, bd, dyd = be.to(x.detach().contiguous(), g.detach(), b.detach(), dy.contiguous())
    stats = be.ops.groupnorm_stats(xd, G, 1e-5)
    # forward consistency of the op under test
    out = be.ops.groupnorm_apply(xd, stats, gd, bd, G, relu, tokens, None if r is None else be.to(r.detach()))
    assert _rel(out.cpu(), y.detach()) < 1e-4
    dx, dg, db, dres = be.ops.groupnorm_backward(xd, stats, gd, bd, dyd, G, relu, tokens, want_residual=res)
    assert _rel(dx.cpu(), x.grad) < 2e-4
    assert _rel(dg.cpu(), g.grad) < 2e-4 and _rel(db.cpu(), b.grad) < 2e-4
    if res:
        assert _rel(dres.cpu(), r.grad) < 1e-5


@pytest.mark.parametrize("act", [1, 2])
def test_activation(be, act):
    x = _t("act_x", (50, 64), act, 2.0).requires_grad_()
    dy = _t("act_dy", (50, 64), act + 5)
    y = F.relu(x) if act == 1 else F.gelu(x)
    y.backward(dy)
    assert _rel(be.ops.act_forward(be.to(x.detach()), act).cpu(), y.detach()) < 1e-5
    assert _rel(be.ops.act_backward(*be.to(x.detach(), dy), act).cpu(), x.grad) < 1e-5


def test_droppath(be):
    B, X, Y, S, C = 2, 3, 4, 5, 32
    ident = _t("dp_i", (B, X, Y, S, C), 1)
    br = _t("dp_b", (B, X, Y, S, C), 2)
    keep = (paramgen.uniform("dp_k", (B, S), 3) < 0.7).float() / 0.7
    ref = ident + br * keep.view(B, 1, 1, S, 1)
    out = be.ops.droppath(*be.to(ident, br, keep.reshape(-1).contiguous()), X * Y, S)
    assert _rel(out.cpu(), ref) < 1e-6
    out = be.ops.droppath(None, *be.to(br, keep.reshape(-1).contiguous()), X * Y, S)
    assert _rel(out.cpu(), br * keep.view(B, 1, 1, S, 1)) < 1e-6


@pytest.mark.parametrize("C,bias", [(64, True), (192, False)])
def test_dualpath_combine_backward(be, C, bias):
    B, X, Y, Z = 2, 3, 5, 4
    tok = _t("dc_t", (B, X, Y, Z + 1, C), 1).requires_grad_()
    bev = _t("dc_b", (B, X, Y, C), 2).requires_grad_()
    w = (_t("dc_w", (C,), 3) * 0.2).requires_grad_()
    b = (_t("dc_bias", (1,), 4)).requires_grad_() if bias else None
    ident = _t("dc_i", (B, X, Y, Z, C), 5)
    s = (tok[:, :, :, :Z] * w).sum(-1, keepdim=True) + (b if bias else 0)
    out = tok[:, :, :, :Z] + torch.sigmoid(s) * bev.unsqueeze(3) + ident
    dout = _t("dc_do", tuple(out.shape), 6)
    out.backward(dout)
    args = be.to(tok.detach(), bev.detach(), w.detach())
    fwd = be.ops.dualpath_combine(*args, None if b is None else be.to(b.detach()), be.to(ident))
    assert _rel(fwd.cpu(), out.detach()) < 1e-5
    dtok, dbev, dw, db = be.ops.dualpath_combine_backward(*args, None if b is None else be.to(b.detach()), be.to(dout))
    assert _rel(dtok.cpu(), tok.grad) < 1e-4 and _rel(dbev.cpu(), bev.grad) < 1e-4
    assert _rel(dw.cpu(), w.grad) < 1e-4
    if bias:
        assert _rel(db.cpu(), b.grad) < 1e-4
    assert float(dtok[:, :, :, Z].abs().max()) == 0.0


@pytest.mark.parametrize("shape,shape2", [((4, 5, 2), (8, 10, 4)), ((3, 4, 2), (7, 9, 5)), ((2, 2, 1), (4, 4, 2))])
def test_upsample_add_backward(be, shape, shape2):
    B, C = 2, 8
    coarse = _t("ua_c", (B, C, *shape), 1).requires_grad_()
    out = F.interpolate(coarse, size=shape2, mode="trilinear", align_corners=False)
    dout = _t("ua_do", (B, *shape2, C), 2)
    out.backward(dout.permute(0, 4, 1, 2, 3))
    dc = be.ops.upsample_add_backward(be.to(dout), (B, *shape, C))
    assert _rel(dc.cpu(), coarse.grad.permute(0, 2, 3, 4, 1)) < 1e-5


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("align,mode,shared", [(False, "border", False), (True, "zeros", False), (False, "zeros", True)])
def test_point_sample_backward(be, align, mode, shared, det, monkeypatch):
    """det: the reproducible form (ops.deterministic: the scatter in 64-bit fixed point scaled by max |dout|, then floats)"""
    monkeypatch.setattr(be.ops, "deterministic", det)
    N, C, X, Y, Z, P = 3, 2, 6, 5, 4, 200
    vol = _t("ps_v", (N, C, X, Y, Z), 1).requires_grad_()
    pts = paramgen.uniform("ps_p", (1 if shared else N, P, 3), 2) * 1.2 - 0.1
    grid = (pts * 2 - 1).view(pts.shape[0], P, 1, 1, 3).expand(N, P, 1, 1, 3)
    out = F.grid_sample(vol, grid, mode="bilinear", padding_mode=mode, align_corners=align).view(N, C, P)
    dout = _t("ps_do", (N, C, P), 3)
    out.backward(dout)
    fwd = be.ops.point_sample_3d(be.to(vol.detach()), be.to(pts.contiguous()), align, mode)
    assert _rel(fwd.cpu(), out.detach()) < 1e-5
    dv = be.ops.point_sample_3d_backward(be.to(dout), be.to(pts.contiguous()), (N, C, X, Y, Z), align, mode)
    assert _rel(dv.cpu(), vol.grad) < 1e-4
    # the voxel-major form [V, ld] the mask-logit contraction consumes, into columns col0 .. of a shared buffer
    ld, col0 = N * C + 5, 3
    shared_buf = torch.zeros((X * Y * Z, ld), device=be.device)
    got = be.ops.point_sample_3d_backward(be.to(dout * 1e-7), be.to(pts.contiguous()), (N, C, X, Y, Z), align, mode,
                                          voxel_major_cols=ld, out=shared_buf, col0=col0)
    ref = vol.grad.reshape(N * C, -1).t() * 1e-7
    assert _rel(got[:, col0:col0 + N * C].cpu(), ref) < 1e-4
    assert float(got[:, :col0].abs().max()) == 0.0 and float(got[:, col0 + N * C:].abs().max()) == 0.0


@pytest.fixture
def wgrad_mode():
    """sets ops.wgrad_f16 for one test and restores it"""
    saved = []

    def set_mode(be, f16):
        saved.append((be.ops, be.ops.wgrad_f16, be.ops.wgrad_f16_linear, be.ops.wgrad_f16_single))
        be.ops.wgrad_f16 = be.ops.wgrad_f16_linear = bool(f16)
        be.ops.wgrad_f16_single = f16 == "x1"       # the convolutions' one-product form (G8 shapes)
    yield set_mode
    for ops, v, vl, vs in saved:
        ops.wgrad_f16, ops.wgrad_f16_linear, ops.wgrad_f16_single = v, vl, vs


def test_point_loss_rows_backward(be):
    R, P = 5, 300
    x = _t("pl_x", (R, P), 1, 2.0).requires_grad_()
    t = (paramgen.uniform("pl_t", (R, P), 2) < 0.3).float()
    s = x.sigmoid()
    rows = torch.stack((F.binary_cross_entropy_with_logits(x, t, reduction="none").sum(1), (s * t).sum(1), s.sum(1),
                        t.sum(1)), 1)
    g = _t("pl_g", (R, 4), 3)
    rows.backward(g)
    assert _rel(be.ops.point_loss_rows(*be.to(x.detach(), t)).cpu(), rows.detach()) < 1e-5
    dx = be.ops.point_loss_rows_backward(*be.to(x.detach(), t, g))
    assert _rel(dx.cpu(), x.grad) < 1e-5


@pytest.mark.parametrize("M,N,K", [(1500, 192, 128), (2100, 128, 384), (4000, 96, 192), (100, 18, 192), (1300, 288, 64),
                                   (1100, 192, 192), (1200, 40, 160),
                                   # >= 8 M-slabs: the per-XCD, tile-class-major workgroup order
                                   (9000, 192, 192), (9100, 160, 64), (8200, 320, 192),
                                   # M <= 2048 with shapes off the 32 x 32 tiles of the small-M kernel; M in (2048, 65536]
                                   # with widths the tile kernel does not take (the 8-lanes-per-output kernel)
                                   (100, 1536, 192), (7, 33, 50), (1024, 18, 18), (2500, 18, 30), (300, 192, 6)])
@pytest.mark.parametrize("f16", [False, True])
def test_linear_wgrad(be, M, N, K, f16, wgrad_mode):
    """f16 = the two-product weight gradient (ops._wgrad_terms: dy as ONE fp16 piece after a power-of-two scale from
    max |dy|, x as fp16 (hi, lo)): 2^-12 per dy element, averaged over the M rows -- bound 3e-4; dy is scaled far
    out of the fp16 range on purpose (the raw values would flush to zero)"""
    wgrad_mode(be, f16)
    dy = _t("wg_dy", (M, N), M) * (3e-7 if f16 else 1.0)
    x = _t("wg_x", (M, K), M + 1)
    dw, db = be.ops.linear_wgrad(*be.to(dy, x))
    assert _rel(dw.cpu(), dy.t() @ x) < (3e-4 if f16 else 1e-4)
    assert _rel(db.cpu(), dy.sum(0)) < 1e-4


@pytest.mark.parametrize("f16", [False, True])
def test_linear_wgrad_strided(be, f16, wgrad_mode):
    """operands that are column blocks of wider matrices (row stride > width); in the fp16 mode the columns OUTSIDE the
    block are 1e6 times larger -- the scale must come from the block's own maximum"""
    wgrad_mode(be, f16)
    M, N, K = 1200, 64, 128
    dyw = _t("wgs_dy", (M, 2 * N), 1)
    dyw[:, :N] *= 1e6
    xw = _t("wgs_x", (M, 3 * K), 2)
    dw, db = be.ops.linear_wgrad(be.to(dyw)[:, N:], be.to(xw)[:, K:2 * K])
    assert _rel(dw.cpu(), dyw[:, N:].t() @ xw[:, K:2 * K]) < (3e-4 if f16 else 1e-4)
    assert _rel(db.cpu(), dyw[:, N:].sum(0)) < 1e-4


CONV_CASES = [
    # B, dims, Cin, Cout, k, stride, dil
    (1, (10, 9, 4), 32, 64, (3, 3, 3), 1, 1),
    (2, (8, 8, 4), 64, 32, (3, 3, 3), 2, 1),
    (1, (12, 11, 1), 32, 32, (3, 3, 1), 1, 3),
    (1, (8, 6, 4), 64, 128, (1, 1, 1), 2, 1),
    (1, (7, 9, 8), 192, 64, (3, 3, 3), 1, 1),
    # 256 rows per parity class: the 128-row tiles of the strided data gradient skip the taps their class cannot reach
    (1, (16, 16, 8), 32, 32, (3, 3, 3), 2, 1),
    # Zo = 16 / 8: the per-column staging path of the pre-split weight gradient (two batches, stride 2, dilation)
    (2, (5, 6, 16), 32, 64, (3, 3, 3), 1, 1),
    (1, (6, 6, 16), 64, 32, (3, 3, 3), 2, 1),
    (1, (6, 5, 8), 32, 32, (3, 3, 3), 1, 2),
    # 192 = 128 + 64 in BOTH tile dimensions (the roofline kernel's shape: tiles 128x128, 128x64, 64x128, 64x64 of one
    # launch, pre-split per-column staging) and 160 = 128 + 32 (a 64-wide remainder tile, half masked)
    (1, (4, 5, 16), 192, 192, (3, 3, 3), 1, 1),
    (1, (5, 4, 8), 160, 96, (3, 3, 3), 1, 1),
    # the G8 path (csrc/wgrad_g8.h: stride 1, 3^3, Z % 8 == 0, channels % 64 == 0; >= 1024 rows): LDS-DMA staging from
    # z-shifted pre-split copies.  Tiles 128x64 / 64x128 (two batches, Z = 16: two z-groups per column), 192x192 with
    # several slabs per XCD and a ragged last stage, Z = 32 (four z-groups per column, a stage spans half a column),
    # 320 = 128 + 192 output channels (two tile classes in one call), a 3x3x3 window that leaves the grid in x and y
    (2, (6, 6, 16), 64, 128, (3, 3, 3), 1, 1),
    (1, (9, 15, 8), 192, 192, (3, 3, 3), 1, 1),
    (1, (4, 9, 32), 128, 64, (3, 3, 3), 1, 1),
    (1, (8, 8, 16), 64, 320, (3, 3, 3), 1, 1),
    (1, (2, 64, 8), 64, 64, (3, 3, 3), 1, 1),
]


def _conv_ref(x, w, k, stride, dil):
    pad = tuple(dil * (kk - 1) // 2 for kk in k)
    return F.conv3d(x, w, stride=stride, dilation=dil, padding=pad)


@pytest.mark.parametrize("f16", [False, True, "x1"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3d_wgrad_dgrad(be, case, f16, wgrad_mode):
    """f16 = True: dy in one fp16 piece x the fp16 (hi, lo) halves of x; "x1": ONE product -- x too in one fp16 piece --
    on the G8 kernel's shapes (the other shapes compute the two-product form: skipped here)"""
    wgrad_mode(be, f16)
    B, dims, Cin, Cout, k, stride, dil = case
    if f16 == "x1" and not (stride == 1 and dil == 1 and k == (3, 3, 3) and dims[2] % 8 == 0 and Cin % 64 == 0 and
                            Cout % 64 == 0 and B * dims[0] * dims[1] * dims[2] >= 1024):
        pytest.skip("not a G8 shape")
    x = _t("cv_x", (B, Cin, *dims), Cin).requires_grad_()
    w = (_t("cv_w", (Cout, Cin, *k), Cout) * (Cin * k[0] * k[1] * k[2]) ** -0.5).requires_grad_()
    y = _conv_ref(x, w, k, stride, dil)
    dy = _t("cv_dy", tuple(y.shape), 9) * (2e-6 if f16 else 1.0)
    y.backward(dy)
    x_cl = x.detach().permute(0, 2, 3, 4, 1).contiguous()
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous()
    dw, _ = be.ops.conv3d_wgrad(be.to(dy_cl), be.to(x_cl), k, stride, dil)
    ref_dw = w.grad.permute(0, 2, 3, 4, 1).reshape(Cout, -1)
    assert _rel(dw.cpu(), ref_dw) < (6e-4 if f16 == "x1" else 3e-4 if f16 else 1e-4)
    if f16 == "x1":
        # ... and it IS the one-product arithmetic: the two-product kernel on x rounded to one fp16 piece agrees to
        # accumulation-order noise
        be.ops.wgrad_f16_single = False
        dw2, _ = be.ops.conv3d_wgrad(be.to(dy_cl), be.to(x_cl.half().float()), k, stride, dil)
        assert _rel(dw.cpu(), dw2.cpu()) < 2e-6
    if f16:
        return                              # (the data gradient does not depend on the mode)
    # data gradient: weight re-laid as [Cin, taps * Cout]
    wt = w.detach().permute(1, 2, 3, 4, 0).reshape(Cin, -1).contiguous()
    sp = be.ops.split_bf16(be.to(wt))
    dx = be.ops.conv3d_dgrad(be.to(dy_cl), sp, (B, *dims, Cin), k, stride, dil)
    assert _rel(dx.cpu(), x.grad.permute(0, 2, 3, 4, 1)) < 1e-4


@pytest.mark.parametrize("on", [True, False])
def test_conv2d_wgrad_on_the_g8_kernel(be, on, wgrad_mode):
    """DepthNet's shape in small: a 3x3 convolution over [B, 16, 44] maps (Z = 1) -- with the 16-extent moved innermost
    the G8 weight-gradient kernel takes it as a [1, 44, 16] volume with a (1, 3, 3) window (ops.wgrad_2d_as_g8); the taps
    must come back in (dx, dy) order.  Off: the register-transposing kernel on the same inputs."""
    wgrad_mode(be, True)
    saved = be.ops.wgrad_2d_as_g8
    be.ops.wgrad_2d_as_g8 = on
    try:
        B, H, W, Cin, Cout = 2, 16, 44, 64, 128
        x = _t("c2_x", (B, Cin, H, W, 1), 3).requires_grad_()
        w = (_t("c2_w", (Cout, Cin, 3, 3, 1), 4) * (9 * Cin) ** -0.5).requires_grad_()
        y = F.conv3d(x, w, padding=(1, 1, 0))
        dy = _t("c2_dy", tuple(y.shape), 5) * 3e-6
        y.backward(dy)
        calls = []
        orig = be.ops._conv3d_wgrad

        def spy(dy_, x_, k, *a, **kw):
            calls.append(tuple(k))
            return orig(dy_, x_, k, *a, **kw)
        be.ops._conv3d_wgrad = spy
        try:
            dw, db = be.ops.conv3d_wgrad(be.to(dy.permute(0, 2, 3, 4, 1).contiguous()), be.to(x.detach().permute(0, 2, 3, 4, 1).contiguous()),
                                         (3, 3, 1), 1, 1, want_bias=True)
        finally:
            be.ops._conv3d_wgrad = orig
        assert calls == [(1, 3, 3) if on else (3, 3, 1)], calls
        ref = w.grad.permute(0, 2, 3, 4, 1).reshape(Cout, -1)
        assert _rel(dw.cpu(), ref) < 6e-4
        assert _rel(db.cpu(), dy.sum((0, 2, 3, 4))) < 1e-5
    finally:
        be.ops.wgrad_2d_as_g8 = saved


@pytest.mark.parametrize("case", [(2, (8, 8, 4), 64, 32, (3, 3, 3), 2, 1), (1, (16, 16, 8), 32, 64, (3, 3, 3), 2, 1),
                                  (1, (8, 6, 4), 64, 128, (1, 1, 1), 2, 1)])
def test_strided_dgrad_split_k(be, case, monkeypatch):
    """the class-major data gradient with its K loop cut in two slices (forced: the cost model only splits at sizes the
    emulator cannot afford): the per-tile row table, the tap offsets and the 16-rows-per-workgroup slab reduction"""
    monkeypatch.setenv("OCCF_GEMM_KSPLIT", "2")
    B, dims, Cin, Cout, k, stride, dil = case
    x = _t("cv_x", (B, Cin, *dims), Cin).requires_grad_()
    w = _t("cv_w", (Cout, Cin, *k), Cout) * (Cin * k[0] * k[1] * k[2]) ** -0.5
    y = _conv_ref(x, w, k, stride, dil)
    dy = _t("cv_dy", tuple(y.shape), 9)
    y.backward(dy)
    dy_cl = dy.permute(0, 2, 3, 4, 1).contiguous()
    wt = w.permute(1, 2, 3, 4, 0).reshape(Cin, -1).contiguous()
    sp = be.ops.split_bf16(be.to(wt))
    dx = be.ops.conv3d_dgrad(be.to(dy_cl), sp, (B, *dims, Cin), k, stride, dil)
    assert _rel(dx.cpu(), x.grad.permute(0, 2, 3, 4, 1)) < 1e-4


@pytest.mark.parametrize("X,Y,S,heads,shift,B", [(14, 14, 3, 1, 0, 1), (10, 9, 2, 2, 3, 2), (7, 16, 1, 4, 3, 1),
                                                 (5, 5, 2, 3, 0, 1)])
def test_window_attention_backward(be, X, Y, S, heads, shift, B):
    from oracle import occformer_ref as O
    C = heads * 32
    wq = paramgen.tensor("wb_qkvw", (3 * C, C), 1, C ** -0.5).requires_grad_()
    bq = paramgen.tensor("wb_qkvb", (3 * C,), 1, 0.3).requires_grad_()
    tab = paramgen.tensor("wb_tab", (169, heads), 1, 0.5).requires_grad_()
    y = paramgen.tensor("wb_tokens", (B * S, X * Y, C), 2).requires_grad_()
    sd = {"a.w_msa.qkv.weight": wq, "a.w_msa.qkv.bias": bq, "a.w_msa.proj.weight": torch.eye(C),
          "a.w_msa.proj.bias": torch.zeros(C), "a.w_msa.relative_position_bias_table": tab}
    ref = O.shift_window_msa(sd, "a.", y, X, Y, heads, shift)            # [(b s), x*y, C]
    dref = paramgen.tensor("wb_do", tuple(ref.shape), 3)
    ref.backward(dref)
    to_k = lambda t: t.view(B, S, X, Y, C).permute(0, 2, 3, 1, 4).reshape(-1, C).contiguous()
    yk = to_k(y.detach())
    qkv = F.linear(yk, wq.detach(), bq.detach()).contiguous()
    args = be.to(qkv, bq.detach(), tab.detach())
    out = be.ops.window_attention(*args, B, X, Y, S, heads, shift)
    dqkv, dpad, dtab = be.ops.window_attention_backward(*args, out, be.to(to_k(dref)), B, X, Y, S, heads, shift)
    dqkv, dpad, dtab = dqkv.cpu(), dpad.cpu(), dtab.cpu()
    assert _rel(dtab, tab.grad) < 2e-4
    assert _rel(dqkv @ wq.detach(), to_k(y.grad)) < 2e-4
    assert _rel(dqkv.t() @ yk, wq.grad) < 2e-4
    assert _rel(dqkv.sum(0) + dpad, bq.grad) < 2e-4
    if X % 7 == 0 and Y % 7 == 0:
        assert float(dpad.abs().max()) == 0.0


@pytest.mark.parametrize("Q,L,heads,masked", [(20, 100, 3, True), (100, 37, 2, True), (100, 100, 6, False),
                                              (7, 300, 1, True),
                                              # several key chunks x tiles, a partial last tile, all 128 query slots
                                              (100, 1234, 2, True), (128, 260, 1, True), (33, 129, 1, False)])
def test_masked_attention_backward(be, Q, L, heads, masked):
    B, E = 2, heads * 32
    q = _t("xb_q", (B, Q, E), 1).requires_grad_()
    k = _t("xb_k", (B, L, E), 2).requires_grad_()
    v = _t("xb_v", (B, L, E), 3).requires_grad_()
    blocked = row_open = None
    bias = None
    if masked:
        blocked = paramgen.uniform("xb_m", (B, Q, L), 4) < 0.6
        blocked[0, 1] = True                                            # fully blocked row -> unmasked
        row_open = (~blocked.all(-1)).int().reshape(-1)
        eff = blocked & ~blocked.all(-1, keepdim=True)
        bias = torch.zeros(B, 1, Q, L).masked_fill(eff.unsqueeze(1), float("-inf"))
    sp = lambda t: t.view(B, -1, heads, 32).transpose(1, 2)
    ref = F.scaled_dot_product_attention(sp(q), sp(k), sp(v), attn_mask=bias).transpose(1, 2).reshape(B, Q, E)
    dout = _t("xb_do", (B, Q, E), 5)
    ref.backward(dout)
    a = be.to(q.detach(), k.detach(), v.detach())
    bl = None if blocked is None else be.to(blocked.to(torch.uint8).contiguous())
    ro = None if row_open is None else be.to(row_open.contiguous())
    out = be.ops.masked_attention(*a, heads, bl, ro)
    assert _rel(out.cpu(), ref.detach()) < 1e-4
    dq, dk, dv = be.ops.masked_attention_backward(*a, heads, out, be.to(dout), bl, ro)
    assert _rel(dq.cpu(), q.grad) < 2e-4 and _rel(dk.cpu(), k.grad) < 2e-4 and _rel(dv.cpu(), v.grad) < 2e-4


@pytest.mark.parametrize("E,heads,shapes", [(96, 8, [(2, 2, 1), (4, 4, 2), (8, 8, 4)]), (48, 4, [(3, 2, 2), (5, 4, 3)]),
                                            (40, 8, [(2, 3, 1), (4, 4, 2), (6, 5, 3)]), (192, 8, [(2, 2, 1), (4, 3, 2)]),
                                            # levels too large for one LDS tile: the tiled value-gradient path with
                                            # margins, and offsets (scale 2 x 4 cells) that leave the region -> fallback
                                            (24, 2, [(11, 10, 2), (22, 20, 4), (44, 40, 8)])])
@pytest.mark.parametrize("records", ["1", "0"])      # the tiles read the gather pass's per-sample records / recompute them
def test_msda3d_backward(be, monkeypatch, E, heads, shapes, records):
    from oracle import occformer_ref as O
    monkeypatch.setenv("OCCF_MSDA_RECORDS", records)
    B, P = 2, 4
    L = len(shapes)
    Nq = sum(x * y * z for x, y, z in shapes)
    value = paramgen.tensor("mb_value", (B, Nq, E), 1).requires_grad_()
    offs = paramgen.tensor("mb_offs", (B, Nq, heads * L * P * 3), 1, 2.0).requires_grad_()
    logits = paramgen.tensor("mb_logits", (B, Nq, heads * L * P), 1).requires_grad_()
    ref_pts = torch.cat([O.reference_points_3d(s) for s in shapes], 0)[None, :, None, :].expand(B, -1, L, -1)
    norm = torch.tensor([[s[2], s[1], s[0]] for s in shapes], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + offs.view(B, Nq, heads, L, P, 3) / norm[None, None, None, :, None, :]
    w = logits.view(B, Nq, heads, L * P).softmax(-1).view(B, Nq, heads, L, P)
    ref = O.msda3d_core(value.view(B, Nq, heads, E // heads), shapes, loc, w)
    dout = paramgen.tensor("mb_do", tuple(ref.shape), 2)
    ref.backward(dout)
    for hm in (False, True):
        v = value.detach()
        if hm:
            v = v.view(B, Nq, heads, E // heads).permute(0, 2, 1, 3).contiguous()
        dv, doff, dlg = be.ops.msda3d_backward(*be.to(v, offs.detach(), logits.detach(), dout), shapes, heads, P,
                                               head_major=hm)
        assert _rel(dv.cpu(), value.grad) < 2e-4
        assert _rel(doff.cpu(), offs.grad) < 2e-4 and _rel(dlg.cpu(), logits.grad) < 2e-4


def test_msda3d_backward_keeps_a_diverged_step_visible(be):
    """a NaN / inf in dout must reach d(value) (the tiled path accumulates in fixed point scaled by max|dout|; fmaxf
    alone would drop the NaN and the float -> int conversion of the contributions is undefined for it)"""
    E, heads, shapes, B, P = 24, 2, [(4, 4, 2), (8, 8, 4)], 1, 4
    L, Nq = len(shapes), sum(x * y * z for x, y, z in shapes)
    value = paramgen.tensor("mn_value", (B, Nq, E), 1)
    offs = paramgen.tensor("mn_offs", (B, Nq, heads * L * P * 3), 1, 1.0)
    logits = paramgen.tensor("mn_logits", (B, Nq, heads * L * P), 1)
    for bad in (float("nan"), float("inf")):
        dout = paramgen.tensor("mn_do", (B, Nq, E), 2).clone()
        dout[0, 7, 3] = bad
        dv, _, _ = be.ops.msda3d_backward(*be.to(value, offs, logits, dout), shapes, heads, P, head_major=False)
        assert not bool(torch.isfinite(dv).all()), bad


@pytest.mark.parametrize("det", [False, True])
@pytest.mark.parametrize("groups,dg", [(4, 1), (2, 2)])
def test_deform_col2im(be, groups, dg, det, monkeypatch):
    from oracle import occformer_ref as O
    monkeypatch.setattr(be.ops, "deterministic", det)     # (det: the data-gradient scatter in 64-bit fixed point)
    BN, C, H, W, K = 2, 32, 6, 7, 3
    x = _t("dc2_x", (BN, C, H, W), 1).requires_grad_()
    off = (_t("dc2_off", (BN, dg * 2 * K * K, H, W), 2) * 1.5).requires_grad_()
    wgt = _t("dc2_w", (C, C // groups, K, K), 3, 0.2).requires_grad_()
    y = O.deform_conv2d(x, off, wgt, 1, 1, 1, groups, dg)
    dy = _t("dc2_dy", tuple(y.shape), 4)
    y.backward(dy)
    # forward columns and the grouped contraction's gradient w.r.t. the columns, in plain torch
    x_cl = x.detach().permute(0, 2, 3, 1).contiguous()
    col = be.ops.deform_im2col(be.to(x_cl), be.to(off.detach()), K, 1, 1, 1, groups, dg).cpu()
    cpg = C // groups
    w_g = wgt.detach().view(groups, C // groups, cpg, K * K).permute(0, 1, 3, 2)      # [g, co, tap, c]
    out = torch.einsum("pgtc,gotc->pgo", col, w_g).reshape(BN, H, W, C).permute(0, 3, 1, 2)
    assert _rel(out, y.detach()) < 1e-4
    dy_p = dy.permute(0, 2, 3, 1).reshape(BN * H * W, groups, C // groups)
    dcol = torch.einsum("pgo,gotc->pgtc", dy_p, w_g).contiguous()
    dx, doff = be.ops.deform_col2im(be.to(x_cl), be.to(off.detach()), be.to(dcol), K, 1, 1, 1, groups, dg)
    assert _rel(dx.cpu().permute(0, 3, 1, 2), x.grad) < 2e-4
    assert _rel(doff.cpu(), off.grad) < 2e-4


# ------------------------------------------------------------------------------------------ composite autograd nodes
@pytest.fixture
def bound_ops(be, monkeypatch):
    import occformer_amd.ops as ops_mod
    monkeypatch.setattr(ops_mod, "_ops", be.ops)
    return be


def test_inproj_node_matches_three_linears(bound_ops):
    """autograd.InProj (one node for nn.MultiheadAttention's q / k / v projections) == F.linear on the three blocks of
    in_proj_weight, forward and every gradient; xq is also used as xk (self-attention passes the same tensor twice)"""
    from occformer_amd import autograd as A
    be = bound_ops
    E, Q, L = 64, 9, 37
    W = _t("ip_w", (3 * E, E), 1, E ** -0.5)
    b = _t("ip_b", (3 * E,), 2, 0.3)
    xq, xv = _t("ip_xq", (2, Q, E), 3), _t("ip_xv", (2, L, E), 4)
    gq, gk, gv = _t("ip_gq", (2, Q, E), 5), _t("ip_gk", (2, Q, E), 6), _t("ip_gv", (2, L, E), 7)
    Wr, br, xqr, xvr = (t.clone().requires_grad_() for t in (W, b, xq, xv))
    (F.linear(xqr, Wr[:E], br[:E]) * gq).sum().add((F.linear(xqr, Wr[E:2 * E], br[E:2 * E]) * gk).sum()).add(
        (F.linear(xvr, Wr[2 * E:], br[2 * E:]) * gv).sum()).backward()
    Wd = torch.nn.Parameter(be.to(W))
    bd = torch.nn.Parameter(be.to(b))
    xqd, xvd = be.to(xq).requires_grad_(), be.to(xv).requires_grad_()
    q, k, v = A.InProj.apply(xqd, xqd, xvd, Wd, bd)
    assert _rel(q.detach().cpu(), F.linear(xq, W[:E], b[:E])) < 1e-5
    assert _rel(v.detach().cpu(), F.linear(xv, W[2 * E:], b[2 * E:])) < 1e-5
    ((q * be.to(gq)).sum() + (k * be.to(gk)).sum() + (v * be.to(gv)).sum()).backward()
    assert _rel(Wd.grad.cpu(), Wr.grad) < 1e-4 and _rel(bd.grad.cpu(), br.grad) < 1e-5
    assert _rel(xqd.grad.cpu(), xqr.grad) < 1e-4 and _rel(xvd.grad.cpu(), xvr.grad) < 1e-4


@pytest.mark.parametrize("align,pad", [(False, "border"), (True, "zeros")])
def test_sampled_mask_logits_joint_matches_dense_autograd(bound_ops, align, pad):
    """autograd.SampledMaskLogitsJoint over three prediction sets == torch.autograd through the DENSE formulation the
    reference uses (einsum('qc,cxyz->qxyz') -> grid_sample at the points): sampled values, d(mask_embed rows) of
    every set and the ONE summed d(mask features)"""
    from occformer_amd import autograd as A
    be = bound_ops
    X, Y, Z, E, P = 5, 4, 3, 16, 40
    V = X * Y * Z
    feat = _t("smj_feat", (V, E), 1)
    sets = [(3, 11), (1, 12), (5, 13)]                      # (matched rows, seed)
    embeds = [_t("smj_e", (n, E), s) for n, s in sets]
    pts = [paramgen.uniform("smj_p", (n, P, 3), s) * 1.2 - 0.1 for n, s in sets]
    douts = [_t("smj_d", (n, P), s + 50) for n, s in sets]
    fr = feat.clone().requires_grad_()
    ers = [e.clone().requires_grad_() for e in embeds]
    total = 0
    refs = []
    for e, p, d in zip(ers, pts, douts):
        dense = (e @ fr.t()).view(e.shape[0], 1, X, Y, Z)
        smp = F.grid_sample(dense, (p * 2 - 1).view(e.shape[0], P, 1, 1, 3), mode="bilinear", padding_mode=pad,
                            align_corners=align).view(e.shape[0], P)
        refs.append(smp.detach())
        total = total + (smp * d).sum()
    total.backward()
    fd = be.to(feat).requires_grad_()
    eds = [be.to(e).requires_grad_() for e in embeds]
    vols = [(e.detach() @ fd.detach().t()).view(e.shape[0], X, Y, Z) for e in eds]      # detached logits of the rows
    outs = A.SampledMaskLogitsJoint.apply(fd, align, pad, len(sets), *vols, *eds, *[be.to(p) for p in pts])
    for o, r in zip(outs, refs):
        assert _rel(o.detach().cpu(), r) < 1e-5
    sum((o * be.to(d)).sum() for o, d in zip(outs, douts)).backward()
    assert _rel(fd.grad.cpu(), fr.grad) < 2e-4
    for ed, er in zip(eds, ers):
        assert _rel(ed.grad.cpu(), er.grad) < 2e-4


def test_depthnet_training_convs_on_the_library_kernels(bound_ops, monkeypatch):
    """OCCF_DEPTHNET_LIB=1: DepthNet's training-mode 2-D convolutions on the library's kernel pairs give the same
    output and the same parameter gradients as the ATen / MIOpen graph (ViewTransformerLSSBEVDepth.py:450-504)"""
    import occformer_amd.view_transformer as vt
    be = bound_ops
    torch.manual_seed(0)
    net = vt.DepthNet(32, 32, 16, 12, cam_channels=27)
    sd = paramgen.fill_state_dict(net.state_dict(), 5)
    net.load_state_dict(sd)
    x = paramgen.tensor("dn_x", (3, 32, 6, 10), 1)
    m = paramgen.tensor("dn_m", (3, 27), 2)
    g = paramgen.tensor("dn_g", (3, 28, 6, 10), 3)

    def run(lib):
        monkeypatch.setattr(vt, "_DEPTHNET_LIB", lib)
        n = vt.DepthNet(32, 32, 16, 12, cam_channels=27)
        n.load_state_dict(sd)
        n = n.to(be.device).train()
        for mod in n.modules():                            # (dropout off: this test is about the convolutions)
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
        xd = be.to(x).requires_grad_()
        y = n(xd, be.to(m))
        (y * be.to(g)).sum().backward()
        return y.detach().cpu(), xd.grad.cpu(), {k: p.grad.cpu() for k, p in n.named_parameters() if p.grad is not None}

    y0, dx0, g0 = run(False)
    y1, dx1, g1 = run(True)
    assert _rel(y1, y0) < 2e-4 and _rel(dx1, dx0) < 2e-3
    num = sum(float((g1[k] - g0[k]).norm() ** 2) for k in g0)
    den = sum(float(g0[k].norm() ** 2) for k in g0)
    assert (num / den) ** 0.5 < 2e-3, (num / den) ** 0.5


@pytest.mark.parametrize("drop", [0.3, 0.0])
def test_swin_block_fused_droppath_nodes(bound_ops, monkeypatch, drop):
    """autograd.ProjDropPath / SwinFfn (DropPath + identity add, GELU + pre-activation, GELU' in the epilogues of the
    streaming linear, csrc/gemm_stream.h) against the node-per-op graph of the same block (Linear, Act, Linear,
    DropPathAdd: the graph the oracle comparisons of tests/test_train_step.py pin) on the same DropPath draws -- the
    SwinBlock of window_attention.py:300-344 in train mode.  Forward, the token gradient and every parameter gradient."""
    from occformer_amd import autograd as A
    from occformer_amd import noise
    from occformer_amd.encoder import SwinBlock
    from occformer_amd.training import DeviceRNG
    be = bound_ops
    monkeypatch.setenv("OCCF_GEMM_STREAM", "64")
    monkeypatch.setenv("OCCF_GEMM_STREAM_WGS", "16")
    C, B, X, Y, S = 64, 1, 7, 9, 3
    blk = SwinBlock(C, C // 32, C, window_size=7, drop_path_rate=drop)
    blk.load_state_dict(paramgen.fill_state_dict(blk.state_dict(), 21))
    blk = blk.to(be.device).train()
    tok = _t("sw_tok", (B, X, Y, S, C), 3)
    g = _t("sw_g", (B, X, Y, S, C), 4)
    lib = be.ops.lib

    def run(fuse):
        monkeypatch.setattr(A, "_SWIN_FUSE", fuse)
        for p in blk.parameters():
            p.grad = None
        x = be.to(tok).requires_grad_()
        noise.set_rng(DeviceRNG(be.device, 17))
        try:
            n0 = lib.occf_linear_stream_launches()
            y = blk(x)
            (y * be.to(g)).sum().backward()
            took = lib.occf_linear_stream_launches() - n0
        finally:
            noise.set_rng(None)
        return y.detach().cpu(), x.grad.cpu(), {k: p.grad.cpu().clone() for k, p in blk.named_parameters()}, took

    y0, dx0, gr0, took0 = run(False)
    y1, dx1, gr1, took1 = run(True)
    # fused: projection, FFN in, FFN out forward + the GELU' data gradient = 4 launches the node-per-op graph does not
    # make through occf_linear_stream_fwd (its own linears go through occf_linear_bf16_fwd, which also streams them)
    assert took1 >= 4
    assert _rel(y1, y0) < 1e-5 and _rel(dx1, dx0) < 1e-4
    for k in gr0:
        assert _rel(gr1[k], gr0[k]) < 2e-4, k
    assert float((y0 - tok.detach()).abs().max().item()) > 1e-3     # (the branches are not dropped altogether)
