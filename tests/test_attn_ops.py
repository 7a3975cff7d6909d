"""Encoder / pixel-decoder / occupancy-decoder kernels
against the oracle.  CPU: host emulation of the kernel sources; -m gpu: gfx950 library."""
import pytest
import torch
import torch.nn.functional as F

from oracle import occformer_ref as O
from tests import paramgen

TOL = dict(atol=2e-5, rtol=1e-4)     # fp32 kernels vs fp32 oracle; north_star bound is 1e-3


@pytest.mark.parametrize("X,Y,S,heads,shift,B", [(14, 14, 3, 1, 0, 1), (10, 9, 2, 2, 3, 2),
                                                 (7, 16, 1, 4, 3, 1), (5, 5, 2, 5, 0, 1)])
def test_window_attention(be, X, Y, S, heads, shift, B):
    C = heads * 32
    sd = {"w.qkv.weight": paramgen.tensor("qkvw", (3 * C, C), 1, C ** -0.5),
          "w.qkv.bias": paramgen.tensor("qkvb", (3 * C,), 1, 0.3),
          "w.proj.weight": torch.eye(C), "w.proj.bias": torch.zeros(C),
          "w.relative_position_bias_table": paramgen.tensor("tab", (169, heads), 1, 0.5)}
    y = paramgen.tensor("tokens", (B * S, X * Y, C), 2)                  # (b s) (x y) c, "after LN"
    ref = O.shift_window_msa(sd, "", y, X, Y, heads, shift) if False else None
    sd2 = {k.replace("w.", "a.w_msa."): v for k, v in sd.items()}
    ref = O.shift_window_msa(sd2, "a.", y, X, Y, heads, shift)           # [(b s), x*y, C]
    yk = y.view(B, S, X, Y, C).permute(0, 2, 3, 1, 4).reshape(-1, C).contiguous()
    qkv = F.linear(yk, sd["w.qkv.weight"], sd["w.qkv.bias"]).contiguous()
    out = be.ops.window_attention(*be.to(qkv, sd["w.qkv.bias"], sd["w.relative_position_bias_table"]),
                                  B, X, Y, S, heads, shift).cpu()
    out = out.view(B, X, Y, S, C).permute(0, 3, 1, 2, 4).reshape(B * S, X * Y, C)
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())


def test_window_mask_and_index_tables():
    """known-answer helpers the kernel re-derives arithmetically"""
    from tests.conftest import golden
    g = golden("tables")
    assert torch.equal(O.rel_pos_index(7).int(), g["rel_pos_index"])
    m = O.shift_window_mask(14, 14, 7, 3)
    assert m.shape == (4, 49, 49) and float(m[0].abs().sum()) == 0.0 and float(m[3].min()) == -100.0


@pytest.mark.parametrize("E,heads,shapes", [(96, 8, [(2, 2, 1), (4, 4, 2), (8, 8, 4)]),
                                            (48, 4, [(3, 2, 2), (5, 4, 3)]),
                                            (40, 8, [(2, 3, 1), (4, 4, 2), (6, 5, 3)]),
                                            (192, 8, [(2, 2, 1), (4, 3, 2)])])     # head dim 24 as in the configs
def test_msda3d(be, E, heads, shapes):
    B, P = 2, 4
    L = len(shapes)
    Nq = sum(x * y * z for x, y, z in shapes)
    value = paramgen.tensor("value", (B, Nq, E), 1)
    offs = paramgen.tensor("offs", (B, Nq, heads * L * P * 3), 1, 2.0)
    logits = paramgen.tensor("logits", (B, Nq, heads * L * P), 1)
    ref_pts = torch.cat([O.reference_points_3d(s) for s in shapes], 0)[None, :, None, :].expand(B, -1, L, -1)
    norm = torch.tensor([[s[2], s[1], s[0]] for s in shapes], dtype=torch.float32)
    loc = ref_pts[:, :, None, :, None, :] + offs.view(B, Nq, heads, L, P, 3) / norm[None, None, None, :, None, :]
    w = logits.view(B, Nq, heads, L * P).softmax(-1).view(B, Nq, heads, L, P)
    ref = O.msda3d_core(value.view(B, Nq, heads, E // heads), shapes, loc, w)
    out = be.ops.msda3d(*be.to(value, offs, logits), shapes, heads, P).cpu()
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())
    vhm = value.view(B, Nq, heads, E // heads).permute(0, 2, 1, 3).contiguous()
    out2 = be.ops.msda3d(*be.to(vhm, offs, logits), shapes, heads, P, head_major=True).cpu()
    assert torch.equal(out2, out)


@pytest.mark.parametrize("shape,target", [((16, 16, 8), (4, 4, 2)), ((16, 16, 8), (8, 8, 4)),
                                          ((10, 7, 5), (3, 2, 2)), ((6, 6, 2), (6, 6, 2))])
def test_mask_pool(be, shape, target):
    B, Q = 2, 5
    mp = paramgen.tensor("mp", (B, Q, *shape), 3, 2.0)
    mp[0, 1] = -mp[0, 1].abs() - 0.1                                     # a fully blocked row
    pooled, blocked, row_open = be.ops.mask_pool(be.to(mp), target)
    ref = F.adaptive_max_pool3d(mp, target).flatten(2)
    assert torch.equal(pooled.cpu(), ref)
    assert torch.equal(blocked.cpu().bool(), ref.sigmoid() < 0.5)
    assert torch.equal(row_open.cpu().view(B, Q).bool(), ~(ref.sigmoid() < 0.5).all(-1))
    assert not bool(row_open.cpu().view(B, Q)[0, 1])


@pytest.mark.parametrize("Q,L,heads,masked", [(20, 70, 3, True), (100, 300, 2, True), (7, 1500, 1, True),
                                              (20, 64, 2, False), (130, 90, 1, True)])
def test_masked_attention(be, Q, L, heads, masked):
    B, E = 2, heads * 32
    q = paramgen.tensor("q", (B, Q, E), 1)
    k = paramgen.tensor("k", (B, L, E), 1)
    v = paramgen.tensor("v", (B, L, E), 1)
    blocked = paramgen.uniform("blk", (B, Q, L), 1) < 0.6
    blocked[0, 0] = True                                                 # all-masked row -> unmasked
    blocked[1, 2, : L - 1] = True                                        # single open key at the end
    blocked[1, 3, 1:] = True                                             # single open key at the start
    fixed = blocked & ~blocked.all(-1, keepdim=True)
    qh = q.view(B, Q, heads, 32).transpose(1, 2) * 32 ** -0.5
    kh = k.view(B, L, heads, 32).transpose(1, 2)
    vh = v.view(B, L, heads, 32).transpose(1, 2)
    att = qh @ kh.transpose(-2, -1)
    if masked:
        att = att.masked_fill(fixed.unsqueeze(1), float("-inf"))
    ref = (att.softmax(-1) @ vh).transpose(1, 2).reshape(B, Q, E)
    if masked:
        row_open = (~blocked.all(-1)).int().view(-1)
        out = be.ops.masked_attention(*be.to(q, k, v), heads, *be.to(blocked.to(torch.uint8).contiguous(), row_open))
    else:
        out = be.ops.masked_attention(*be.to(q, k, v), heads)
    assert torch.allclose(out.cpu(), ref, **TOL), float((out.cpu() - ref).abs().max())


@pytest.mark.parametrize("shape,occ,Q,K", [((8, 8, 4), (16, 16, 8), 12, 17), ((5, 6, 3), (9, 12, 5), 12, 17),
                                           ((4, 4, 2), (4, 4, 2), 12, 17),      # same grid: identity kernel
                                           ((6, 5, 4), (6, 5, 4), 21, 19),      # identity, wide class bucket
                                           ((2, 4, 16), (4, 8, 32), 7, 17),     # Z2 = 32: LDS-staged kernel
                                           ((5, 6, 16), (10, 12, 32), 9, 17),   # LDS-staged, several workgroups, odd Q
                                           ((4, 6, 16), (10, 16, 32), 5, 20),   # LDS-staged, more than 2x upsampling
                                           ((3, 3, 3), (3, 3, 3), 9, 17)])      # odd voxel count: resampler
def test_upsample_classify(be, shape, occ, Q, K):
    B = 2
    mp = paramgen.tensor("mp", (B, Q, *shape), 5, 2.0)
    cls = paramgen.tensor("cls", (B, Q, K + 1), 5)
    ref = O.format_results(cls, F.interpolate(mp, size=occ, mode="trilinear", align_corners=True))
    out = be.ops.upsample_classify(*be.to(mp, cls), occ).cpu()
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())


def test_lidarseg_sample(be):
    B, Q, K = 2, 70, 17
    shape = (8, 6, 4)
    pc_range = [-8.0, -8.0, -2.0, 8.0, 8.0, 2.0]
    mp = paramgen.tensor("mp", (B, Q, *shape), 6, 2.0)
    cls = paramgen.tensor("cls", (B, Q, K + 1), 6)
    lo, hi = torch.tensor(pc_range[:3]), torch.tensor(pc_range[3:])
    pts = [paramgen.uniform(f"p{b}", (50 + b, 3), 6) * (hi - lo) * 1.2 + lo - 0.1 * (hi - lo) for b in range(B)]
    ref = O.lidarseg_points(cls, mp, pts, pc_range)
    rows = []
    for b, p in enumerate(pts):
        g = (p - lo) / (hi - lo) * 2 - 1
        rows.append(torch.cat((torch.full((p.shape[0], 1), float(b)), g), 1))
    out = be.ops.lidarseg_sample(*be.to(mp, cls, torch.cat(rows, 0).contiguous())).cpu()
    assert torch.allclose(out, ref, **TOL), float((out - ref).abs().max())


@pytest.mark.parametrize("E", [64, 192])         # 192 channels: the streaming kernel (where the geometry allows)
@pytest.mark.parametrize("shape,target,Q", [((8, 8, 16), (4, 4, 8), 20), ((8, 8, 16), (2, 2, 2), 100),
                                            ((4, 16, 8), (2, 4, 2), 9), ((8, 8, 16), (8, 8, 16), 128),
                                            ((4, 32, 4), (1, 2, 1), 3), ((2, 4, 32), (1, 2, 4), 33),
                                            ((16, 8, 16), (2, 2, 8), 70)])
def test_mask_gemm_pool_fused(be, monkeypatch, shape, target, Q, E):
    """fused GEMM+pool == (same split-bf16 GEMM, then the pooling kernel), bit for bit; x-window slices,
    several windows per tile, window = voxel"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    B = 2
    X, Y, Z = shape
    me = paramgen.tensor("me", (B, Q, E), 1)
    feat = paramgen.tensor("mf", (B, X * Y * Z, E), 2)
    sp = be.ops.split_bf16(be.to(feat))
    mp = torch.empty(B, Q, X * Y * Z)
    mpd = be.to(mp)
    for b in range(B):
        be.ops.linear(be.to(me)[b], be.to(feat)[b], out=mpd[b], w_split=(sp[0][b], sp[1][b]), allow_small=False)
    ref_pooled, ref_blocked, ref_open = be.ops.mask_pool(mpd.view(B, Q, X, Y, Z), target)
    pooled, blocked, row_open = be.ops.mask_gemm_pool(be.to(me), sp, shape, target)
    again = be.ops.mask_gemm_pool(be.to(me), sp, shape, target)         # the other traversal direction
    assert all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(again, (pooled, blocked, row_open)))
    assert torch.equal(pooled.cpu(), ref_pooled.cpu())
    assert torch.equal(blocked.cpu(), ref_blocked.cpu()) and torch.equal(row_open.cpu(), ref_open.cpu())
    # and against plain fp32 torch within the split-bf16 accuracy
    full = torch.einsum("bqe,bve->bqv", me, feat).view(B, Q, X, Y, Z)
    assert torch.allclose(pooled.cpu(), F.adaptive_max_pool3d(full, target).flatten(2), atol=3e-4, rtol=1e-4)


@pytest.mark.parametrize("shape,target", [((6, 10, 8), (3, 5, 4)), ((5, 7, 4), (2, 3, 2)), ((8, 8, 16), (3, 4, 8))])
def test_mask_gemm_pool_declines_non_uniform_windows(be, monkeypatch, shape, target):
    """overlapping adaptive windows / tiles that straddle windows: the fused kernel declines (None)
    and the head falls back to GEMM + mask_pool"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    X, Y, Z = shape
    me = paramgen.tensor("me", (1, 5, 64), 1)
    feat = paramgen.tensor("mf", (1, X * Y * Z, 64), 2)
    assert be.ops.mask_gemm_pool(be.to(me), be.ops.split_bf16(be.to(feat)), shape, target) is None


@pytest.mark.parametrize("X,Y,S,shift,B", [(14, 14, 3, 0, 1), (10, 9, 2, 3, 2), (7, 16, 1, 3, 1), (5, 5, 2, 0, 1)])
@pytest.mark.parametrize("packed", [True, False])      # weights in MFMA-fragment order / row-major
def test_swin_attention_fused(be, monkeypatch, X, Y, S, shift, B, packed):
    """x + proj(window_msa(layernorm(x))) in one kernel == the oracle's LayerNorm -> ShiftWindowMSA -> residual
    (C = 128 / 4 heads; padded, shifted and partial windows)"""
    monkeypatch.setattr(be.ops, "precision", "bf16x3")
    monkeypatch.setattr(be.ops, "swin_frag", packed)
    C, heads = 128, 4
    sd = {"a.w_msa.qkv.weight": paramgen.tensor("fqkvw", (3 * C, C), 1, C ** -0.5),
          "a.w_msa.qkv.bias": paramgen.tensor("fqkvb", (3 * C,), 1, 0.3),
          "a.w_msa.proj.weight": paramgen.tensor("fpw", (C, C), 1, C ** -0.5),
          "a.w_msa.proj.bias": paramgen.tensor("fpb", (C,), 1, 0.2),
          "a.w_msa.relative_position_bias_table": paramgen.tensor("ftab", (169, heads), 1, 0.5)}
    g = 1 + 0.2 * paramgen.tensor("flg", (C,), 2)
    bt = 0.1 * paramgen.tensor("flb", (C,), 3)
    x = paramgen.tensor("ftok", (B * S, X * Y, C), 2, 2.0) + 0.5           # (b s) (x y) c
    ref = x + O.shift_window_msa(sd, "a.", F.layer_norm(x, (C,), g, bt, 1e-5), X, Y, heads, shift)
    xk = x.view(B, S, X, Y, C).permute(0, 2, 3, 1, 4).reshape(-1, C).contiguous()
    ops = be.ops
    wq, wp = be.to(sd["a.w_msa.qkv.weight"], sd["a.w_msa.proj.weight"])
    out = ops.swin_attention_fused(be.to(xk), *be.to(g, bt), 1e-5, ops.split_bf16(wq), be.to(sd["a.w_msa.qkv.bias"]),
                                   be.to(sd["a.w_msa.relative_position_bias_table"]), ops.split_bf16(wp),
                                   be.to(sd["a.w_msa.proj.bias"]), B, X, Y, S, heads, shift)
    assert out is not None
    out = out.cpu().view(B, X, Y, S, C).permute(0, 3, 1, 2, 4).reshape(B * S, X * Y, C)
    assert torch.allclose(out, ref, atol=2e-4, rtol=2e-4), float((out - ref).abs().max())


@pytest.mark.parametrize("B,Q,E,H,n_cls", [(1, 100, 192, 1536, 18), (2, 37, 64, 96, 21)])
def test_decoder_rows_kernels(be, B, Q, E, H, n_cls):
    """csrc/decoder_rows.hip: the decoder layer's per-query chain as two kernels (K1 after the cross-attention, K2 after
    the self-attention, K2 in head-only mode for the prediction set in front of layer 0) against the same chain in
    fp64 torch -- mmcv's DetrTransformerDecoderLayer order ('cross_attn','norm','self_attn','norm','ffn','norm') with
    the positional term added to q and k only, and forward_head's post_norm / cls_embed / mask_embed
    (mask2former_nusc_occ.py:426-456, 640-667).  Rows that do not fill the last 16-row workgroup; two samples (qpos per
    sample); a class count that is not a multiple of 16."""
    import torch.nn.functional as F
    ops = be.ops
    t = lambda n, shp, sc=1.0: paramgen.tensor("dr." + n, shp, 1, sc)
    lin = lambda n, N, K: (t(n + ".w", (N, K), K ** -0.5), t(n + ".b", (N,), 0.2))
    ln = lambda n: (1 + t(n + ".g", (E,), 0.2), t(n + ".bt", (E,), 0.2), 1e-5)
    W = {n: lin(n, *nk) for n, nk in dict(out0=(E, E), qk=(2 * E, E), v=(E, E), out1=(E, E), ffn1=(H, E), ffn2=(E, H),
                                           cls=(n_cls, E), me0=(E, E), me1=(E, E), me2=(E, E), xq=(E, E)).items()}
    N = {n: ln(n) for n in ("ln0", "ln1", "ln2", "post")}
    o, q, o2 = t("o", (B, Q, E)), t("q", (B, Q, E)), t("o2", (B, Q, E))
    qpos = t("qpos", (Q, E))
    d64 = lambda x: x.double()
    L = lambda x, n: F.linear(x, d64(W[n][0]), d64(W[n][1]))
    LN = lambda x, n: F.layer_norm(x, (E,), d64(N[n][0]), d64(N[n][1]), N[n][2])
    pk = lambda n: (ops.decoder_rows_pack(be.to(W[n][0])), be.to(W[n][1]))
    nd = lambda n: (be.to(N[n][0]), be.to(N[n][1]), N[n][2])
    layer = dict(out0=pk("out0"), ln0=nd("ln0"), qk=pk("qk"), v=pk("v"), out1=pk("out1"), ln1=nd("ln1"), ffn1=pk("ffn1"),
                 ffn2=pk("ffn2"), ln2=nd("ln2"), H=H)
    head = dict(post=nd("post"), cls=pk("cls"), n_cls=n_cls, me0=pk("me0"), me1=pk("me1"), me2=pk("me2"))
    od, qd, o2d, qposd = be.to(o, q, o2, qpos)
    # K1
    q1r = LN(L(d64(o), "out0") + d64(q), "ln0")
    qkr = L(q1r + d64(qpos), "qk")
    q1, qs, ks, vs = ops.decoder_rows_k1(od, qd, qposd, layer)
    tol = lambda ref: 3e-5 * float(ref.abs().max())
    assert float((q1.cpu() - q1r).abs().max()) < tol(q1r)
    assert float((qs.cpu() - qkr[..., :E]).abs().max()) < tol(qkr) and float((ks.cpu() - qkr[..., E:]).abs().max()) < tol(qkr)
    vr = L(q1r, "v")
    assert float((vs.cpu() - vr).abs().max()) < tol(vr)
    # K2, whole chain (on the kernel's own q1 as the residual operand)
    q1x = d64(q1.cpu())
    q2r = LN(L(d64(o2), "out1") + q1x, "ln1")
    q3r = LN(L(F.relu(L(q2r, "ffn1")), "ffn2") + q2r, "ln2")
    dr = LN(q3r, "post")
    clsr = L(dr, "cls")
    mer = L(F.relu(L(F.relu(L(dr, "me0")), "me1")), "me2")
    qxr = L(q3r + d64(qpos), "xq")
    q3, cls, me, qx = ops.decoder_rows_k2(o2d, q1, qposd, layer, head, pk("xq"))
    for got, ref in ((q3, q3r), (cls, clsr), (me, mer), (qx, qxr)):
        assert tuple(got.shape) == tuple(ref.shape)
        assert float((got.cpu() - ref).abs().max()) < tol(ref)
    # K2, head only, no next layer
    none, cls0, me0, qx0 = ops.decoder_rows_k2(None, qd, qposd, None, head, None)
    d0 = LN(d64(q), "post")
    assert none is None and qx0 is None
    assert float((cls0.cpu() - L(d0, "cls")).abs().max()) < tol(clsr)
    me0r = L(F.relu(L(F.relu(L(d0, "me0")), "me1")), "me2")
    assert float((me0.cpu() - me0r).abs().max()) < tol(me0r)
