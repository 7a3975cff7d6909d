"""torch.autograd.Function wrappers that pair every forward entry point of liboccformer_hip.so with its hand-written
backward (csrc/bwd_elem.hip, wgrad.hip, attn_bwd.hip, msda3d.hip, dcn.hip, lss.hip) -- the training step of the path.

The reference's counterpart is ATen autograd through the PyTorch modules plus ``QuickCumsumCuda``
(mmdet3d/ops/bev_pool/bev_pool.py:37-80); ``loss.backward()`` of ``OccupancyFormer.forward_train``
(projects/mmdet3d_plugin/occformer/detectors/occupancyformer.py:132-199) runs through these Functions here.
torch is used for the graph bookkeeping, device memory and streams only; every gradient is computed by a kernel of
the library.  Parameters enter as Function inputs so autograd accumulates into their ``.grad`` (what DDP hooks).
"""
import os

import torch

from . import fused, noise
from .ops import get_ops

_WT_CACHE = {}
_FLIP_CACHE = {}
_DG_CACHE = {}


def _split(t):
    ops = get_ops()
    return None if ops.precision == "f32" else ops.split_bf16(t)


_w2d = fused._w2d


def _wt(weight):
    """W^T [K, N] (+ bf16 split) of a linear weight [N, K(, 1, 1, 1)], cached per parameter version"""
    hit = fused.prepared(weight, "wt")
    if hit is not None:
        return hit

    def make():
        wt = _w2d(weight.detach()).t().contiguous()
        return wt, _split(wt)
    return fused._versioned(_WT_CACHE, weight, make)


class Linear(torch.autograd.Function):
    """y = act(x W^T + b) [+ residual]; act in {0, 1 = ReLU} (ReLU only without residual).  ``weight`` is the
    PARAMETER ([N, K], or a 1x1(x1) convolution weight [N, K, 1, 1(, 1)]): its identity / version keys the cached
    bf16 split and transpose."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, residual, rows=None, head_major=None):
        """``rows = (lo, hi)``: use weight[lo:hi] / bias[lo:hi] (the q / k / v blocks of nn.MultiheadAttention's
        in_proj_weight); the gradients come back full-size, zero outside the block.
        ``head_major = (rows per batch, head_dim)``: x [B, rows, K] -> y [B, N / head_dim, rows, head_dim] written by
        the GEMM epilogue (the layout the deformable-attention sampler gathers from)"""
        assert not (act and residual is not None) and act in (0, 1)
        ops = get_ops()
        w2 = _w2d(weight.detach())
        b = None if bias is None else bias.detach()
        sp = fused.split_weight(weight, _w2d)
        if rows is not None:
            lo, hi = rows
            w2, b = w2[lo:hi], (None if b is None else b[lo:hi])
            sp = None if sp is None else (sp[0][lo:hi], sp[1][lo:hi])
        if head_major is not None:
            assert not act and residual is None and rows is None
            y = ops.linear(x, w2, b, 0, None, w_split=sp, head_major=head_major)
        else:
            y = ops.linear(x, w2, b, act, None if residual is None else residual.detach(), w_split=sp, allow_small=True)
        ctx.save_for_backward(x, weight, y if act else None)
        ctx.has_bias, ctx.has_res, ctx.act, ctx.rows = bias is not None, residual is not None, act, rows
        ctx.hm = head_major is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        ops = get_ops()
        Nfull, K = weight.shape[0], weight.numel() // weight.shape[0]
        lo, hi = ctx.rows if ctx.rows is not None else (0, Nfull)
        N = hi - lo
        # head-major output: the gradient [B, heads, rows, dh] is normally the permuted view of a token-major tensor
        g = dy.permute(0, 2, 1, 3).contiguous() if ctx.hm else dy.contiguous()
        if ctx.act:
            g = ops.act_backward(y, g, 1)
        g2, x2 = g.reshape(-1, N), x.reshape(-1, K)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt, sp = _wt(weight)
            if ctx.rows is not None:                     # W[lo:hi]^T = columns lo:hi of W^T: made contiguous per call
                wt = wt[:, lo:hi].contiguous()
                sp = _split(wt)
            dx = ops.linear(g2, wt, None, w_split=sp).view(x.shape)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.linear_wgrad(g2, x2 if x2.stride(1) == 1 else x2.contiguous(), want_bias=ctx.has_bias)
            if ctx.rows is not None:
                full = torch.zeros((Nfull, K), dtype=dw.dtype, device=dw.device)
                full[lo:hi] = dw
                dw = full
                if db is not None:
                    fb = torch.zeros((Nfull,), dtype=db.dtype, device=db.device)
                    fb[lo:hi] = db
                    db = fb
            dw = dw.view(weight.shape)
        return dx, dw, db, None, (dy if ctx.has_res else None), None, None


class InProj(torch.autograd.Function):
    """nn.MultiheadAttention's input projections: q = xq Wq^T + bq, k = xk Wk^T + bk, v = xv Wv^T + bv with
    W = in_proj_weight [3E, E] = (Wq | Wk | Wv).  One node instead of three row-sliced ``Linear``s, so the weight / bias
    gradients are assembled once (one concatenation) instead of three zero-filled full-size gradients summed by
    autograd -- the decoder calls this 18 times per training step."""

    @staticmethod
    def forward(ctx, xq, xk, xv, weight, bias):
        ops = get_ops()
        E = weight.shape[1]
        w, b = weight.detach(), bias.detach()
        sp = fused.split_weight(weight, _w2d)
        outs = []
        for i, x in enumerate((xq, xk, xv)):
            lo, hi = i * E, (i + 1) * E
            outs.append(ops.linear(x, w[lo:hi], b[lo:hi], 0, None, w_split=None if sp is None else (sp[0][lo:hi], sp[1][lo:hi]),
                                   allow_small=True))
        ctx.save_for_backward(xq, xk, xv, weight)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        xq, xk, xv, weight = ctx.saved_tensors
        ops = get_ops()
        E = weight.shape[1]
        wt, _ = _wt(weight)                                # [E, 3E]
        dxs, dws, dbs = [], [], []
        for i, (x, g) in enumerate(((xq, dq), (xk, dk), (xv, dv))):
            lo, hi = i * E, (i + 1) * E
            g2, x2 = g.contiguous().reshape(-1, E), x.reshape(-1, E)
            dx = None
            if ctx.needs_input_grad[i]:
                wti = wt[:, lo:hi].contiguous()
                dx = ops.linear(g2, wti, None, w_split=_split(wti)).view(x.shape)
            dxs.append(dx)
            dw, db = ops.linear_wgrad(g2, x2 if x2.stride(1) == 1 else x2.contiguous(), want_bias=True)
            dws.append(dw)
            dbs.append(db)
        return dxs[0], dxs[1], dxs[2], torch.cat(dws, 0), torch.cat(dbs, 0)


def linear(x, lin, act=0, residual=None, heavy_gate=False):
    """``heavy_gate``: how the comparison tap (noise.relu_gate) classifies this layer's ReLU units"""
    y = Linear.apply(x, lin.weight, lin.bias, act, residual, None)
    return noise.relu_gate(y, heavy_gate) if act == 1 else y


def _conv_geometry(conv):
    ks = tuple(conv.kernel_size) + (1,) * (3 - len(conv.kernel_size))
    pad = tuple(conv.padding) + (0,) * (3 - len(conv.padding))
    stride, dil = conv.stride[0], conv.dilation[0]
    assert all(s == stride for s in conv.stride) and all(d == dil for d in conv.dilation) and conv.groups == 1
    return ks, stride, dil, pad


class Conv3d(torch.autograd.Function):
    """channels-last convolution y = conv(x) + b on the implicit-GEMM / halo kernels; weight in nn.Conv layout"""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, ks, stride, dil, pad, fork=False, gn=None):
        """``fork``: also return x_cl itself (-> (x_cl, y)) for the residual connection around the convolution: the node
        then receives the residual gradient too and the data-gradient convolution adds it in its epilogue (a separate
        ATen add of two [1, 200, 200, 16, 128] gradients per Dualpath / ASPP block otherwise)"""
        ops = get_ops()
        w2 = fused.tap_major_of(weight)
        # ``gn = (groups, eps)``: the GroupNorm that follows takes its statistics from this convolution's epilogue
        # (``ops.last_gn_stats`` right after the call; None when the launch shape has no such epilogue)
        y = ops.conv3d(x_cl, w2, ks, stride, dil, pad, None if bias is None else bias.detach(),
                       w_split=fused.split_weight(weight, fused._tap_layout), gn=gn)
        ctx.save_for_backward(x_cl, weight)
        ctx.geom = (ks, stride, dil, pad)
        ctx.has_bias = bias is not None
        ctx.fork = fork
        if fork:
            ctx.set_materialize_grads(False)
            return x_cl, y
        return y

    @staticmethod
    def backward(ctx, *grads):
        dres, dy = grads if ctx.fork else (None, grads[0])
        if dy is None:
            return dres, None, None, None, None, None, None, None, None
        x_cl, weight = ctx.saved_tensors
        ks, stride, dil, pad = ctx.geom
        ops = get_ops()
        g = dy.contiguous()
        Cout, Cin = weight.shape[0], weight.shape[1]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if stride == 1:
                # the data gradient of a stride-1 convolution is a convolution of dy with the taps flipped and the
                # channel roles swapped (padding dil*(k-1) - pad): the 3^3 case runs on the LDS-halo kernel
                def make():
                    w5 = weight.detach() if weight.dim() == 5 else weight.detach().unsqueeze(-1)
                    wf = w5.flip(2, 3, 4).permute(1, 2, 3, 4, 0).reshape(Cin, -1).contiguous()   # [Cin, taps*Cout]
                    return wf, _split(wf)
                wf, sp = fused.prepared(weight, "flip") or fused._versioned(_FLIP_CACHE, weight, make)
                dpad = tuple(dil * (k - 1) - p for k, p in zip(ks, pad))
                # (act_f16: dy may enter as ONE fp16 piece where the Winograd kernel takes the shape -- ops.dgrad_f16)
                dx = ops.conv3d(g, wf, ks, 1, dil, dpad, None, w_split=sp,
                                residual=None if dres is None else dres.contiguous(), act_f16=ops.dgrad_f16)
                dres = None
            else:
                def make():
                    w5 = weight.detach() if weight.dim() == 5 else weight.detach().unsqueeze(-1)
                    wt = w5.permute(1, 2, 3, 4, 0).reshape(Cin, -1).contiguous()                   # [Cin, taps*Cout]
                    return wt, ops.split_bf16(wt)
                wt, sp = fused.prepared(weight, "dg") or fused._versioned(_DG_CACHE, weight, make)
                dx = ops.conv3d_dgrad(g, sp, tuple(x_cl.shape), ks, stride, dil, pad)
        if dres is not None:
            dx = dres if dx is None else dx + dres
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw2, db = ops.conv3d_wgrad(g, x_cl, ks, stride, dil, pad, want_bias=ctx.has_bias)
            dw = dw2.view(Cout, *ks, Cin).permute(0, 4, 1, 2, 3)
            if weight.dim() == 4:
                dw = dw.squeeze(-1)
            # (canonical strides also for size-1 kernel dims: a [Cout, Cin, 1, 1, 1] gradient permuted out of the
            # tap-major layout "is contiguous" with strides [Cin, 1, Cin, Cin, Cin], which DDP's bucket views refuse --
            # r05a: "Grad strides do not match bucket view strides", i.e. a copy per such gradient and step)
            dw = dw.contiguous().reshape(-1).view(weight.shape)
        return dx, dw, db, None, None, None, None, None, None


_GN_EPILOGUE = os.environ.get("OCCF_TRAIN_GN_EPILOGUE", "1") == "1"


def conv_fork(x_cl, conv_mod, gn=None):
    """-> (x_cl as the residual operand, conv(x_cl), GroupNorm statistics of the output for ``gn`` or None); see
    Conv3d.forward"""
    ks, stride, dil, pad = _conv_geometry(conv_mod)
    ops = get_ops()
    ident, y = Conv3d.apply(x_cl, conv_mod.weight, conv_mod.bias, ks, stride, dil, pad, True,
                            None if gn is None or not _GN_EPILOGUE else (gn.num_groups, gn.eps))
    stats, ops.last_gn_stats = ops.last_gn_stats, None
    return ident, y, (stats if gn is not None else None)


def conv(x_cl, conv_mod):
    """nn.Conv3d / nn.Conv2d module on channels-last [B, X, Y, Z, C] (no activation; bias included)"""
    ks, stride, dil, pad = _conv_geometry(conv_mod)
    if ks == (1, 1, 1) and stride == 1 and x_cl.is_contiguous():
        return Linear.apply(x_cl, conv_mod.weight, conv_mod.bias, 0, None, None)   # [Cout, Cin, 1, 1(, 1)] read as [Cout, Cin]
    return Conv3d.apply(x_cl, conv_mod.weight, conv_mod.bias, ks, stride, dil, pad)


class GroupNorm(torch.autograd.Function):
    """occf_groupnorm_apply: y = relu?(GN(x)) [+ residual], token mode appends the z-mean slot"""

    @staticmethod
    def forward(ctx, x_cl, weight, bias, groups, eps, relu, tokens, residual, stats=None):
        ops = get_ops()
        x_cl = x_cl.contiguous()
        if stats is None:
            stats = ops.groupnorm_stats(x_cl, groups, eps)
        y = ops.groupnorm_apply(x_cl, stats, weight.detach(), bias.detach(), groups, relu, tokens,
                                None if residual is None else residual.detach().contiguous())
        ctx.save_for_backward(x_cl, stats, weight, bias)
        ctx.cfg = (groups, relu, tokens, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_cl, stats, weight, bias = ctx.saved_tensors
        groups, relu, tokens, has_res = ctx.cfg
        dx, dg, db, dres = get_ops().groupnorm_backward(x_cl, stats, weight.detach(), bias.detach(), dy.contiguous(),
                                                        groups, relu, tokens, want_residual=has_res)
        return dx, dg, db, None, None, None, None, dres, None


def group_norm(x_cl, gn, relu=False, tokens=False, residual=None, stats=None, gate_class=False):
    """``gate_class``: how the comparison tap classifies this map's ReLU units (noise.gates_wanted: False = light,
    "bev" = a BEV-ASPP map)"""
    y = GroupNorm.apply(x_cl, gn.weight, gn.bias, gn.num_groups, gn.eps, relu, tokens, residual, stats)
    if relu and noise.gates_wanted(gate_class):
        noise.relu_gate(y, gate_class, lambda: _gn_relu_gate(x_cl, gn, y, tokens, residual, stats))
    return y


def _gn_relu_gate(x_cl, gn, y, tokens, residual, stats):
    """comparison runs only: the ReLU gate of y = relu(GN(x)) [+ residual | + token slot] in the reference's
    channel-first layout"""
    if residual is not None:
        ops = get_ops()
        x = x_cl.detach().contiguous()
        st = stats if stats is not None else ops.groupnorm_stats(x, gn.num_groups, gn.eps)
        y = ops.groupnorm_apply(x, st, gn.weight.detach(), gn.bias.detach(), gn.num_groups, True, False, None)
    elif tokens:
        y = y.detach()[:, :, :, :-1]
    return (y.detach() > 0).movedim(-1, 1)


def conv_gn(x_cl, conv_mod, gn, relu=False, tokens=False, residual=None, gate_class=False):
    """conv -> GroupNorm (-> ReLU / token buffer / + residual); the statistics come from the convolution's epilogue
    where the launch has one (as in the inference path, fused.conv_gn) instead of a separate pass over its output"""
    ks, stride, dil, pad = _conv_geometry(conv_mod)
    if (ks == (1, 1, 1) and stride == 1 and x_cl.is_contiguous()) or not _GN_EPILOGUE:
        return group_norm(conv(x_cl, conv_mod), gn, relu, tokens, residual, gate_class=gate_class)
    ops = get_ops()
    y = Conv3d.apply(x_cl, conv_mod.weight, conv_mod.bias, ks, stride, dil, pad, False, (gn.num_groups, gn.eps))
    stats, ops.last_gn_stats = ops.last_gn_stats, None
    return group_norm(y, gn, relu, tokens, residual, stats, gate_class=gate_class)


class LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        return get_ops().layernorm(x, weight.detach(), bias.detach(), eps)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx, dg, db = get_ops().layernorm_backward(x, weight.detach(), dy.contiguous(), ctx.eps)
        return dx, dg, db, None


def layernorm(x, ln):
    return LayerNorm.apply(x, ln.weight, ln.bias, ln.eps)


class LayerNormFork(torch.autograd.Function):
    """(x, LN(x)) for a pre-norm residual block  x -> x + f(LN(x)):  the first output is x itself, to be used as the
    residual operand.  One node then receives BOTH gradients of x and the LayerNorm backward kernel adds the residual
    one in its own pass -- as two consumers of x, autograd summed them with a separate ATen add (three passes over
    [680 000, 128] per norm at the 200-grid, 2 ms per training step)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        assert x.is_contiguous()
        ctx.save_for_backward(x, weight)
        ctx.eps = eps
        ctx.set_materialize_grads(False)
        return x, get_ops().layernorm(x, weight.detach(), bias.detach(), eps)

    @staticmethod
    def backward(ctx, dres, dy):
        x, weight = ctx.saved_tensors
        if dy is None:
            return dres, None, None, None
        dx, dg, db = get_ops().layernorm_backward(x, weight.detach(), dy.contiguous(), ctx.eps,
                                                  addend=None if dres is None else dres.contiguous())
        return dx, dg, db, None


def layernorm_fork(x, ln):
    """-> (x as the residual operand, LN(x)); see LayerNormFork"""
    return LayerNormFork.apply(x.contiguous(), ln.weight, ln.bias, ln.eps)


class Act(torch.autograd.Function):
    """1 = ReLU, 2 = exact GELU"""

    @staticmethod
    def forward(ctx, x, act):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.act = act
        return get_ops().act_forward(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return get_ops().act_backward(x, dy.contiguous(), ctx.act), None


class DropPathAdd(torch.autograd.Function):
    """identity + branch * scale[sample] over the token buffer (sample = (batch, slice))"""

    @staticmethod
    def forward(ctx, identity, branch, scale, XY, S):
        ctx.save_for_backward(scale)
        ctx.geom = (XY, S)
        return get_ops().droppath(identity.contiguous(), branch.contiguous(), scale, XY, S)

    @staticmethod
    def backward(ctx, dy):
        (scale,) = ctx.saved_tensors
        dy = dy.contiguous()
        return dy, get_ops().droppath(None, dy, scale, *ctx.geom), None, None, None


# ---- the Swin block's two DropPath branches on the streaming kernel's epilogues (csrc/gemm_stream.h) -----------------
# window_attention.py:300-344 (SwinBlock.forward): x = x + drop_path(attn(norm1(x))); x = x + drop_path(ffn(norm2(x)))
# with mmcv's FFN = Linear, GELU, Linear.  As separate nodes (Linear, Act, Linear, DropPathAdd) the branch output, the
# pre-activation and the activation's gradient each make a round trip through HBM per block ([680 000, 128] at stage 0);
# here the DropPath scale + identity add ride in the projection's epilogue, the GELU in the first FFN linear's (which
# also writes the pre-activation the backward needs) and GELU' in the epilogue of the data gradient through the second.
_SWIN_FUSE = os.environ.get("OCCF_TRAIN_SWIN_FUSE", "1") == "1"


def _wgrad(g2, x2, weight, has_bias):
    dw, db = get_ops().linear_wgrad(g2, x2 if x2.stride(1) == 1 else x2.contiguous(), want_bias=has_bias)
    return dw.view(weight.shape), db


class ProjDropPath(torch.autograd.Function):
    """out = identity + scale[sample] * (x W^T + b)   (scale None: plain residual add)"""

    @staticmethod
    def forward(ctx, identity, x, weight, bias, scale, XY, S):
        ops = get_ops()
        sp = fused.split_weight(weight, _w2d)
        out = ops.linear_stream(x, sp, bias.detach(), 0, residual=identity.detach(), row_scale=scale, XY=XY, S=S)
        if out is None:
            # outside the streaming kernel's envelope after all (stream_fusable asks for the default arithmetic; the
            # envelope switches are re-read per call): the unfused sequence, same result
            y = ops.linear(x, weight.detach(), bias.detach(), w_split=sp)
            out = identity.detach() + y if scale is None else ops.droppath(identity.detach(), y, scale, XY, S)
        ctx.save_for_backward(x, weight, scale)
        ctx.geom = (XY, S)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, weight, scale = ctx.saved_tensors
        ops = get_ops()
        dy = dy.contiguous()
        db = dy if scale is None else ops.droppath(None, dy, scale, *ctx.geom)
        dw, dbias = _wgrad(db, x, weight, True)
        wt, sp = _wt(weight)
        dx = ops.linear(db, wt, None, w_split=sp)
        return dy, dx, dw, dbias, None, None, None


class SwinFfn(torch.autograd.Function):
    """out = identity + scale[sample] * (GELU(x W1^T + b1) W2^T + b2)"""

    @staticmethod
    def forward(ctx, identity, x, w1, b1, w2, b2, scale, XY, S):
        ops = get_ops()
        sp1, sp2 = fused.split_weight(w1, _w2d), fused.split_weight(w2, _w2d)
        r = ops.linear_stream(x, sp1, b1.detach(), 2, pre_out=True)
        if r is None:                                   # (the unfused sequence: see ProjDropPath)
            z = ops.linear(x, w1.detach(), b1.detach(), w_split=sp1)
            r = (ops.act_forward(z, 2), z)
        f, z = r
        out = ops.linear_stream(f, sp2, b2.detach(), 0, residual=identity.detach(), row_scale=scale, XY=XY, S=S)
        if out is None:
            y = ops.linear(f, w2.detach(), b2.detach(), w_split=sp2)
            out = identity.detach() + y if scale is None else ops.droppath(identity.detach(), y, scale, XY, S)
        ctx.save_for_backward(x, z, f, w1, w2, scale)
        ctx.geom = (XY, S)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, z, f, w1, w2, scale = ctx.saved_tensors
        ops = get_ops()
        dy = dy.contiguous()
        db = dy if scale is None else ops.droppath(None, dy, scale, *ctx.geom)
        dw2, db2 = _wgrad(db, f, w2, True)
        w2t, sp2 = _wt(w2)
        dz = ops.linear_stream(db, sp2, None, 3, aux=z)             # (db W2) * GELU'(z)
        if dz is None:
            dz = ops.act_backward(z, ops.linear(db, w2t, None, w_split=sp2), 2)
        dw1, db1 = _wgrad(dz, x, w1, True)
        w1t, sp1 = _wt(w1)
        dx = ops.linear(dz, w1t, None, w_split=sp1)
        return dy, dx, dw1, db1, dw2, db2, None, None, None


def stream_fusable(M, *dims):
    """do the Swin block's fused nodes apply?  (bf16 arithmetic, the rows and widths gemm_stream.h takes)"""
    ops = get_ops()
    if not _SWIN_FUSE or ops.precision == "f32" or not hasattr(ops, "linear_stream"):
        return False
    return all(ops.lib.occf_linear_stream_takes(M, n, k) == 1 for n, k in dims)


class TokenBevSlot(torch.autograd.Function):
    """(tok, tok[:, :, :, Z:Z+1]) of the token buffer [B, X, Y, Z + 1, C]: the z-mean slot feeds the BEV ASPP while the
    whole buffer goes on to the soft-gated fusion.  As a plain slice, the slot's gradient came back as a zero-filled
    copy of the WHOLE buffer (fill + strided copy + add: three passes over [680 000, 128] per Dualpath block); here
    it is added into the slot rows of the buffer's gradient, which this node alone receives."""

    @staticmethod
    def forward(ctx, tok, Z):
        ctx.Z = Z
        ctx.set_materialize_grads(False)
        return tok, tok[:, :, :, Z:Z + 1].contiguous()

    @staticmethod
    def backward(ctx, dtok, dslot):
        if dslot is None:
            return dtok, None
        Z = ctx.Z
        if dtok is None:
            shape = list(dslot.shape)
            shape[3] = Z + 1
            dtok = dslot.new_zeros(shape)
        elif not (getattr(dtok, "_occf_owned", False) and dtok.is_contiguous()):
            # never write to an incoming gradient that may be shared (another consumer of ``tok``, a tensor hook, a
            # retained graph: the autograd contract).  Only the buffer DualpathCombine.backward has just allocated and
            # tagged -- handed over by the engine untouched, one consumer -- is completed in place (one slot of Z + 1
            # instead of a pass over the whole token buffer)
            dtok = dtok.clone(memory_format=torch.contiguous_format)
        dtok[:, :, :, Z:Z + 1] += dslot
        return dtok, None


class WindowAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, qkv_bias, table, B, X, Y, S, heads, shift):
        ops = get_ops()
        qkv = qkv.contiguous()
        out = ops.window_attention(qkv, qkv_bias.detach(), table.detach(), B, X, Y, S, heads, shift)
        ctx.save_for_backward(qkv, qkv_bias, table, out)
        ctx.geom = (B, X, Y, S, heads, shift)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qkv_bias, table, out = ctx.saved_tensors
        dqkv, dpad, dtab = get_ops().window_attention_backward(qkv, qkv_bias.detach(), table.detach(), out,
                                                               dout.contiguous(), *ctx.geom)
        return dqkv, dpad, dtab, None, None, None, None, None, None


class DualpathCombine(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, bev, weight, bias, identity):
        """tok [B,X,Y,Z+1,C], bev [B,X,Y,C], weight = combine_coeff.weight [1,C,1,1,1], identity [B,X,Y,Z,C]"""
        w = weight.detach().reshape(-1)
        ctx.save_for_backward(tok, bev, weight, bias)
        return get_ops().dualpath_combine(tok, bev, w, None if bias is None else bias.detach(), identity.contiguous())

    @staticmethod
    def backward(ctx, dout):
        tok, bev, weight, bias = ctx.saved_tensors
        dtok, dbev, dw, db = get_ops().dualpath_combine_backward(tok, bev, weight.detach().reshape(-1),
                                                                 None if bias is None else bias.detach(),
                                                                 dout.contiguous())
        dtok._occf_owned = True                  # fresh, exclusively this graph's (see TokenBevSlot.backward)
        return dtok, dbev, dw.view(weight.shape), (db if bias is not None else None), dout


class MSDA(torch.autograd.Function):
    """sampling core; ``ol`` = the fused offset / logit projection output [B, Nq, n_off + n_logits]"""

    @staticmethod
    def forward(ctx, value, ol, n_off, shapes, heads, points, head_major=False):
        """value [B, Nq, E], or head-major [B, heads, Nq, E / heads] (gathers of one head are contiguous rows)"""
        ops = get_ops()
        value, ol = value.contiguous(), ol.contiguous()
        ctx.save_for_backward(value, ol)
        ctx.cfg = (n_off, tuple(shapes), heads, points, bool(head_major))
        return ops.msda3d(value, ol[..., :n_off], ol[..., n_off:], shapes, heads, points, head_major=bool(head_major))

    @staticmethod
    def backward(ctx, dout):
        value, ol = ctx.saved_tensors
        n_off, shapes, heads, points, hm = ctx.cfg
        d_ol = torch.empty_like(ol)
        dvalue, _, _ = get_ops().msda3d_backward(value, ol[..., :n_off], ol[..., n_off:], dout.contiguous(), shapes,
                                                 heads, points, head_major=hm, d_ol=d_ol, n_off=n_off)
        if hm:                                        # token-major [B, Nq, E] seen in the input's layout
            B, Nq, E = dvalue.shape
            dvalue = dvalue.view(B, Nq, heads, E // heads).permute(0, 2, 1, 3)
        return dvalue, d_ol, None, None, None, None, None


class UpsampleAdd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, coarse, lateral):
        ctx.cshape = tuple(coarse.shape)
        return get_ops().upsample_add(coarse.contiguous(), lateral.contiguous())

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        return get_ops().upsample_add_backward(dout, ctx.cshape), dout


class MaskedAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, heads, blocked, row_open):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        out = get_ops().masked_attention(q, k, v, heads, blocked, row_open)
        ctx.save_for_backward(q, k, v, out)
        ctx.mask = (blocked, row_open)
        ctx.heads = heads
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out = ctx.saved_tensors
        dq, dk, dv = get_ops().masked_attention_backward(q, k, v, ctx.heads, out, dout.contiguous(), *ctx.mask)
        return dq, dk, dv, None, None, None


class PointSample3d(torch.autograd.Function):
    """point_sample_3d with a gradient w.r.t. the volume (the points come from no_grad sampling)"""

    @staticmethod
    def forward(ctx, vol, pts, align_corners, padding_mode):
        pts = pts.contiguous()
        ctx.save_for_backward(pts)
        ctx.cfg = (tuple(vol.shape), align_corners, padding_mode)
        return get_ops().point_sample_3d(vol.contiguous(), pts, align_corners, padding_mode)

    @staticmethod
    def backward(ctx, dout):
        (pts,) = ctx.saved_tensors
        shape, align, mode = ctx.cfg
        return get_ops().point_sample_3d_backward(dout.contiguous(), pts, shape, align, mode), None, None, None


class PointLossRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        logits, targets = logits.contiguous(), targets.contiguous()
        ctx.save_for_backward(logits, targets)
        return get_ops().point_loss_rows(logits, targets)

    @staticmethod
    def backward(ctx, grows):
        logits, targets = ctx.saved_tensors
        return get_ops().point_loss_rows_backward(logits, targets, grows.contiguous()), None


class DeformConv(torch.autograd.Function):
    """DCNv1 / DCNv2: bilinear im2col (csrc/dcn.hip) + grouped contraction; x_cl [BN, H, W, C], offset
    [BN, dg*2*K*K, Ho, Wo], weight [Cout, Cin/groups, K, K] (, mask [BN, dg*K*K, Ho, Wo]: the sigmoid-ed modulation of
    DCNv2) -> [BN*Ho*Wo, Cout]"""

    @staticmethod
    def forward(ctx, x_cl, offset, weight, K, pad, groups, dgroups, mask=None, stride=1):
        ops = get_ops()
        x_cl, offset = x_cl.contiguous(), offset.contiguous()
        mask = None if mask is None else mask.contiguous()
        col = ops.deform_im2col(x_cl, offset, K, stride, pad, 1, groups, dgroups, mask=mask)
        Cout = weight.shape[0]
        out = torch.empty((col.shape[0], Cout), dtype=x_cl.dtype, device=x_cl.device)
        og = Cout // groups
        for g, wg in enumerate(weight.detach().chunk(groups, 0)):
            w2 = wg.permute(0, 2, 3, 1).reshape(og, -1).contiguous()
            ops.linear(col[:, g].flatten(1), w2, out=out[:, g * og:(g + 1) * og], w_split=_split(w2))
        ctx.save_for_backward(x_cl, offset, weight, col, mask)
        ctx.cfg = (K, pad, groups, dgroups, stride)
        return out

    @staticmethod
    def backward(ctx, dout):
        x_cl, offset, weight, col, mask = ctx.saved_tensors
        K, pad, groups, dgroups, stride = ctx.cfg
        ops = get_ops()
        dout = dout.contiguous()
        Cout = weight.shape[0]
        og, cpg = Cout // groups, weight.shape[1]
        dcol = torch.empty_like(col)
        dws = []
        for g, wg in enumerate(weight.detach().chunk(groups, 0)):
            w2 = wg.permute(0, 2, 3, 1).reshape(og, -1).contiguous()              # [og, K*K*cpg]
            wt = w2.t().contiguous()
            dy_g = dout[:, g * og:(g + 1) * og]
            ops.linear(dy_g, wt, None, out=dcol[:, g].flatten(1), w_split=_split(wt))
            dw2, _ = ops.linear_wgrad(dy_g, col[:, g].flatten(1), want_bias=False)
            dws.append(dw2.view(og, K, K, cpg).permute(0, 3, 1, 2))
        dw = torch.cat(dws, 0).contiguous()
        if mask is not None:
            dx, doff, dmask = ops.deform_col2im(x_cl, offset, dcol, K, stride, pad, 1, groups, dgroups, mask=mask)
            return dx, doff, dw, None, None, None, None, dmask, None
        dx, doff = ops.deform_col2im(x_cl, offset, dcol, K, stride, pad, 1, groups, dgroups)
        return dx, doff, dw, None, None, None, None, None, None


class SampledMaskLogitsJoint(torch.autograd.Function):
    """``SampledMaskLogits`` for ALL prediction sets of one image at once (the ten sets share the mask features).
    One prediction set at a time, the backward wrote a dense [V, E] mask-feature gradient per set (491 MB at the
    200-grid), autograd summed the ten of them (nine more read-modify-write passes) and every set re-read the mask
    features for its mask_embed gradient: ~10 ms per training step.  Joined, the matched rows of all sets are columns
    of ONE voxel-major [V, cols] scatter buffer and the two contractions run once.

    apply(feat_tok [V, E], align_corners, padding_mode, k, vol_rows_0..k-1 [n_i, X, Y, Z] (detached logits),
          embed_rows_0..k-1 [n_i, E], pts_0..k-1 [n_i, P, 3]) -> k tensors [n_i, P]"""

    @staticmethod
    def forward(ctx, feat_tok, align_corners, padding_mode, k, *args):
        vols, embeds = args[:k], args[k:2 * k]
        pts = [p.contiguous() for p in args[2 * k:3 * k]]
        ctx.save_for_backward(feat_tok, *embeds, *pts)
        ctx.cfg = ([tuple(v.shape) for v in vols], align_corners, padding_mode, k)
        ops = get_ops()
        return tuple(ops.point_sample_3d(v.unsqueeze(1).contiguous(), p, align_corners, padding_mode)[:, 0]
                     for v, p in zip(vols, pts))

    @staticmethod
    def backward(ctx, *douts):
        shapes, align, mode, k = ctx.cfg
        saved = ctx.saved_tensors
        feat_tok, embeds, pts = saved[0], saved[1:1 + k], saved[1 + k:1 + 2 * k]
        ops = get_ops()
        cols, c = [], 0
        for shp in shapes:
            cols.append(c)
            c += (shp[0] + 3) // 4 * 4
        total = max(32, (c + 31) // 32 * 32)
        X, Y, Z = shapes[0][1:]
        dmt = torch.zeros((X * Y * Z, total), dtype=feat_tok.dtype, device=feat_tok.device)
        for i in range(k):
            if douts[i] is not None:
                ops.point_sample_3d_backward(douts[i].contiguous().unsqueeze(1), pts[i], (shapes[i][0], 1, X, Y, Z), align,
                                             mode, voxel_major_cols=total, out=dmt, col0=cols[i])
        d_feat, d_embeds = None, [None] * k
        if any(ctx.needs_input_grad[4 + k:4 + 2 * k]):
            d_all = ops.linear_wgrad(dmt, feat_tok.contiguous(), want_bias=False)[0]                 # [total, E]
            d_embeds = [d_all[cols[i]:cols[i] + shapes[i][0]].contiguous() for i in range(k)]
        if ctx.needs_input_grad[0]:
            et = torch.zeros((feat_tok.shape[1], total), dtype=feat_tok.dtype, device=feat_tok.device)
            for i in range(k):
                et[:, cols[i]:cols[i] + shapes[i][0]] = embeds[i].detach().t()
            # (with the split the contraction runs on the bf16x3 streaming kernel; without it ops.linear took the exact
            # fp32 MFMA kernel: 0.87 ms at 64 TF for [640 000, 224] x [224, 192], r06c)
            d_feat = ops.linear(dmt, et, None, w_split=_split(et), allow_small=False)                # [V, E]
        return (d_feat, None, None, None) + (None,) * k + tuple(d_embeds) + (None,) * k


class SampledMaskLogits(torch.autograd.Function):
    """point_sample_3d(mask_pred[rows], coords) as a function of (mask_embed rows, mask features): the mask logits
    are einsum('qc,cxyz->qxyz') (mask2former_nusc_occ.py:455), and trilinear sampling is linear in the volume, so the
    loss gradient reaches ``mask_embed`` and the mask features through the SAMPLED rows only.  The forward reads the
    (detached, already materialised) logits of the matched rows; the backward scatters the [rows, P] gradient into a
    voxel-major [V, rows] buffer and contracts it twice -- the dense [Q, X, Y, Z] gradient of the reference (256 MB per
    prediction set at the 200-grid) never exists.

    vol_rows [n, X, Y, Z] detached logits of the matched rows; embed_rows [n, E]; feat_tok [V, E]; pts [n, P, 3]"""

    @staticmethod
    def forward(ctx, vol_rows, embed_rows, feat_tok, pts, align_corners, padding_mode):
        pts = pts.contiguous()
        ctx.save_for_backward(embed_rows, feat_tok, pts)
        ctx.cfg = (tuple(vol_rows.shape), align_corners, padding_mode)
        return get_ops().point_sample_3d(vol_rows.unsqueeze(1).contiguous(), pts, align_corners, padding_mode)[:, 0]

    @staticmethod
    def backward(ctx, dout):
        embed_rows, feat_tok, pts = ctx.saved_tensors
        (n, X, Y, Z), align, mode = ctx.cfg
        ops = get_ops()
        npad = (n + 3) // 4 * 4
        # dMP^T [V, npad]: the matched rows' logit gradient, voxel-major
        dmt = ops.point_sample_3d_backward(dout.contiguous().unsqueeze(1), pts, (n, 1, X, Y, Z), align, mode,
                                           voxel_major_cols=npad)
        d_embed = d_feat = None
        if ctx.needs_input_grad[1]:
            d_embed = ops.linear_wgrad(dmt, feat_tok.contiguous(), want_bias=False)[0][:n].contiguous()   # [n, E]
        if ctx.needs_input_grad[2]:
            et = torch.zeros((embed_rows.shape[1], npad), dtype=embed_rows.dtype, device=embed_rows.device)
            et[:, :n] = embed_rows.detach().t()
            d_feat = ops.linear(dmt, et, None, w_split=_split(et), allow_small=False)                      # [V, E]
        return None, d_embed, d_feat, None, None, None
