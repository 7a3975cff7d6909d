"""Dual-path 3-D encoder, registry name ``OccupancyEncoder``.

Host-side mirror of projects/mmdet3d_plugin/occformer/backbones/{occnet.py,
dualpath_block.py, modules/window_attention.py, modules/aspp.py}: same constructor keys,
same output list, same state-dict names.

Layout: voxel tensors are logical [B, C, X, Y, Z] over channels-last memory
[B, X, Y, Z, C].  A block builds ONE token buffer [B, X, Y, Z+1, C] (the Z height slices
plus the BEV mean slice) and runs the shared SwinBlock on it in place of the reference's
rearrange / cat / permute().contiguous() round trips; windows, padding and the cyclic shift
are index arithmetic inside the window-attention kernel (csrc/window_attn.hip).
"""
import torch
import torch.utils.checkpoint
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as A
from . import fused, noise
from .ops import get_ops
from .registry import BACKBONES


def build_norm(cfg, channels):
    """mmcv build_norm_layer for the norm types the OccFormer configs use."""
    cfg = dict(cfg)
    t = cfg.pop("type")
    requires_grad = cfg.pop("requires_grad", True)
    if t == "GN":
        layer = nn.GroupNorm(cfg["num_groups"], channels, eps=cfg.get("eps", 1e-5))
    elif t == "LN":
        layer = nn.LayerNorm(channels, eps=cfg.get("eps", 1e-5))
    elif t == "BN3d":
        layer = nn.BatchNorm3d(channels, eps=cfg.get("eps", 1e-5))
    elif t in ("BN", "BN2d"):
        layer = nn.BatchNorm2d(channels, eps=cfg.get("eps", 1e-5))
    else:
        raise KeyError(f"norm type {t}")
    for p in layer.parameters():
        p.requires_grad = requires_grad
    return layer


# ------------------------------------------------------------------ SwinBlock on the token buffer
class _WindowMSA(nn.Module):
    def __init__(self, dims, heads, ws=7):
        super().__init__()
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * ws - 1) ** 2, heads))
        t = torch.arange(ws * ws)
        r, c = t // ws, t % ws
        idx = (2 * ws - 1) * (r[:, None] - r[None, :] + ws - 1) + (c[:, None] - c[None, :] + ws - 1)
        self.register_buffer("relative_position_index", idx)    # kept for checkpoint compatibility
        self.qkv = nn.Linear(dims, dims * 3)
        self.proj = nn.Linear(dims, dims)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=0.02)


class _ShiftWindowMSA(nn.Module):
    def __init__(self, dims, heads, ws, shift):
        super().__init__()
        self.w_msa = _WindowMSA(dims, heads, ws)
        self.shift_size = shift


class _FFN(nn.Module):
    """mmcv FFN parameter layout: layers.0.0 = fc1, layers.1 = fc2."""

    def __init__(self, dims, hidden, act="gelu"):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dims, hidden)), nn.Linear(hidden, dims))
        self.act = act

    def forward(self, x):
        h = self.layers[0][0](x)
        h = F.gelu(h) if self.act == "gelu" else F.relu(h)
        return self.layers[1](h)


class SwinBlock(nn.Module):
    """window_attention.py:346-372 on tokens [B, X, Y, S, C]."""

    def __init__(self, embed_dims, num_heads, feedforward_channels, window_size=7, shift=False,
                 drop_path_rate=0.2):
        super().__init__()
        assert window_size == 7 and embed_dims == num_heads * 32
        self.heads = num_heads
        self.norm1 = nn.LayerNorm(embed_dims)
        self.attn = _ShiftWindowMSA(embed_dims, num_heads, window_size, window_size // 2 if shift else 0)
        self.norm2 = nn.LayerNorm(embed_dims)
        self.ffn = _FFN(embed_dims, feedforward_channels)
        self.drop_path_rate = drop_path_rate

    def forward(self, tok):
        """tok [B, X, Y, S, C] contiguous -> same shape.  LN / qkv / proj+residual / FFN(+GELU,
        +residual) all run in the HIP kernels; DropPath is the identity in eval mode."""
        if self.training:
            return self._forward_train(tok)
        B, X, Y, S, C = tok.shape
        t = tok.reshape(-1, C)
        m = self.attn.w_msa
        ops = get_ops()
        y = None
        if m.qkv.bias is not None and m.proj.bias is not None:
            y = ops.swin_attention_fused(t, self.norm1.weight.detach(), self.norm1.bias.detach(), self.norm1.eps,
                                         fused.split_weight(m.qkv.weight), m.qkv.bias.detach(),
                                         m.relative_position_bias_table.detach(), fused.split_weight(m.proj.weight),
                                         m.proj.bias.detach(), B, X, Y, S, self.heads, self.attn.shift_size)
        if y is not None:
            t = fused.mlp(y, self.ffn.layers[0][0], self.ffn.layers[1], act=2, ln=self.norm2, ln_mode=1)
            return t.view(B, X, Y, S, C)
        qkv = fused.linear(fused.layernorm(t, self.norm1), m.qkv)
        a = get_ops().window_attention(qkv, m.qkv.bias.detach(), m.relative_position_bias_table.detach(),
                                       B, X, Y, S, self.heads, self.attn.shift_size)
        t = fused.linear(a, m.proj, residual=t)
        t = fused.mlp(t, self.ffn.layers[0][0], self.ffn.layers[1], act=2, ln=self.norm2, ln_mode=1)
        return t.view(B, X, Y, S, C)


    def _drop_path(self, B, S, device):
        """one DropPath draw of the block: the reference's batch is cat(B BEV maps, (b z) height slices)
        (dualpath_block.py:72-76), i.e. noise index b -> slice (b, S-1) and B + b*Z + z -> slice (b, z); returned in
        token-buffer order [b*S + s]"""
        sc = noise.drop_path_scale(B * S, self.drop_path_rate, device)
        if sc is None:
            return None
        Z = S - 1
        return torch.cat((sc[B:].view(B, Z), sc[:B].view(B, 1)), 1).reshape(-1).contiguous()

    def _forward_train(self, tok):
        """the same block as a differentiable graph of the library's forward / backward kernel pairs
        (occformer_amd/autograd.py), with the DropPath of the training mode (window_attention.py:311,332)"""
        B, X, Y, S, C = tok.shape
        t = tok.reshape(-1, C)
        m = self.attn.w_msa
        t, n1 = A.layernorm_fork(t, self.norm1)           # (t continues as the residual operand)
        qkv = A.linear(n1, m.qkv)
        a = A.WindowAttention.apply(qkv, m.qkv.bias, m.relative_position_bias_table, B, X, Y, S, self.heads,
                                    self.attn.shift_size)
        lin1, lin2 = self.ffn.layers[0][0], self.ffn.layers[1]
        H = lin1.weight.shape[0]
        # the large stages run both DropPath branches on the streaming kernel's epilogues (autograd.ProjDropPath /
        # SwinFfn: no separate DropPath / GELU / GELU' passes); the small ones keep the node-per-op graph
        fusable = m.proj.bias is not None and lin1.bias is not None and lin2.bias is not None and \
            A.stream_fusable(t.shape[0], (C, C), (H, C), (C, H))
        sc = self._drop_path(B, S, tok.device)
        if fusable:
            t = A.ProjDropPath.apply(t, a, m.proj.weight, m.proj.bias, sc, X * Y, S)
        elif sc is None:
            t = A.linear(a, m.proj, residual=t)
        else:
            t = A.DropPathAdd.apply(t, A.linear(a, m.proj), sc, X * Y, S)
        t, n2 = A.layernorm_fork(t, self.norm2)
        sc = self._drop_path(B, S, tok.device)
        if fusable:
            t = A.SwinFfn.apply(t, n2, lin1.weight, lin1.bias, lin2.weight, lin2.bias, sc, X * Y, S)
        else:
            f = A.Act.apply(A.linear(n2, lin1), 2)
            if sc is None:
                t = A.linear(f, lin2, residual=t)
            else:
                t = A.DropPathAdd.apply(t, A.linear(f, lin2), sc, X * Y, S)
        return t.view(B, X, Y, S, C)


# ------------------------------------------------------------------ BEV ASPP (global path)
class _AtrousGN(nn.Module):
    def __init__(self, cin, cout, k, dil, groups):
        super().__init__()
        self.atrous_conv = nn.Conv2d(cin, cout, k, padding=0 if k == 1 else dil, dilation=dil, bias=False)
        self.bn = nn.GroupNorm(groups, cout)      # named 'bn' in the reference, GroupNorm in practice
        nn.init.kaiming_normal_(self.atrous_conv.weight)

    def forward(self, x_cl):
        if self.training:
            return A.conv_gn(x_cl, self.atrous_conv, self.bn, relu=True, gate_class="bev")
        return fused.conv_gn(x_cl, self.atrous_conv, self.bn, relu=True)


class _ASPP(nn.Module):
    """aspp.py:49-122 on channels-last BEV maps [B, X, Y, 1, C]."""

    def __init__(self, ch, groups, dilations, dropout):
        super().__init__()
        self.aspp1 = _AtrousGN(ch, ch, 1, dilations[0], groups)
        self.aspp2 = _AtrousGN(ch, ch, 3, dilations[1], groups)
        self.aspp3 = _AtrousGN(ch, ch, 3, dilations[2], groups)
        self.aspp4 = _AtrousGN(ch, ch, 3, dilations[3], groups)
        self.global_avg_pool = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Conv2d(ch, ch, 1, bias=False),
                                             nn.GroupNorm(groups, ch), nn.ReLU(inplace=True))
        self.conv1 = nn.Conv2d(ch * 5, ch, 1, bias=False)
        self.bn1 = nn.GroupNorm(groups, ch)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x_cl):
        B, X, Y, _, C = x_cl.shape
        # image-level branch: mean over (X, Y) -> 1x1 conv -> GN -> ReLU -> broadcast
        # (bilinear upsampling of a 1x1 map with align_corners=True is a broadcast)
        branches = (self.aspp1(x_cl), self.aspp2(x_cl), self.aspp3(x_cl), self.aspp4(x_cl))   # (the reference's order)
        g = x_cl.mean((1, 2), keepdim=True)
        # (noise.relu_gate: the comparison tap -- C units that scale the whole map; a no-op unless a comparison records)
        g = noise.relu_gate(F.relu(self.global_avg_pool[2](self.global_avg_pool[1](g.reshape(B, C, 1, 1)))))
        g = g.view(B, 1, 1, 1, C).expand(B, X, Y, 1, C)
        y = torch.cat((*branches, g), -1)
        if self.training:
            y = A.conv_gn(y, self.conv1, self.bn1, relu=True, gate_class="bev")
            mask = noise.dropout_mask((B, C, X, Y), self.dropout.p, x_cl.device)     # aspp.py:103,122
            if mask is not None:
                y = y * mask.permute(0, 2, 3, 1).unsqueeze(3)
            return x_cl + y
        y = fused.conv_gn(y, self.conv1, self.bn1, relu=True, residual=x_cl)
        return y                                   # = x + dropout(relu(gn(conv1(cat))))  (eval)


class BottleNeckASPP(nn.Module):
    """aspp.py:134-172."""

    def __init__(self, inplanes, reduction=4, dilations=(1, 6, 12, 18), norm_cfg=None, dropout=0.1):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="GN", num_groups=32, requires_grad=True)
        assert norm_cfg["type"] == "GN"
        ch = inplanes // reduction
        groups = norm_cfg["num_groups"]
        self.input_conv = nn.Sequential(nn.Conv2d(inplanes, ch, 1, bias=False), nn.GroupNorm(groups, ch),
                                        nn.ReLU(inplace=True))
        self.aspp = _ASPP(ch, ch // 2 if ch <= groups else groups, list(dilations), dropout)
        self.output_conv = nn.Sequential(nn.Conv2d(ch, inplanes, 1, bias=False),
                                         nn.GroupNorm(groups, inplanes), nn.ReLU(inplace=True))

    def forward(self, x_cl):
        """x_cl [B, X, Y, 1, C] (any row stride) -> contiguous [B, X, Y, 1, C]."""
        if self.training:
            # (gate_class "bev": the comparison tap's class of the BEV ASPP's maps -- a no-op unless a comparison records)
            y = A.conv_gn(x_cl, self.input_conv[0], self.input_conv[1], relu=True, gate_class="bev")
            y = self.aspp(y)
            return A.conv_gn(y, self.output_conv[0], self.output_conv[1], relu=True, residual=x_cl, gate_class="bev")
        y = fused.conv_gn(x_cl, self.input_conv[0], self.input_conv[1], relu=True)
        y = self.aspp(y)
        return fused.conv_gn(y, self.output_conv[0], self.output_conv[1], relu=True, residual=x_cl.contiguous())


# ------------------------------------------------------------------ the block and the encoder
class DualpathTransformerBlock(nn.Module):
    """dualpath_block.py:13-82."""

    def __init__(self, in_channels, channels, stride=1, norm_cfg=None, coeff_bias=True, aspp_drop=0.1,
                 layer_index=0, **kwargs):
        super().__init__()
        self.stride = stride
        self.channels = channels
        self.shift = layer_index % 2 == 1
        if stride > 1:
            self.downsample = nn.Sequential(nn.Conv3d(in_channels, channels, 1, stride=stride, bias=False),
                                            build_norm(norm_cfg, channels))
        else:
            self.downsample = nn.Identity()
        self.input_conv = nn.Sequential(nn.Conv3d(in_channels, channels, 3, padding=1, stride=stride, bias=False),
                                        build_norm(norm_cfg, channels), nn.ReLU(inplace=True))
        self.bev_encoder = SwinBlock(channels, channels // 32, channels, window_size=7, drop_path_rate=0.2,
                                     shift=self.shift)
        self.aspp = BottleNeckASPP(channels, norm_cfg=norm_cfg, dropout=aspp_drop)
        self.combine_coeff = nn.Conv3d(channels, 1, 1, bias=coeff_bias)

    def forward(self, x):
        """x logical [B, Cin, X, Y, Z] -> logical [B, C, X', Y', Z'] over channels-last memory."""
        if not isinstance(self.input_conv[1], nn.GroupNorm):
            raise NotImplementedError("the HIP path implements the GroupNorm blocks of the OccFormer configs")
        if self.stride == 1 and x.shape[1] != self.channels:
            # dualpath_block.py:36-42,82: the skip is nn.Identity() unless stride > 1 -- the reference's final addition
            # fails on these shapes as well
            raise RuntimeError(f"DualpathTransformerBlock: identity skip needs in_channels == channels at stride 1, got "
                               f"{x.shape[1]} -> {self.channels}")
        ops = get_ops()
        x_cl = fused.channels_last_view(x.float())
        if self.training:
            if self.stride > 1:
                tok = A.conv_gn(x_cl, self.input_conv[0], self.input_conv[1], relu=True, tokens=True)
                ident = A.conv_gn(x_cl, self.downsample[0], self.downsample[1])
            else:       # x_cl is also the identity operand: its two gradients meet inside the convolution's backward
                ident, y, st = A.conv_fork(x_cl, self.input_conv[0], self.input_conv[1])
                tok = A.group_norm(y, self.input_conv[1], relu=True, tokens=True, stats=st)
            Z = tok.shape[3] - 1
            tok, slot = A.TokenBevSlot.apply(self.bev_encoder(tok), Z)
            bev = self.aspp(slot)
            out = A.DualpathCombine.apply(tok, bev.reshape(*bev.shape[:3], -1), self.combine_coeff.weight,
                                          self.combine_coeff.bias, ident)
            return out.permute(0, 4, 1, 2, 3)
        # 3^3 conv (its epilogue also yields the GroupNorm statistics) -> GN + ReLU -> token buffer [B,X,Y,Z+1,C]
        tok = fused.conv_gn(x_cl, self.input_conv[0], self.input_conv[1], relu=True, tokens=True)
        Z = tok.shape[3] - 1
        tok = self.bev_encoder(tok)
        bev = self.aspp(tok[:, :, :, Z:Z + 1])                                      # [B,X,Y,1,C]
        if self.stride > 1:
            ident = fused.conv_gn(x_cl, self.downsample[0], self.downsample[1])
        else:
            ident = x_cl
        cw = self.combine_coeff
        out = ops.dualpath_combine(tok, bev.reshape(*bev.shape[:3], -1), cw.weight.detach().reshape(-1),
                                   None if cw.bias is None else cw.bias.detach(), ident)
        return out.permute(0, 4, 1, 2, 3)


@BACKBONES.register_module()
class OccupancyEncoder(nn.Module):
    """occnet.py:12-75."""

    def __init__(self, in_channels, num_stage=4, block_numbers=(2, 2, 2, 2),
                 block_inplanes=(64, 128, 256, 512), block_strides=(1, 2, 2, 2), out_indices=(0, 1, 2, 3),
                 norm_cfg=None, with_cp=True, **kwargs):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="BN3d", requires_grad=True)
        self.out_indices = tuple(out_indices)
        self.with_cp = with_cp
        self.layers = nn.ModuleList()
        index = 0
        for i in range(num_stage):
            blocks = []
            for j in range(block_numbers[i]):
                blocks.append(DualpathTransformerBlock(
                    in_channels, block_inplanes[i], stride=block_strides[i] if j == 0 else 1,
                    norm_cfg=norm_cfg, layer_index=index, **kwargs))
                in_channels = block_inplanes[i]
                index += 1
            self.layers.append(nn.Sequential(*blocks))

    def forward(self, x):
        outs = []
        for i, layer in enumerate(self.layers):
            # with_cp (occnet.py:67-68) exists to fit 24 GB cards; with 288 GB of HBM every activation stays resident
            x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return outs
