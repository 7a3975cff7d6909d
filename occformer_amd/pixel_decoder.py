"""3-D multi-scale deformable-attention pixel decoder, registry names
``MSDeformAttnPixelDecoder3D``, ``MultiScaleDeformableAttention3D``,
``SinePositionalEncoding3D``.

Host-side mirror of projects/mmdet3d_plugin/occformer/necks/{multiscale_deformattn_3d.py,
multi_scale_deform_attn_3d.py} and mask2former/positional_encodings/positional_encoding.py,
with the mmcv transformer bricks they sit on (BaseTransformerLayer / DetrTransformerEncoder /
FFN, mmcv-full 1.4.0) folded into plain modules that keep the reference's state-dict names.
Tokens are batch-first [B, Nq, E]; the sampling core is csrc/msda3d.hip.
"""
import math

import torch
import torch.nn as nn

from . import autograd as A
from . import fused
from .encoder import _FFN
from .ops import get_ops
from .registry import ATTENTION, NECKS, POSITIONAL_ENCODING


@POSITIONAL_ENCODING.register_module()
class SinePositionalEncoding3D(nn.Module):
    """positional_encoding.py:11-108.  ``forward(mask)`` keeps the reference signature
    ([B, X, Y, Z] bool, True = padded); ``for_shape`` is the all-valid fast path (the only
    case the OccFormer forward uses) and is cached per shape: it is input independent."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6,
                 offset=0.0, init_cfg=None):
        super().__init__()
        self.num_feats = int(num_feats)
        self.temperature = temperature
        self.normalize = normalize
        self.scale = scale
        self.eps = eps
        self.offset = offset
        self._cache = {}

    def forward(self, mask, stride=None):
        not_mask = 1 - mask.to(torch.int)
        embeds = [not_mask.cumsum(d, dtype=torch.float32) for d in (1, 2, 3)]
        if self.normalize:
            embeds = [(e + self.offset) / (e.narrow(d, e.shape[d] - 1, 1) + self.eps) * self.scale
                      for e, d in zip(embeds, (1, 2, 3))]
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / self.num_feats)
        parts = []
        for e in embeds:
            p = e[..., None] / dim_t
            parts.append(torch.stack((p[..., 0::2].sin(), p[..., 1::2].cos()), dim=5).flatten(4))
        return torch.cat(parts, dim=4).permute(0, 4, 1, 2, 3)

    def for_shape(self, shape, device):
        """[X*Y*Z, 3*num_feats] tokens (channels-last) for an all-valid volume."""
        key = (tuple(shape), str(device))
        if key not in self._cache:
            mask = torch.zeros((1, *shape), dtype=torch.bool, device=device)
            self._cache[key] = self.forward(mask)[0].flatten(1).t().contiguous()
        return self._cache[key]


@ATTENTION.register_module()
class MultiScaleDeformableAttention3D(nn.Module):
    """multi_scale_deform_attn_3d.py:83-286 (batch-first tokens here)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64,
                 dropout=0.1, batch_first=False, norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError("embed_dims must be divisible by num_heads")
        self.embed_dims, self.num_heads = embed_dims, num_heads
        self.num_levels, self.num_points = num_levels, num_points
        self.dropout = nn.Dropout(dropout)
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 3)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        nn.init.zeros_(self.sampling_offsets.weight)
        th = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        g = torch.stack((th.cos(), th.sin(), (th.sin() + th.cos()) / 2), -1)
        g = (g / g.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 3)
        g = g.repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            g[:, :, i, :] *= i + 1
        self.sampling_offsets.bias.data = g.reshape(-1)
        nn.init.zeros_(self.attention_weights.weight)
        nn.init.zeros_(self.attention_weights.bias)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.zeros_(self.value_proj.bias)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.zeros_(self.output_proj.bias)

    def _offset_logit_weights(self):
        """sampling_offsets and attention_weights share their input: one fused GEMM, N = H*L*P*4."""
        a, b = self.sampling_offsets, self.attention_weights
        key = fused.param_version(a.weight, b.weight, a.bias, b.bias) + (get_ops().precision, id(get_ops()))
        if getattr(self, "_fused_key", None) != key:
            self._fused_w = torch.cat((a.weight.detach(), b.weight.detach()), 0).contiguous()
            self._fused_b = torch.cat((a.bias.detach(), b.bias.detach()), 0).contiguous()
            self._fused_split = None if get_ops().precision == "f32" else get_ops().split_bf16(self._fused_w)
            self._fused_key = key
        return self._fused_w, self._fused_b

    def forward(self, query, query_pos, level_shapes):
        """query/query_pos [B, Nq, E]; queries are the cells of ``level_shapes`` in order."""
        if self.training:
            return self._forward_train(query, query_pos, level_shapes)
        ops = get_ops()
        B, Nq, E = query.shape
        dh = E // self.num_heads
        w, b = self._offset_logit_weights()
        # (query + pos) W + b = query W + (pos W + b): the positional half only depends on the grid and the
        # weights, so it is kept resident (140 MB per layer at the 200-grid) and enters as the GEMM residual
        pkey = (self._fused_key, query_pos.data_ptr(), query_pos._version, tuple(query_pos.shape))
        if getattr(self, "_pos_key", None) != pkey:
            self._pos_ol = ops.linear(query_pos.contiguous(), w, b, w_split=self._fused_split)
            self._pos_key = pkey
        ol = ops.linear(query, w, None, residual=self._pos_ol, w_split=self._fused_split)
        n_off = self.sampling_offsets.out_features
        if ops.head_major_supported(B * Nq, E, E, dh):
            # the value projection writes the head-major layout the sampler gathers from
            value_hm = fused.linear(query, self.value_proj, head_major=(Nq, dh))
        else:
            value = fused.linear(query, self.value_proj)
            value_hm = value.view(B, Nq, self.num_heads, dh).permute(0, 2, 1, 3).contiguous()
        # offsets and logits stay column blocks of the fused projection output (row stride = its width)
        out = ops.msda3d(value_hm, ol[..., :n_off], ol[..., n_off:], level_shapes, self.num_heads,
                         self.num_points, head_major=True)
        return fused.linear(out, self.output_proj, residual=query)       # dropout = identity (eval)


    def _forward_train(self, query, query_pos, level_shapes):
        """differentiable graph of forward / backward kernel pairs (occformer_amd/autograd.py)"""
        if self.dropout.p > 0:
            raise NotImplementedError("MultiScaleDeformableAttention3D dropout > 0 (every OccFormer config uses 0.0)")
        qp = query + query_pos
        # one projection for offsets | logits (the two weights concatenated per step: 288 x 192, its gradient splits back
        # through the cat) instead of two GEMM pairs, a [B, Nq, 288] concatenation and two data gradients summed
        w = torch.cat((self.sampling_offsets.weight, self.attention_weights.weight), 0)
        b = torch.cat((self.sampling_offsets.bias, self.attention_weights.bias), 0)
        ol = A.Linear.apply(qp, w, b, 0, None, None)
        B, Nq, E = query.shape
        dh = E // self.num_heads
        hm = get_ops().head_major_supported(B * Nq, E, E, dh)
        if hm:      # the value projection writes the head-major layout the sampler (forward and backward) gathers from
            value = A.Linear.apply(query, self.value_proj.weight, self.value_proj.bias, 0, None, None, (Nq, dh))
        else:
            value = A.linear(query, self.value_proj)
        out = A.MSDA.apply(value, ol, self.sampling_offsets.out_features, tuple(level_shapes), self.num_heads,
                           self.num_points, hm)
        return A.linear(out, self.output_proj, residual=query)


class _EncoderLayer(nn.Module):
    """mmcv BaseTransformerLayer with operation_order ('self_attn','norm','ffn','norm')."""

    def __init__(self, attn_cfg, ffn_channels, embed_dims):
        super().__init__()
        cfg = dict(attn_cfg)
        cfg.pop("type", None)
        self.attentions = nn.ModuleList([MultiScaleDeformableAttention3D(**cfg)])
        self.ffns = nn.ModuleList([_FFN(embed_dims, ffn_channels, act="relu")])
        self.norms = nn.ModuleList([nn.LayerNorm(embed_dims), nn.LayerNorm(embed_dims)])

    def forward(self, x, pos, level_shapes):
        ffn = self.ffns[0].layers
        if self.training:
            x = A.layernorm(self.attentions[0](x, pos, level_shapes), self.norms[0])
            y = A.linear(A.linear(x, ffn[0][0], act=1), ffn[1], residual=x)
            return A.layernorm(y, self.norms[1])
        x = fused.layernorm(self.attentions[0](x, pos, level_shapes), self.norms[0])
        return fused.mlp(x, ffn[0][0], ffn[1], act=1, ln=self.norms[1], ln_mode=2)


class _Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        tl = cfg["transformerlayers"]
        assert tuple(tl["operation_order"]) == ("self_attn", "norm", "ffn", "norm")
        attn = tl["attn_cfgs"]
        E = attn["embed_dims"]
        ffn_ch = tl.get("feedforward_channels", tl.get("ffn_cfgs", {}).get("feedforward_channels", 1024))
        self.layers = nn.ModuleList([_EncoderLayer(attn, ffn_ch, E) for _ in range(cfg["num_layers"])])
        self.embed_dims = E


class _ConvModule(nn.Module):
    """mmcv ConvModule (conv -> GN -> optional ReLU) with the sub-module names `.conv`, `.gn`."""

    def __init__(self, cin, cout, k, padding, bias, groups, act):
        super().__init__()
        self.conv = nn.Conv3d(cin, cout, k, padding=padding, bias=bias)
        self.gn = nn.GroupNorm(groups, cout)
        self.act = act

    def forward(self, x_cl):
        """channels-last [B, X, Y, Z, Cin] -> contiguous [B, X, Y, Z, Cout]"""
        if self.training:
            return A.conv_gn(x_cl, self.conv, self.gn, relu=self.act)
        return fused.conv_gn(x_cl, self.conv, self.gn, relu=self.act)


@NECKS.register_module()
class MSDeformAttnPixelDecoder3D(nn.Module):
    """multiscale_deformattn_3d.py:20-249."""

    def __init__(self, in_channels=(256, 512, 1024, 2048), strides=(4, 8, 16, 32), feat_channels=256,
                 out_channels=256, num_outs=3, conv_cfg=None, norm_cfg=None, act_cfg=None, encoder=None,
                 positional_encoding=None, init_cfg=None):
        super().__init__()
        norm_cfg = norm_cfg or dict(type="GN", num_groups=32)
        groups = norm_cfg["num_groups"]
        self.strides = list(strides)
        self.num_input_levels = len(in_channels)
        self.num_encoder_levels = encoder["transformerlayers"]["attn_cfgs"]["num_levels"]
        n_in, n_enc = self.num_input_levels, self.num_encoder_levels
        self.input_convs = nn.ModuleList([
            _ConvModule(in_channels[i], feat_channels, 1, 0, True, groups, False)
            for i in range(n_in - 1, n_in - n_enc - 1, -1)])
        self.encoder = _Encoder(encoder)
        pe = dict(positional_encoding)
        pe.pop("type", None)
        self.postional_encoding = SinePositionalEncoding3D(**pe)        # (sic) reference attribute name
        self.level_encoding = nn.Embedding(n_enc, feat_channels)
        self.lateral_convs = nn.ModuleList()
        self.output_convs = nn.ModuleList()
        for i in range(n_in - n_enc - 1, -1, -1):
            self.lateral_convs.append(_ConvModule(in_channels[i], feat_channels, 1, 0, False, groups, False))
            self.output_convs.append(_ConvModule(feat_channels, feat_channels, 3, 1, False, groups, True))
        self.mask_feature = nn.Conv3d(feat_channels, out_channels, 1)
        self.num_outs = num_outs

    def init_weights(self):
        for m in self.input_convs:
            nn.init.xavier_uniform_(m.conv.weight)
            nn.init.zeros_(m.conv.bias)
        for m in list(self.lateral_convs) + list(self.output_convs) + [self.mask_feature]:
            conv = m.conv if hasattr(m, "conv") else m
            nn.init.kaiming_uniform_(conv.weight, a=1)
            if conv.bias is not None:
                nn.init.zeros_(conv.bias)
        nn.init.normal_(self.level_encoding.weight, 0, 1)
        for p in self.encoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)
        for layer in self.encoder.layers:
            layer.attentions[0].init_weights()

    def forward(self, feats):
        ops = get_ops()
        B = feats[0].shape[0]
        n_in, n_enc = self.num_input_levels, self.num_encoder_levels
        feats_cl = [fused.channels_last_view(f.float()) for f in feats]
        toks, poss, shapes = [], [], []
        for i in range(n_enc):
            f = feats_cl[n_in - 1 - i]
            y = self.input_convs[i](f)                                   # [B, X, Y, Z, E]
            shp = tuple(f.shape[1:4])
            pe = self.postional_encoding.for_shape(shp, f.device) + self.level_encoding.weight[i]
            toks.append(y.reshape(B, -1, y.shape[-1]))
            poss.append(pe.unsqueeze(0).expand(B, -1, -1))
            shapes.append(shp)
        x = torch.cat(toks, 1).contiguous()
        lw = self.level_encoding.weight
        pkey = (tuple(shapes), B, fused.param_version(lw), str(x.device))
        if self.training:
            pos = torch.cat(poss, 1).contiguous()        # carries the level-encoding gradient
        else:
            if getattr(self, "_pos_cache_key", None) != pkey:
                self._pos_cache = torch.cat(poss, 1).contiguous().detach()
                self._pos_cache_key = pkey
            pos = self._pos_cache
        for layer in self.encoder.layers:
            x = layer(x, pos, shapes)
        outs, start = [], 0
        for shp in shapes:
            n = shp[0] * shp[1] * shp[2]
            outs.append(x[:, start:start + n].reshape(B, *shp, -1))      # channels-last [B, X, Y, Z, E]
            start += n
        for j, i in enumerate(range(n_in - n_enc - 1, -1, -1)):
            cur = self.lateral_convs[j](feats_cl[i])
            y = A.UpsampleAdd.apply(outs[-1], cur) if self.training else ops.upsample_add(outs[-1].contiguous(), cur)
            outs.append(self.output_convs[j](y))
        outs[-1] = A.conv(outs[-1], self.mask_feature) if self.training else fused.conv(outs[-1], self.mask_feature)
        # logical [B, E, X, Y, Z] views over channels-last memory, fine -> coarse
        return [o.permute(0, 4, 1, 2, 3) for o in outs[::-1]]
