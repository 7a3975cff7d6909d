"""Mask2Former-style occupancy decoder, registry names ``Mask2FormerNuscOccHead`` and
``Mask2FormerOccHead``.

Host-side mirror of projects/mmdet3d_plugin/occformer/mask2former/{mask2former_nusc_occ.py,
mask2former_occ.py}: same constructor keys, ``forward`` / ``simple_test`` contract and
state-dict names; the mmcv/mmdet bricks underneath
(DetrTransformerDecoder(+Layer), MultiheadAttention -> nn.MultiheadAttention, FFN) are
folded into plain modules.  Tokens are batch-first.  Kernels: csrc/mask_head.hip.
"""
import numpy as np
import os

import torch
import torch.nn as nn

from . import autograd as A
from . import fused
from . import noise
from .encoder import _FFN
from .ops import get_ops
from .pixel_decoder import SinePositionalEncoding3D
from .registry import HEADS
from .training import (KittiTrainingMixin, LazyMask, NuscTrainingMixin, OccHeadTrainingMixin,
                       semantic_kitti_class_frequencies)


class _MHAParams(nn.Module):
    """Parameter holder with torch.nn.MultiheadAttention's names (in_proj_*, out_proj.*)."""

    def __init__(self, E):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = nn.Linear(E, E)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.zeros_(self.out_proj.bias)


class _MHA(nn.Module):
    """mmcv MultiheadAttention: positional terms are added to q and k, never to v; returns
    identity + attention output (dropout 0 in every OccFormer config)."""

    def __init__(self, E, heads):
        super().__init__()
        self.attn = _MHAParams(E)
        self.heads = heads
        self.E = E

    def forward(self, query, key, value, query_pos, key_pos, blocked=None, row_open=None, key_with_pos=None,
                kv=None):
        """``key_with_pos``: ``key + key_pos`` computed by the caller (the level tokens and their encodings
        are the same for every layer that attends to that level); ``kv``: this layer's projected keys / values
        when the caller projected a level's tokens for all of its layers in one GEMM"""
        E = self.E
        ops = get_ops()
        if self.training:
            W, bv = self.attn.in_proj_weight, self.attn.in_proj_bias
            if key_with_pos is None:
                key_with_pos = key + key_pos if key_pos is not None else key
            q, k, v = A.InProj.apply(query + query_pos, key_with_pos, value, W, bv)
            o = A.MaskedAttention.apply(q, k, v, self.heads, blocked, row_open)
            return A.linear(o, self.attn.out_proj, residual=query)
        w, b = self.attn.in_proj_weight.detach(), self.attn.in_proj_bias.detach()
        sp = fused.split_weight(self.attn.in_proj_weight)
        part = (lambda lo, hi: None) if sp is None else (lambda lo, hi: (sp[0][lo:hi], sp[1][lo:hi]))
        q = ops.linear(query + query_pos, w[:E], b[:E], w_split=part(0, E))
        if kv is not None:
            k, v = kv
        else:
            if key_with_pos is None:
                key_with_pos = key + key_pos if key_pos is not None else key
            k = ops.linear(key_with_pos, w[E:2 * E], b[E:2 * E], w_split=part(E, 2 * E))
            v = ops.linear(value, w[2 * E:], b[2 * E:], w_split=part(2 * E, 3 * E))
        o = ops.masked_attention(q, k, v, self.heads, blocked, row_open)
        return fused.linear(o, self.attn.out_proj, residual=query.contiguous())


class _DecoderLayer(nn.Module):
    """DetrTransformerDecoderLayer with operation_order
    ('cross_attn','norm','self_attn','norm','ffn','norm')."""

    def __init__(self, E, heads, ffn_channels):
        super().__init__()
        self.attentions = nn.ModuleList([_MHA(E, heads), _MHA(E, heads)])
        self.ffns = nn.ModuleList([_FFN(E, ffn_channels, act="relu")])
        self.norms = nn.ModuleList([nn.LayerNorm(E) for _ in range(3)])

    def forward(self, q, qpos, key, key_pos, blocked, row_open, key_with_pos=None, kv=None):
        if self.training:
            q = A.layernorm(self.attentions[0](q, key, key, qpos, key_pos, blocked, row_open, key_with_pos),
                            self.norms[0])
            q = A.layernorm(self.attentions[1](q, q, q, qpos, qpos), self.norms[1])
            ffn = self.ffns[0].layers
            y = A.linear(A.linear(q, ffn[0][0], act=1, heavy_gate=True), ffn[1], residual=q)
            return A.layernorm(y, self.norms[2])
        q = fused.layernorm(self.attentions[0](q, key, key, qpos, key_pos, blocked, row_open, key_with_pos, kv),
                            self.norms[0])
        q = fused.layernorm(self.attentions[1](q, q, q, qpos, qpos), self.norms[1])
        ffn = self.ffns[0].layers
        y = fused.linear(fused.linear(q, ffn[0][0], act=1), ffn[1], residual=q)
        return fused.layernorm(y, self.norms[2])


class _Decoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        tl = cfg["transformerlayers"]
        assert tuple(tl["operation_order"]) == ("cross_attn", "norm", "self_attn", "norm", "ffn", "norm")
        E = tl["attn_cfgs"]["embed_dims"]
        heads = tl["attn_cfgs"]["num_heads"]
        ffn = tl.get("feedforward_channels", tl.get("ffn_cfgs", {}).get("feedforward_channels", 2048))
        self.layers = nn.ModuleList([_DecoderLayer(E, heads, ffn) for _ in range(cfg["num_layers"])])
        self.post_norm = nn.LayerNorm(E)
        self.embed_dims = E


class _Mask2FormerOccBase(OccHeadTrainingMixin, nn.Module):
    # OCCF_LAZY_LOGITS (default 1): the training graph contracts the full-resolution mask logits only where the losses
    # read them (attention mask from the fused contraction + pooling kernel, ~20 matched rows per set on demand) instead
    # of one dense [Q, X, Y, Z] volume per prediction set: same gradients, 2.4 GB less memory, 141.0 vs 141.9 ms per
    # step (r04i; in round 2 it had measured even and stayed off).  0 = the dense volumes.
    lazy_train_logits = os.environ.get("OCCF_LAZY_LOGITS", "1") == "1"

    def __init__(self, feat_channels, out_channels, num_occupancy_classes=20, num_queries=100,
                 num_transformer_feat_level=3, enforce_decoder_input_project=False,
                 transformer_decoder=None, positional_encoding=None, pooling_attn_mask=True,
                 point_cloud_range=None, padding_mode="border", sample_weight_gamma=0.25,
                 loss_cls=None, loss_mask=None, loss_dice=None, train_cfg=None, test_cfg=None,
                 init_cfg=None, align_corners=True, **kwargs):
        super().__init__()
        self.num_occupancy_classes = self.num_classes = num_occupancy_classes
        self.num_queries = num_queries
        self.point_cloud_range = point_cloud_range
        self.num_transformer_feat_level = num_transformer_feat_level
        self.num_heads = transformer_decoder["transformerlayers"]["attn_cfgs"]["num_heads"]
        self.num_transformer_decoder_layers = transformer_decoder["num_layers"]
        self.transformer_decoder = _Decoder(transformer_decoder)
        self.decoder_embed_dims = E = self.transformer_decoder.embed_dims
        self.decoder_input_projs = nn.ModuleList()
        for _ in range(num_transformer_feat_level):
            if E != feat_channels or enforce_decoder_input_project:
                self.decoder_input_projs.append(nn.Conv3d(feat_channels, E, kernel_size=1))
            else:
                self.decoder_input_projs.append(nn.Identity())
        pe = dict(positional_encoding)
        pe.pop("type", None)
        self.decoder_positional_encoding = SinePositionalEncoding3D(**pe)
        self.query_embed = nn.Embedding(num_queries, feat_channels)
        self.query_feat = nn.Embedding(num_queries, feat_channels)
        self.level_embed = nn.Embedding(num_transformer_feat_level, feat_channels)
        self.cls_embed = nn.Linear(feat_channels, self.num_classes + 1)
        self.mask_embed = nn.Sequential(nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, feat_channels), nn.ReLU(inplace=True),
                                        nn.Linear(feat_channels, out_channels))
        self.test_cfg = test_cfg
        self.loss_cfgs = dict(cls=loss_cls, mask=loss_mask, dice=loss_dice)
        self.pooling_attn_mask = pooling_attn_mask
        self.align_corners = align_corners
        self.padding_mode = padding_mode
        self.sample_weight_gamma = sample_weight_gamma
        if not pooling_attn_mask:
            raise NotImplementedError("only preserve-pooling attention masks (every OccFormer config)")
        self._init_training(train_cfg, loss_cls, loss_mask, loss_dice)

    def init_weights(self):
        for p in self.transformer_decoder.parameters():
            if p.dim() > 1:
                nn.init.xavier_normal_(p)

    # -- mask2former_nusc_occ.py:426-471
    def forward_head(self, decoder_out, mask_feat_tok, vol_shape, target_shape, mask_feat_split=None,
                     want_mask=True, want_attn=True):
        """decoder_out [B, Q, E]; mask_feat_tok [B, V, E] channels-last tokens.
        Returns cls [B,Q,K+1], mask_pred [B,Q,X,Y,Z] (None if not wanted), (blocked u8 [B,Q,L],
        row_open) (None if not wanted).  When only the attention mask is needed the contraction
        and the preserve-pooling run fused and the full-resolution logits are never written."""
        ops = get_ops()
        if self.training:
            return self._forward_head_train(decoder_out, mask_feat_tok, vol_shape, target_shape, mask_feat_split,
                                            want_attn)
        d = fused.layernorm(decoder_out.contiguous(), self.transformer_decoder.post_norm)
        cls_pred = fused.linear(d, self.cls_embed)
        me = self.mask_embed
        mask_embed = fused.linear(fused.linear(fused.linear(d, me[0], act=1), me[2], act=1), me[4])
        return self._head_from_embed(cls_pred, mask_embed, mask_feat_tok, vol_shape, target_shape, mask_feat_split,
                                     want_mask, want_attn)

    def _head_from_embed(self, cls_pred, mask_embed, mask_feat_tok, vol_shape, target_shape, mask_feat_split, want_mask,
                         want_attn):
        """forward_head from the class logits / mask embeddings on (inference): the mask contraction and the
        preserve-pooling attention mask"""
        ops = get_ops()
        d = mask_embed
        B, Q = mask_embed.shape[:2]
        if not want_mask and want_attn and mask_feat_split is not None and Q <= 128 and ops.use_fused_mask_pool:
            fusedp = ops.mask_gemm_pool(mask_embed, mask_feat_split, vol_shape, target_shape)
            if fusedp is not None:
                return cls_pred, None, (fusedp[1], fusedp[2])
        # einsum('bqc,bcxyz->bqxyz'): per batch a [Q, E] x [V, E]^T GEMM whose "weight" is the
        # channels-last mask feature itself
        mask_pred = torch.empty((B, Q, mask_feat_tok.shape[1]), dtype=d.dtype, device=d.device)
        for b in range(B):
            sp = None if mask_feat_split is None else (mask_feat_split[0][b], mask_feat_split[1][b])
            ops.linear(mask_embed[b], mask_feat_tok[b], out=mask_pred[b], w_split=sp)
        mask_pred = mask_pred.view(B, Q, *vol_shape)
        am = None
        if want_attn:
            _, blocked, row_open = ops.mask_pool(mask_pred.detach(), target_shape)
            am = (blocked, row_open)
        return cls_pred, mask_pred, am

    def _forward_head_train(self, decoder_out, mask_feat_tok, vol_shape, target_shape, mask_feat_split, want_attn):
        """training-step variant: class / mask-embedding branches as differentiable kernel pairs; the mask logits
        are materialised DETACHED (attention mask, target assignment, importance sampling) and handed on as a
        ``LazyMask`` whose gradient route is the point-sampled contraction (training.LazyMask)"""
        ops = get_ops()
        d = A.layernorm(decoder_out, self.transformer_decoder.post_norm)
        cls_pred = A.linear(d, self.cls_embed)
        me = self.mask_embed
        mask_embed = A.linear(A.linear(A.linear(d, me[0], act=1, heavy_gate=True), me[2], act=1, heavy_gate=True), me[4])
        B, Q = mask_embed.shape[:2]
        with torch.no_grad():
            dense, am = None, None
            if want_attn and mask_feat_split is not None and Q <= 128 and ops.use_fused_mask_pool and self.lazy_train_logits:
                # attention mask from the fused contraction + pooling kernel: the full-resolution logits of this set
                # are only contracted where the losses need them (training.LazyMask)
                fusedp = ops.mask_gemm_pool(mask_embed.detach(), mask_feat_split, vol_shape, target_shape)
                if fusedp is not None:
                    am = (fusedp[1], fusedp[2])
            if am is None and want_attn or not self.lazy_train_logits:
                dense = torch.empty((B, Q, mask_feat_tok.shape[1]), dtype=d.dtype, device=d.device)
                for b in range(B):
                    sp = None if mask_feat_split is None else (mask_feat_split[0][b], mask_feat_split[1][b])
                    ops.linear(mask_embed[b].detach(), mask_feat_tok[b].detach(), out=dense[b], w_split=sp,
                               allow_small=False)
                dense = dense.view(B, Q, *vol_shape)
                if want_attn and am is None:
                    _, blocked, row_open = ops.mask_pool(dense, target_shape)
                    am = (blocked, row_open)
        if am is not None:
            noise.tape_mask(am[0])                  # (comparison tap: a no-op unless a comparison records)
        return cls_pred, LazyMask(dense, mask_embed, mask_feat_tok, vol_shape, mask_feat_split), am

    def _project_level_tokens(self, keys, keys_pp):
        """Key / value projections of the cross-attentions (mask2former_nusc_occ.py:657-667 via
        nn.MultiheadAttention's in_proj): layers lv, lv + n_levels, ... attend to the same level tokens, so their
        key (and value) projections are ONE GEMM per level with the layers' weights stacked along N; the
        epilogue writes the stack "head-major", i.e. as one contiguous [L, E] matrix per layer.
        -> per level (K [B, n, L, E], V [B, n, L, E]), or None when the stacked form does not apply."""
        ops = get_ops()
        nl = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        E = self.decoder_embed_dims
        B = keys[0].shape[0]
        if B != 1 or len(layers) % nl or ops.precision == "f32" or os.environ.get("OCCF_STACK_KV", "1") != "1" or \
                self.training:
            return None
        params = [l.attentions[0].attn.in_proj_weight for l in layers] + \
                 [l.attentions[0].attn.in_proj_bias for l in layers]
        ver = fused.param_version(*params) + (ops.precision, id(ops))
        cache = getattr(self, "_kv_stack", None)
        if cache is None or cache[0] != ver:
            per_level = []
            with torch.no_grad():
                for lv in range(nl):
                    ls = layers[lv::nl]
                    wk = torch.cat([l.attentions[0].attn.in_proj_weight.detach()[E:2 * E] for l in ls]).contiguous()
                    wv = torch.cat([l.attentions[0].attn.in_proj_weight.detach()[2 * E:] for l in ls]).contiguous()
                    bk = torch.cat([l.attentions[0].attn.in_proj_bias.detach()[E:2 * E] for l in ls]).contiguous()
                    bv = torch.cat([l.attentions[0].attn.in_proj_bias.detach()[2 * E:] for l in ls]).contiguous()
                    per_level.append((wk, bk, ops.split_bf16(wk), wv, bv, ops.split_bf16(wv)))
            cache = (ver, per_level)
            self._kv_stack = cache
        out = []
        for lv in range(nl):
            wk, bk, spk, wv, bv, spv = cache[1][lv]
            L = keys[lv].shape[1]
            if not ops.head_major_supported(L, wk.shape[0], E, E):
                return None
            k = ops.linear(keys_pp[lv].reshape(L, E), wk, bk, w_split=spk, head_major=(L, E))
            v = ops.linear(keys[lv].reshape(L, E), wv, bv, w_split=spv, head_major=(L, E))
            out.append((k, v))
        return out

    def _decoder_rows_pack(self):
        """every weight of the decoder's per-query chain in the fragment order csrc/decoder_rows.hip reads (per weight
        version; inference only), or None when the kernels do not apply (arithmetic mode, switch, widths)"""
        ops = get_ops()
        dec = self.transformer_decoder
        E = self.decoder_embed_dims
        if not ops.use_decoder_rows or ops.precision != "bf16x3" or E % 32 or self.training:
            return None
        params = list(dec.parameters()) + list(self.cls_embed.parameters()) + list(self.mask_embed.parameters())
        ver = fused.param_version(*params) + (id(ops),)
        cache = getattr(self, "_rows_pack", None)
        if cache is not None and cache[0] == ver:
            return cache[1]
        with torch.no_grad():
            def lin(w, b):
                f = ops.decoder_rows_pack(w.detach().float().contiguous())
                return None if f is None else (f, None if b is None else b.detach().float().contiguous())

            def ln(n):
                return (n.weight.detach(), n.bias.detach(), n.eps)
            layers = []
            for l in dec.layers:
                xa, sa = l.attentions[0].attn, l.attentions[1].attn
                f1, f2 = l.ffns[0].layers[0][0], l.ffns[0].layers[1]
                if l.ffns[0].act != "relu":
                    layers = None
                    break
                layers.append(dict(
                    xq=lin(xa.in_proj_weight[:E], xa.in_proj_bias[:E]), out0=lin(xa.out_proj.weight, xa.out_proj.bias),
                    ln0=ln(l.norms[0]), qk=lin(sa.in_proj_weight[:2 * E], sa.in_proj_bias[:2 * E]),
                    v=lin(sa.in_proj_weight[2 * E:], sa.in_proj_bias[2 * E:]),
                    out1=lin(sa.out_proj.weight, sa.out_proj.bias), ln1=ln(l.norms[1]),
                    ffn1=lin(f1.weight, f1.bias), ffn2=lin(f2.weight, f2.bias), ln2=ln(l.norms[2]), H=f1.weight.shape[0]))
            me = self.mask_embed
            head = dict(post=ln(dec.post_norm), cls=lin(self.cls_embed.weight, self.cls_embed.bias),
                        n_cls=self.cls_embed.weight.shape[0], me0=lin(me[0].weight, me[0].bias),
                        me1=lin(me[2].weight, me[2].bias), me2=lin(me[4].weight, me[4].bias))
        ok = layers is not None and all(v is not None for l in layers for v in l.values()) and \
            all(v is not None for v in head.values()) and me[4].weight.shape[0] == E and \
            all(l["H"] % 32 == 0 and (2 * E * 4 + 2 * (E + 8) * 2 + 2 * (l["H"] + 8) * 2) * 16 + 24 * 1024 <= 160 * 1024 for l in layers)
        pack = (layers, head) if ok else None
        self._rows_pack = (ver, pack)
        return pack

    def _cross_kv(self, layer, key, key_with_pos):
        """this layer's projected cross-attention keys / values (when the stacked per-level projection does not apply)"""
        ops = get_ops()
        a = layer.attentions[0].attn
        E = self.decoder_embed_dims
        w, b = a.in_proj_weight.detach(), a.in_proj_bias.detach()
        sp = fused.split_weight(a.in_proj_weight)
        part = (lambda lo, hi: None) if sp is None else (lambda lo, hi: (sp[0][lo:hi], sp[1][lo:hi]))
        k = ops.linear(key_with_pos, w[E:2 * E], b[E:2 * E], w_split=part(E, 2 * E))
        v = ops.linear(key, w[2 * E:], b[2 * E:], w_split=part(2 * E, 3 * E))
        return k, v

    def _forward_rows(self, pack, q, qpos, keys, keys_pp, kv_all, mask_tok, vol_shape, shapes, mf_split, last_only):
        """the decoder loop of ``forward`` (inference) on csrc/decoder_rows.hip: per layer cross-attention -> K1 ->
        self-attention -> K2 -> mask contraction + pooling; the same values as the launch-per-op loop"""
        ops = get_ops()
        layers_pk, head = pack
        nl = self.num_transformer_feat_level
        layers = self.transformer_decoder.layers
        n_layers = len(layers)
        q = q.contiguous()
        qpos = qpos[0].contiguous()                                   # [Q, E]
        cls_list, mask_list = [], []
        _, cls, me, qx = ops.decoder_rows_k2(None, q, qpos, None, head, layers_pk[0]["xq"])
        cls, mp, am = self._head_from_embed(cls, me, mask_tok, vol_shape, shapes[0], mf_split, not last_only, True)
        if not last_only:
            cls_list.append(cls)
            mask_list.append(mp)
        for i, layer in enumerate(layers):
            lv = i % nl
            if kv_all is not None:
                j = i // nl
                k, v = kv_all[lv][0][:, j], kv_all[lv][1][:, j]
            else:
                k, v = self._cross_kv(layer, keys[lv], keys_pp[lv])
            o = ops.masked_attention(qx, k, v, self.num_heads, am[0], am[1])
            q1, qs, ks, vs = ops.decoder_rows_k1(o, q, qpos, layers_pk[i])
            o2 = ops.masked_attention(qs, ks, vs, self.num_heads, None, None)
            last = i == n_layers - 1
            q, cls, me, qx = ops.decoder_rows_k2(o2, q1, qpos, layers_pk[i], head,
                                                 None if last else layers_pk[i + 1]["xq"], want_q=not last)
            cls, mp, am = self._head_from_embed(cls, me, mask_tok, vol_shape, shapes[(i + 1) % nl], mf_split,
                                                last or not last_only, not last)
            if last or not last_only:
                cls_list.append(cls)
                mask_list.append(mp)
        return cls_list, mask_list

    # -- mask2former_nusc_occ.py:589-689
    def forward(self, voxel_feats, img_metas=None, last_only=False, **kwargs):
        """Reference contract: (list[10] cls, list[10] mask_pred).  ``last_only=True`` (what
        ``simple_test`` needs) returns one-element lists holding the final layer's predictions and
        skips materialising the nine intermediate mask tensors."""
        mask_features = voxel_feats[0]
        memories = voxel_feats[:0:-1]
        B, E = mask_features.shape[:2]
        vol_shape = tuple(mask_features.shape[-3:])
        mask_tok = fused.channels_last_view(mask_features.float()).reshape(B, -1, E)
        keys, key_pos, shapes = [], [], []
        for i in range(self.num_transformer_feat_level):
            m = self.decoder_input_projs[i](memories[i])
            shp = tuple(m.shape[-3:])
            t = fused.channels_last_view(m.float()).reshape(B, -1, E) + self.level_embed.weight[i]
            keys.append(t)
            key_pos.append(self.decoder_positional_encoding.for_shape(shp, m.device).unsqueeze(0))
            shapes.append(shp)
        keys_pp = [k + kp for k, kp in zip(keys, key_pos)]          # once per level, shared by its 3 layers
        q = self.query_feat.weight.unsqueeze(0).expand(B, -1, -1)
        qpos = self.query_embed.weight.unsqueeze(0).expand(B, -1, -1)
        # the mask features are the "weight" of ten contractions: split them to bf16 (hi, lo) once
        mf_split = None if get_ops().precision == "f32" else get_ops().split_bf16(mask_tok.detach().contiguous())
        if self.training:
            last_only = False
            mask_tok = mask_tok.contiguous()
        n_layers = len(self.transformer_decoder.layers)
        cls_list, mask_list = [], []
        rows_pack = None if self.training else self._decoder_rows_pack()
        if rows_pack is not None:
            return self._forward_rows(rows_pack, q, qpos, keys, keys_pp, self._project_level_tokens(keys, keys_pp),
                                      mask_tok, vol_shape, shapes, mf_split, last_only)
        cls, mp, am = self.forward_head(q, mask_tok, vol_shape, shapes[0], mf_split, want_mask=not last_only)
        if not last_only:
            cls_list.append(cls)
            mask_list.append(mp)
        kv_all = self._project_level_tokens(keys, keys_pp)
        for i, layer in enumerate(self.transformer_decoder.layers):
            lv = i % self.num_transformer_feat_level
            kv = None
            if kv_all is not None:
                j = i // self.num_transformer_feat_level
                kv = (kv_all[lv][0][:, j], kv_all[lv][1][:, j])
            q = layer(q, qpos, keys[lv], key_pos[lv], am[0], am[1], keys_pp[lv], kv)
            last = i == n_layers - 1
            cls, mp, am = self.forward_head(q, mask_tok, vol_shape,
                                            shapes[(i + 1) % self.num_transformer_feat_level], mf_split,
                                            want_mask=last or not last_only, want_attn=not last)
            if last or not last_only:
                cls_list.append(cls)
                mask_list.append(mp)
        return cls_list, mask_list

    # -- mask2former_nusc_occ.py:691-696 at mask resolution (used by callers that want it)
    def format_results(self, mask_cls_results, mask_pred_results):
        return get_ops().upsample_classify(mask_pred_results.contiguous(), mask_cls_results.contiguous(),
                                           mask_pred_results.shape[-3:])

    def _output_voxels(self, cls, mask_pred, occ_size):
        if not self.align_corners:
            raise NotImplementedError("the fused resample + classify kernel implements align_corners=True "
                                      "(the value of every OccFormer config)")
        return get_ops().upsample_classify(mask_pred.contiguous(), cls.contiguous(), tuple(occ_size))


@HEADS.register_module()
class Mask2FormerNuscOccHead(NuscTrainingMixin, _Mask2FormerOccBase):
    # -- mask2former_nusc_occ.py:505-542 (eval branch)
    def forward_lidarseg(self, cls_preds, mask_preds, points, img_metas=None):
        """As the reference: the class volume at mask resolution (format_results), then a trilinear
        grid_sample of it at the LiDAR points, then softmax.  The class volume comes from the one-tap classify
        kernel (256 MB of logits read once, sequentially) and the point kernel gathers 8 taps x K classes per
        point -- instead of 8 taps x 100 queries per point straight from the query logits."""
        if self.padding_mode != "border" or not self.align_corners:
            raise NotImplementedError("lidarseg sampling is built for border padding / align_corners=True")
        ops = get_ops()
        # (cached per device: building a device tensor from a Python list is a blocking host-to-device copy,
        # i.e. a host sync in the middle of an otherwise asynchronous forward)
        key = (tuple(float(v) for v in img_metas[0]["pc_range"]), str(mask_preds.device))
        cache = self.__dict__.setdefault("_pc_range_cache", {})
        if key not in cache:
            pc = torch.tensor(key[0], dtype=torch.float32, device=mask_preds.device)
            cache[key] = (pc[:3].clone(), (pc[3:] - pc[:3]).clone())
        lo, ext = cache[key]
        vol = self.format_results(cls_preds, mask_preds)                       # [B, K, X, Y, Z]
        logits = []
        for b, p in enumerate(points):
            # grid_sample's (x, y, z) order for a [.., X, Y, Z] volume is the reversed axis order; the point
            # kernel takes [0, 1] coordinates (align_corners / border clamp applied inside)
            g = ((p[:, :3].float() - lo) / ext).flip(-1).contiguous()      # (flip, not a list index: no H2D copy)
            logits.append(ops.point_sample_3d(vol[b:b + 1], g[None], True, "border")[0].t())
        return torch.softmax(torch.cat(logits, 0), dim=1)

    # -- mask2former_nusc_occ.py:698-745
    def simple_test(self, voxel_feats, img_metas, points=None, **kwargs):
        all_cls, all_masks = self(voxel_feats, img_metas, last_only=True)
        cls, mp = all_cls[-1], all_masks[-1]
        res = {"output_voxels": [self._output_voxels(cls, mp, img_metas[0]["occ_size"])],
               "output_points": None}
        if points is not None:
            res["output_points"] = self.forward_lidarseg(cls, mp, points, img_metas)
        return res


@HEADS.register_module()
class Mask2FormerOccHead(KittiTrainingMixin, _Mask2FormerOccBase):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        # mask2former_occ.py:133-142: SemanticKITTI class weights 1/log(freq), normalised to class 0,
        # background weight kept from the config
        w = 1 / np.log(semantic_kitti_class_frequencies)
        self.class_weight = (w / w[0]).tolist() + [self.class_weight[-1]]
        self.get_sampling_weights()

    # -- mask2former_occ.py:673-703
    def simple_test(self, voxel_feats, img_metas, **kwargs):
        all_cls, all_masks = self(voxel_feats, img_metas, last_only=True)
        return {"output_voxels": [self._output_voxels(all_cls[-1], all_masks[-1], img_metas[0]["occ_size"])],
                "output_points": None}
