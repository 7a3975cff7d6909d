"""Host-side helpers that bind nn.Module parameters (kept in the reference's layout and
names for checkpoint compatibility) to the channels-last HIP kernels."""
import os
import weakref

import torch

from .ops import get_ops

_TAP_CACHE = {}
_SPLIT_CACHE = {}

# ---- validity of everything derived from parameter VALUES (bf16 hi/lo splits, tap-major / transposed / flipped
# layouts, folded BatchNorms, stacked projections, positional projections) ---------------------------------------
# ``param._version`` alone is NOT a valid key: in-place updates that go through ``.data`` or through the fused
# optimizer kernels (torch.optim.AdamW(fused=True): _fused_adamw_ mutates the storage without a version bump on this
# torch build) leave it unchanged.  Every derived value is therefore also keyed on a process-wide EPOCH that is bumped
#   * after every ``torch.optim.Optimizer.step`` (a global post-step hook, registered below at import),
#   * at the start of every ``OccupancyFormer.forward_train`` call (one "prepare weights" pass per training step:
#     each weight is re-laid / re-split once, on first use, by the kernels in gemm_bf16.hip -- covers hand-written
#     update loops that touch ``p.data``),
#   * by ``invalidate_caches()`` (public; call it after any other out-of-band write to a parameter).
# ``OCCF_CACHE_CHECK=1`` additionally stores a content checksum with every entry and asserts it on every hit
# (device-side ``torch._assert_async``: no host sync), to catch writers that bypass all of the above.
_EPOCH = [0]
_CHECK = os.environ.get("OCCF_CACHE_CHECK", "0") == "1"


def invalidate_caches():
    """declare every parameter-derived cache entry stale (cheap: a counter bump; entries are rebuilt on next use)"""
    _EPOCH[0] += 1


def epoch():
    return _EPOCH[0]


def param_version(*params):
    """cache key component for values derived from ``params`` -- use this, never ``_version`` alone"""
    return tuple((p._version, p.data_ptr()) for p in params) + (_EPOCH[0],)


def _post_step_hook(optimizer, args, kwargs):
    invalidate_caches()


try:        # every torch optimizer instance, present and future, in this process
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg
    _hook_handle = _reg(_post_step_hook)
except ImportError:      # pragma: no cover  (torch < 2.0)
    _hook_handle = None


def _checksum(param):
    p = param.detach()
    return torch.stack((p.double().sum(), p.double().abs().sum()))


def _versioned(cache, param, make):
    """value derived from a Parameter, recomputed when the parameter may have changed: its ``_version`` bumped
    (load_state_dict, foreach optimizers), the process-wide epoch moved (see above), or ``id(param)`` got reused."""
    if not param.is_leaf:
        return make()                   # a graph intermediate (e.g. two weights concatenated per step): never cached
    key = id(param)
    hit = cache.get(key)
    ver = (param._version, param.data_ptr(), _EPOCH[0])
    if hit is not None and hit[0] == ver and hit[2]() is param:
        if _CHECK:
            torch._assert_async(bool_all(hit[3] == _checksum(param)),
                                "occformer_amd: a cached weight layout is stale (parameter written behind the cache)")
        return hit[1]
    val = make()
    cache[key] = (ver, val, weakref.ref(param), _checksum(param) if _CHECK else None)
    return val


def bool_all(t):
    return t.all()


# ---- all weight-derived layouts of a step in ONE launch ------------------------------------------------------------
class _WeightPrep:
    """The layouts the kernels consume -- 'split' (hi, lo of the weight read as [N, K]), 'tap' ([Cout, taps*Cin] + split),
    'wt' (W^T [K, N] + split: data gradient of a linear), 'flip' / 'dg' ([Cin, taps*Cout] with / without flipped taps
    + split: stride-1 / strided data gradient of a convolution) -- as persistent buffers refreshed together by
    csrc/prep.hip (one table of strided-gather descriptors, one launch) whenever any parameter may have changed
    (``param_version``).  Rebuilding them one by one after every optimizer step cost ~1 200 small launches per step."""

    KINDS = ("split", "tap", "wt", "flip", "dg")

    def __init__(self):
        self.entries = {}
        self.table = None
        self.table_for = None

    @staticmethod
    def _spec(param, kind):
        """-> (dims[5], strides[5] of the INPUT in elements, base offset, output shape) or None if not expressible"""
        if not param.is_contiguous() or param.dim() not in (2, 4, 5) or param.numel() >= 2 ** 31:
            return None
        shp = list(param.shape)
        N = shp[0]
        C = shp[1]
        k = shp[2:] + [1] * (5 - len(shp)) if len(shp) > 2 else [1, 1, 1]
        kx, ky, kz = k
        taps = kx * ky * kz
        sN, sC, sx, sy, sz = C * taps, taps, ky * kz, kz, 1
        if kind == "split":
            spec = ((1, 1, 1, N, C * taps), (0, 0, 0, C * taps, 1), 0, (N, C * taps))
        elif kind == "tap":
            spec = ((N, kx, ky, kz, C), (sN, sx, sy, sz, sC), 0, (N, taps * C))
        elif kind == "wt":
            if taps != 1:
                return None
            spec = ((1, 1, 1, C, N), (0, 0, 0, 1, C), 0, (C, N))
        elif kind == "dg":
            spec = ((C, kx, ky, kz, N), (sC, sx, sy, sz, sN), 0, (C, taps * N))
        elif kind == "flip":
            spec = ((C, kx, ky, kz, N), (sC, -sx, -sy, -sz, sN), (kx - 1) * sx + (ky - 1) * sy + (kz - 1) * sz,
                    (C, taps * N))
        else:
            raise KeyError(kind)
        return None if spec[0][4] % 2 else spec

    def get(self, param, kind):
        """-> (fp32 layout or None for 'split', (hi, lo)); None when this (parameter, kind) cannot go through the
        table (non-leaf, odd inner dimension, CPU tensor under the GPU library ...) -- the caller then derives it the
        slow way"""
        if not param.is_leaf or param.dtype != torch.float32:
            return None
        ops = get_ops()
        key = (id(param), kind)
        e = self.entries.get(key)
        if e is not None and (e["ref"]() is not param or e["ptr"] != param.data_ptr() or e["shape"] != tuple(param.shape)
                              or e["ops"] is not ops):
            e = None
        if e is None:
            spec = self._spec(param, kind)
            if spec is None:
                return None
            dims, strides, base, oshape = spec
            dev = param.device
            e = dict(ref=weakref.ref(param), ptr=param.data_ptr(), shape=tuple(param.shape), ops=ops, kind=kind,
                     dims=dims, strides=strides, base=base, stamp=None,
                     mode=1 if kind in ("wt", "dg", "flip") else 0,      # transposes go through LDS tiles (prep.hip)
                     f32=None if kind == "split" else torch.empty(oshape, dtype=torch.float32, device=dev),
                     hi=torch.empty(oshape, dtype=torch.int16, device=dev),
                     lo=torch.empty(oshape, dtype=torch.int16, device=dev))
            self.entries[key] = e
            self.table = None
            # a NEW layout is derived on its own (a two-row table, one small launch): refreshing the whole table for
            # every first use made the first step O(entries^2) -- 386 launches x 0.71 ms = 17 % of a profiled run (r03x)
            self._refresh_one(ops, e, param)
        if e["stamp"] != (param._version, _EPOCH[0]):
            self._refresh(ops, param.device)
        elif _CHECK:
            torch._assert_async(bool_all(e["sum"] == _checksum(param)),
                                "occformer_amd: a prepared weight layout is stale (parameter written behind the cache)")
        return e["f32"], (e["hi"], e["lo"])

    @staticmethod
    def _span(e):
        """pairs of the table span of an entry (whole 512-pair slots; csrc/prep.hip): mode 1 = one slot per 32 x 32 tile
        of the transposed output [M][N], mode 0 = the output pairs in order"""
        d = e["dims"]
        if e["mode"] == 1:
            m = d[0] * d[1] * d[2] * d[3]
            return ((m + 31) // 32) * ((d[4] + 31) // 32) * 512
        n = d[0] * d[1] * d[2] * d[3] * d[4]
        return (n // 2 + 511) // 512 * 512

    def _refresh_one(self, ops, e, p):
        span = self._span(e)
        rows = [[e["ptr"] + 4 * e["base"], 0 if e["f32"] is None else e["f32"].data_ptr(), e["hi"].data_ptr(),
                 e["lo"].data_ptr(), 0, *e["dims"], *e["strides"], e["mode"]],
                [0, 0, 0, 0, span] + [1] * 5 + [0] * 5 + [0]]
        ops.prep_weights(torch.tensor(rows, dtype=torch.int64).to(p.device), 1, span)
        e["stamp"] = (p._version, _EPOCH[0])
        if _CHECK:
            e["sum"] = _checksum(p)

    def _refresh(self, ops, device):
        live = []
        for key, e in list(self.entries.items()):
            p = e["ref"]()
            if p is None or p.data_ptr() != e["ptr"] or tuple(p.shape) != e["shape"]:
                del self.entries[key]
                self.table = None
            elif e["ops"] is ops and p.device == device:
                live.append((e, p))
        if not live:
            return
        ids = tuple(id(e) for e, _ in live)
        if self.table is None or self.table_for != (ids, str(device)):
            rows, pair0 = [], 0
            for e, p in live:
                rows.append([e["ptr"] + 4 * e["base"], 0 if e["f32"] is None else e["f32"].data_ptr(),
                             e["hi"].data_ptr(), e["lo"].data_ptr(), pair0, *e["dims"], *e["strides"], e["mode"]])
                pair0 += self._span(e)
            rows.append([0, 0, 0, 0, pair0] + [1] * 5 + [0] * 5 + [0])
            # (a pageable upload: only when the set of prepared layouts changes, i.e. during the first one or two steps)
            self.table = torch.tensor(rows, dtype=torch.int64).to(device)
            self.table_for = (ids, str(device))
            self.total = pair0
        ops.prep_weights(self.table, len(live), self.total)
        for e, p in live:
            e["stamp"] = (p._version, _EPOCH[0])
            if _CHECK:
                e["sum"] = _checksum(p)
            for derived in ("_occf_halo_pack", "_occf_wino_pack", "_occf_wino_pack_f16"):   # fragment orders of the halo / Winograd kernels:
                if hasattr(e["hi"], derived):                        # derived from hi / lo / the fp32 layout
                    delattr(e["hi"], derived)


_PREP = _WeightPrep()


def prepared(param, kind):
    """(fp32 layout | None, (hi, lo)) of ``param`` in layout ``kind`` from the one-launch table, or None (see
    ``_WeightPrep.get``); exact-fp32 mode has no splits and stays on the per-parameter caches"""
    if get_ops().precision == "f32":
        return None
    return _PREP.get(param, kind)


def _w2d(w):
    return w.reshape(w.shape[0], -1)


def split_weight(param, as_2d=None):
    """(hi, lo) bf16 split of a weight for the bf16 matrix-core path, or None in exact-fp32 mode.  ``as_2d``: None /
    ``_w2d`` (the weight read as [N, K]) or ``_tap_layout`` (tap-major) go through the one-launch table."""
    ops = get_ops()
    if ops.precision == "f32":
        return None
    kind = "split" if as_2d is None or as_2d is _w2d else "tap" if as_2d is _tap_layout else None
    if kind is not None:
        hit = _PREP.get(param, kind)
        if hit is not None:
            return hit[1]
    return _versioned(_SPLIT_CACHE, param,
                      lambda: ops.split_bf16(param.detach() if as_2d is None else as_2d(param.detach())))


def _tap_layout(w):
    if w.dim() == 4:
        w = w.unsqueeze(-1)
    return w.permute(0, 2, 3, 4, 1).reshape(w.shape[0], -1).contiguous()


def tap_major(conv):
    """Conv weight [Cout, Cin, kX, kY(, kZ)] -> [Cout, taps*Cin] (k = tap*Cin + cin), cached
    until the parameter is modified."""
    return tap_major_of(conv.weight)


def tap_major_of(weight):
    hit = prepared(weight, "tap")
    if hit is not None:
        return hit[0]
    return _versioned(_TAP_CACHE, weight, lambda: _tap_layout(weight.detach()))


def channels_last_view(x):
    """logical [B, C, X, Y, Z] -> [B, X, Y, Z, C] with unit channel stride (copy only if needed)."""
    v = x.permute(0, 2, 3, 4, 1)
    return v if v.is_contiguous() else v.contiguous()


def linear(x, lin, act=0, residual=None, head_major=None):
    return get_ops().linear(x, lin.weight.detach(), None if lin.bias is None else lin.bias.detach(), act,
                            residual, w_split=split_weight(lin.weight), head_major=head_major)


def layernorm(x, ln):
    return get_ops().layernorm(x, ln.weight.detach(), ln.bias.detach(), ln.eps)


def conv(x_cl, conv, act=0, gn=None):
    """nn.Conv3d / nn.Conv2d module applied to a channels-last [B, X, Y, Z, C] tensor.  ``gn``: the
    nn.GroupNorm that follows -- its statistics are then taken from the epilogue of the convolution
    (``get_ops().last_gn_stats``) when the kernel supports it."""
    ks = tuple(conv.kernel_size) + (1,) * (3 - len(conv.kernel_size))
    pad = tuple(conv.padding) + (0,) * (3 - len(conv.padding))
    stride = conv.stride[0]
    dil = conv.dilation[0]
    assert all(s == stride for s in conv.stride) and all(d == dil for d in conv.dilation) and conv.groups == 1
    bias = None if conv.bias is None else conv.bias.detach()
    if ks == (1, 1, 1) and stride == 1 and x_cl.is_contiguous():
        rows = x_cl.numel() // (x_cl.shape[0] * x_cl.shape[-1])
        return get_ops().linear(x_cl, conv.weight.detach().reshape(conv.out_channels, -1), bias, act,
                                w_split=split_weight(conv.weight, _w2d),
                                gn=None if gn is None else (gn.num_groups, gn.eps, rows))
    return get_ops().conv3d(x_cl, tap_major(conv), ks, stride, dil, pad, bias, act,
                            w_split=split_weight(conv.weight, _tap_layout),
                            gn=None if gn is None else (gn.num_groups, gn.eps))


def conv_gn(x_cl, conv_mod, gn, relu=False, tokens=False, residual=None):
    """conv -> GroupNorm (-> ReLU / token buffer / + residual) with the statistics from the conv epilogue"""
    ops = get_ops()
    y = conv(x_cl, conv_mod, gn=gn)
    return group_norm(y, gn, relu, tokens, residual, stats=ops.last_gn_stats)


_FOLD_CACHE = {}


def conv_bn(x_cl, conv_mod, bn=None, act=0, residual=None):
    """nn.Conv2d/3d (+ eval-mode BatchNorm folded into weight and bias) on channels-last [B, X, Y, Z, C]:
    y = act(conv'(x) + b') [+ residual], W' = W * gamma / sqrt(var + eps), b' = beta + (b - mean) * gamma / sqrt(..)."""
    ops = get_ops()
    w = conv_mod.weight
    deps = [w] + ([] if conv_mod.bias is None else [conv_mod.bias]) + \
        ([] if bn is None else [bn.weight, bn.bias, bn.running_mean, bn.running_var])
    ver = param_version(*deps) + (ops.precision, id(ops))
    hit = _FOLD_CACHE.get(id(w))
    if hit is None or hit[0] != ver or hit[2]() is not w:
        with torch.no_grad():
            wt = _tap_layout(w.detach()).float()
            b = None if conv_mod.bias is None else conv_mod.bias.detach().float()
            if bn is not None:
                if bn.training:
                    raise NotImplementedError("BatchNorm folding is the inference path; call .eval()")
                sc = bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)
                wt = (wt * sc[:, None]).contiguous()
                b = bn.bias.detach() - bn.running_mean.detach() * sc + (0 if b is None else b * sc)
            split = None if ops.precision == "f32" else ops.split_bf16(wt)
        hit = (ver, (wt, None if b is None else b.contiguous(), split), weakref.ref(w))
        _FOLD_CACHE[id(w)] = hit
    wt, b, split = hit[1]
    ks = tuple(conv_mod.kernel_size) + (1,) * (3 - len(conv_mod.kernel_size))
    pad = tuple(conv_mod.padding) + (0,) * (3 - len(conv_mod.padding))
    stride, dil = conv_mod.stride[0], conv_mod.dilation[0]
    assert all(v == stride for v in conv_mod.stride) and all(d == dil for d in conv_mod.dilation) and conv_mod.groups == 1
    if ks == (1, 1, 1) and stride == 1 and x_cl.is_contiguous():
        return ops.linear(x_cl, wt, b, act, residual, w_split=split)
    return ops.conv3d(x_cl, wt, ks, stride, dil, pad, b, act, residual, w_split=split)


def group_norm(x_cl, gn, relu=False, tokens=False, residual=None, stats=None):
    """nn.GroupNorm module on a contiguous channels-last tensor [B, ..., Z, C]."""
    ops = get_ops()
    if stats is None:
        stats = ops.groupnorm_stats(x_cl, gn.num_groups, gn.eps)
    return ops.groupnorm_apply(x_cl, stats, gn.weight.detach(), gn.bias.detach(), gn.num_groups, relu, tokens,
                               residual)


def require_eval(module):
    if module.training:
        raise NotImplementedError(f"{type(module).__name__}: the HIP forward path is inference-only in this "
                                  "round (backward kernels are the next scope row); call .eval()")


def mlp(x, fc1, fc2, act, ln=None, ln_mode=0):
    """Transformer feed-forward half on tokens [..., C]: one fused kernel when the shape allows
    (hidden activation stays in LDS), otherwise LN / linear / linear."""
    ops = get_ops()
    C, H = fc1.in_features, fc1.out_features
    if ops.mlp_fused_supported(C, H) and x.is_contiguous():
        return ops.mlp_fused(x, None if ln is None else ln.weight.detach(), None if ln is None else ln.bias.detach(),
                             split_weight(fc1.weight), fc1.bias.detach(), split_weight(fc2.weight),
                             fc2.bias.detach(), act, ln_mode, 1e-5 if ln is None else ln.eps)
    h = layernorm(x, ln) if ln_mode == 1 else x
    y = linear(linear(h, fc1, act=act), fc2, residual=x)
    return layernorm(y, ln) if ln_mode == 2 else y
