"""Thin host-side marshalling from torch tensors to the C ABI of liboccformer_hip.so.

Tensors are only used for device memory, streams and shapes; every computation happens in
the hand-written gfx950 kernels behind ``include/occformer_hip.h``.  ``ops`` (the module
level instance) is bound to the real library and refuses non-GPU tensors.
"""
import ctypes
import os

import torch

from . import _lib


try:
    _raw_stream = torch._C._cuda_getCurrentRawStream          # (device index) -> hipStream_t as int
except AttributeError:      # pragma: no cover
    def _raw_stream(dev):
        return torch.cuda.current_stream(dev).cuda_stream


class OccfError(RuntimeError):
    pass


class HipOps:
    """One method per C entry point.  ``strict=True`` (the product binding) requires
    GPU-resident contiguous tensors; the test-suite builds a non-strict binding around
    the host-emulation build of the same kernel sources to check index math on CPU."""

    def __init__(self, lib, strict=True):
        self.lib = lib
        self.strict = strict
        self.last_flops = 0          # set by the MFMA-bound ops (for bench.py's roofline)
        # arithmetic of the dense contractions: "f32" = exact fp32 MFMA; "bf16x3" = 3-term bf16
        # split (fp32-class accuracy, matrix-core rate); "bf16" = plain bf16 products
        self.precision = os.environ.get("OCCF_PRECISION", "bf16x3")
        self.wgrad_f16 = os.environ.get("OCCF_WGRAD_F16", "1") != "0"
        # (the linears' weight gradients are bound by their operand staging: two products are no faster than three, r06a)
        self.wgrad_f16_linear = os.environ.get("OCCF_WGRAD_F16_LINEAR", "0") == "1"
        # the convolutions' weight gradients on ONE product (dy and x each one fp16 piece) where the G8 kernel applies:
        # scripts/precision_probe.py wg1c -- whole gradient 4.5e-5 -> 5.0e-5, worst parameter 4.4e-4 -> 5.9e-4
        self.wgrad_f16_single = os.environ.get("OCCF_WGRAD_F16_SINGLE", "1") != "0"
        self.use_halo_conv = os.environ.get("OCCF_HALO_CONV", "1") == "1"
        self.halo_frag = os.environ.get("OCCF_HALO_FRAG", "1") == "1"
        # Winograd F(2, 3) along x for the stride-1 3^3 convolutions (csrc/conv_wino.hip); 0 = the direct halo kernel
        self.use_wino = os.environ.get("OCCF_WINO", "1") == "1"
        # the data gradients of those convolutions on two fp16-piece products (dy in ONE piece; OCCF_DGRAD_F16=0: three)
        self.dgrad_f16 = os.environ.get("OCCF_DGRAD_F16", "1") == "1"
        # ... and on ONE product: the transformed filters too as one fp16 piece (csrc/conv_wino.hip W1;
        # scripts/precision_probe.py cd1: whole gradient 5.0e-5 -> 5.2e-5, worst parameter 5.9e-4 -> 6.9e-4)
        self.dgrad_f16_single = os.environ.get("OCCF_DGRAD_F16_SINGLE", "1") == "1"
        # 2-D 3x3 weight gradients with a first extent of 8 / 16 / 32 / 64 as [1, Y, X] volumes on the G8 kernel
        self.wgrad_2d_as_g8 = os.environ.get("OCCF_WGRAD_2D_G8", "1") != "0"
        # the decoder's per-query chain as two kernels per layer in inference (csrc/decoder_rows.hip); 0 = one launch per op
        self.use_decoder_rows = os.environ.get("OCCF_DECODER_ROWS", "1") == "1"
        # OCCF_DETERMINISTIC=1: every scatter sum of the backward in a fixed order or in integer fixed point (two runs of
        # a seeded step give the same bits; tests/test_train_step.py::test_training_step_is_reproducible)
        self.deterministic = os.environ.get("OCCF_DETERMINISTIC", "0") == "1"
        self.swin_frag = os.environ.get("OCCF_SWIN_FRAG", "1") == "1"
        # fused mask contraction + preserve-pooling (the intermediate mask logits are never written)
        self.use_fused_mask_pool = os.environ.get("OCCF_FUSED_MASK_POOL", "1") == "1"
        self._mgp_reverse = 0
        self.use_fused_swin = os.environ.get("OCCF_FUSED_SWIN", "1") == "1"
        self.use_fused_mlp = os.environ.get("OCCF_FUSED_MLP", "1") == "1"

    # ------------------------------------------------------------------ plumbing
    def _stream(self):
        """the raw HIP stream torch currently launches on (torch.cuda.current_stream() builds a Stream object through
        four Python layers: 9 us per call x 2 700 library launches = 8 ms of host time per training step, r03d)"""
        if self.strict:
            return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
        return ctypes.c_void_p(0)

    def _ptr(self, t, dtype=None, n=None):
        """device pointer of a contiguous tensor (None -> NULL); ``n`` = the element count the kernel will address
        through it: the kernels do no bounds checks, an operand of another size is refused here"""
        if t is None:
            return ctypes.c_void_p(0)
        if n is not None and t.numel() != n:
            raise OccfError(f"operand of {t.numel()} elements (shape {tuple(t.shape)}) where the kernel addresses {n}")
        if self.strict and not t.is_cuda:
            raise OccfError("occformer_amd ops need GPU tensors (no CPU path exists)")
        if not t.is_contiguous():
            raise OccfError("non-contiguous tensor passed to a HIP op")
        if dtype is not None and t.dtype != dtype:
            raise OccfError(f"expected {dtype}, got {t.dtype}")
        return ctypes.c_void_p(t.data_ptr())

    def _call(self, name, *args):
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise OccfError(f"{name} failed with code {rc}")

    f32 = torch.float32
    i32 = torch.int32

    # ------------------------------------------------------------------ view transform
    def bev_pool_forward(self, x, geom, interval_lengths, interval_starts, b, d, h, w):
        """Same signature/argument order as bev_pool_ext.bev_pool_forward
        (mmdet3d/ops/bev_pool/src/bev_pool.cpp:22-28)."""
        n, c = x.shape
        out = torch.empty((b, d, h, w, c), dtype=x.dtype, device=x.device)
        self._call("occf_bev_pool_fwd", self._ptr(x, self.f32), self._ptr(geom, self.i32, 4 * n),
                   self._ptr(interval_starts, self.i32), self._ptr(interval_lengths, self.i32, interval_starts.numel()),
                   self._ptr(out), int(b), int(d), int(h), int(w), n, c,
                   interval_starts.numel(), self._stream())
        return out

    def bev_pool_backward(self, out_grad, geom, interval_lengths, interval_starts, b, d, h, w):
        n = geom.shape[0]
        c = out_grad.shape[4]
        x_grad = torch.empty((n, c), dtype=out_grad.dtype, device=out_grad.device)
        self._call("occf_bev_pool_bwd", self._ptr(out_grad, self.f32, int(b) * int(d) * int(h) * int(w) * c),
                   self._ptr(geom, self.i32, 4 * n),
                   self._ptr(interval_starts, self.i32), self._ptr(interval_lengths, self.i32, interval_starts.numel()),
                   self._ptr(x_grad), int(b), int(d), int(h), int(w), n, c,
                   interval_starts.numel(), self._stream())
        return x_grad

    def lss_voxel_index(self, frustum, cam, bda, grid, B, N, X, Y, Z, bda4):
        DHW = frustum.numel() // 3
        vox = torch.empty((B * N * DHW,), dtype=self.i32, device=frustum.device)
        self._call("occf_lss_voxel_index", self._ptr(frustum, self.f32), self._ptr(cam, self.f32),
                   self._ptr(bda, self.f32), self._ptr(grid, self.f32), self._ptr(vox), B, N, DHW,
                   X, Y, Z, int(bda4), self._stream())
        return vox

    def lift_splat_forward(self, depth, feat_cl, offsets, sorted_pts, n_vox):
        """depth [BN, D, HW] f32, feat_cl [BN, HW, C] f32 -> [n_vox, C]."""
        BN, D, HW = depth.shape
        C = feat_cl.shape[-1]
        if tuple(feat_cl.shape[:2]) != (BN, HW) or offsets.numel() != n_vox + 1 or sorted_pts.numel() != BN * D * HW:
            raise OccfError("lift_splat: depth / feature / CSR sizes disagree (the kernel does no bounds checks)")
        out = torch.empty((n_vox, C), dtype=depth.dtype, device=depth.device)
        self._call("occf_lift_splat_fwd", self._ptr(depth, self.f32), self._ptr(feat_cl, self.f32),
                   self._ptr(offsets, self.i32), self._ptr(sorted_pts, self.i32), self._ptr(out),
                   n_vox, BN, D, HW, C, self._stream())
        return out

    def lift_splat_backward(self, out_grad, depth, feat_cl, vox):
        BN, D, HW = depth.shape
        C = feat_cl.shape[-1]
        if tuple(feat_cl.shape[:2]) != (BN, HW) or vox.numel() != BN * D * HW:
            raise OccfError("lift_splat: depth / feature / voxel-index sizes disagree")
        d_depth = torch.empty_like(depth)
        d_feat = torch.empty_like(feat_cl)
        self._call("occf_lift_splat_bwd", self._ptr(out_grad, self.f32), self._ptr(depth, self.f32),
                   self._ptr(feat_cl, self.f32), self._ptr(vox, self.i32), self._ptr(d_depth),
                   self._ptr(d_feat), vox.numel(), BN, D, HW, C, self._stream())
        return d_depth, d_feat

    # ------------------------------------------------------------------ encoder
    def window_attention(self, qkv, qkv_bias, bias_table, B, X, Y, S, heads, shift):
        """qkv [B*X*Y*S, 3C] -> attention output [B*X*Y*S, C] (before proj)."""
        C = qkv.shape[1] // 3
        out = torch.empty((qkv.shape[0], C), dtype=qkv.dtype, device=qkv.device)
        self._call("occf_window_attn_fwd", self._ptr(qkv, self.f32, B * X * Y * S * 3 * C), self._ptr(qkv_bias, self.f32, 3 * C),
                   self._ptr(bias_table, self.f32), self._ptr(out), B, X, Y, S, C, heads, int(shift),
                   self._stream())
        return out

    # ------------------------------------------------------------------ pixel decoder
    def _swin_pack(self, split, rows, C):
        """(hi, lo) of a [rows, 128] weight in MFMA-fragment order for the fused Swin kernel (cached on the split
        tensor, which is cached per weight version)"""
        hi, lo = split
        pk = getattr(hi, "_occf_swin_pack", None)
        if pk is None:
            fh, fl = torch.empty_like(hi), torch.empty_like(lo)
            self._call("occf_swin_attn_pack", self._ptr(hi), self._ptr(lo), self._ptr(fh), self._ptr(fl), rows, C,
                       self._stream())
            pk = (fh, fl)
            hi._occf_swin_pack = pk
        return pk

    def swin_attention_fused(self, x, ln_w, ln_b, eps, wqkv_split, bqkv, bias_table, wproj_split, bproj, B, X, Y, S,
                             heads, shift):
        """x [B*X*Y*S, C] -> x + proj(window_msa(layernorm(x))) in one kernel, or None when the shape / precision
        mode is outside the fused kernel (C = 128, 4 heads, 3-term split)."""
        C = x.shape[1]
        if not self.use_fused_swin or self.precision != "bf16x3" or C != 128 or heads != 4 or wqkv_split is None:
            return None
        out = torch.empty_like(x)
        self.last_flops = 2 * x.shape[0] * C * 4 * C + 4 * x.shape[0] * 49 * C
        packed = 0
        if self.swin_frag:
            wqkv_split, wproj_split, packed = self._swin_pack(wqkv_split, 3 * C, C), self._swin_pack(wproj_split, C, C), 1
        rc = self.lib.occf_swin_attn_fused_fwd(
            self._ptr(x, self.f32), self._ptr(ln_w, self.f32), self._ptr(ln_b, self.f32), float(eps),
            self._ptr(wqkv_split[0]), self._ptr(wqkv_split[1]), self._ptr(bqkv, self.f32),
            self._ptr(bias_table, self.f32), self._ptr(wproj_split[0]), self._ptr(wproj_split[1]),
            self._ptr(bproj, self.f32), self._ptr(out), B, X, Y, S, C, heads, int(shift), packed, self._stream())
        if rc == -2:
            return None
        if rc != 0:
            raise OccfError(f"occf_swin_attn_fused_fwd failed with code {rc}")
        return out

    def msda3d(self, value, offsets, logits, level_shapes, heads, points, head_major=False):
        """value [B, Nq, E] (or [B, heads, Nq, E/heads] with head_major); offsets [B, Nq, heads*L*P*3];
        logits [B, Nq, heads*L*P] -> [B, Nq, E].  offsets / logits may be column slices of one tensor
        (unit channel stride, any row stride)."""
        if head_major:
            B, _, Nq, dh = value.shape
            E = heads * dh
        else:
            B, Nq, E = value.shape
        L = len(level_shapes)
        arr = (ctypes.c_int32 * (3 * L))(*[int(v) for s in level_shapes for v in s])
        out = torch.empty((B, Nq, E), dtype=value.dtype, device=value.device)
        for t in (offsets, logits):
            if t.stride(-1) != 1 or t.stride(0) != t.stride(1) * Nq or (self.strict and not t.is_cuda) or \
                    t.dtype != self.f32:
                raise OccfError("msda3d: offsets/logits must be fp32 GPU rows with unit channel stride")
        self._call("occf_msda3d_fwd", self._ptr(value, self.f32), ctypes.c_void_p(offsets.data_ptr()),
                   ctypes.c_void_p(logits.data_ptr()), self._ptr(out),
                   ctypes.cast(arr, ctypes.c_void_p), L, B, Nq, heads, E // heads, points, int(head_major),
                   offsets.stride(1), logits.stride(1), self._stream())
        return out

    # ------------------------------------------------------------------ occupancy decoder
    def mask_pool(self, mask_pred, target):
        """mask_pred [B, Q, X, Y, Z] -> pooled [B, Q, L], blocked u8 [B, Q, L], row_open i32 [B*Q]."""
        B, Q, X, Y, Z = mask_pred.shape
        ox, oy, oz = (int(t) for t in target)
        L = ox * oy * oz
        pooled = torch.empty((B, Q, L), dtype=mask_pred.dtype, device=mask_pred.device)
        blocked = torch.empty((B, Q, L), dtype=torch.uint8, device=mask_pred.device)
        row_open = torch.empty((B * Q,), dtype=self.i32, device=mask_pred.device)
        self._call("occf_mask_pool_fwd", self._ptr(mask_pred, self.f32), self._ptr(pooled),
                   self._ptr(blocked), self._ptr(row_open), B * Q, X, Y, Z, ox, oy, oz, self._stream())
        return pooled, blocked, row_open

    def mask_gemm_pool(self, mask_embed, feat_split, vol_shape, target):
        """mask_embed [B, Q, E], feat_split = (hi, lo) of [B, V, E] -> pooled [B,Q,L], blocked, row_open
        without materialising the [B, Q, X, Y, Z] logits; None when the pooling geometry is not uniform
        (the caller then runs the GEMM and mask_pool).  Needs a bf16 precision mode."""
        B, Q, E = mask_embed.shape
        X, Y, Z = (int(v) for v in vol_shape)
        ox, oy, oz = (int(t) for t in target)
        need = self.lib.occf_mask_gemm_pool_workspace(B, Q, E, X, Y, Z, ox, oy, oz)
        if need <= 0 or self.precision == "f32":
            return None
        L = ox * oy * oz
        ws = torch.empty((need,), dtype=self.f32, device=mask_embed.device)
        pooled = torch.empty((B, Q, L), dtype=self.f32, device=mask_embed.device)
        blocked = torch.empty((B, Q, L), dtype=torch.uint8, device=mask_embed.device)
        row_open = torch.empty((B * Q,), dtype=self.i32, device=mask_embed.device)
        self.last_flops = 2 * B * Q * X * Y * Z * E
        self._call("occf_mask_gemm_pool_fwd", self._ptr(mask_embed, self.f32), self._ptr(feat_split[0]),
                   self._ptr(feat_split[1]), self._ptr(pooled), self._ptr(blocked), self._ptr(row_open),
                   self._ptr(ws), B, Q, E, X, Y, Z, ox, oy, oz, 3 if self.precision == "bf16x3" else 1,
                   self._mgp_reverse, self._stream())
        self._mgp_reverse ^= 1          # next call walks the features the other way (cache reuse, same results)
        return pooled, blocked, row_open

    def masked_attention(self, q, k, v, heads, blocked=None, row_open=None):
        """q [B, Q, E]; k, v [B, L, E] -> [B, Q, E] (all projected; before out_proj)."""
        B, Q, E = q.shape
        L = k.shape[1]
        need = self.lib.occf_masked_xattn_workspace(B, Q, L, heads)
        ws = torch.empty((need,), dtype=self.f32, device=q.device)
        out = torch.empty_like(q)
        self._call("occf_masked_xattn_fwd", self._ptr(q, self.f32), self._ptr(k, self.f32, B * L * E),
                   self._ptr(v, self.f32, B * L * E), self._ptr(blocked, torch.uint8, B * Q * L),
                   self._ptr(row_open, self.i32, B * Q),
                   self._ptr(out), self._ptr(ws), need, B, Q, L, E, heads, self._stream())
        return out

    # ---- the decoder's per-query chain as two kernels per layer (csrc/decoder_rows.hip, inference)
    def decoder_rows_pack(self, w):
        """weight [N, K] fp32 -> the (hi, lo) fragment pair csrc/decoder_rows.hip reads, or None (K % 32 != 0)"""
        N, K = w.shape
        n = self.lib.occf_decoder_rows_pack_elems(N, K)
        if n <= 0 or w.stride(1) != 1:
            return None
        fh = torch.empty((n,), dtype=torch.int16, device=w.device)
        fl = torch.empty((n,), dtype=torch.int16, device=w.device)
        self._call("occf_decoder_rows_pack", self._ptr(w, self.f32), w.stride(0), N, K, self._ptr(fh), self._ptr(fl),
                   self._stream())
        return fh, fl

    @staticmethod
    def _pair(f):
        return None if f is None else (ctypes.c_void_p * 2)(f[0].data_ptr(), f[1].data_ptr())

    def decoder_rows_k1(self, attn_out, q_in, qpos, pk):
        """pk: dict of (fragment pair, bias) per linear + LayerNorm triples (head.py: _decoder_rows_pack)
        -> (q1, qs, ks, vs), each [B, Q, E]"""
        B, Q, E = q_in.shape
        out = torch.empty((4, B, Q, E), dtype=self.f32, device=q_in.device)
        if self.strict and not (attn_out.is_cuda and attn_out.is_contiguous() and q_in.is_contiguous()):
            raise OccfError("decoder_rows_k1: contiguous GPU tensors expected")
        # the static arguments (weights, biases, norms) as ctypes objects built ONCE per weight pack: with ~20 / ~45
        # arguments per call the per-call conversions cost more host time than the launches they replace (r06d)
        args = pk.get("_k1_args")
        if args is None:
            g, bt, eps = pk["ln0"]
            args = [None, None, self._ptr(qpos, self.f32, Q * E), 0, E, Q, self._pair(pk["out0"][0]),
                    self._ptr(pk["out0"][1]), self._ptr(g), self._ptr(bt), ctypes.c_float(float(eps)),
                    self._pair(pk["qk"][0]), self._ptr(pk["qk"][1]), self._pair(pk["v"][0]), self._ptr(pk["v"][1]),
                    None, None, None, None, None]
            pk["_k1_args"] = args
            pk["_k1_qpos"] = qpos.data_ptr()
        if pk["_k1_qpos"] != qpos.data_ptr():
            args[2] = self._ptr(qpos, self.f32, Q * E)
            pk["_k1_qpos"] = qpos.data_ptr()
        p0, step = out.data_ptr(), B * Q * E * 4
        args[0], args[1], args[3] = attn_out.data_ptr(), q_in.data_ptr(), B * Q
        args[15], args[16], args[17], args[18] = p0, p0 + step, p0 + 2 * step, p0 + 3 * step
        args[19] = self._stream()
        rc = self.lib.occf_decoder_rows_k1(*args)
        if rc != 0:
            raise OccfError(f"occf_decoder_rows_k1 failed with code {rc}")
        return out[0], out[1], out[2], out[3]

    def decoder_rows_k2(self, attn_out, q_in, qpos, pk, head, nxt, want_q=True):
        """``pk`` None: head only (q_in = the queries).  head: post_norm / cls / mask_embed pack; nxt: the NEXT layer's
        cross-attention query projection (pair, bias) or None.  -> (q3 or None, cls [B, Q, n_cls], mask_embed, qx or None)"""
        B, Q, E = q_in.shape
        dev = q_in.device
        n_cls = head["n_cls"]
        if self.strict and not (q_in.is_cuda and q_in.is_contiguous() and (attn_out is None or attn_out.is_contiguous())):
            raise OccfError("decoder_rows_k2: contiguous GPU tensors expected")
        q3 = torch.empty((B, Q, E), dtype=self.f32, device=dev) if (pk is not None and want_q) else None
        cls = torch.empty((B, Q, n_cls), dtype=self.f32, device=dev)
        me = torch.empty((B, Q, E), dtype=self.f32, device=dev)
        qx = torch.empty((B, Q, E), dtype=self.f32, device=dev) if nxt is not None else None
        holder = pk if pk is not None else head
        key = ("_k2_args", id(head), id(nxt), qpos.data_ptr())
        args = holder.get(key)
        if args is None:
            z = ctypes.c_void_p(0)
            lin = lambda e: [self._pair(e[0]), self._ptr(e[1])] if e is not None else [z, z]
            ln = lambda e: [self._ptr(e[0]), self._ptr(e[1]), ctypes.c_float(float(e[2]))] if e is not None else [z, z, ctypes.c_float(0.0)]
            H = pk["H"] if pk is not None else 32
            args = [1 if pk is not None else 0, None, None, self._ptr(qpos, self.f32, Q * E), 0, E, H, Q,
                    *lin(pk["out1"] if pk else None), *ln(pk["ln1"] if pk else None),
                    *lin(pk["ffn1"] if pk else None), *lin(pk["ffn2"] if pk else None), *ln(pk["ln2"] if pk else None),
                    *ln(head["post"]), *lin(head["cls"]), n_cls, *lin(head["me0"]), *lin(head["me1"]), *lin(head["me2"]),
                    *lin(nxt), None, None, None, None, None]
            holder[key] = args
        n = len(args)
        args[1] = attn_out.data_ptr() if pk is not None else None
        args[2], args[4] = q_in.data_ptr(), B * Q
        args[n - 5] = q3.data_ptr() if q3 is not None else None
        args[n - 4], args[n - 3] = cls.data_ptr(), me.data_ptr()
        args[n - 2] = qx.data_ptr() if qx is not None else None
        args[n - 1] = self._stream()
        rc = self.lib.occf_decoder_rows_k2(*args)
        if rc != 0:
            raise OccfError(f"occf_decoder_rows_k2 failed with code {rc}")
        return q3, cls, me, qx

    def upsample_classify(self, mask_pred, cls, occ_size):
        B, Q, X, Y, Z = mask_pred.shape
        K = cls.shape[-1] - 1
        X2, Y2, Z2 = (int(t) for t in occ_size)
        out = torch.empty((B, K, X2, Y2, Z2), dtype=mask_pred.dtype, device=mask_pred.device)
        ws = torch.empty((B * Q * 24,), dtype=self.f32, device=mask_pred.device)
        self._call("occf_upsample_classify_fwd", self._ptr(mask_pred, self.f32), self._ptr(cls, self.f32, B * Q * (K + 1)),
                   self._ptr(out), self._ptr(ws), B, Q, K, X, Y, Z, X2, Y2, Z2, self._stream())
        return out

    def lidarseg_sample(self, mask_pred, cls, pts):
        """pts [P, 4] = (batch, gx, gy, gz) -> class probabilities [P, K]."""
        B, Q, X, Y, Z = mask_pred.shape
        K = cls.shape[-1] - 1
        P = pts.shape[0]
        out = torch.empty((P, K), dtype=mask_pred.dtype, device=mask_pred.device)
        self._call("occf_lidarseg_sample_fwd", self._ptr(mask_pred, self.f32), self._ptr(cls, self.f32, B * Q * (K + 1)),
                   self._ptr(pts, self.f32, 4 * P), self._ptr(out), P, B, Q, K, X, Y, Z, self._stream())
        return out

    # ------------------------------------------------------------------ dense contractions
    def split_bf16(self, w):
        """fp32 tensor -> (hi, lo) bf16 bit patterns (int16 tensors of the same shape)."""
        w = w.contiguous()
        hi = torch.empty(w.shape, dtype=torch.int16, device=w.device)
        lo = torch.empty(w.shape, dtype=torch.int16, device=w.device)
        self._call("occf_split_bf16", self._ptr(w, self.f32), self._ptr(hi), self._ptr(lo), w.numel(),
                   self._stream())
        return hi, lo

    def prep_weights(self, table, n, total_pairs):
        """run a device-resident descriptor table of strided-gather + split jobs (csrc/prep.hip) in one launch"""
        self._call("occf_prep_weights", self._ptr(table, torch.int64), int(n), int(total_pairs), self._stream())

    def _splitk_workspace(self, M, N, K, device):
        need = self.lib.occf_gemm_bf16_workspace(M, N, K)
        if need <= 0:
            return None, 0
        return torch.empty((need,), dtype=self.f32, device=device), need

    def _bf16_terms(self, K, rows, w_split):
        if self.precision == "f32" or w_split is None or K % 32 != 0 or rows < 64:
            return 0
        return 3 if self.precision == "bf16x3" else 1

    def head_major_supported(self, M, N, K, dh):
        """can ``linear(..., head_major=(rows, dh))`` be used for this shape / precision mode"""
        return self.precision != "f32" and K % 32 == 0 and max(M, N) >= 64 and dh % 4 == 0 and N % dh == 0

    def _gn_finalize(self, partial, B, nblk, C, G, count, eps):
        stats = torch.empty((B, G, 2), dtype=self.f32, device=partial.device)
        self._call("occf_groupnorm_finalize", self._ptr(partial), self._ptr(stats), B, nblk, C, G, float(count),
                   float(eps), self._stream())
        return stats

    def linear(self, x, weight, bias=None, act=0, residual=None, out=None, w_split=None, allow_small=True,
               head_major=None, gn=None):
        """x [..., K] (rows contiguous along K) @ weight[N, K]^T -> [..., N].  ``w_split`` = the
        (hi, lo) bf16 split of ``weight`` enables the bf16 matrix-core path (see ``precision``).
        ``gn = (groups, eps, rows_per_batch)``: also leave the GroupNorm statistics of the output in
        ``self.last_gn_stats`` ([B, G, 2], from the GEMM epilogue's partial sums; None if the shape /
        precision mode does not allow it -- the caller then runs groupnorm_stats)."""
        self.last_gn_stats = None
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = x.reshape(-1, K)
        M = x2.shape[0]
        if out is None:
            out = torch.empty((M, N), dtype=x.dtype, device=x.device)
        r2 = residual.reshape(-1, N) if residual is not None else None
        if weight.numel() != N * K or (bias is not None and bias.numel() != N) or out.numel() != M * N or \
                (r2 is not None and r2.shape[0] != M):
            raise OccfError(f"linear: x [{M}, {K}], weight {tuple(weight.shape)}, bias / residual / out sizes disagree")
        self.last_flops = 2 * M * N * K
        for t in (x2, out, r2):
            if t is not None and (t.stride(1) != 1 or (self.strict and not t.is_cuda)):
                raise OccfError("linear: rows must be channel-contiguous GPU tensors")
        rp = ctypes.c_void_p(r2.data_ptr()) if r2 is not None else ctypes.c_void_p(0)
        if head_major is None and M * N <= 262144 and ((allow_small and M <= 128) or K % 4 != 0 or x2.stride(0) % 4 != 0):
            self._call("occf_linear_small_fwd", ctypes.c_void_p(x2.data_ptr()), self._ptr(weight, self.f32),
                       self._ptr(bias), rp, ctypes.c_void_p(out.data_ptr()), M, N, K, x2.stride(0),
                       out.stride(0), r2.stride(0) if r2 is not None else 0, int(act), self._stream())
            return out.view(*x.shape[:-1], N)
        terms = self._bf16_terms(K, max(M, N), w_split)
        if head_major is not None:
            # head_major = (rows per batch, head_dim): out[b, h, q, d] written by the GEMM epilogue
            rows, dh = head_major
            if not terms or residual is not None:
                raise OccfError("head-major output needs the bf16 GEMM path and no residual")
            self._call("occf_linear_bf16_fwd", ctypes.c_void_p(x2.data_ptr()), self._ptr(w_split[0]),
                       self._ptr(w_split[1]), self._ptr(bias), ctypes.c_void_p(0), ctypes.c_void_p(out.data_ptr()),
                       M, N, K, x2.stride(0), N, 0, int(act), terms, ctypes.c_void_p(0), 0, int(dh), int(rows),
                       ctypes.c_void_p(0), self._stream())
            return out.view(M // rows, N // dh, rows, dh)
        if terms:
            if gn is not None and residual is None and act == 0 and self.lib.occf_gemm_bf16_workspace(M, N, K) == 0:
                G, eps, rows = gn
                if (rows % 128 == 0 or M == rows) and M % rows == 0 and N % G == 0:
                    nblk = (rows + 127) // 128
                    part = torch.empty((M // rows * nblk * N * 2,), dtype=self.f32, device=x.device)
                    rc = self.lib.occf_linear_bf16_fwd(
                        ctypes.c_void_p(x2.data_ptr()), self._ptr(w_split[0]), self._ptr(w_split[1]),
                        self._ptr(bias), rp, ctypes.c_void_p(out.data_ptr()), M, N, K, x2.stride(0), out.stride(0),
                        0, int(act), terms, ctypes.c_void_p(0), 0, 0, 0, self._ptr(part), self._stream())
                    if rc == 0:
                        self.last_gn_stats = self._gn_finalize(part, M // rows, nblk, N, G, rows * (N // G), eps)
                        return out.view(*x.shape[:-1], N)
                    if rc != -2:
                        raise OccfError(f"occf_linear_bf16_fwd failed with code {rc}")
            ws, nws = self._splitk_workspace(M, N, K, x.device)
            self._call("occf_linear_bf16_fwd", ctypes.c_void_p(x2.data_ptr()), self._ptr(w_split[0]),
                       self._ptr(w_split[1]), self._ptr(bias), rp, ctypes.c_void_p(out.data_ptr()), M, N, K,
                       x2.stride(0), out.stride(0), r2.stride(0) if r2 is not None else 0, int(act), terms,
                       self._ptr(ws), nws, 0, 0, ctypes.c_void_p(0), self._stream())
        else:
            self._call("occf_linear_fwd", ctypes.c_void_p(x2.data_ptr()), self._ptr(weight, self.f32),
                       self._ptr(bias), rp, ctypes.c_void_p(out.data_ptr()), M, N, K, x2.stride(0),
                       out.stride(0), r2.stride(0) if r2 is not None else 0, int(act), self._stream())
        return out.view(*x.shape[:-1], N)

    def linear_stream(self, x, w_split, bias=None, act=0, residual=None, aux=None, pre_out=False, row_scale=None,
                      XY=1, S=1):
        """csrc/gemm_stream.h with the training graph's epilogues (occf_linear_stream_fwd): x [M, K] (rows contiguous)
        -> out [M, N], or (out, pre) with ``pre_out``; ``row_scale`` [samples] with the token-buffer geometry (XY, S);
        ``act = 3``: out = (x W^T + b) * GELU'(aux).  None when the shape is outside the kernel's envelope (the caller
        then runs the unfused sequence)."""
        M, K = x.shape
        N = w_split[0].shape[0]
        if x.stride(1) != 1 or w_split[0].numel() != N * K or self.precision == "f32":
            return None
        if self.strict and not x.is_cuda:
            raise OccfError("occformer_amd ops need GPU tensors (no CPU path exists)")
        side = aux if act == 3 else residual
        if side is not None and (tuple(side.shape) != (M, N) or side.stride(1) != 1):
            raise OccfError("linear_stream: residual / aux must be [M, N] with unit column stride")
        if row_scale is not None and (M % (int(XY) * int(S)) or row_scale.numel() != M // int(XY)):
            raise OccfError(f"linear_stream: row_scale needs one entry per (sample, slice): {M} rows, XY = {XY}, S = {S}, "
                            f"{row_scale.numel()} entries")
        out = torch.empty((M, N), dtype=self.f32, device=x.device)
        pre = torch.empty((M, N), dtype=self.f32, device=x.device) if pre_out else None
        rc = self.lib.occf_linear_stream_fwd(
            ctypes.c_void_p(x.data_ptr()), self._ptr(w_split[0]), self._ptr(w_split[1]), self._ptr(bias, self.f32, N),
            ctypes.c_void_p(side.data_ptr() if side is not None else 0), self._ptr(out), self._ptr(pre),
            self._ptr(row_scale, self.f32), M, N, K, x.stride(0), N, side.stride(0) if side is not None else 0, int(act),
            1 if self.precision == "bf16" else 3, int(XY) * int(S), int(S), self._stream())
        if rc == -2:
            return None
        if rc != 0:
            raise OccfError(f"occf_linear_stream_fwd failed with code {rc}")
        self.last_flops = 2 * M * N * K
        return (out, pre) if pre_out else out

    def _halo_fragments(self, w_split, Cin, Cout):
        """the pre-split weight in MFMA-fragment order for the halo kernel (cached ON the split tensor, which is
        itself cached per weight version); None when switched off (OCCF_HALO_FRAG=0) or not packable"""
        if not self.halo_frag or w_split is None:
            return None
        hi, lo = w_split
        pk = getattr(hi, "_occf_halo_pack", None)
        if pk is None:
            n = self.lib.occf_conv3x3x3_halo_pack_elems(Cin, Cout)
            if n <= 0 or hi.numel() != n:
                return None
            fh, fl = torch.empty_like(hi), torch.empty_like(lo)
            self._call("occf_conv3x3x3_halo_pack", self._ptr(hi), self._ptr(lo), self._ptr(fh), self._ptr(fl), Cin, Cout,
                       self._stream())
            pk = (fh, fl)
            hi._occf_halo_pack = pk
        return pk

    def _wino_fragments(self, weight_tap, w_split, Cin, Cout, f16=False):
        """the x-transformed filters U (hi, lo) in MFMA-fragment order for the Winograd kernel (csrc/conv_wino.hip),
        cached ON the split tensor like the direct kernel's fragments; None when switched off (``use_wino``, OCCF_WINO=0)
        or outside the envelope.  ``f16``: fp16 (hi, lo) halves for the two-product form"""
        if not self.use_wino or w_split is None or weight_tap.dtype != self.f32 or not weight_tap.is_contiguous():
            return None
        hi = w_split[0]
        attr = "_occf_wino_pack_f16" if f16 else "_occf_wino_pack"
        pk = getattr(hi, attr, None)
        if pk is None:
            n = self.lib.occf_conv3x3x3_wino_pack_elems(Cin, Cout)
            if n <= 0:
                return None
            fh = torch.empty((n,), dtype=hi.dtype, device=hi.device)
            fl = torch.empty((n,), dtype=hi.dtype, device=hi.device)
            self._call("occf_conv3x3x3_wino_pack", self._ptr(weight_tap, self.f32), self._ptr(fh), self._ptr(fl), Cin,
                       Cout, int(f16), self._stream())
            pk = (fh, fl)
            setattr(hi, attr, pk)
        return pk

    def absmax_slot(self, x2):
        """scale slot of a contiguous [rows, cols] tensor for the two-product fp16 kernels: slot[0] = the bit pattern of
        max |x| (csrc/occf_absmax.h: two launches, no atomics, no host synchronisation)"""
        slot = torch.empty((self.lib.occf_absmax_slot_words(),), dtype=self.i32, device=x2.device)
        self._call("occf_absmax_f32", self._ptr(x2, self.f32), x2.shape[0], x2.shape[1], x2.stride(0), self._ptr(slot),
                   self._stream())
        return slot

    def conv3d(self, x_cl, weight_tap, ksize, stride=1, dil=1, pad=None, bias=None, act=0, residual=None,
               w_split=None, gn=None, act_f16=False):
        """x_cl [B, Xi, Yi, Zi, Cin] (any strides with unit channel stride) -> [B, Xo, Yo, Zo, Cout].
        ``gn = (groups, eps)``: GroupNorm statistics of the output -> ``self.last_gn_stats`` (see linear).
        ``act_f16``: the caller allows x_cl to enter as ONE fp16 piece (two products per product, csrc/conv_wino.hip's
        F16 variant) -- the data gradients of the stride-1 3^3 convolutions; ignored where no such kernel applies."""
        self.last_gn_stats = None
        B, Xi, Yi, Zi, Cin = x_cl.shape
        assert x_cl.stride(4) == 1
        kX, kY, kZ = ksize
        if pad is None:
            pad = tuple(dil * (k - 1) // 2 for k in ksize)
        Cout = weight_tap.shape[0]
        Xo = (Xi + 2 * pad[0] - dil * (kX - 1) - 1) // stride + 1
        Yo = (Yi + 2 * pad[1] - dil * (kY - 1) - 1) // stride + 1
        Zo = (Zi + 2 * pad[2] - dil * (kZ - 1) - 1) // stride + 1
        out = torch.empty((B, Xo, Yo, Zo, Cout), dtype=x_cl.dtype, device=x_cl.device)
        if weight_tap.numel() != Cout * kX * kY * kZ * Cin or (bias is not None and bias.numel() != Cout) or \
                (residual is not None and residual.numel() != out.numel()):
            raise OccfError(f"conv3d: x {tuple(x_cl.shape)}, weight {tuple(weight_tap.shape)} for taps {tuple(ksize)}, "
                            "bias / residual sizes disagree")
        self.last_flops = 2 * B * Xo * Yo * Zo * Cout * kX * kY * kZ * Cin
        if self.strict and not x_cl.is_cuda:
            raise OccfError("occformer_amd ops need GPU tensors (no CPU path exists)")
        geom = (B, Xi, Yi, Zi, Cin, Cout, kX, kY, kZ, int(stride), int(dil), pad[0], pad[1], pad[2],
                x_cl.stride(0), x_cl.stride(1), x_cl.stride(2), x_cl.stride(3), int(act))
        terms = self._bf16_terms(Cin, B * Xo * Yo * Zo, w_split)
        want_gn = gn is not None and residual is None and act == 0
        if terms and self.use_halo_conv and (kX, kY, kZ) == (3, 3, 3) and stride == 1 and dil == 1 \
                and tuple(pad) == (1, 1, 1):
            halo_args = (ctypes.c_void_p(x_cl.data_ptr()), self._ptr(w_split[0]), self._ptr(w_split[1]),
                         self._ptr(bias), self._ptr(residual), self._ptr(out), B, Xi, Yi, Zi, Cin, Cout,
                         x_cl.stride(0), x_cl.stride(1), x_cl.stride(2), x_cl.stride(3), int(act), terms)
            # Winograd F(2, 3) along x first (3-term mode only): 2/3 of the matrix-core products
            f16 = bool(act_f16) and terms == 3 and x_cl.is_contiguous()
            wino = self._wino_fragments(weight_tap, w_split, Cin, Cout, f16) if terms == 3 else None
            if wino is not None:
                # (one-product form: the lo fragments are not passed)
                w_lo = ctypes.c_void_p(0) if (f16 and self.dgrad_f16_single) else self._ptr(wino[1])
                wargs = (ctypes.c_void_p(x_cl.data_ptr()), self._ptr(wino[0]), w_lo, self._ptr(bias),
                         self._ptr(residual), self._ptr(out), B, Xi, Yi, Zi, Cin, Cout, x_cl.stride(0), x_cl.stride(1),
                         x_cl.stride(2), x_cl.stride(3), int(act))
                # (slot_t stays referenced until the launches below are queued: a buffer allocated in between -- the
                # GroupNorm partials, which the same kernel WRITES -- must not land on its memory)
                slot_t = self.absmax_slot(x_cl.view(-1, Cin)) if f16 else None
                slot = self._ptr(slot_t) if f16 else ctypes.c_void_p(0)
                rc = -2
                if want_gn:
                    nblk = self.lib.occf_conv3x3x3_wino_gn_blocks(Xi, Yi, Zi)
                    if nblk > 0 and Cout % gn[0] == 0:
                        part = torch.empty((B * nblk * Cout * 2,), dtype=self.f32, device=x_cl.device)
                        rc = self.lib.occf_conv3x3x3_wino_fwd(*wargs, self._ptr(part), slot, self._stream())
                        if rc == 0:
                            self.last_gn_stats = self._gn_finalize(part, B, nblk, Cout, gn[0],
                                                                   Xo * Yo * Zo * (Cout // gn[0]), gn[1])
                            return out
                if rc == -2:
                    rc = self.lib.occf_conv3x3x3_wino_fwd(*wargs, ctypes.c_void_p(0), slot, self._stream())
                if rc == 0:
                    return out
                if rc != -2:
                    raise OccfError(f"occf_conv3x3x3_wino_fwd failed with code {rc}")
            frag = self._halo_fragments(w_split, Cin, Cout)
            frag_args = (self._ptr(frag[0]), self._ptr(frag[1])) if frag else (ctypes.c_void_p(0), ctypes.c_void_p(0))
            rc = -2
            if want_gn:
                nblk = self.lib.occf_conv3x3x3_halo_gn_blocks(Xi, Yi, Zi)
                if nblk > 0 and Cout % gn[0] == 0:
                    part = torch.empty((B * nblk * Cout * 2,), dtype=self.f32, device=x_cl.device)
                    rc = self.lib.occf_conv3x3x3_halo_fwd(*halo_args, self._ptr(part), *frag_args, self._stream())
                    if rc == 0:
                        self.last_gn_stats = self._gn_finalize(part, B, nblk, Cout, gn[0],
                                                               Xo * Yo * Zo * (Cout // gn[0]), gn[1])
                        return out
            if rc == -2:
                rc = self.lib.occf_conv3x3x3_halo_fwd(*halo_args, ctypes.c_void_p(0), *frag_args, self._stream())
            if rc == 0:
                return out
            if rc != -2:                     # -2 = shape outside the halo kernel's envelope
                raise OccfError(f"occf_conv3x3x3_halo_fwd failed with code {rc}")
        if terms:
            V = Xo * Yo * Zo
            # (shapes that profit from split-K keep it; their statistics come from the separate pass)
            if want_gn and (V % 128 == 0 or B == 1) and Cout % gn[0] == 0 and \
                    self.lib.occf_gemm_bf16_workspace(B * V, Cout, kX * kY * kZ * Cin) == 0:
                nblk = (V + 127) // 128
                part = torch.empty((B * nblk * Cout * 2,), dtype=self.f32, device=x_cl.device)
                rc = self.lib.occf_conv3d_bf16_fwd(
                    ctypes.c_void_p(x_cl.data_ptr()), self._ptr(w_split[0]), self._ptr(w_split[1]), self._ptr(bias),
                    self._ptr(residual), self._ptr(out), *geom, terms, ctypes.c_void_p(0), 0, self._ptr(part),
                    self._stream())
                if rc == 0:
                    self.last_gn_stats = self._gn_finalize(part, B, nblk, Cout, gn[0], V * (Cout // gn[0]), gn[1])
                    return out
                if rc != -2:
                    raise OccfError(f"occf_conv3d_bf16_fwd failed with code {rc}")
            ws, nws = self._splitk_workspace(B * Xo * Yo * Zo, Cout, kX * kY * kZ * Cin, x_cl.device)
            self._call("occf_conv3d_bf16_fwd", ctypes.c_void_p(x_cl.data_ptr()), self._ptr(w_split[0]),
                       self._ptr(w_split[1]), self._ptr(bias), self._ptr(residual), self._ptr(out), *geom, terms,
                       self._ptr(ws), nws, ctypes.c_void_p(0), self._stream())
        else:
            self._call("occf_conv3d_fwd", ctypes.c_void_p(x_cl.data_ptr()), self._ptr(weight_tap, self.f32),
                       self._ptr(bias), self._ptr(residual), self._ptr(out), *geom, self._stream())
        return out

    def mlp_fused_supported(self, C, H):
        return self.use_fused_mlp and self.precision != "f32" and C in (128, 192, 256) and H % 128 == 0

    def mlp_fused(self, x, ln_w, ln_b, w1_split, b1, w2_split, b2, act, ln_mode, eps=1e-5):
        """out = LNpost?(x + W2.act(W1.LNpre?(x) + b1) + b2); x [..., C] contiguous."""
        C = x.shape[-1]
        H = w1_split[0].shape[0]
        out = torch.empty_like(x)
        self.last_flops = 4 * (x.numel() // C) * C * H
        self._call("occf_mlp_fused_fwd", self._ptr(x, self.f32), self._ptr(ln_w), self._ptr(ln_b),
                   self._ptr(w1_split[0]), self._ptr(w1_split[1]), self._ptr(b1), self._ptr(w2_split[0]),
                   self._ptr(w2_split[1]), self._ptr(b2), self._ptr(out), x.numel() // C, C, H, int(act),
                   int(ln_mode), float(eps), 3 if self.precision == "bf16x3" else 1, self._stream())
        return out

    # ------------------------------------------------------------------ norms / fusion
    def groupnorm_stats(self, x_cl, groups, eps=1e-5):
        """x_cl [B, ..., C] contiguous channels-last -> stats [B, G, 2]."""
        B, C = x_cl.shape[0], x_cl.shape[-1]
        V = x_cl.numel() // (B * C)
        need = self.lib.occf_groupnorm_workspace(B, V, C, groups)
        ws = torch.empty((need,), dtype=self.f32, device=x_cl.device)
        stats = torch.empty((B, groups, 2), dtype=self.f32, device=x_cl.device)
        self._call("occf_groupnorm_stats", self._ptr(x_cl, self.f32), self._ptr(stats), self._ptr(ws), B, V, C,
                   groups, float(eps), self._stream())
        return stats

    def groupnorm_apply(self, x_cl, stats, gamma, beta, groups, relu=False, tokens=False, residual=None):
        """x_cl [B, P..., Z, C]; tokens=True appends the z-mean slot -> [B, P..., Z+1, C]."""
        B, C, Z = x_cl.shape[0], x_cl.shape[-1], x_cl.shape[-2]
        P = x_cl.numel() // (B * C * Z)
        shape = list(x_cl.shape)
        if tokens:
            shape[-2] = Z + 1
        out = torch.empty(shape, dtype=x_cl.dtype, device=x_cl.device)
        self._call("occf_groupnorm_apply", self._ptr(x_cl, self.f32), self._ptr(stats, self.f32, B * groups * 2),
                   self._ptr(gamma, self.f32, C), self._ptr(beta, self.f32, C),
                   self._ptr(residual, None, None if residual is None else x_cl.numel()), self._ptr(out),
                   B, P, Z, C, groups, int(relu), int(tokens), self._stream())
        return out

    def layernorm(self, x, gamma, beta, eps=1e-5):
        C = x.shape[-1]
        out = torch.empty_like(x)
        self._call("occf_layernorm_fwd", self._ptr(x, self.f32), self._ptr(gamma, self.f32, C),
                   self._ptr(beta, self.f32, C), self._ptr(out), x.numel() // C, C, float(eps), self._stream())
        return out

    def dualpath_combine(self, tokens, bev, w, b, identity):
        """tokens [B, X, Y, Z+1, C], bev [B, X, Y, C], identity [B, X, Y, Z, C] -> [B, X, Y, Z, C]."""
        B, X, Y, Zs, C = tokens.shape
        out = torch.empty((B, X, Y, Zs - 1, C), dtype=tokens.dtype, device=tokens.device)
        self._call("occf_dualpath_combine", self._ptr(tokens, self.f32), self._ptr(bev, self.f32, B * X * Y * C),
                   self._ptr(w, self.f32, C), self._ptr(b), self._ptr(identity, self.f32, out.numel()), self._ptr(out),
                   B * X * Y, Zs - 1, C, self._stream())
        return out

    def upsample_add(self, coarse, lateral):
        """coarse [B, X, Y, Z, C] trilinearly resized (align_corners=False) onto lateral [B, X2, Y2, Z2, C]."""
        B, X, Y, Z, C = coarse.shape
        _, X2, Y2, Z2, C2 = lateral.shape
        if C2 != C or lateral.shape[0] != B:
            raise OccfError(f"upsample_add: coarse {tuple(coarse.shape)} vs lateral {tuple(lateral.shape)}")
        out = torch.empty_like(lateral)
        self._call("occf_upsample_add", self._ptr(coarse, self.f32), self._ptr(lateral, self.f32), self._ptr(out),
                   B, X, Y, Z, X2, Y2, Z2, C, self._stream())
        return out

    def scale_shift_act(self, y, scale, shift, residual=None, relu=True):
        """IN PLACE on a channels_last [N, C, H, W] (or row-major [..., C]) feature map, fp32 or bf16:
        y = act(y * scale[c] + shift[c] (+ residual)) -- eval BatchNorm + identity add + ReLU of the image branch"""
        cl = y.dim() == 4 and y.is_contiguous(memory_format=torch.channels_last)
        C = y.shape[1] if cl else y.shape[-1]
        if not (cl or y.is_contiguous()) or y.dtype not in (torch.float32, torch.bfloat16):
            raise OccfError("scale_shift_act: a channels_last / row-major fp32 or bf16 tensor expected")
        if residual is not None and (residual.shape != y.shape or residual.dtype != y.dtype or residual.stride() != y.stride()):
            raise OccfError("scale_shift_act: the residual must have the layout of y")
        if self.strict and not y.is_cuda:
            raise OccfError("occformer_amd ops need GPU tensors (no CPU path exists)")
        self._call("occf_scale_shift_act", ctypes.c_void_p(y.data_ptr()), self._ptr(scale, self.f32, C),
                   self._ptr(shift, self.f32, C), ctypes.c_void_p(residual.data_ptr() if residual is not None else 0),
                   y.numel() // C, C, int(bool(relu)), int(y.dtype == torch.bfloat16), self._stream())
        return y

    def deform_im2col(self, x_cl, offset, K, stride, pad, dil, groups, deform_groups, mask=None):
        """x_cl [BN, H, W, C], offset [BN, dg*2*K*K, Ho, Wo] (, mask [BN, dg*K*K, Ho, Wo]: DCNv2)
        -> col [BN*Ho*Wo, groups, K*K, C/groups]."""
        BN, H, W, C = x_cl.shape
        Ho, Wo = offset.shape[-2:]
        col = torch.empty((BN * Ho * Wo, groups, K * K, C // groups), dtype=x_cl.dtype, device=x_cl.device)
        if mask is None:
            self._call("occf_deform_im2col", self._ptr(x_cl, self.f32), self._ptr(offset, self.f32), self._ptr(col),
                       BN, H, W, C, K, stride, pad, dil, groups, deform_groups, self._stream())
        else:
            self._call("occf_modulated_deform_im2col", self._ptr(x_cl, self.f32), self._ptr(offset, self.f32),
                       self._ptr(mask, self.f32), self._ptr(col), BN, H, W, C, K, stride, pad, dil, groups,
                       deform_groups, self._stream())
        return col

    # ------------------------------------------------------------------ training-time sampling
    def point_sample_3d(self, vol, pts, align_corners=False, padding_mode="zeros"):
        """vol [N, C, X, Y, Z]; pts [N, P, 3] (or [1, P, 3] shared by all N) in [0, 1], last dim in
        grid_sample order (z, y, x) -> [N, C, P]."""
        N, C, X, Y, Z = vol.shape
        P = pts.shape[1]
        shared = pts.shape[0] == 1 and N > 1
        out = torch.empty((N, C, P), dtype=vol.dtype, device=vol.device)
        self._call("occf_point_sample_3d_fwd", self._ptr(vol, self.f32), self._ptr(pts, self.f32), self._ptr(out),
                   N, C, X, Y, Z, P, int(shared), int(align_corners), int(padding_mode == "border"),
                   self._stream())
        return out

    def point_sample_3d_rows(self, vol, rows, pts, align_corners=False, padding_mode="zeros"):
        """vol [R, X, Y, Z] (contiguous); rows int64 [N] in [0, R); pts [N, P, 3] (or [1, P, 3] shared) -> [N, P]:
        point_sample_3d(vol[rows].unsqueeze(1), pts)[:, 0] without materialising vol[rows]"""
        R, X, Y, Z = vol.shape
        N, P = rows.shape[0], pts.shape[1]
        shared = pts.shape[0] == 1 and N > 1
        if not vol.is_contiguous() or rows.dtype != torch.int64 or not (shared or pts.shape[0] == N):
            raise OccfError("point_sample_3d_rows: contiguous [R, X, Y, Z] volume, int64 rows, [N | 1, P, 3] points")
        out = torch.empty((N, P), dtype=vol.dtype, device=vol.device)
        if N == 0:
            return out
        self._call("occf_point_sample_3d_rows_fwd", self._ptr(vol, self.f32), self._ptr(rows.contiguous(), torch.int64),
                   self._ptr(pts.contiguous(), self.f32), self._ptr(out), N, X, Y, Z, P, int(shared),
                   int(align_corners), int(padding_mode == "border"), self._stream())
        return out

    def point_sample_tokens(self, tok, dims, pts, align_corners=False, padding_mode="zeros"):
        """tok [V = X*Y*Z, C] channels-last volume (unit column stride); pts [P, 3] in [0, 1] (grid_sample order)
        -> [P, C]"""
        X, Y, Z = dims
        V, C = tok.shape
        if V != X * Y * Z or tok.stride(1) != 1:
            raise OccfError("point_sample_tokens: tok must be [X*Y*Z, C] with unit column stride")
        P = pts.shape[0]
        out = torch.empty((P, C), dtype=tok.dtype, device=tok.device)
        self._ptr(pts, self.f32)
        self._call("occf_point_sample_tokens_fwd", ctypes.c_void_p(tok.data_ptr()), self._ptr(pts, self.f32),
                   self._ptr(out), X, Y, Z, C, tok.stride(0), P, int(align_corners), int(padding_mode == "border"),
                   self._stream())
        return out

    def sample_without_replacement(self, weights, uniforms, k, exponential=False):
        """weights [R, V] or [1, V]; uniforms [R, V] in (0, 1] (or Exp(1) draws with exponential=True)
        -> int64 indices [R, k] (unordered)."""
        R, V = uniforms.shape
        need = self.lib.occf_sample_wor_workspace(R, V)
        ws = torch.empty((need,), dtype=self.f32, device=uniforms.device)
        out = torch.empty((R, k), dtype=torch.int64, device=uniforms.device)
        self._call("occf_sample_wor_fwd", self._ptr(weights, self.f32), self._ptr(uniforms, self.f32),
                   self._ptr(out), self._ptr(ws), R, V, int(k), int(weights.shape[0] == 1 and R > 1),
                   int(exponential), self._stream())
        # (the compaction hands out slots in arrival order: the SET is reproducible, the order is not -- sorted in the
        # reproducible mode so that everything summed over these indices adds up in the same order)
        return out.sort(dim=1).values if self.deterministic else out

    def topk_smallest_abs(self, values, k):
        """values [R, V] -> int64 indices [R, k] of the k smallest |values| per row (unordered)."""
        R, V = values.shape
        ws = torch.empty((self.lib.occf_sample_wor_workspace(R, V),), dtype=self.f32, device=values.device)
        out = torch.empty((R, k), dtype=torch.int64, device=values.device)
        self._call("occf_topk_smallest_abs_fwd", self._ptr(values, self.f32), self._ptr(out), self._ptr(ws), R, V,
                   int(k), self._stream())
        return out.sort(dim=1).values if self.deterministic else out

    def point_loss_rows(self, logits, targets):
        """logits/targets [R, P] -> [R, 4] = sums of {BCE, sigmoid*t, sigmoid, t}."""
        R, P = logits.shape
        out = torch.empty((R, 4), dtype=self.f32, device=logits.device)
        self._call("occf_point_loss_rows_fwd", self._ptr(logits, self.f32), self._ptr(targets, self.f32),
                   self._ptr(out), R, P, self._stream())
        return out

    def hungarian(self, cost):
        """cost [Q, G] or [P, Q, G] fp32 -> (match_gt [.., G] int32: query of GT g, assigned_gt [.., Q] int32:
        0 = background, g + 1 = matched to GT g) -- scipy.optimize.linear_sum_assignment on the device."""
        c = cost.detach().to(self.f32).contiguous()
        lead = c.shape[:-2]
        Q, G = c.shape[-2:]
        P = 1
        for d in lead:
            P *= int(d)
        match = torch.empty((*lead, G), dtype=torch.int32, device=c.device)
        assigned = torch.empty((*lead, Q), dtype=torch.int32, device=c.device)
        self._call("occf_hungarian_fwd", self._ptr(c, self.f32), self._ptr(match), self._ptr(assigned), P, Q, G,
                   self._stream())
        return match, assigned

    # ------------------------------------------------------------------ input pipeline / evaluation (csrc/pipeline.hip)
    def lidar_depth(self, points, cam, N, H, W, kitti):
        """points [P, >=3] fp32 (rows may be wider: x, y, z, ...), cam [N, 36] packed camera constants
        (occformer_amd.pipeline.pack_depth_cameras) -> gt_depths [N, H, W]: nearest valid LiDAR depth per pixel"""
        pts = points if points.stride(-1) == 1 and points.dtype == self.f32 else points.to(self.f32).contiguous()
        if pts.dim() != 2 or pts.shape[1] < 3 or pts.stride(1) != 1:
            raise OccfError("lidar_depth: points [P, >=3] with unit column stride")
        P = pts.shape[0]
        out = torch.empty((N, H, W), dtype=self.f32, device=pts.device)
        ws = torch.empty((N * H * W,), dtype=torch.int32, device=pts.device)
        if not pts.is_contiguous():                      # a column block of a wider matrix: pass its row stride
            base, ld = pts, pts.stride(0)
            ptr = ctypes.c_void_p(base.data_ptr())
            if self.strict and not base.is_cuda:
                raise OccfError("occformer_amd ops need GPU tensors (no CPU path exists)")
        else:
            ptr, ld = self._ptr(pts, self.f32), pts.shape[1]
        self._call("occf_lidar_depth_fwd", ptr, int(ld), self._ptr(cam.contiguous(), self.f32), self._ptr(out),
                   self._ptr(ws), P, int(N), int(H), int(W), int(bool(kitti)), self._stream())
        return out

    def ssc_confusion(self, counts, target, pred=None, scores=None, nonempty=None, nonsurface=None, num_classes=None):
        """accumulate into ``counts`` [C*C + 3] int64 (conf[t, p] then completion tp, fp, fn); target uint8 [B, ...],
        pred int64 labels of the same shape OR scores [B, C, ...] fp32 (arg-max over C inside the kernel)"""
        C = int(num_classes if num_classes is not None else scores.shape[1])
        B = target.shape[0]
        V = target.numel() // B
        as_u8 = lambda t: None if t is None else (t if t.dtype == torch.uint8 else t.to(torch.uint8)).contiguous()
        # (the converted copies must outlive the launch: keep them in locals, not as temporaries of the call expression)
        tg, ne, ns = as_u8(target), as_u8(nonempty), as_u8(nonsurface)
        pr = None if pred is None else pred.to(torch.int64).contiguous()
        sc = None if scores is None else scores.contiguous()
        self._call("occf_ssc_confusion_fwd", self._ptr(pr, torch.int64), self._ptr(sc, self.f32), self._ptr(tg),
                   self._ptr(ne), self._ptr(ns), self._ptr(counts, torch.int64), int(B), int(V), C, self._stream())
        return counts

    # ------------------------------------------------------------------ backward kernels (training step)
    def _ws(self, n, device):
        return torch.empty((max(int(n), 1),), dtype=self.f32, device=device)

    def colsum(self, x2):
        """x2 [M, N] (unit column stride) -> [N] column sums (bias gradients)"""
        M, N = x2.shape
        out = torch.empty((N,), dtype=self.f32, device=x2.device)
        ws = self._ws(self.lib.occf_colsum_workspace(M, N), x2.device)
        if x2.stride(1) != 1:
            raise OccfError("colsum: rows must be channel-contiguous")
        self._call("occf_colsum", ctypes.c_void_p(x2.data_ptr()), self._ptr(out), self._ptr(ws), M, N, x2.stride(0),
                   self._stream())
        return out

    def layernorm_backward(self, x, gamma, dy, eps=1e-5, addend=None):
        """-> (dx [+ addend: the residual connection's gradient, same pass], dgamma, dbeta)"""
        C = x.shape[-1]
        M = x.numel() // C
        dx = torch.empty_like(x)
        dg = torch.empty((C,), dtype=self.f32, device=x.device)
        db = torch.empty((C,), dtype=self.f32, device=x.device)
        ws = self._ws(self.lib.occf_layernorm_bwd_workspace(M, C), x.device)
        if addend is not None and (tuple(addend.shape) != tuple(x.shape) or not addend.is_contiguous()):
            raise OccfError("layernorm_backward: addend must be a contiguous tensor of x's shape")
        self._call("occf_layernorm_bwd", self._ptr(x, self.f32), self._ptr(gamma, self.f32), self._ptr(dy, self.f32),
                   self._ptr(addend, self.f32), self._ptr(dx), self._ptr(dg), self._ptr(db), self._ptr(ws), M, C,
                   float(eps), self._stream())
        return dx, dg, db

    def groupnorm_backward(self, x_cl, stats, gamma, beta, dy, groups, relu=False, tokens=False, want_residual=False):
        """backward of groupnorm_apply -> (dx, dgamma, dbeta, dresidual or None)"""
        B, C, Z = x_cl.shape[0], x_cl.shape[-1], x_cl.shape[-2]
        P = x_cl.numel() // (B * C * Z)
        dx = torch.empty_like(x_cl)
        dg = torch.empty((C,), dtype=self.f32, device=x_cl.device)
        db = torch.empty((C,), dtype=self.f32, device=x_cl.device)
        dres = torch.empty_like(x_cl) if want_residual else None
        ws = self._ws(self.lib.occf_groupnorm_bwd_workspace(B, P * Z, C, groups), x_cl.device)
        if dy.numel() != B * P * (Z + 1 if tokens else Z) * C:
            raise OccfError("groupnorm_backward: dy has the wrong size")
        self._call("occf_groupnorm_bwd", self._ptr(x_cl, self.f32), self._ptr(stats, self.f32),
                   self._ptr(gamma, self.f32), self._ptr(beta, self.f32), self._ptr(dy, self.f32), self._ptr(dx),
                   self._ptr(dg), self._ptr(db), self._ptr(dres), self._ptr(ws), B, P, Z, C, groups, int(relu),
                   int(tokens), self._stream())
        return dx, dg, db, dres

    def act_forward(self, x, act):
        y = torch.empty_like(x)
        self._call("occf_act_fwd", self._ptr(x, self.f32), self._ptr(y), x.numel(), int(act), self._stream())
        return y

    def act_backward(self, x, dy, act):
        dx = torch.empty_like(x)
        self._call("occf_act_bwd", self._ptr(x, self.f32), self._ptr(dy, self.f32), self._ptr(dx), x.numel(), int(act),
                   self._stream())
        return dx

    def droppath(self, identity, branch, scale, XY, S):
        """out = identity + branch * scale[b*S + s] for token rows ((b*XY + xy)*S + s); identity may be None"""
        C = branch.shape[-1]
        out = torch.empty_like(branch)
        self._call("occf_droppath", self._ptr(identity), self._ptr(branch, self.f32), self._ptr(scale, self.f32),
                   self._ptr(out), branch.numel() // C, C, int(XY), int(S), self._stream())
        return out

    def dualpath_combine_backward(self, tokens, bev, w, b, dout):
        B, X, Y, Zs, C = tokens.shape
        BP = B * X * Y
        dtok = torch.empty_like(tokens)
        dbev = torch.empty((B, X, Y, C), dtype=self.f32, device=tokens.device)
        dw = torch.empty((C,), dtype=self.f32, device=tokens.device)
        dbias = torch.empty((1,), dtype=self.f32, device=tokens.device)
        ws = self._ws(self.lib.occf_dualpath_combine_bwd_workspace(BP, C), tokens.device)
        self._call("occf_dualpath_combine_bwd", self._ptr(tokens, self.f32), self._ptr(bev, self.f32, BP * C),
                   self._ptr(w, self.f32, C), self._ptr(b), self._ptr(dout, self.f32, BP * (Zs - 1) * C),
                   self._ptr(dtok), self._ptr(dbev),
                   self._ptr(dw), self._ptr(dbias), self._ptr(ws), BP, Zs - 1, C, self._stream())
        return dtok, dbev, dw, dbias

    def upsample_add_backward(self, dout, coarse_shape):
        B, X, Y, Z, C = coarse_shape
        _, X2, Y2, Z2, C2 = dout.shape
        if C2 != C or dout.shape[0] != B:
            raise OccfError(f"upsample_add_backward: dout {tuple(dout.shape)} vs coarse {tuple(coarse_shape)}")
        dc = torch.empty(tuple(coarse_shape), dtype=self.f32, device=dout.device)
        self._call("occf_upsample_add_bwd", self._ptr(dout, self.f32), self._ptr(dc), B, X, Y, Z, X2, Y2, Z2, C,
                   self._stream())
        return dc

    def point_sample_3d_backward(self, dout, pts, vol_shape, align_corners=False, padding_mode="zeros",
                                 voxel_major_cols=0, out=None, col0=0):
        """-> dvol [N, C, X, Y, Z], or with ``voxel_major_cols`` = ld > 0 the voxel-major [X*Y*Z, ld] (column n*C+c,
        the columns beyond N*C stay zero).  ``out`` (voxel-major only): accumulate into columns col0 + n*C + c of an
        existing zero-initialised [X*Y*Z, ld] buffer (several calls share one buffer)."""
        N, C, X, Y, Z = vol_shape
        P = pts.shape[1]
        shared = pts.shape[0] == 1 and N > 1
        shape = (X * Y * Z, int(voxel_major_cols)) if voxel_major_cols else tuple(vol_shape)
        if self.deterministic and P > 0:
            # reproducible sums: the scatter in 64-bit fixed point into a buffer of this call's own columns, then floats
            ld = N * C if voxel_major_cols else 0
            acc = torch.zeros((X * Y * Z, ld) if ld else tuple(vol_shape), dtype=torch.int64, device=dout.device)
            slot = torch.empty((self.lib.occf_absmax_slot_words(),), dtype=self.i32, device=dout.device)
            self._call("occf_absmax_flat", self._ptr(dout, self.f32), dout.numel(), self._ptr(slot), self._stream())
            self._call("occf_point_sample_3d_bwd_fx", self._ptr(dout, self.f32), self._ptr(pts, self.f32), self._ptr(acc),
                       self._ptr(slot), N, C, X, Y, Z, P, int(shared), int(align_corners), int(padding_mode == "border"),
                       ld, self._stream())
            part = torch.empty(acc.shape, dtype=self.f32, device=dout.device)
            self._call("occf_fx_to_f32", self._ptr(acc), self._ptr(part), acc.numel(), self._ptr(slot), self._stream())
            if not voxel_major_cols:
                return part
            if out is None:
                out = torch.zeros(shape, dtype=self.f32, device=dout.device)
            elif tuple(out.shape) != shape or col0 < 0 or col0 + N * C > voxel_major_cols:
                raise OccfError("point_sample_3d_backward: out must be the voxel-major [X*Y*Z, ld] buffer")
            out[:, col0:col0 + N * C] = part
            return out
        if out is not None:
            if not voxel_major_cols or tuple(out.shape) != shape or col0 < 0 or col0 + N * C > voxel_major_cols:
                raise OccfError("point_sample_3d_backward: out must be the voxel-major [X*Y*Z, ld] buffer")
            self._ptr(out, self.f32)
            dvol = out
            self._call("occf_point_sample_3d_bwd", self._ptr(dout, self.f32), self._ptr(pts, self.f32),
                       ctypes.c_void_p(out.data_ptr() + 4 * int(col0)), N, C, X, Y, Z, P, int(shared),
                       int(align_corners), int(padding_mode == "border"), int(voxel_major_cols), self._stream())
            return dvol
        dvol = torch.zeros(shape, dtype=self.f32, device=dout.device)
        self._call("occf_point_sample_3d_bwd", self._ptr(dout, self.f32), self._ptr(pts, self.f32), self._ptr(dvol),
                   N, C, X, Y, Z, P, int(shared), int(align_corners), int(padding_mode == "border"),
                   int(voxel_major_cols), self._stream())
        return dvol

    def point_loss_rows_backward(self, logits, targets, grad_rows):
        R, P = logits.shape
        dx = torch.empty_like(logits)
        self._call("occf_point_loss_rows_bwd", self._ptr(logits, self.f32), self._ptr(targets, self.f32),
                   self._ptr(grad_rows, self.f32), self._ptr(dx), R, P, self._stream())
        return dx

    def _grad_terms(self):
        return 1 if self.precision == "bf16" else 3

    def _wgrad_terms(self):
        """weight-gradient contractions: 2 = two fp16-piece products per product (dy as ONE fp16 piece after a per-tensor
        power-of-two scale, x as fp16 (hi, lo)) -- weight gradients are leaf quantities, the 2^-12 rounding of dy averages
        over the contraction and propagates nowhere (whole gradient 1.2e-4 against 5.5e-5 with three bf16
        products).  ``OCCF_WGRAD_F16=0`` keeps the three-term bf16 split."""
        if self.precision == "bf16x3" and self.wgrad_f16:
            return 2
        return self._grad_terms()

    def linear_wgrad(self, dy2, x2, want_bias=True):
        """dy2 [M, N], x2 [M, K] (unit column strides) -> (dW [N, K], db [N] or None)"""
        M, N = dy2.shape
        K = x2.shape[1]
        if x2.shape[0] != M or dy2.stride(1) != 1 or x2.stride(1) != 1:
            raise OccfError("linear_wgrad: row-major operands with equal row counts expected")
        dw = torch.empty((N, K), dtype=self.f32, device=x2.device)
        db = torch.empty((N,), dtype=self.f32, device=x2.device) if want_bias else None
        need = self.lib.occf_linear_wgrad_workspace(M, N, K)
        ws = self._ws(need, x2.device)
        self.last_flops = 2 * M * N * K
        self._call("occf_linear_wgrad", ctypes.c_void_p(dy2.data_ptr()), ctypes.c_void_p(x2.data_ptr()),
                   self._ptr(dw), self._ptr(db), self._ptr(ws), need, M, N, K, dy2.stride(0), x2.stride(0),
                   self._wgrad_terms() if self.wgrad_f16_linear else self._grad_terms(), self._stream())
        return dw, db

    def conv3d_wgrad(self, dy, x_cl, ksize, stride=1, dil=1, pad=None, want_bias=False):
        """dy [B, Xo, Yo, Zo, Cout] contiguous, x_cl [B, Xi, Yi, Zi, Cin] (unit channel stride)
        -> (dW tap-major [Cout, taps*Cin], db or None)"""
        B, Xi, Yi, Zi, Cin = x_cl.shape
        Cout = dy.shape[-1]
        kX, kY, kZ = ksize
        if pad is None:
            pad = tuple(dil * (k - 1) // 2 for k in ksize)
        if (self.wgrad_2d_as_g8 and Zi == 1 and (kX, kY, kZ) == (3, 3, 1) and stride == 1 and dil == 1
                and tuple(pad) == (1, 1, 0) and Xi in (8, 16, 32, 64) and Cin % 64 == 0 and Cout % 64 == 0
                and B * Xi * Yi >= 1024 and self._wgrad_terms() == 2 and dy.shape[1:4] == x_cl.shape[1:4]):
            # a 2-D 3x3 convolution whose FIRST spatial extent is 8 / 16 / 32 / 64 (DepthNet's 16 x 44 feature maps): with
            # that axis moved innermost it is a [1, Y, X] volume with a (1, 3, 3) window -- the G8 kernel's shape (groups
            # of 8 consecutive rows along the innermost axis; csrc/wgrad_g8.h) instead of the register-transposing kernel
            # (VALU-bound: 0.18 ms per call against ~0.05 for two 8 MB transposes + the G8 launch).  Taps come back as
            # (dy, dx) and are swapped to (dx, dy).
            xp = x_cl.permute(0, 3, 2, 1, 4).contiguous()
            dp = dy.permute(0, 3, 2, 1, 4).contiguous()
            flops = 2 * dy.numel() * 9 * Cin
            dwp, _ = self._conv3d_wgrad(dp, xp, (1, 3, 3), 1, 1, (0, 1, 1), False)
            self.last_flops = flops
            db = dy.reshape(-1, Cout).sum(0) if want_bias else None
            return dwp.view(Cout, 3, 3, Cin).transpose(1, 2).reshape(Cout, 9 * Cin), db
        return self._conv3d_wgrad(dy, x_cl, ksize, stride, dil, pad, want_bias)

    def _conv3d_wgrad(self, dy, x_cl, ksize, stride, dil, pad, want_bias):
        B, Xi, Yi, Zi, Cin = x_cl.shape
        Cout = dy.shape[-1]
        kX, kY, kZ = ksize
        dw = torch.empty((Cout, kX * kY * kZ * Cin), dtype=self.f32, device=dy.device)
        db = torch.empty((Cout,), dtype=self.f32, device=dy.device) if want_bias else None
        geom = (B, Xi, Yi, Zi, Cin, Cout, kX, kY, kZ, int(stride), int(dil), pad[0], pad[1], pad[2])
        need = self.lib.occf_conv3d_wgrad_workspace(*geom)
        ws = self._ws(need, dy.device)
        self.last_flops = 2 * dy.numel() * kX * kY * kZ * Cin
        self._call("occf_conv3d_wgrad", self._ptr(dy, self.f32), ctypes.c_void_p(x_cl.data_ptr()), self._ptr(dw),
                   self._ptr(db), self._ptr(ws), need, *geom, x_cl.stride(0), x_cl.stride(1), x_cl.stride(2),
                   x_cl.stride(3), 4 if (self._wgrad_terms() == 2 and self.wgrad_f16_single) else self._wgrad_terms(),
                   self._stream())
        return dw, db

    def conv3d_dgrad(self, dy, wt_split, in_shape, ksize, stride=1, dil=1, pad=None):
        """dy [B, Xo, Yo, Zo, Cout] contiguous; wt_split = (hi, lo) of the weight as [Cin, taps*Cout]
        -> dx [B, Xi, Yi, Zi, Cin]"""
        B, Xi, Yi, Zi, Cin = in_shape
        Cout = dy.shape[-1]
        kX, kY, kZ = ksize
        if pad is None:
            pad = tuple(dil * (k - 1) // 2 for k in ksize)
        dx = torch.empty(tuple(in_shape), dtype=self.f32, device=dy.device)
        K = kX * kY * kZ * Cout
        ws, nws = self._splitk_workspace(B * Xi * Yi * Zi, Cin, K, dy.device)
        # (the multiply-adds of the forward convolution: every (output voxel, tap, channel pair) once -- NOT input voxels x
        # taps, which counts the structural zeros of a strided convolution's gradient)
        self.last_flops = 2 * dy.shape[0] * dy.shape[1] * dy.shape[2] * dy.shape[3] * Cin * K
        self._call("occf_conv3d_bf16_dgrad", self._ptr(dy, self.f32), self._ptr(wt_split[0]), self._ptr(wt_split[1]),
                   self._ptr(dx), B, Xi, Yi, Zi, Cin, Cout, kX, kY, kZ, int(stride), int(dil), pad[0], pad[1], pad[2],
                   self._grad_terms(), self._ptr(ws), nws, self._stream())
        return dx

    def window_attention_backward(self, qkv, qkv_bias, bias_table, attn_out, dout, B, X, Y, S, heads, shift):
        """-> (dqkv [n_tok, 3C], dqkv_bias_pad [3C] (padded-token part only), dbias_table [169, heads])"""
        C = qkv.shape[1] // 3
        dqkv = torch.empty_like(qkv)
        dpad = torch.zeros((3 * C,), dtype=self.f32, device=qkv.device)
        dtab = torch.empty_like(bias_table)
        ws = self._ws(self.lib.occf_window_attn_bwd_workspace(B, X, Y, S, heads), qkv.device)
        self._call("occf_window_attn_bwd", self._ptr(qkv, self.f32), self._ptr(qkv_bias, self.f32),
                   self._ptr(bias_table, self.f32), self._ptr(attn_out, self.f32), self._ptr(dout, self.f32),
                   self._ptr(dqkv), self._ptr(dpad), self._ptr(dtab), self._ptr(ws), B, X, Y, S, C, heads, int(shift),
                   self._stream())
        return dqkv, dpad, dtab

    def masked_attention_backward(self, q, k, v, heads, out, dout, blocked=None, row_open=None):
        B, Q, E = q.shape
        L = k.shape[1]
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        ws = self._ws(self.lib.occf_masked_xattn_bwd_workspace(B, Q, L, heads), q.device)
        self._call("occf_masked_xattn_bwd", self._ptr(q, self.f32), self._ptr(k, self.f32), self._ptr(v, self.f32),
                   self._ptr(blocked, torch.uint8), self._ptr(row_open, self.i32), self._ptr(out, self.f32),
                   self._ptr(dout, self.f32), self._ptr(dq), self._ptr(dk), self._ptr(dv), self._ptr(ws), B, Q, L, E,
                   heads, self._stream())
        return dq, dk, dv

    def msda3d_backward(self, value, offsets, logits, dout, level_shapes, heads, points, head_major=False, d_ol=None,
                        n_off=None):
        """-> (dvalue [B, Nq, E] token-major, doffsets, dlogits).  ``d_ol`` [B, Nq, n_off + n_logits]: write the two
        parameter-side gradients as column blocks of this one tensor (the fused offset/logit projection)."""
        if head_major:
            B, _, Nq, dh = value.shape
            E = heads * dh
        else:
            B, Nq, E = value.shape
        L = len(level_shapes)
        arr = (ctypes.c_int32 * (3 * L))(*[int(v) for s in level_shapes for v in s])
        dvalue = torch.zeros((B, Nq, E), dtype=self.f32, device=value.device)
        if d_ol is None:
            doff = torch.empty((B, Nq, heads * L * points * 3), dtype=self.f32, device=value.device)
            dlg = torch.empty((B, Nq, heads * L * points), dtype=self.f32, device=value.device)
        else:
            doff, dlg = d_ol[..., :n_off], d_ol[..., n_off:]
        need = self.lib.occf_msda3d_bwd_workspace(ctypes.cast(arr, ctypes.c_void_p), L, B, heads, E // heads)
        ws = self._ws(need, value.device) if need > 0 else None
        self._call("occf_msda3d_bwd", self._ptr(value, self.f32), ctypes.c_void_p(offsets.data_ptr()),
                   ctypes.c_void_p(logits.data_ptr()), self._ptr(dout, self.f32), self._ptr(dvalue),
                   ctypes.c_void_p(doff.data_ptr()), ctypes.c_void_p(dlg.data_ptr()), ctypes.cast(arr, ctypes.c_void_p),
                   L, B, Nq, heads, E // heads, points, int(head_major), offsets.stride(1), logits.stride(1),
                   doff.stride(1), dlg.stride(1), self._ptr(ws), need, self._stream())
        return dvalue, doff, dlg

    def deform_col2im(self, x_cl, offset, dcol, K, stride, pad, dil, groups, deform_groups, mask=None):
        """backward of deform_im2col -> (dx [BN, H, W, C], doffset like offset[, dmask like mask: DCNv2])"""
        BN, H, W, C = x_cl.shape
        doff = torch.empty_like(offset)
        if self.deterministic:
            acc = torch.zeros(x_cl.shape, dtype=torch.int64, device=x_cl.device)
            slot = torch.empty((self.lib.occf_absmax_slot_words(),), dtype=self.i32, device=x_cl.device)
            self._call("occf_absmax_flat", self._ptr(dcol, self.f32), dcol.numel(), self._ptr(slot), self._stream())
            dmask = torch.empty_like(mask) if mask is not None else None
            self._call("occf_deform_col2im_fx", self._ptr(x_cl, self.f32), self._ptr(offset, self.f32),
                       self._ptr(mask, self.f32) if mask is not None else ctypes.c_void_p(0), self._ptr(dcol, self.f32),
                       self._ptr(acc), self._ptr(doff), self._ptr(dmask), self._ptr(slot), BN, H, W, C, K, stride, pad,
                       dil, groups, deform_groups, self._stream())
            dx = torch.empty_like(x_cl)
            self._call("occf_fx_to_f32", self._ptr(acc), self._ptr(dx), acc.numel(), self._ptr(slot), self._stream())
            return (dx, doff, dmask) if mask is not None else (dx, doff)
        dx = torch.zeros_like(x_cl)
        if mask is not None:
            dmask = torch.empty_like(mask)
            self._call("occf_modulated_deform_col2im", self._ptr(x_cl, self.f32), self._ptr(offset, self.f32),
                       self._ptr(mask, self.f32), self._ptr(dcol, self.f32), self._ptr(dx), self._ptr(doff),
                       self._ptr(dmask), BN, H, W, C, K, stride, pad, dil, groups, deform_groups, self._stream())
            return dx, doff, dmask
        self._call("occf_deform_col2im", self._ptr(x_cl, self.f32), self._ptr(offset, self.f32),
                   self._ptr(dcol, self.f32), self._ptr(dx), self._ptr(doff), BN, H, W, C, K, stride, pad, dil, groups,
                   deform_groups, self._stream())
        return dx, doff


    # ------------------------------------------------------------------ img_inputs producer (csrc/pipeline.hip)
    def image_resample(self, img, bounds, kk, out_size, vertical):
        """one pass of Pillow's separable resampling: img uint8 [H, W, C]; bounds int32 [out_size, 2], kk int32
        [out_size, ksize] on the device -> uint8 [H, out_size, C] (horizontal) or [out_size, W, C] (vertical)"""
        H, W, C = img.shape
        if img.dtype != torch.uint8 or bounds.dtype != torch.int32 or kk.dtype != torch.int32:
            raise OccfError("image_resample: uint8 image, int32 tables")
        out = torch.empty((out_size, W, C) if vertical else (H, out_size, C), dtype=torch.uint8, device=img.device)
        self._call("occf_image_resample_fwd", self._ptr(img, torch.uint8), self._ptr(out), self._ptr(bounds), self._ptr(kk),
                   int(kk.shape[1]), H, W, C, int(out_size), int(bool(vertical)), self._stream())
        return out

    def image_crop_rotate_normalize(self, img, crop_xy, out_wh, flip, rot_mode, affine, mean, stdinv, to_rgb,
                                    want_canvas=False):
        """img uint8 [Hn, Wn, 3] -> (float32 [3, fH, fW], uint8 canvas [fH, fW, 3] or None); see the header"""
        Hn, Wn, C = img.shape
        if C != 3 or img.dtype != torch.uint8:
            raise OccfError("image_crop_rotate_normalize: uint8 [H, W, 3] image")
        fW, fH = int(out_wh[0]), int(out_wh[1])
        out = torch.empty((3, fH, fW), dtype=self.f32, device=img.device)
        canvas = torch.empty((fH, fW, 3), dtype=torch.uint8, device=img.device) if want_canvas else None
        aff = (ctypes.c_int64 * 6)(*[int(v) for v in (affine if affine is not None else [0] * 6)])
        mean_h = (ctypes.c_float * 3)(*[float(v) for v in mean])
        std_h = (ctypes.c_float * 3)(*[float(v) for v in stdinv])
        self._call("occf_image_crop_rotate_normalize_fwd", self._ptr(img, torch.uint8), self._ptr(out), self._ptr(canvas),
                   Hn, Wn, int(crop_xy[0]), int(crop_xy[1]), fW, fH, int(bool(flip)), int(rot_mode),
                   ctypes.cast(aff, ctypes.c_void_p), ctypes.cast(mean_h, ctypes.c_void_p),
                   ctypes.cast(std_h, ctypes.c_void_p), int(bool(to_rgb)), self._stream())
        return out, canvas


_ops = None


def get_ops():
    """The product binding (real gfx950 library, GPU tensors only)."""
    global _ops
    if _ops is None:
        _ops = HipOps(_lib.get(), strict=True)
    return _ops
