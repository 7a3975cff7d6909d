"""Noise source of the training step.  Every stochastic op of the path's training forward -- DropPath of the shared
SwinBlock (window_attention.py:311,332), the Dropout of the BEV ASPP (aspp.py:103,122) and of DepthNet's ASPP
(ViewTransformerLSSBEVDepth.py:407), the point / class-guided sampling of the occupancy heads (mmdet_utils.py:91-246)
-- draws from ONE injectable object with ``rand`` / ``randperm`` / ``exponential`` methods, in a fixed call order,
so that a step is reproducible and can be replayed against the CPU oracle on identical draws (bitwise parity with
torch's own generator streams is not defined across devices)."""
import torch

_rng = None


def set_rng(rng):
    """install a noise source (None = a seeded device generator created on first use)"""
    global _rng
    _rng = rng


def get_rng(device):
    global _rng
    if _rng is None:
        from .training import DeviceRNG
        _rng = DeviceRNG(device)
    return _rng


def drop_path_scale(n_samples, drop_prob, device):
    """mmcv DropPath: per sample floor(keep + U) / keep; None when the op is the identity"""
    if drop_prob <= 0.0:
        return None
    keep = 1.0 - drop_prob
    u = get_rng(device).rand(n_samples).to(device=device, dtype=torch.float32)
    return torch.floor(keep + u) / keep


def dropout_mask(shape, p, device):
    """nn.Dropout as an explicit mask: (U >= p) / (1 - p), drawn in the REFERENCE's tensor layout ``shape``"""
    if p <= 0.0:
        return None
    u = get_rng(device).rand(*shape).to(device=device, dtype=torch.float32)
    return (u >= p).float() / (1.0 - p)


# ---- comparison tap (tests, bench.py's check): the ReLU gates this implementation used in the training graph ----
# Two levels.  "heavy" = the units whose gate moves every upstream gradient (decoder-head MLPs, DepthNet, the ASPP's
# image-level vector): what the full-size comparisons force into the oracle.  "all" = every ReLU of the path (encoder /
# pixel-decoder GroupNorm + ReLU maps, the pixel decoder's FFNs): what the tiny configurations need, where ONE gate of a
# 10^4-unit map weighs 1e-3 of the whole gradient.  The taps sit in the op layer (autograd.Linear / autograd.GroupNorm:
# one hook per op) and in one helper per ATen-composed module (view_transformer._relu, encoder._ASPP); call ORDER is
# the oracle's evaluation order, masks are in the reference's [B, C, ...] layout.
_gates = None
_level = None


def record_gates(on=True):
    """start (``True`` / "heavy", "heavy+bev" or "all" -> the list that fills up, in call order) or stop (``False``)
    recording.  "heavy+bev" = the heavy units plus the BEV ASPP's GroupNorm + ReLU maps (class "bev": small maps on the
    coarse stages, where one unit is a visible share of the convolution's gradient -- oracle.occformer_ref.forced_gates)"""
    global _gates, _level
    _gates = [] if on else None
    _level = None if not on else (on if on in ("all", "heavy+bev") else "heavy")
    return _gates


def gates_wanted(heavy):
    """``heavy``: True (heavy unit), "bev" (a BEV-ASPP map) or False (any other ReLU of the path)"""
    if _gates is None:
        return False
    return heavy is True or _level == "all" or (heavy == "bev" and _level == "heavy+bev")


def relu_gate(h, heavy=True, gate=None):
    """``h``: a ReLU OUTPUT of the training graph; a no-op unless a comparison asked for the gates
    (oracle.occformer_ref.forced_gates explains why).  ``gate``: a callable -> bool tensor for the ops whose output is
    not the bare ReLU (GroupNorm + ReLU + residual, token buffers); evaluated only while recording."""
    if gates_wanted(heavy):
        _gates.append((h.detach() > 0) if gate is None else gate())
    return h


def tape_mask(blocked):
    """the decoder's boolean attention mask of the training graph (mask2former_nusc_occ.py:457-469: pooled mask logits,
    ``sigmoid < 0.5``): the other discrete decision on fp32 values next to the ReLU gates -- a pooled logit within rounding
    of zero blocks a key in one implementation and not in the other, and at a coarse level one key is a visible share
    of a query's attention.  Taped in call order with the heavy gates; a no-op unless a comparison records."""
    if gates_wanted(True):
        _gates.append(blocked.detach() != 0)
    return blocked


class RecordedRNG:
    """wraps a noise source and tapes every draw as (kind, CPU tensor) in call order -- the tape a comparison feeds to
    the CPU oracle so that it repeats THIS step's noise (the converse of replaying the oracle's tape on the device; one
    oracle pass instead of two).  A host copy per draw: comparison runs only."""

    def __init__(self, inner):
        self.inner, self.tape = inner, []

    def _keep(self, kind, t):
        self.tape.append((kind, t.detach().cpu()))
        return t

    def rand(self, *shape):
        return self._keep("rand", self.inner.rand(*shape))

    def randperm(self, n):
        return self._keep("randperm", self.inner.randperm(n))

    def exponential(self, shape, dtype=torch.float32):
        return self._keep("exponential", self.inner.exponential(shape, dtype))
