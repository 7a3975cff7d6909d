"""EfficientNet image backbone of the SemanticKITTI configs (``CustomEfficientNet``, SURVEY.md §8f rank 3: image
branch glue, outside the hand-written kernel scope -- it runs on PyTorch-ROCm / MIOpen like the ResNet of the
nuScenes configs and exists so that ``projects/configs/occformer_kitti/*.py`` load unchanged and released
checkpoints map onto it key by key).

Behaviour restated from the reference's ``P/occformer/backbones/efficientnet.py:231-533`` (the mmcls EfficientNet:
Tan & Le 2019 compound scaling of the MBConv stage table, TensorFlow-style "same" padding, Swish, squeeze-excite
with ratio 0.25 of the block input, BatchNorm eps 1e-3) and written from that description:

* stage table of the b-family (kernel, channels, stride, expand ratio, repeats); width x w -> multiples of 8,
  depth x d -> ceil;
* ``layers[0]`` = stem conv, ``layers[-1]`` = 1x1 head conv; in between one ``nn.Sequential`` per resolution: a
  stage with stride 1 (other than the first) is appended to the preceding stride-2 stage, which is what gives the
  checkpoint its ``layers.<i>.<j>.`` numbering (b7: 7 entries, outputs (2..6) = 48, 80, 224, 640, 2560 channels);
* block = [expand 1x1 conv-BN-Swish unless the expand ratio is 1] -> depthwise kxk conv-BN-Swish -> SE -> linear
  1x1 conv-BN, + identity when stride 1 and channels match (stochastic depth only in training).

EdgeTPU variants ('es', 'em', 'el') are not built (no OccFormer config uses them).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from .registry import BACKBONES

# (kernel, channels, stride, expand ratio, repeats) of EfficientNet-B0 (Tan & Le 2019, table 1)
_B0_STAGES = [(3, 16, 1, 1, 1), (3, 24, 2, 6, 2), (5, 40, 2, 6, 2), (3, 80, 2, 6, 3), (5, 112, 1, 6, 3),
              (5, 192, 2, 6, 4), (3, 320, 1, 6, 1)]
_STEM, _HEAD = 32, 1280
# (width multiplier, depth multiplier) of the compound-scaled family
_SCALING = {"b0": (1.0, 1.0), "b1": (1.0, 1.1), "b2": (1.1, 1.2), "b3": (1.2, 1.4), "b4": (1.4, 1.8),
            "b5": (1.6, 2.2), "b6": (1.8, 2.6), "b7": (2.0, 3.1), "b8": (2.2, 3.6)}


def _round_channels(c, divisor=8, min_ratio=0.9):
    """nearest multiple of ``divisor`` (at least one), never more than 10 % below ``c``"""
    new = max(divisor, int(c + divisor / 2) // divisor * divisor)
    return new + divisor if new < min_ratio * c else new


class _SamePadConv2d(nn.Conv2d):
    """TensorFlow "SAME" convolution: the output has ceil(size / stride) positions; the total padding that needs is
    split with the extra row / column at the bottom / right (so it depends on the input size)."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, k, stride=stride, padding=0, groups=groups, bias=bias)

    def forward(self, x):
        pads = []
        for size, k, s, d in zip(x.shape[-2:][::-1], self.kernel_size[::-1], self.stride[::-1], self.dilation[::-1]):
            total = max((math.ceil(size / s) - 1) * s + (k - 1) * d + 1 - size, 0)
            pads += [total // 2, total - total // 2]
        if any(pads):
            x = F.pad(x, pads)
        return F.conv2d(x, self.weight, self.bias, self.stride, 0, self.dilation, self.groups)


class _ConvBN(nn.Module):
    """conv -> BatchNorm(eps 1e-3) -> optional Swish; parameter names ``conv.*`` / ``bn.*`` as in the checkpoints"""

    def __init__(self, cin, cout, k, stride=1, groups=1, act=True):
        super().__init__()
        self.conv = _SamePadConv2d(cin, cout, k, stride, groups)
        self.bn = nn.BatchNorm2d(cout, eps=1e-3)
        self.act = act

    def forward(self, x):
        x = self.bn(self.conv(x))
        return x * torch.sigmoid(x) if self.act else x


class _Conv1x1(nn.Module):
    """the two 1x1 convolutions (with bias, no norm) of the squeeze-excite gate: ``conv.*``"""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, 1)

    def forward(self, x):
        return self.conv(x)


class _SqueezeExcite(nn.Module):
    def __init__(self, channels, squeezed):
        super().__init__()
        self.conv1 = _Conv1x1(channels, squeezed)
        self.conv2 = _Conv1x1(squeezed, channels)

    def forward(self, x):
        g = self.conv1(x.mean((2, 3), keepdim=True))
        g = self.conv2(g * torch.sigmoid(g))
        return x * torch.sigmoid(g)


class _MBConv(nn.Module):
    """inverted residual block: ``expand_conv`` (absent when the expand ratio is 1), ``depthwise_conv``, ``se``,
    ``linear_conv``"""

    def __init__(self, cin, cout, k, stride, expand, drop_path, with_cp):
        super().__init__()
        mid = int(cin * expand)
        self.shortcut = stride == 1 and cin == cout
        self.drop_path = drop_path
        self.with_cp = with_cp
        if mid != cin:
            self.expand_conv = _ConvBN(cin, mid, 1)
        self.depthwise_conv = _ConvBN(mid, mid, k, stride, groups=mid)
        self.se = _SqueezeExcite(mid, int(mid / (expand * 4)))          # a quarter of the block's input channels
        self.linear_conv = _ConvBN(mid, cout, 1, act=False)

    def _body(self, x):
        y = self.expand_conv(x) if hasattr(self, "expand_conv") else x
        y = self.linear_conv(self.se(self.depthwise_conv(y)))
        if not self.shortcut:
            return y
        if self.training and self.drop_path > 0:                         # stochastic depth, per sample
            keep = 1.0 - self.drop_path
            mask = (keep + torch.rand((y.shape[0], 1, 1, 1), dtype=y.dtype, device=y.device)).floor()
            y = y.div(keep) * mask
        return x + y

    def forward(self, x):
        if self.with_cp and x.requires_grad:
            return cp.checkpoint(self._body, x, use_reentrant=False)
        return self._body(x)


@BACKBONES.register_module()
class CustomEfficientNet(nn.Module):
    """``forward(x [B, 3, H, W]) -> tuple`` of the feature maps after ``layers[i]``, ``i in out_indices``."""

    def __init__(self, arch="b0", drop_path_rate=0.0, out_indices=(6,), frozen_stages=0, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, norm_eval=False, with_cp=False, init_cfg=None, pretrained=None):
        super().__init__()
        if arch not in _SCALING:
            raise NotImplementedError(f"EfficientNet arch {arch!r}: only the b-family is built")
        for name, cfg, want in (("conv_cfg", conv_cfg, "Conv2dAdaptivePadding"), ("act_cfg", act_cfg, "Swish")):
            if cfg is not None and cfg.get("type") != want:
                raise NotImplementedError(f"CustomEfficientNet is built with {name} type {want!r}")
        if norm_cfg is not None and (norm_cfg.get("type") not in ("BN", "BN2d") or norm_cfg.get("eps", 1e-3) != 1e-3):
            raise NotImplementedError("CustomEfficientNet is built with BatchNorm, eps 1e-3")
        width, depth = _SCALING[arch]
        # resolution groups: a stride-1 stage (other than the first) joins the group before it
        groups = []
        for i, (k, c, s, e, n) in enumerate(_B0_STAGES):
            blocks = [(k, _round_channels(c * width), s if j == 0 else 1, e) for j in range(int(math.ceil(n * depth)))]
            if s == 1 and i > 0:
                groups[-1] += blocks
            else:
                groups.append(blocks)
        n_layers = len(groups) + 2
        self.out_indices = tuple(out_indices)
        if any(i not in range(n_layers) for i in self.out_indices):
            raise ValueError(f"out_indices must lie in range(0, {n_layers})")
        if frozen_stages not in range(n_layers + 1):
            raise ValueError(f"frozen_stages must lie in range(0, {n_layers + 1})")
        self.frozen_stages = frozen_stages
        self.norm_eval = norm_eval
        cin = _round_channels(_round_channels(_STEM * width))
        layers = [_ConvBN(3, cin, 3, 2)]
        total = sum(len(g) for g in groups)
        bi = 0
        for gi, blocks in enumerate(groups):
            if gi > max(self.out_indices) - 1:           # layers past the last requested output are not built
                break
            seq = []
            for (k, cout, s, e) in blocks:
                dp = drop_path_rate * bi / (total - 1) if total > 1 else 0.0
                seq.append(_MBConv(cin, cout, k, s, e, dp, with_cp))
                cin = cout
                bi += 1
            layers.append(nn.Sequential(*seq))
        if len(layers) < max(self.out_indices) + 1:
            layers.append(_ConvBN(cin, _round_channels(_HEAD * width), 1))
        self.layers = nn.ModuleList(layers)
        self._init_weights()

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.ones_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        outs = []
        for i, layer in enumerate(self.layers):
            x = layer(x)
            if i in self.out_indices:
                outs.append(x)
        return tuple(outs)

    def train(self, mode=True):
        super().train(mode)
        for i in range(self.frozen_stages):
            self.layers[i].eval()
            for p in self.layers[i].parameters():
                p.requires_grad = False
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self
