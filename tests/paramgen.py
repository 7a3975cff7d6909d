"""Deterministic parameter / input generators shared by the golden-vector script, the
oracle tests and the GPU parity tests.  Every tensor is drawn from its own CPU
generator seeded by crc32(key) ^ seed, so a key's value does not depend on which other
keys exist; the golden fixtures store a checksum of the generated parameters so RNG
drift would be detected rather than silently accepted."""
import math
import zlib

import torch

_SKIP = ("num_batches_tracked", "relative_position_index", ".dx", ".bx", ".nx", ".frustum")


def _gen(key, seed):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    return g


def fill_state_dict(sd, seed=0):
    """Return a new dict with every float parameter/buffer of ``sd`` re-drawn."""
    out = {}
    for k, v in sd.items():
        if any(k.endswith(s) or k == s.lstrip(".") for s in _SKIP) or not v.is_floating_point():
            out[k] = v.detach().clone()
            continue
        g = _gen(k, seed)
        shape = tuple(v.shape)
        if k.endswith("running_var"):
            t = torch.rand(shape, generator=g) + 0.5
        elif k.endswith("running_mean"):
            t = torch.randn(shape, generator=g) * 0.1
        elif k.endswith("relative_position_bias_table"):
            t = torch.randn(shape, generator=g) * 0.5
        elif "sampling_offsets.bias" in k:
            t = torch.randn(shape, generator=g) * 1.5
        elif k.endswith("conv_offset.weight"):
            t = torch.randn(shape, generator=g) * 0.02
        elif k.endswith("conv_offset.bias"):
            t = torch.randn(shape, generator=g) * 0.3
        elif v.dim() >= 2:
            fan_in = v[0].numel()
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
            if k.endswith("query_feat.weight") or k.endswith("query_embed.weight") or \
                    k.endswith("level_embed.weight") or k.endswith("level_encoding.weight"):
                t = torch.randn(shape, generator=g)
        elif any(s in k for s in ("norm", ".gn.", ".bn", "bn1", "bn2", ".ln.")) and k.endswith("weight") \
                or (v.dim() == 1 and k.endswith(".1.weight")) or (v.dim() == 1 and k.endswith(".2.weight")):
            t = 1.0 + 0.2 * torch.randn(shape, generator=g)
        else:
            t = torch.randn(shape, generator=g) * 0.1
        out[k] = t.to(v.dtype)
    return out


def checksum(sd):
    s = 0.0
    for k in sorted(sd):
        if sd[k].is_floating_point():
            s += float(sd[k].double().abs().sum())
    return s


def tensor(key, shape, seed=0, scale=1.0):
    return torch.randn(tuple(shape), generator=_gen("input:" + key, seed)) * scale


def uniform(key, shape, seed=0):
    return torch.rand(tuple(shape), generator=_gen("input:" + key, seed))


def camera_rig(B, N, H, W, focal, seed=0, kitti=False, augment=True):
    """Synthetic surround rig: yaw-spread cameras at t=(1.5*cos, 1.5*sin, 1.5),
    optical axis = camera +z pointing outward horizontally; optional image/BEV augmentation."""
    yaws = torch.tensor([55.0, 0.0, -55.0, 110.0, 180.0, -110.0])[:N] if N <= 6 else \
        torch.linspace(0, 360, N + 1)[:N]
    rots, trans = [], []
    for yaw in yaws.tolist():
        a = math.radians(yaw)
        fwd = torch.tensor([math.cos(a), math.sin(a), 0.0])
        right = torch.tensor([math.sin(a), -math.cos(a), 0.0])
        down = torch.tensor([0.0, 0.0, -1.0])
        rots.append(torch.stack((right, down, fwd), 1))     # cam(x right, y down, z fwd) -> ego
        trans.append(torch.tensor([1.5 * math.cos(a), 1.5 * math.sin(a), 1.5]))
    rots = torch.stack(rots).unsqueeze(0).repeat(B, 1, 1, 1)
    trans = torch.stack(trans).unsqueeze(0).repeat(B, 1, 1)
    K = torch.tensor([[focal, 0.0, W / 2.0], [0.0, focal, H / 2.0 - 0.1 * H], [0.0, 0.0, 1.0]])
    if kitti:
        K4 = torch.eye(4)
        K4[:3, :3] = K
        K4[:3, 3] = torch.tensor([4.0, 0.2, 0.003])
        intr = K4.view(1, 1, 4, 4).repeat(B, N, 1, 1)
    else:
        intr = K.view(1, 1, 3, 3).repeat(B, N, 1, 1)
    post_rots = torch.eye(3).view(1, 1, 3, 3).repeat(B, N, 1, 1)
    post_trans = torch.zeros(B, N, 3)
    bda = torch.eye(4 if kitti else 3).view(1, -1, 4 if kitti else 3).repeat(B, 1, 1)
    if augment:
        g = _gen("rig", seed)
        for b in range(B):
            for n in range(N):
                s = 1.0 + 0.08 * float(torch.randn((), generator=g))
                ang = math.radians(3.0 * float(torch.randn((), generator=g)))
                R2 = torch.tensor([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]]) * s
                post_rots[b, n, :2, :2] = R2
                post_trans[b, n, :2] = torch.randn(2, generator=g) * 3.0
            flip = torch.diag(torch.tensor([-1.0 if b % 2 else 1.0, 1.0, 1.0])) * 1.03
            bda[b, :3, :3] = flip
            if kitti:
                bda[b, :3, 3] = torch.tensor([0.1, -0.2, 0.05])
    return rots, trans, intr, post_rots, post_trans, bda
