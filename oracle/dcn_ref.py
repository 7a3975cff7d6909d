"""TEST INFRASTRUCTURE (oracle): modulated deformable convolution (DCNv2, mmcv ``ModulatedDeformConv2dPack`` of the
R101-DCN image backbone, occformer_nusc_r101_896x1600.py:78-79) stated with ``F.grid_sample`` -- the CPU statement the
library's im2col / col2im kernels (csrc/dcn.hip) are checked against.  tests/test_image_backbone.py pins this
statement itself to explicit bilinear loops.  Never imported by the product."""
import torch
import torch.nn.functional as F


def modulated_deform_conv2d(x, offset, mask, weight, bias, k, stride, padding, dilation, dg):
    """x [B, C, H, W]; offset [B, dg * 2 * k*k, Ho, Wo] (all dy/dx pairs of a group, tap-major); mask
    [B, dg * k*k, Ho, Wo] (already sigmoid-ed); weight [Cout, C, k, k] -> [B, Cout, Ho, Wo]"""
    B, C, H, W = x.shape
    Ho, Wo = offset.shape[-2:]
    ys = (torch.arange(Ho, device=x.device, dtype=x.dtype) * stride - padding).view(1, Ho, 1)
    xs = (torch.arange(Wo, device=x.device, dtype=x.dtype) * stride - padding).view(1, 1, Wo)
    offset = offset.view(B, dg, k * k, 2, Ho, Wo)
    mask = mask.view(B, dg, k * k, Ho, Wo)
    cpg = C // dg
    cols = []
    for t in range(k * k):
        ky, kx = divmod(t, k)
        per_group = []
        for g in range(dg):
            py = ys + ky * dilation + offset[:, g, t, 0]
            px = xs + kx * dilation + offset[:, g, t, 1]
            # pixel coordinates -> grid_sample's align_corners=True convention (zeros outside)
            grid = torch.stack((2 * px / max(W - 1, 1) - 1, 2 * py / max(H - 1, 1) - 1), -1)
            smp = F.grid_sample(x[:, g * cpg:(g + 1) * cpg], grid, mode="bilinear", padding_mode="zeros",
                                align_corners=True)
            per_group.append(smp * mask[:, g, t].unsqueeze(1))
        cols.append(torch.cat(per_group, 1))
    col = torch.stack(cols, 2)                                        # [B, C, k*k, Ho, Wo]
    out = torch.einsum("bcthw,oct->bohw", col, weight.flatten(2))
    return out if bias is None else out + bias.view(1, -1, 1, 1)


def module_reference(m, x, offset, mask):
    """the stand-in tests install as ``ModulatedDeformConv2dPack.cpu_reference``"""
    return modulated_deform_conv2d(x, offset, mask, m.weight, m.bias, m.k, m.stride, m.padding, m.dilation, m.dg)
