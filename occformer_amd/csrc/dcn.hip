// Deformable-convolution im2col (DCNv1) for DepthNet's depth branch.
//
// Reference: mmcv-full 1.4.0 `deform_conv2d` (third-party CUDA op, not under /root/reference;
// call site projects/mmdet3d_plugin/occformer/image2bev/ViewTransformerLSSBEVDepth.py:479-487:
// 3x3, padding 1, groups 4, deform_groups 1, im2col_step 128).  Published algorithm: for output
// pixel (h, w) and tap (ky, kx) sample the input bilinearly (zero outside) at
// (h*stride - pad + ky*dil + dy, w*stride - pad + kx*dil + dx) with (dy, dx) =
// offset[2*(dg*K*K + tap) + {0, 1}], then contract with the grouped weight.
//
// Input x channels-last [BN, H, W, C]; offsets in the conv_offset layout [BN, dg*2*K*K, Ho, Wo];
// output columns [BN*Ho*Wo][groups][K*K][C/groups] so that every conv group's K-slab is one
// contiguous row segment for the MFMA GEMM that follows (occf_linear_*).
// Thread = (output pixel, tap, channel quad); gathers are 16-B channel-contiguous.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// `mask` != NULL: DCNv2 (mmcv ModulatedDeformConv2dPack, the R101-DCN image backbone of
// occformer_nusc_r101_896x1600.py:78-79): the sample of tap t is scaled by mask[bn, dg*K*K + t, ho, wo]
// (already sigmoid-ed by the caller, as mmcv does).
__global__ void __launch_bounds__(256) deform_im2col_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
    float* __restrict__ col, int BN, int H,
    int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil, int groups, int dgroups) {
  const int Q = C / 4, KK = K * K;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)BN * Ho * Wo * KK * Q;
  if (gid >= total) return;
  const int cq = (int)(gid % Q);
  long r = gid / Q;
  const int t = (int)(r % KK);
  r /= KK;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int bn = (int)(r / Ho);
  const int c = cq * 4;
  const int dg = c / (C / dgroups);
  const int ky = t / K, kx = t % K;
  const long obase = (((long)bn * dgroups + dg) * 2 * KK + 2 * t) * Ho * Wo + (long)ho * Wo + wo;
  const float py = (float)(ho * stride - pad + ky * dil) + offset[obase];
  const float px = (float)(wo * stride - pad + kx * dil) + offset[obase + (long)Ho * Wo];
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    const float ly = py - fy, lx = px - fx;
    const float* xb = x + (long)bn * H * W * C + c;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const float wgt = ((k >> 1) ? ly : 1.f - ly) * ((k & 1) ? lx : 1.f - lx);
      const float4 v = *(const float4*)(xb + ((long)yy * W + xx) * C);
      acc[0] = fmaf(wgt, v.x, acc[0]); acc[1] = fmaf(wgt, v.y, acc[1]);
      acc[2] = fmaf(wgt, v.z, acc[2]); acc[3] = fmaf(wgt, v.w, acc[3]);
    }
  }
  if (mask) {
    const float mk = mask[(((long)bn * dgroups + dg) * KK + t) * Ho * Wo + (long)ho * Wo + wo];
    acc[0] *= mk; acc[1] *= mk; acc[2] *= mk; acc[3] *= mk;
  }
  const int cpg = C / groups;
  const int g = c / cpg, cg = c - g * cpg;
  const long pix = ((long)bn * Ho + ho) * Wo + wo;
  *(float4*)(col + (pix * groups + g) * KK * cpg + (long)t * cpg + cg) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

extern "C" int occf_deform_im2col(const float* x, const float* offset, float* col, int BN, int H, int W, int C,
                                  int K, int stride, int pad, int dil, int groups, int deform_groups,
                                  void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || deform_groups <= 0) return OCCF_EINVAL;
  if (C % groups || C % deform_groups || (C / groups) % 4 || (C / deform_groups) % 4) return OCCF_ESHAPE;
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const long total = (long)BN * Ho * Wo * K * K * (C / 4);
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     offset, (const float*)nullptr, col, BN, H, W, C, Ho, Wo, K, stride, pad, dil, groups, deform_groups);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_modulated_deform_im2col(const float* x, const float* offset, const float* mask, float* col, int BN,
                                            int H, int W, int C, int K, int stride, int pad, int dil, int groups,
                                            int deform_groups, void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || deform_groups <= 0 || !mask) return OCCF_EINVAL;
  if (C % groups || C % deform_groups || (C / groups) % 4 || (C / deform_groups) % 4) return OCCF_ESHAPE;
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const long total = (long)BN * Ho * Wo * K * K * (C / 4);
  hipLaunchKernelGGL(deform_im2col_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     offset, mask, col, BN, H, W, C, Ho, Wo, K, stride, pad, dil, groups, deform_groups);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the deformable im2col (mmcv `deformable_col2im` + `deformable_col2im_coord`): dcol in the forward's
// column layout -> dx (bilinear scatter, float atomics; ZERO-FILLED by the caller) and doffset[BN, dg*2*K*K, Ho, Wo]
// (every element written).  One wave per (output pixel, tap, deform group): lanes stride over the channel quads of
// the group, the two coordinate gradients are wave reductions.
// `mask` != NULL (DCNv2, mmcv `modulated_deformable_col2im` + `_coord`): the forward sample was scaled by the modulation
// mk = mask[bn, dg*K*K + t, ho, wo], so dx and doffset carry the factor mk and dmask[bn, dg*K*K + t, ho, wo] =
// <dcol, unscaled sample> (every element written).
// FX: dx is a zero-filled int64 buffer, the scatter runs in fixed point (occf_scatter_add: reproducible sums)
template <bool FX>
__global__ void __launch_bounds__(256) deform_col2im_kernel(
    const float* __restrict__ x, const float* __restrict__ offset, const float* __restrict__ mask,
    const float* __restrict__ dcol, void* __restrict__ dx, float* __restrict__ doffset, float* __restrict__ dmask,
    int BN, int H, int W, int C, int Ho, int Wo, int K, int stride, int pad, int dil, int groups, int dgroups,
    const uint32_t* __restrict__ slot) {
  float fx_inv = 1.f;
  const float fx_scale = FX ? occf_fx_scale(slot[0], fx_inv) : 1.f;
  const int KK = K * K;
  const int lane = threadIdx.x & 63;
  const long wid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long total = (long)BN * Ho * Wo * KK * dgroups;
  if (wid >= total) return;
  long r = wid;
  const int dg = (int)(r % dgroups);
  r /= dgroups;
  const int t = (int)(r % KK);
  r /= KK;
  const int wo = (int)(r % Wo);
  r /= Wo;
  const int ho = (int)(r % Ho);
  const int bn = (int)(r / Ho);
  const int ky = t / K, kx = t % K;
  const long obase = (((long)bn * dgroups + dg) * 2 * KK + 2 * t) * Ho * Wo + (long)ho * Wo + wo;
  const float py = (float)(ho * stride - pad + ky * dil) + offset[obase];
  const float px = (float)(wo * stride - pad + kx * dil) + offset[obase + (long)Ho * Wo];
  float gy = 0.f, gx = 0.f, gm = 0.f;
  const long mbase = (((long)bn * dgroups + dg) * KK + t) * Ho * Wo + (long)ho * Wo + wo;
  const float mk = mask ? mask[mbase] : 1.f;
  if (py > -1.f && px > -1.f && py < (float)H && px < (float)W) {
    const float fy = floorf(py), fx = floorf(px);
    const int y0 = (int)fy, x0 = (int)fx;
    const float ly = py - fy, lx = px - fx;
    const int cd = C / dgroups, cpg = C / groups;
    const long pix = ((long)bn * Ho + ho) * Wo + wo;
    for (int cq = lane; cq < cd / 4; cq += 64) {
      const int c = dg * cd + cq * 4;
      const int g = c / cpg, cg = c - g * cpg;
      const float4 gc = *(const float4*)(dcol + (pix * groups + g) * KK * cpg + (long)t * cpg + cg);
      const float* xb = x + (long)bn * H * W * C + c;
      const long db = (long)bn * H * W * C + c;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float wy = (k >> 1) ? ly : 1.f - ly, wx = (k & 1) ? lx : 1.f - lx;
        const float4 v = *(const float4*)(xb + ((long)yy * W + xx) * C);
        const float dot = (gc.x * v.x + gc.y * v.y) + (gc.z * v.z + gc.w * v.w);
        gy = fmaf(((k >> 1) ? 1.f : -1.f) * wx, dot, gy);
        gx = fmaf(((k & 1) ? 1.f : -1.f) * wy, dot, gx);
        gm = fmaf(wy * wx, dot, gm);
        const long d = db + ((long)yy * W + xx) * C;
        const float wgt = wy * wx * mk;
        occf_scatter_add<FX>(dx, d + 0, wgt * gc.x, fx_scale);
        occf_scatter_add<FX>(dx, d + 1, wgt * gc.y, fx_scale);
        occf_scatter_add<FX>(dx, d + 2, wgt * gc.z, fx_scale);
        occf_scatter_add<FX>(dx, d + 3, wgt * gc.w, fx_scale);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    gy += __shfl_xor(gy, o);
    gx += __shfl_xor(gx, o);
    gm += __shfl_xor(gm, o);
  }
  if (lane == 0) {
    doffset[obase] = gy * mk;
    doffset[obase + (long)Ho * Wo] = gx * mk;
    if (dmask) dmask[mbase] = gm;
  }
}

extern "C" int occf_deform_col2im(const float* x, const float* offset, const float* dcol, float* dx, float* doffset,
                                  int BN, int H, int W, int C, int K, int stride, int pad, int dil, int groups,
                                  int deform_groups, void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || deform_groups <= 0) return OCCF_EINVAL;
  if (C % groups || C % deform_groups || (C / groups) % 4 || (C / deform_groups) % 4) return OCCF_ESHAPE;
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const long waves = (long)BN * Ho * Wo * K * K * deform_groups;
  hipLaunchKernelGGL(deform_col2im_kernel<false>, dim3(occf_cdiv(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     offset, (const float*)nullptr, dcol, (void*)dx, doffset, (float*)nullptr, BN, H, W, C, Ho, Wo, K,
                     stride, pad, dil, groups, deform_groups, (const uint32_t*)nullptr);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_modulated_deform_col2im(const float* x, const float* offset, const float* mask, const float* dcol,
                                            float* dx, float* doffset, float* dmask, int BN, int H, int W, int C, int K,
                                            int stride, int pad, int dil, int groups, int deform_groups, void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || deform_groups <= 0 || !mask || !dmask)
    return OCCF_EINVAL;
  if (C % groups || C % deform_groups || (C / groups) % 4 || (C / deform_groups) % 4) return OCCF_ESHAPE;
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const long waves = (long)BN * Ho * Wo * K * K * deform_groups;
  hipLaunchKernelGGL(deform_col2im_kernel<false>, dim3(occf_cdiv(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     offset, mask, dcol, (void*)dx, doffset, dmask, BN, H, W, C, Ho, Wo, K, stride, pad, dil, groups,
                     deform_groups, (const uint32_t*)nullptr);
  OCCF_LAUNCH_CHECK();
}

// both col2im forms with the data-gradient scatter in 64-bit fixed point (reproducible): acc = zero-filled int64 buffer
// shaped like dx, slot = scale slot holding max |dcol|; mask / dmask NULL = DCNv1.  occf_fx_to_f32 converts afterwards.
extern "C" int occf_deform_col2im_fx(const float* x, const float* offset, const float* mask, const float* dcol,
                                     long long* acc, float* doffset, float* dmask, const uint32_t* slot, int BN, int H,
                                     int W, int C, int K, int stride, int pad, int dil, int groups, int deform_groups,
                                     void* stream) {
  if (BN <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0 || groups <= 0 || deform_groups <= 0 || !acc || !slot)
    return OCCF_EINVAL;
  if ((mask == nullptr) != (dmask == nullptr)) return OCCF_EINVAL;
  if (C % groups || C % deform_groups || (C / groups) % 4 || (C / deform_groups) % 4) return OCCF_ESHAPE;
  const int Ho = (H + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const int Wo = (W + 2 * pad - dil * (K - 1) - 1) / stride + 1;
  const long waves = (long)BN * Ho * Wo * K * K * deform_groups;
  hipLaunchKernelGGL(deform_col2im_kernel<true>, dim3(occf_cdiv(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     offset, mask, dcol, (void*)acc, doffset, dmask, BN, H, W, C, Ho, Wo, K, stride, pad, dil, groups,
                     deform_groups, slot);
  OCCF_LAUNCH_CHECK();
}
