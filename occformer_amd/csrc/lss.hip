// LSS view transform on gfx950: frustum -> voxel index, sorted-interval pooling
// (drop-in for the reference's only native op, bev_pool) and the fused lift+splat that
// never materialises the [B,N,D,fH,fW,C] volume.
//
// Reference: mmdetection3d/mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:20-98,
//            projects/mmdet3d_plugin/occformer/image2bev/ViewTransformerLSSVoxel.py:77-121,
//            .../ViewTransformerLSSBEVDepth.py:117-150.
//
// All three pooling kernels are HBM-write bound: every output row is written exactly
// once, channel-contiguous (a row of C floats is covered by C/4 adjacent lanes with
// 16-byte accesses); the gathers hit a <=2 MB feature map that lives in L2.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// ---------------------------------------------------------------------------------------
// bev_pool at the reference's op boundary.
// One lane-group of C/4 lanes (16 B per lane) per interval; rows of an interval are added
// sequentially in row order with un-fused fp32 adds, i.e. the same arithmetic, in the
// same order, as bev_pool_cuda.cu:37-41 -> bit-identical results.
template <int VEC>
__global__ void __launch_bounds__(256) bev_pool_fwd_kernel(
    const float* __restrict__ x, const int* __restrict__ geom, const int* __restrict__ starts,
    const int* __restrict__ lengths, float* __restrict__ out, int d, int h, int w, int c,
    int n_intervals) {
  const int lanes_per_row = c / VEC;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long interval = gid / lanes_per_row;
  const int cv = (int)(gid % lanes_per_row) * VEC;
  if (interval >= n_intervals) return;
  const int s = starts[interval];
  const int len = lengths[interval];
  const int* g = geom + (long)s * 4;
  const long obase = ((((long)g[3] * d + g[2]) * h + g[0]) * w + g[1]) * c + cv;
  const float* px = x + (long)s * c + cv;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  for (int i = 0; i < len; ++i) {
    float r[VEC];
    if (VEC == 4) {
      const float4 t = *(const float4*)(px + (long)i * c);
      r[0] = t.x; r[1 % VEC] = t.y; r[2 % VEC] = t.z; r[3 % VEC] = t.w;
    } else {
#pragma unroll
      for (int v = 0; v < VEC; ++v) r[v] = px[(long)i * c + v];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) acc[v] = occf_fadd(acc[v], r[v]);
  }
  if (VEC == 4) {
    *(float4*)(out + obase) = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) out[obase + v] = acc[v];
  }
}

template <int VEC>
__global__ void __launch_bounds__(256) bev_pool_bwd_kernel(
    const float* __restrict__ out_grad, const int* __restrict__ geom,
    const int* __restrict__ starts, const int* __restrict__ lengths, float* __restrict__ x_grad,
    int d, int h, int w, int c, int n_intervals) {
  const int lanes_per_row = c / VEC;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long interval = gid / lanes_per_row;
  const int cv = (int)(gid % lanes_per_row) * VEC;
  if (interval >= n_intervals) return;
  const int s = starts[interval];
  const int len = lengths[interval];
  const int* g = geom + (long)s * 4;
  const long obase = ((((long)g[3] * d + g[2]) * h + g[0]) * w + g[1]) * c + cv;
  float r[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) r[v] = out_grad[obase + v];
  float* px = x_grad + (long)s * c + cv;
  for (int i = 0; i < len; ++i) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) px[(long)i * c + v] = r[v];
  }
}

extern "C" int occf_bev_pool_fwd(const float* x, const int32_t* geom, const int32_t* interval_starts,
                                 const int32_t* interval_lengths, float* out, int b, int d, int h,
                                 int w, int n, int c, int n_intervals, void* stream) {
  if (b <= 0 || d <= 0 || h <= 0 || w <= 0 || c <= 0 || n < 0 || n_intervals < 0) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
#ifndef OCCF_EMU
  hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)b * d * h * w * c, st);
  if (e != hipSuccess) return (int)e;
#else
  memset(out, 0, sizeof(float) * (size_t)b * d * h * w * c);
#endif
  if (n_intervals == 0) return 0;
  if (c % 4 == 0) {
    const long threads = (long)n_intervals * (c / 4);
    hipLaunchKernelGGL(bev_pool_fwd_kernel<4>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st, x,
                       geom, interval_starts, interval_lengths, out, d, h, w, c, n_intervals);
  } else {
    const long threads = (long)n_intervals * c;
    hipLaunchKernelGGL(bev_pool_fwd_kernel<1>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st, x,
                       geom, interval_starts, interval_lengths, out, d, h, w, c, n_intervals);
  }
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_bev_pool_bwd(const float* out_grad, const int32_t* geom,
                                 const int32_t* interval_starts, const int32_t* interval_lengths,
                                 float* x_grad, int b, int d, int h, int w, int n, int c,
                                 int n_intervals, void* stream) {
  if (b <= 0 || d <= 0 || h <= 0 || w <= 0 || c <= 0 || n < 0 || n_intervals < 0) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  // every row of x_grad belongs to exactly one interval, so no zero fill is needed
  if (n_intervals == 0) return 0;
  if (c % 4 == 0) {
    const long threads = (long)n_intervals * (c / 4);
    hipLaunchKernelGGL(bev_pool_bwd_kernel<4>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st,
                       out_grad, geom, interval_starts, interval_lengths, x_grad, d, h, w, c,
                       n_intervals);
  } else {
    const long threads = (long)n_intervals * c;
    hipLaunchKernelGGL(bev_pool_bwd_kernel<1>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st,
                       out_grad, geom, interval_starts, interval_lengths, x_grad, d, h, w, c,
                       n_intervals);
  }
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Frustum point -> voxel id.  Per camera the host passes (all fp32, computed with the same
// torch ops as the reference): inv(post_rots) [9], post_trans [3], rots @ inv(K) [9],
// trans [3], K[:3,3] shift [3] (zeros unless KITTI), and per batch the 3x4 bda.  The point
// math follows get_geometry's operation order with un-fused mul/add; quantisation is
// trunc-toward-zero of (p - lo) / dx exactly as `.long()` does in voxel_pooling.
// cam layout: 27 floats per (b, n): ipr[9] pt[3] comb[9] tr[3] shift[3]
// Voxel id is the channels-last row index ((b*X + x)*Y + y)*Z + z, or -1 if out of range.
__device__ __forceinline__ float occf_dot3(const float* m, float x, float y, float z) {
  return occf_fadd(occf_fadd(occf_fmul(m[0], x), occf_fmul(m[1], y)), occf_fmul(m[2], z));
}

__global__ void __launch_bounds__(256) lss_voxel_index_kernel(
    const float* __restrict__ frustum, const float* __restrict__ cam, const float* __restrict__ bda,
    const float* __restrict__ grid, int32_t* __restrict__ vox, int B, int N, int DHW, int X, int Y,
    int Z, int bda4) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N * DHW;
  if (gid >= total) return;
  const int pt = (int)(gid % DHW);
  const int bn = (int)(gid / DHW);
  const int b = bn / N;
  const float* cm = cam + (long)bn * 27;
  const float* f = frustum + (long)pt * 3;
  float px = occf_fadd(f[0], -cm[9]), py = occf_fadd(f[1], -cm[10]), pz = occf_fadd(f[2], -cm[11]);
  float qx = occf_dot3(cm + 0, px, py, pz), qy = occf_dot3(cm + 3, px, py, pz),
        qz = occf_dot3(cm + 6, px, py, pz);
  qx = occf_fmul(qx, qz);
  qy = occf_fmul(qy, qz);
  qx = occf_fadd(qx, -cm[24]);
  qy = occf_fadd(qy, -cm[25]);
  qz = occf_fadd(qz, -cm[26]);
  float ex = occf_fadd(occf_dot3(cm + 12, qx, qy, qz), cm[21]);
  float ey = occf_fadd(occf_dot3(cm + 15, qx, qy, qz), cm[22]);
  float ez = occf_fadd(occf_dot3(cm + 18, qx, qy, qz), cm[23]);
  const float* bm = bda + (long)b * 12;
  float gx, gy, gz;
  if (bda4) {
    gx = occf_fadd(occf_dot3(bm + 0, ex, ey, ez), bm[3]);
    gy = occf_fadd(occf_dot3(bm + 4, ex, ey, ez), bm[7]);
    gz = occf_fadd(occf_dot3(bm + 8, ex, ey, ez), bm[11]);
  } else {
    gx = occf_dot3(bm + 0, ex, ey, ez);
    gy = occf_dot3(bm + 4, ex, ey, ez);
    gz = occf_dot3(bm + 8, ex, ey, ez);
  }
  // grid: lo[3], dx[3], nx[3] (nx as float, compared as in the reference)
  const float fx = occf_fadd(gx, -grid[0]) / grid[3];
  const float fy = occf_fadd(gy, -grid[1]) / grid[4];
  const float fz = occf_fadd(gz, -grid[2]) / grid[5];
  // trunc toward zero; clamp huge/NaN values out of range before the int conversion
  const bool finite = (fx > -2.0e9f && fx < 2.0e9f && fy > -2.0e9f && fy < 2.0e9f && fz > -2.0e9f &&
                       fz < 2.0e9f);
  int32_t v = -1;
  if (finite) {
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    if (ix >= 0 && (float)ix < grid[6] && iy >= 0 && (float)iy < grid[7] && iz >= 0 &&
        (float)iz < grid[8] && ix < X && iy < Y && iz < Z)
      v = ((b * X + ix) * Y + iy) * Z + iz;
  }
  vox[gid] = v;
}

extern "C" int occf_lss_voxel_index(const float* frustum, const float* cam, const float* bda,
                                    const float* grid, int32_t* vox, int B, int N, int DHW, int X,
                                    int Y, int Z, int bda4, void* stream) {
  if (B <= 0 || N <= 0 || DHW <= 0 || X <= 0 || Y <= 0 || Z <= 0) return OCCF_EINVAL;
  if ((long)B * X * Y * Z >= 2147483647L) return OCCF_ESHAPE;
  const long total = (long)B * N * DHW;
  hipLaunchKernelGGL(lss_voxel_index_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0,
                     (hipStream_t)stream, frustum, cam, bda, grid, vox, B, N, DHW, X, Y, Z, bda4);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Fused lift + splat.  Points are given as a CSR over ALL voxels in channels-last row
// order: voxel v owns sorted_pts[offsets[v] .. offsets[v+1]) (ascending original point
// index = the order a stable sort of the reference's ranks produces).  For point
// p = (bn, d, hw): contribution = depth[bn, d, hw] * feat[bn, hw, :]  (feat channels-last),
// rounded as a product and then added sequentially -- the same arithmetic as
// `depth.unsqueeze(1) * feat.unsqueeze(2)` followed by bev_pool's in-order sum.
// Every output row (also empty voxels) is written exactly once: no memset pass.
template <int VEC>
__global__ void __launch_bounds__(256) lift_splat_fwd_kernel(
    const float* __restrict__ depth, const float* __restrict__ feat,
    const int32_t* __restrict__ offsets, const int32_t* __restrict__ sorted_pts,
    float* __restrict__ out, long n_vox, int D, int HW, int C) {
  const int lanes_per_row = C / VEC;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long v = gid / lanes_per_row;
  const int cv = (int)(gid % lanes_per_row) * VEC;
  if (v >= n_vox) return;
  const int s = offsets[v], e = offsets[v + 1];
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  const int DHW = D * HW;
  for (int i = s; i < e; ++i) {
    const int p = sorted_pts[i];
    const int bn = p / DHW;
    const int hw = (p - bn * DHW) % HW;
    const float dp = depth[p];
    const float* fr = feat + ((long)bn * HW + hw) * C + cv;
    float r[VEC];
    if (VEC == 4) {
      const float4 t = *(const float4*)fr;
      r[0] = t.x; r[1 % VEC] = t.y; r[2 % VEC] = t.z; r[3 % VEC] = t.w;
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) r[k] = fr[k];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = occf_fadd(acc[k], occf_fmul(dp, r[k]));
  }
  float* o = out + v * C + cv;
  if (VEC == 4) {
    *(float4*)o = make_float4(acc[0], acc[1 % VEC], acc[2 % VEC], acc[3 % VEC]);
  } else {
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = acc[k];
  }
}

extern "C" int occf_lift_splat_fwd(const float* depth, const float* feat, const int32_t* offsets,
                                   const int32_t* sorted_pts, float* out, long n_vox, int BN, int D,
                                   int HW, int C, void* stream) {
  if (n_vox <= 0 || BN <= 0 || D <= 0 || HW <= 0 || C <= 0) return OCCF_EINVAL;
  if ((long)BN * D * HW >= 2147483647L) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  if (C % 4 == 0) {
    const long threads = n_vox * (C / 4);
    hipLaunchKernelGGL(lift_splat_fwd_kernel<4>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st,
                       depth, feat, offsets, sorted_pts, out, n_vox, D, HW, C);
  } else {
    const long threads = n_vox * C;
    hipLaunchKernelGGL(lift_splat_fwd_kernel<1>, dim3(occf_cdiv(threads, 256)), dim3(256), 0, st,
                       depth, feat, offsets, sorted_pts, out, n_vox, D, HW, C);
  }
  OCCF_LAUNCH_CHECK();
}

// Backward of the fused lift+splat (training): one wave per PIXEL (bn, hw), which walks the D depth bins of its ray
// in order -- the D points of a pixel are the only contributions to that pixel's feature gradient, so it accumulates in
// registers in a fixed order (no atomics, no memset, bit-reproducible; round 6: the first version was one wave per
// point with float atomicAdd into d_feat, one of the five non-deterministic sums of the training step).
//   d_depth[p]          = sum_c g[vox(p), c] * feat[bn, hw, c]
//   d_feat[bn, hw, c]   = sum_d depth[p(d)] * g[vox(p(d)), c]
__global__ void __launch_bounds__(256) lift_splat_bwd_kernel(
    const float* __restrict__ out_grad, const float* __restrict__ depth,
    const float* __restrict__ feat, const int32_t* __restrict__ vox, float* __restrict__ d_depth,
    float* __restrict__ d_feat, long n_pix, int D, int HW, int C) {
  const int lane = threadIdx.x & 63;
  const long px = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (px >= n_pix) return;
  const int bn = (int)(px / HW), hw = (int)(px - (long)bn * HW);
  const float* fr = feat + px * C;
  float* df = d_feat + px * C;
  for (int c0 = 0; c0 < C; c0 += 256) {                      // 4 channels per lane and pass (C = 128: one pass)
    float f[4], acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + lane + 64 * e;
      f[e] = c < C ? fr[c] : 0.f;
      acc[e] = 0.f;
    }
    for (int d = 0; d < D; ++d) {
      const long p = ((long)bn * D + d) * HW + hw;
      const int v = vox[p];                                   // (wave-uniform)
      float dd = 0.f;
      if (v >= 0) {
        const float dp = depth[p];
        const float* g = out_grad + (long)v * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + lane + 64 * e;
          const float gv = c < C ? g[c] : 0.f;
          dd += gv * f[e];
          acc[e] += dp * gv;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) dd += __shfl_xor(dd, o);
      if (lane == 0) {
        if (c0 == 0) d_depth[p] = dd;
        else d_depth[p] += dd;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = c0 + lane + 64 * e;
      if (c < C) df[c] = acc[e];
    }
  }
}

extern "C" int occf_lift_splat_bwd(const float* out_grad, const float* depth, const float* feat,
                                   const int32_t* vox, float* d_depth, float* d_feat, long n_pts,
                                   int BN, int D, int HW, int C, void* stream) {
  if (n_pts <= 0 || BN <= 0 || D <= 0 || HW <= 0 || C <= 0) return OCCF_EINVAL;
  if (n_pts != (long)BN * D * HW) return OCCF_EINVAL;       // (vox / depth cover every frustum point)
  hipStream_t st = (hipStream_t)stream;
  const long n_pix = (long)BN * HW;
  hipLaunchKernelGGL(lift_splat_bwd_kernel, dim3(occf_cdiv(n_pix * 64, 256)), dim3(256), 0, st,
                     out_grad, depth, feat, vox, d_depth, d_feat, n_pix, D, HW, C);
  OCCF_LAUNCH_CHECK();
}
