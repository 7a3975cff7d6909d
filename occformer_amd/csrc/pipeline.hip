// Input-pipeline and evaluation ops on the device: the two per-sample stages of the reference that
// run on the HOST between the data loader and the model / after it --
//   * CreateDepthFromLiDAR  (projects/mmdet3d_plugin/datasets/pipelines/lidar2depth.py:15-87): LiDAR points projected
//     into every camera, nearest return per pixel -> the sparse depth maps gt_depths [N, H, W] that supervise DepthNet;
//   * SSCMetrics.update     (projects/mmdet3d_plugin/utils/ssc_metric.py:62-175): completion TP / FP / FN and the
//     per-class TP / FP / FN of the semantic-scene-completion score from the predicted and target label volumes
//     (apis/test.py:64-67: y_pred = argmax(output_voxels, 1)) --
// so that an iteration (training or evaluation) has no CPU stage.  Integer / index work: bit-exact.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

__device__ __forceinline__ float pl_dot3(const float* m, float x, float y, float z) {
  return occf_fadd(occf_fadd(occf_fmul(m[0], x), occf_fmul(m[1], y)), occf_fmul(m[2], z));
}

// one thread per (point, camera).  cam: 32 floats per camera =
//   inv_rots[9] (row-major) | trans[3] | intr[16]: 3x3 row-major in the first 9 (kitti = 0) or 4x4 row-major
//   (kitti = 1) | post_rots[0:2, 0:2] (4) | post_trans[0:2] (2)  -> 9 + 3 + 16 + 4 = 32 (post_trans overlays the unused
//   tail of intr for the 3x3 form; see the host packer)
// Arithmetic in the reference's order (lidar2depth.py:21-41; un-fused multiplies and adds, ATen's small-matmul order):
//   c = inv_rots @ (p - trans); q = K @ c (K @ [c, 1] for the 4x4 form); d = q.z; uv = q.xy / d;
//   uv' = post_rots[:2, :2] @ uv + post_trans[:2]
// valid (:59-63): 0 <= u' <= W - 1, 0 <= v' <= H - 1, d > 0 on the UNROUNDED pixel; the pixel is round-half-even of
// (v', u') (:75-76).  The reference sorts by descending depth and lets the last write win = the MINIMUM depth per
// pixel: an unsigned atomicMin on the bit pattern (positive floats order like unsigned integers); keys start at
// 0xFFFFFFFF and the finalize pass turns untouched pixels into 0.
struct PlCam { float v[36]; };

__global__ void __launch_bounds__(256) lidar_depth_scatter_kernel(const float* __restrict__ pts, long pts_ld,
                                                                  const float* __restrict__ cam, unsigned* __restrict__ key,
                                                                  long P, int N, int H, int W, int kitti) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P * N) return;
  const long p = gid / N;
  const int n = (int)(gid % N);
  const float* cm = cam + (long)n * 36;
  const float* pt = pts + p * pts_ld;
  const float x = occf_fadd(pt[0], -cm[9]), y = occf_fadd(pt[1], -cm[10]), z = occf_fadd(pt[2], -cm[11]);
  const float cx = pl_dot3(cm + 0, x, y, z), cy = pl_dot3(cm + 3, x, y, z), cz = pl_dot3(cm + 6, x, y, z);
  float qx, qy, qd;
  const float* K = cm + 12;
  if (kitti) {                     // 4x4 @ [c, 1]: ((k0*cx + k1*cy) + k2*cz) + k3*1
    qx = occf_fadd(pl_dot3(K + 0, cx, cy, cz), occf_fmul(K[3], 1.0f));
    qy = occf_fadd(pl_dot3(K + 4, cx, cy, cz), occf_fmul(K[7], 1.0f));
    qd = occf_fadd(pl_dot3(K + 8, cx, cy, cz), occf_fmul(K[11], 1.0f));
  } else {
    qx = pl_dot3(K + 0, cx, cy, cz);
    qy = pl_dot3(K + 3, cx, cy, cz);
    qd = pl_dot3(K + 6, cx, cy, cz);
  }
  const float u = qx / qd, v = qy / qd;
  const float* R = cm + 28;
  const float* T = cm + 32;
  const float uu = occf_fadd(occf_fadd(occf_fmul(R[0], u), occf_fmul(R[1], v)), T[0]);
  const float vv = occf_fadd(occf_fadd(occf_fmul(R[2], u), occf_fmul(R[3], v)), T[1]);
  // (comparisons are false for NaN: a point at d == 0 is dropped as in the reference)
  const bool ok = uu >= 0.f && vv >= 0.f && uu <= (float)(W - 1) && vv <= (float)(H - 1) && qd > 0.f;
  if (!ok) return;
  const int ui = (int)rintf(uu), vi = (int)rintf(vv);                  // round half to even = torch.round
  atomicMin(key + ((long)n * H + vi) * W + ui, occf_f2u(qd));
}

__global__ void __launch_bounds__(256) lidar_depth_finalize_kernel(const unsigned* __restrict__ key,
                                                                   float* __restrict__ out, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned k = key[i];
  out[i] = k == 0xFFFFFFFFu ? 0.f : occf_u2f(k);
}

__global__ void __launch_bounds__(256) pl_fill_u32_kernel(unsigned* __restrict__ p, long n, unsigned v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

extern "C" int occf_lidar_depth_fwd(const float* points, long points_ld, const float* cam, float* gt_depths,
                                    uint32_t* workspace, long P, int N, int H, int W, int kitti, void* stream) {
  if (P < 0 || N <= 0 || H <= 0 || W <= 0 || points_ld < 3) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)N * H * W;
  hipLaunchKernelGGL(pl_fill_u32_kernel, dim3(occf_cdiv(n, 256)), dim3(256), 0, st, workspace, n, 0xFFFFFFFFu);
  if (P > 0)
    hipLaunchKernelGGL(lidar_depth_scatter_kernel, dim3(occf_cdiv(P * N, 256)), dim3(256), 0, st, points, points_ld, cam,
                       workspace, P, N, H, W, kitti);
  hipLaunchKernelGGL(lidar_depth_finalize_kernel, dim3(occf_cdiv(n, 256)), dim3(256), 0, st, workspace, gt_depths, n);
  OCCF_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------- SSC confusion
// counts[C*C + 3] (int64, ACCUMULATED: the caller keeps them across update() calls): conf[t * C + p] over the voxels
// of the semantic mask, then completion (tp, fp, fn) over the completion mask.  Per voxel, as ssc_metric.py does it
// (including what its in-place edits amount to): ignore = (target == 255); (t, p) = ignore ? (0, 0) : (target, pred);
//   completion mask = !ignore & nonempty & nonsurface  (:64-70, 108-137): tp / fp / fn of (t > 0) vs (p > 0);
//   semantic mask   = nonempty                          (:75-82, 139-175: `y_true != 255` is re-evaluated AFTER
//                     get_score_completion has overwritten the ignored targets with 0, so the ignored voxels are
//                     counted as class-0 hits unless `nonempty` removes them).
// pred: int64 labels [B*V], or (scores != NULL) the first arg-max over C class scores [B, C, V] (apis/test.py:64).
// One LDS histogram per workgroup (integer LDS atomics), one int64 global atomic per non-zero bin: order-independent.
__global__ void __launch_bounds__(256) ssc_confusion_kernel(const long* __restrict__ pred, const float* __restrict__ scores,
                                                            const uint8_t* __restrict__ target,
                                                            const uint8_t* __restrict__ nonempty,
                                                            const uint8_t* __restrict__ nonsurface,
                                                            unsigned long long* __restrict__ counts, long BV, long V, int C) {
  __shared__ unsigned hist[32 * 32 + 3];
  for (int i = threadIdx.x; i < C * C + 3; i += blockDim.x) hist[i] = 0u;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < BV; i += (long)gridDim.x * blockDim.x) {
    int t = target[i];
    int p;
    if (scores) {
      const long b = i / V, v = i - b * V;
      const float* s = scores + b * C * V + v;
      float best = s[0];
      p = 0;
      for (int c = 1; c < C; ++c) {
        const float sv = s[(long)c * V];
        if (sv > best || (sv != sv && best == best)) { best = sv; p = c; }      // first maximum; NaN wins (torch.argmax)
      }
    } else {
      p = (int)pred[i];
    }
    const bool ignore = t == 255;
    if (ignore) { t = 0; p = 0; }
    const bool ne = nonempty ? nonempty[i] != 0 : true;
    const bool ns = nonsurface ? nonsurface[i] != 0 : true;
    if (!ignore && ne && ns) {
      const bool bt = t > 0, bp = p > 0;
      if (bt && bp) atomicAdd(&hist[C * C + 0], 1u);
      else if (!bt && bp) atomicAdd(&hist[C * C + 1], 1u);
      else if (bt && !bp) atomicAdd(&hist[C * C + 2], 1u);
    }
    if (ne && (unsigned)t < (unsigned)C && (unsigned)p < (unsigned)C) atomicAdd(&hist[t * C + p], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < C * C + 3; i += blockDim.x)
    if (hist[i]) atomicAdd(counts + i, (unsigned long long)hist[i]);
}

extern "C" int occf_ssc_confusion_fwd(const int64_t* pred, const float* scores, const uint8_t* target,
                                      const uint8_t* nonempty, const uint8_t* nonsurface, int64_t* counts, long B, long V,
                                      int C, void* stream) {
  if (B <= 0 || V <= 0 || C <= 0 || C > 32 || (!pred && !scores)) return OCCF_EINVAL;
  const long BV = B * V;
  // <= 2^32 - 1 voxels per workgroup histogram bin: 1024 workgroups x 256 threads, grid-stride
  long wgs = (BV + 256L * 16 - 1) / (256L * 16);
  if (wgs > 2048) wgs = 2048;
  if (wgs < 1) wgs = 1;
  hipLaunchKernelGGL(ssc_confusion_kernel, dim3((unsigned)wgs), dim3(256), 0, (hipStream_t)stream, (const long*)pred,
                     scores, target, nonempty, nonsurface, (unsigned long long*)counts, BV, V, C);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------
// img_inputs producer: the image half of LoadMultiViewImageFromFiles_OccFormer.img_transform_core + mmlabNormalize
// (datasets/pipelines/loading_nusc_imgs.py:57-64, 179-193) on decoded uint8 frames.  The reference runs PIL inside the
// data-loader workers: Image.resize (bicubic with antialiasing: separable, coefficients normalised and quantised to
// 22-bit fixed point, a uint8-rounded intermediate between the horizontal and the vertical pass -- Pillow
// src/libImaging/Resample.c), crop (zero outside), horizontal flip, Image.rotate (nearest neighbour about the centre in
// 16.16 fixed point, zero fill -- Geometry.c affine_fixed), then (x - mean) / std with the BGR -> RGB swap.  Restated
// bit for bit (tests compare with Pillow itself); the coefficient tables come from the host (doubles, as in Pillow).
#define OCCF_PIL_PRECISION_BITS 22

// one pass of the separable resampling: in [H][W][C] uint8 -> horizontal: [H][out_size][C], vertical: [out_size][W][C];
// bounds [out_size][2] = (first tap, tap count), kk [out_size][ksize] int32 coefficients
__global__ void __launch_bounds__(256) image_resample_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out,
                                                             const int32_t* __restrict__ bounds,
                                                             const int32_t* __restrict__ kk, int ksize, int H, int W,
                                                             int C, int out_size, int vertical) {
  const long total = vertical ? (long)out_size * W * C : (long)H * out_size * C;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int c = (int)(gid % C);
  long r = gid / C;
  int o, fixed;                                           // output index along the resampled axis, index along the other
  if (vertical) { fixed = (int)(r % W); o = (int)(r / W); } else { o = (int)(r % out_size); fixed = (int)(r / out_size); }
  const int first = bounds[o * 2 + 0], n = bounds[o * 2 + 1];
  const int32_t* k = kk + (long)o * ksize;
  int acc = 1 << (OCCF_PIL_PRECISION_BITS - 1);
  for (int t = 0; t < n; ++t) {
    const long src = vertical ? ((long)(first + t) * W + fixed) * C + c : ((long)fixed * W + first + t) * C + c;
    acc += (int)in[src] * k[t];
  }
  int v = acc >> OCCF_PIL_PRECISION_BITS;                 // clip8
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  out[gid] = (uint8_t)v;
}

extern "C" int occf_image_resample_fwd(const uint8_t* in, uint8_t* out, const int32_t* bounds, const int32_t* kk,
                                       int ksize, int H, int W, int C, int out_size, int vertical, void* stream) {
  if (H <= 0 || W <= 0 || C <= 0 || out_size <= 0 || ksize <= 0) return OCCF_EINVAL;
  const long total = vertical ? (long)out_size * W * C : (long)H * out_size * C;
  hipLaunchKernelGGL(image_resample_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, in, out, bounds,
                     kk, ksize, H, W, C, out_size, vertical);
  OCCF_LAUNCH_CHECK();
}

// crop (cx0, cy0, fW x fH; zero outside the resized frame) -> flip -> rotate (mode 0 none, 1 = 180 degrees, 2 = the
// 16.16 fixed-point affine map a[0..5]: x_in = (a2 + a1 y + a0 x) >> 16, y_in = (a5 + a4 y + a3 x) >> 16, zero outside)
// -> canvas uint8 [fH][fW][3] (optional) and out[c][y][x] = (float(v) - mean[c]) * stdinv[c], v = channel 2 - c if to_rgb
struct ImgXform { long a[6]; float mean[3], stdinv[3]; };
__global__ void __launch_bounds__(256) image_crop_rotate_normalize_kernel(const uint8_t* __restrict__ in,
                                                                          float* __restrict__ out,
                                                                          uint8_t* __restrict__ canvas, int Hn, int Wn,
                                                                          int cx0, int cy0, int fW, int fH, int flip,
                                                                          int rot_mode, int to_rgb, ImgXform t) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)fW * fH) return;
  const int x = (int)(gid % fW), y = (int)(gid / fW);
  long xin = x, yin = y;
  if (rot_mode == 1) {
    xin = fW - 1 - x;
    yin = fH - 1 - y;
  } else if (rot_mode == 2) {
    xin = (t.a[2] + t.a[1] * y + t.a[0] * x) >> 16;
    yin = (t.a[5] + t.a[4] * y + t.a[3] * x) >> 16;
  }
  int px[3] = {0, 0, 0};
  if (xin >= 0 && xin < fW && yin >= 0 && yin < fH) {
    const long sx = cx0 + (flip ? fW - 1 - xin : xin), sy = cy0 + yin;
    if (sx >= 0 && sx < Wn && sy >= 0 && sy < Hn) {
      const uint8_t* p = in + (sy * Wn + sx) * 3;
      px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
    }
  }
  if (canvas) {
    uint8_t* q = canvas + gid * 3;
    q[0] = (uint8_t)px[0]; q[1] = (uint8_t)px[1]; q[2] = (uint8_t)px[2];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)px[to_rgb ? 2 - c : c];
    out[(long)c * fW * fH + gid] = occf_fmul(occf_fadd(v, -t.mean[c]), t.stdinv[c]);
  }
}

extern "C" int occf_image_crop_rotate_normalize_fwd(const uint8_t* in, float* out, uint8_t* canvas, int Hn, int Wn,
                                                    int cx0, int cy0, int fW, int fH, int flip, int rot_mode,
                                                    const int64_t* affine, const float* mean, const float* stdinv,
                                                    int to_rgb, void* stream) {
  if (Hn <= 0 || Wn <= 0 || fW <= 0 || fH <= 0 || rot_mode < 0 || rot_mode > 2 || !mean || !stdinv ||
      (rot_mode == 2 && !affine))
    return OCCF_EINVAL;
  ImgXform t;
  for (int i = 0; i < 6; ++i) t.a[i] = rot_mode == 2 ? (long)affine[i] : 0;
  for (int i = 0; i < 3; ++i) { t.mean[i] = mean[i]; t.stdinv[i] = stdinv[i]; }
  hipLaunchKernelGGL(image_crop_rotate_normalize_kernel, dim3(occf_cdiv((long)fW * fH, 256)), dim3(256), 0,
                     (hipStream_t)stream, in, out, canvas, Hn, Wn, cx0, cy0, fW, fH, flip, rot_mode, to_rgb, t);
  OCCF_LAUNCH_CHECK();
}
