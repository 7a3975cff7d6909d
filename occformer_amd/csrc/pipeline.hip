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
