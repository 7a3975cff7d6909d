// Backward kernels of the HBM-bound ops of the path (training step): LayerNorm, GroupNorm
// (+ReLU, + token-buffer build, + residual), activations, stochastic depth, the soft-gated dual-path
// fusion, the FPN upsample+add, point sampling and the point-loss row sums, plus the column sums that are
// the bias gradients of every linear / convolution.
//
// What the reference runs here is ATen autograd through nn.LayerNorm / nn.GroupNorm / F.gelu / DropPath
// (window_attention.py:311,332,352-361), dualpath_block.py:65-82, F.interpolate (multiscale_deformattn_3d.py
// :238-243), F.grid_sample (mmdet_utils.py:21-47) and the BCE / Dice losses (mask2former_nusc_occ.py:396-417).
// All channels-last, 16-byte accesses, deterministic two-stage reductions for the parameter gradients
// (no float atomics except the point scatter, whose reference counterpart -- grid_sample's backward -- is
// an atomic scatter as well).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define BE_MAXV 4   // float4 per lane: 64 lanes * 4 * 4 = 1024 channels

// ======================================================================================= column sums
// out[N] = sum_m x[m, n]  (bias gradients).  Stage 1: one workgroup per row block, threads = channel quads x
// row threads; stage 2: fixed-order sum of the per-block partials in double.
__global__ void __launch_bounds__(256) colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                             long M, int N, long ldx, int rows_per_block) {
  __shared__ float s[256][4];
  const int Q = (N + 3) / 4;
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  const bool vec = (N % 4 == 0) && (ldx % 4 == 0);
  for (int q0 = 0; q0 < Q; q0 += 256) {
    // layout of a pass: QQ = min(Q - q0, 256) quads, R = 256 / QQ row threads
    const int QQ = Q - q0 < 256 ? Q - q0 : 256;
    const int R = 256 / QQ;
    const int cq = tid % QQ, rt = tid / QQ;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (rt < R) {
      const int c0 = (q0 + cq) * 4;
      if (vec) {
#pragma unroll 4
        for (long r = r0 + rt; r < r1; r += R) {
          const float4 v = *(const float4*)(x + r * ldx + c0);
          a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
        }
      } else {
        for (long r = r0 + rt; r < r1; r += R)
          for (int e = 0; e < 4; ++e)
            if (c0 + e < N) a[e] += x[r * ldx + c0 + e];
      }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) s[tid][e] = a[e];
    __syncthreads();
    if (tid < QQ) {
      float t[4] = {0.f, 0.f, 0.f, 0.f};
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) t[e] += s[r * QQ + tid][e];
      for (int e = 0; e < 4; ++e)
        if ((q0 + tid) * 4 + e < N) partial[(long)blockIdx.x * N + (q0 + tid) * 4 + e] = t[e];
    }
  }
}

// out[c] = sum_k partial[k * ld + c * stride + offset] (double accumulation, fixed order).  64 columns x 16 row
// groups per workgroup: coalesced loads across the columns, 16 independent partial sums per column in flight
// (a one-thread-per-column loop over ~4000 partial rows cost 0.2 ms per call and 30 ms per training step);
// blockIdx.y = batch element (strides in_bs / out_bs).
__global__ void __launch_bounds__(1024) reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               long nblk, int C, long ld, int stride, int offset,
                                                               long in_bs, long out_bs, float* __restrict__ out2) {
  // out2 != NULL: the C columns are (a, b) pairs -- even columns go to out[c / 2], odd ones to out2[c / 2] (dgamma and
  // dbeta of a LayerNorm from one launch instead of two)
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  const float* p = partial + (long)blockIdx.y * in_bs;
  double s = 0.0;
  if (c < C) {
    long k = ty;
    for (; k + 48 < nblk; k += 64) {
      const float v0 = p[k * ld + (long)c * stride + offset], v1 = p[(k + 16) * ld + (long)c * stride + offset];
      const float v2 = p[(k + 32) * ld + (long)c * stride + offset], v3 = p[(k + 48) * ld + (long)c * stride + offset];
      s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (; k < nblk; k += 16) s += (double)p[k * ld + (long)c * stride + offset];
  }
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][tx];
    if (out2) ((c & 1) ? out2 : out)[(long)blockIdx.y * out_bs + (c >> 1)] = (float)t;
    else out[(long)blockIdx.y * out_bs + c] = (float)t;
  }
}
static void occf_reduce_partials(const float* partial, float* out, long nblk, int C, long ld, int stride, int offset,
                                 hipStream_t st, int batch = 1, long in_bs = 0, long out_bs = 0, float* out2 = nullptr) {
  hipLaunchKernelGGL(reduce_partials_kernel, dim3(occf_cdiv(C, 64), batch), dim3(1024), 0, st, partial, out, nblk, C, ld,
                     stride, offset, in_bs, out_bs, out2);
}

extern "C" long occf_colsum_workspace(long M, int N) { return (long)occf_cdiv(M, 512) * N; }

extern "C" int occf_colsum(const float* x, float* out, float* workspace, long M, int N, long ldx, void* stream) {
  if (M <= 0 || N <= 0 || ldx < N) return OCCF_EINVAL;
  const int rows = 512;
  const int nblk = occf_cdiv(M, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk), dim3(256), 0, st, x, workspace, M, N, ldx, rows);
  occf_reduce_partials(workspace, out, (long)nblk, N, (long)N, 1, 0, st);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= LayerNorm backward
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  dgamma = sum dy * xhat, dbeta = sum dy.
// One wave per row (statistics recomputed from x), a wave walks `rows_per_wave` rows and keeps the parameter
// gradients of its channels in registers; workgroup partials -> partial[block][C][2].
__global__ void __launch_bounds__(256) layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ dy,
                                                            const float* __restrict__ addend, float* __restrict__ dx,
                                                            float* __restrict__ partial, long M, int C, float eps,
                                                            int rows_per_wave) {
  __shared__ float red[4][1024][2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Q = C / 4;
  const long w0 = ((long)blockIdx.x * 4 + wave) * rows_per_wave;
  float4 gm[BE_MAXV];
  float dg[BE_MAXV][4], db[BE_MAXV][4];
#pragma unroll
  for (int i = 0; i < BE_MAXV; ++i) {
    const int q = lane + i * 64;
    gm[i] = q < Q ? *(const float4*)(gamma + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[i][e] = db[i][e] = 0.f;
  }
  for (int rr = 0; rr < rows_per_wave; ++rr) {
    const long row = w0 + rr;
    if (row >= M) break;
    float4 v[BE_MAXV], g[BE_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      const int q = lane + i * 64;
      if (q < Q) {
        v[i] = *(const float4*)(x + row * C + q * 4);
        g[i] = *(const float4*)(dy + row * C + q * 4);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
      } else {
        v[i] = g[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      if (lane + i * 64 < Q) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q2 += (a * a + b * b) + (c * c + d * d);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    const float rstd = 1.0f / sqrtf(q2 / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    float xh[BE_MAXV][4], gg[BE_MAXV][4];
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      const float xv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      const float dv[4] = {g[i].x, g[i].y, g[i].z, g[i].w};
      const float gv[4] = {gm[i].x, gm[i].y, gm[i].z, gm[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[i][e] = (lane + i * 64 < Q) ? (xv[e] - mean) * rstd : 0.f;
        gg[i][e] = dv[e] * gv[e];
        s1 += gg[i][e];
        s2 = fmaf(gg[i][e], xh[i][e], s2);
        dg[i][e] = fmaf(dv[e], xh[i][e], dg[i][e]);
        db[i][e] += dv[e];
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    s1 /= (float)C;
    s2 /= (float)C;
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      const int q = lane + i * 64;
      if (q < Q) {
        float4 o = make_float4(rstd * (gg[i][0] - s1 - xh[i][0] * s2), rstd * (gg[i][1] - s1 - xh[i][1] * s2),
                               rstd * (gg[i][2] - s1 - xh[i][2] * s2), rstd * (gg[i][3] - s1 - xh[i][3] * s2));
        if (addend) {
          const float4 a = *(const float4*)(addend + row * C + q * 4);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *(float4*)(dx + row * C + q * 4) = o;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BE_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        red[wave][q * 4 + e][0] = dg[i][e];
        red[wave][q * 4 + e][1] = db[i][e];
      }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float* o = partial + ((long)blockIdx.x * C + c) * 2;
    o[0] = (red[0][c][0] + red[1][c][0]) + (red[2][c][0] + red[3][c][0]);
    o[1] = (red[0][c][1] + red[1][c][1]) + (red[2][c][1] + red[3][c][1]);
  }
}


// C = 64 * NV (128 / 192 / 256): 16 lanes per row, NV float4 per lane, 4 rows per wave at a time and `iters` such
// groups per wave -- every lane busy, NV x 2 independent 16-byte loads in flight per lane (the one-wave-per-row
// kernel above walks its rows serially with half of the lanes idle at C = 128: 0.4 TB/s on 680 000 x 128)
template <int NV>
__global__ void __launch_bounds__(256) layernorm16_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ dy,
                                                              const float* __restrict__ addend, float* __restrict__ dx,
                                                              float* __restrict__ partial, long M, float eps, int iters) {
  constexpr int C = 64 * NV;
  __shared__ float red[16][C][2];
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;          // 16 row groups per workgroup
  float4 gm[NV];
  float dg[NV][4], db[NV][4];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    gm[j] = *(const float4*)(gamma + (sub + 16 * j) * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) dg[j][e] = db[j][e] = 0.f;
  }
  const long row0 = ((long)blockIdx.x * 16 + grp) * iters;
  for (int it = 0; it < iters; ++it) {
    const long row = row0 + it;
    const bool ok = row < M;
    const long rc = ok ? row : M - 1;
    float4 v[NV], g[NV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      v[j] = *(const float4*)(x + rc * C + (sub + 16 * j) * 4);
      g[j] = *(const float4*)(dy + rc * C + (sub + 16 * j) * 4);
      s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q2 = 0.f;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q2 += (a * a + b * b) + (c * c + d * d);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
    const float rstd = 1.0f / sqrtf(q2 / (float)C + eps);
    float s1 = 0.f, s2 = 0.f;
    float xh[NV][4], gg[NV][4];
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const float xv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
      const float dv[4] = {g[j].x, g[j].y, g[j].z, g[j].w};
      const float gv[4] = {gm[j].x, gm[j].y, gm[j].z, gm[j].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[j][e] = (xv[e] - mean) * rstd;
        gg[j][e] = dv[e] * gv[e];
        s1 += gg[j][e];
        s2 = fmaf(gg[j][e], xh[j][e], s2);
        if (ok) {
          dg[j][e] = fmaf(dv[e], xh[j][e], dg[j][e]);
          db[j][e] += dv[e];
        }
      }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      s1 += __shfl_xor(s1, o);
      s2 += __shfl_xor(s2, o);
    }
    s1 /= (float)C;
    s2 /= (float)C;
    if (ok) {
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        float4 o = make_float4(rstd * (gg[j][0] - s1 - xh[j][0] * s2), rstd * (gg[j][1] - s1 - xh[j][1] * s2),
                               rstd * (gg[j][2] - s1 - xh[j][2] * s2), rstd * (gg[j][3] - s1 - xh[j][3] * s2));
        if (addend) {
          const float4 a = *(const float4*)(addend + row * C + (sub + 16 * j) * 4);
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        *(float4*)(dx + row * C + (sub + 16 * j) * 4) = o;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NV; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      red[grp][(sub + 16 * j) * 4 + e][0] = dg[j][e];
      red[grp][(sub + 16 * j) * 4 + e][1] = db[j][e];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      a += red[r][c][0];
      b += red[r][c][1];
    }
    float* o = partial + ((long)blockIdx.x * C + c) * 2;
    o[0] = a;
    o[1] = b;
  }
}

static int occf_ln16_iters(long M) {
  long it = (M + 16L * 1024 - 1) / (16L * 1024);     // ~1024 workgroups (4 per CU), 1024 partial rows to reduce
  return it < 1 ? 1 : (int)it;
}

static int occf_ln_rows_per_wave(long M) {
  long r = (M + 4 * 1024 - 1) / (4 * 1024);   // ~1024 workgroups
  return r < 1 ? 1 : (int)r;
}
extern "C" long occf_layernorm_bwd_workspace(long M, int C) {
  if (C == 128 || C == 192 || C == 256) return (long)occf_cdiv(M, 16L * occf_ln16_iters(M)) * C * 2;
  const int rpw = occf_ln_rows_per_wave(M);
  return (long)occf_cdiv(M, 4L * rpw) * C * 2;
}

extern "C" int occf_layernorm_bwd(const float* x, const float* gamma, const float* dy, const float* addend, float* dx,
                                  float* dgamma, float* dbeta, float* workspace, long M, int C, float eps,
                                  void* stream) {
  if (M <= 0 || C % 4 != 0 || C > 1024) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  int nblk;
  if (C == 128 || C == 192 || C == 256) {
    const int iters = occf_ln16_iters(M);
    nblk = occf_cdiv(M, 16L * iters);
    if (C == 128) hipLaunchKernelGGL(layernorm16_bwd_kernel<2>, dim3(nblk), dim3(256), 0, st, x, gamma, dy, addend, dx, workspace, M, eps, iters);
    else if (C == 192) hipLaunchKernelGGL(layernorm16_bwd_kernel<3>, dim3(nblk), dim3(256), 0, st, x, gamma, dy, addend, dx, workspace, M, eps, iters);
    else hipLaunchKernelGGL(layernorm16_bwd_kernel<4>, dim3(nblk), dim3(256), 0, st, x, gamma, dy, addend, dx, workspace, M, eps, iters);
  } else {
    const int rpw = occf_ln_rows_per_wave(M);
    nblk = occf_cdiv(M, 4L * rpw);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), 0, st, x, gamma, dy, addend, dx, workspace, M, C, eps, rpw);
  }
  occf_reduce_partials(workspace, dgamma, (long)nblk, 2 * C, (long)C * 2, 1, 0, st, 1, 0, 0, dbeta);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= GroupNorm backward
// Forward (occf_groupnorm_apply): y = relu?(xhat * gamma + beta) [+ residual], token mode appends slot Z =
// mean_z(y).  With dyt[z] = dy[z] + dy[Z] / Z (token mode) and g = dyt * [y > 0] (ReLU):
//   A[b, c] = sum_v g,  Bc[b, c] = sum_v g * xhat;   dgamma = sum_b Bc, dbeta = sum_b A
//   s1[b, grp] = sum_{c in grp} gamma_c A[b, c] / n,  s2 = sum gamma_c Bc[b, c] / n   (n = V * C / G)
//   dx = rstd * (g * gamma - s1 - xhat * s2);          d_residual = dyt
// stage 1 (partial sums per row block) / stage 2 (per batch-channel + per group + parameters) / stage 3 (apply).
// IDX: uint32_t row arithmetic whenever B * V * (C / 4) < 2^31 (see droppath_kernel: the token mode divides per row)
template <typename IDX>
__device__ __forceinline__ void gnb_load(const float* __restrict__ x, const float* __restrict__ dy, long b, long V,
                                         IDX r, int Z, int C, int c0, int tokens, float4& xv, float4& g) {
  xv = *(const float4*)(x + (b * V + (long)r) * C + c0);
  if (tokens) {
    const IDX p = r / (IDX)Z;
    const int z = (int)(r - p * (IDX)Z);
    const long Zs = Z + 1;
    const long Vs = (V / Z) * Zs;
    g = *(const float4*)(dy + (b * Vs + (long)p * Zs + z) * C + c0);
    const float4 m = *(const float4*)(dy + (b * Vs + (long)p * Zs + Z) * C + c0);
    const float inv = 1.0f / (float)Z;
    g.x = fmaf(m.x, inv, g.x); g.y = fmaf(m.y, inv, g.y); g.z = fmaf(m.z, inv, g.z); g.w = fmaf(m.w, inv, g.w);
  } else {
    g = *(const float4*)(dy + (b * V + (long)r) * C + c0);
  }
}

template <typename IDX>
__global__ void __launch_bounds__(256) gn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ dy,
                                                             float* __restrict__ partial, long V, int Z, int C, int G,
                                                             int relu, int tokens, int rows_per_block) {
  __shared__ float sA[256][4], sB[256][4];
  const int Q = C / 4;
  const int R = 256 / Q;
  const int tid = threadIdx.x;
  const int cq = tid % Q, rt = tid / Q;
  const int b = blockIdx.y;
  const IDX r0 = (IDX)blockIdx.x * (IDX)rows_per_block;
  IDX r1 = r0 + (IDX)rows_per_block;
  if (r1 > (IDX)V) r1 = (IDX)V;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, bb[4] = {0.f, 0.f, 0.f, 0.f};
  if (rt < R) {
    const int cg = C / G, c0 = cq * 4;
    float mean[4], rstd[4], gm[4], bt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float* s = stats + ((long)b * G + (c0 + e) / cg) * 2;
      mean[e] = s[0];
      rstd[e] = s[1];
      gm[e] = gamma[c0 + e];
      bt[e] = beta[c0 + e];
    }
#pragma unroll 4
    for (IDX r = r0 + (IDX)rt; r < r1; r += (IDX)R) {
      float4 xv, g;
      gnb_load<IDX>(x, dy, b, V, r, Z, C, c0, tokens, xv, g);
      const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
      const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float xh = (xx[e] - mean[e]) * rstd[e];
        // the ReLU mask from the forward's own expression y = fma(x, rstd*gamma, beta - mean*rstd*gamma)
        const float sc = rstd[e] * gm[e];
        const float ge = (relu && fmaf(xx[e], sc, bt[e] - mean[e] * sc) <= 0.f) ? 0.f : gg[e];
        a[e] += ge;
        bb[e] = fmaf(ge, xh, bb[e]);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { sA[tid][e] = a[e]; sB[tid][e] = bb[e]; }
  __syncthreads();
  if (tid < Q) {
    float ta[4] = {0.f, 0.f, 0.f, 0.f}, tb[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 4; ++e) { ta[e] += sA[r * Q + tid][e]; tb[e] += sB[r * Q + tid][e]; }
    float* o = partial + (((long)b * gridDim.x + blockIdx.x) * C + tid * 4) * 2;
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[e * 2] = ta[e]; o[e * 2 + 1] = tb[e]; }
  }
}

// chan -> group sums gs[B][G][2] = (s1, s2) (already divided by n) and dgamma / dbeta
__global__ void __launch_bounds__(256) gn_bwd_groups_kernel(const float* __restrict__ chan, const float* __restrict__ gamma,
                                                            float* __restrict__ gs, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int B, int C, int G, double n) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B * G) {
    const int b = t / G, g = t % G, cg = C / G;
    double s1 = 0.0, s2 = 0.0;
    for (int c = g * cg; c < (g + 1) * cg; ++c) {
      s1 += (double)gamma[c] * (double)chan[((long)b * C + c) * 2];
      s2 += (double)gamma[c] * (double)chan[((long)b * C + c) * 2 + 1];
    }
    gs[(long)t * 2] = (float)(s1 / n);
    gs[(long)t * 2 + 1] = (float)(s2 / n);
  }
  if (t < C) {
    double a = 0.0, q = 0.0;
    for (int b = 0; b < B; ++b) {
      a += (double)chan[((long)b * C + t) * 2];
      q += (double)chan[((long)b * C + t) * 2 + 1];
    }
    dbeta[t] = (float)a;
    dgamma[t] = (float)q;
  }
}

template <typename IDX>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dy, const float* __restrict__ gs,
                                                           float* __restrict__ dx, float* __restrict__ dres, int B, long V,
                                                           int Z, int C, int G, int relu, int tokens) {
  const int Q = C / 4;
  const IDX gid = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x;
  if (gid >= (IDX)((long)B * V * Q)) return;
  const IDX br = gid / (IDX)Q;
  const int cq = (int)(gid - br * (IDX)Q);
  const IDX bq = B == 1 ? (IDX)0 : br / (IDX)V;
  const IDX r = br - bq * (IDX)V;
  const long b = (long)bq;
  const int c0 = cq * 4, cg = C / G;
  float4 xv, g;
  gnb_load<IDX>(x, dy, b, V, r, Z, C, c0, tokens, xv, g);
  if (dres) *(float4*)(dres + (b * V + (long)r) * C + c0) = g;
  const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
  const float gg[4] = {g.x, g.y, g.z, g.w};
  float o[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int grp = (c0 + e) / cg;
    const float* s = stats + (b * G + grp) * 2;
    const float* t = gs + (b * G + grp) * 2;
    const float gm = gamma[c0 + e];
    const float xh = (xx[e] - s[0]) * s[1];
    const float sc = s[1] * gm;
    const float ge = (relu && fmaf(xx[e], sc, beta[c0 + e] - s[0] * sc) <= 0.f) ? 0.f : gg[e];
    o[e] = s[1] * (ge * gm - t[0] - xh * t[1]);
  }
  *(float4*)(dx + (b * V + (long)r) * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
}

// The same pass with the thread layout of the partial kernel: a workgroup owns `rows_per_block` rows, a thread ONE channel
// quad of every R-th row -- its 4 x (mean, rstd, s1, s2, gamma, beta) are loaded once instead of once per float4 (the
// thread-per-float4 form above issues 24 parameter loads for 3 payload accesses), and 4 rows of loads are in flight.
__global__ void __launch_bounds__(256) gn_bwd_apply_rows_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ dy,
                                                                const float* __restrict__ gs, float* __restrict__ dx,
                                                                float* __restrict__ dres, long V, int Z, int C, int G,
                                                                int relu, int tokens, int rows_per_block) {
  const int Q = C / 4;
  const int R = 256 / Q;
  const int tid = threadIdx.x;
  const int cq = tid % Q, rt = tid / Q;
  if (rt >= R) return;
  const long b = blockIdx.y;
  const uint32_t r0 = blockIdx.x * (uint32_t)rows_per_block;
  uint32_t r1 = r0 + (uint32_t)rows_per_block;
  if (r1 > (uint32_t)V) r1 = (uint32_t)V;
  const int cg = C / G, c0 = cq * 4;
  float mean[4], rstd[4], gm[4], sh[4], t0[4], t1[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int grp = (c0 + e) / cg;
    const float* s = stats + (b * G + grp) * 2;
    const float* t = gs + (b * G + grp) * 2;
    mean[e] = s[0];
    rstd[e] = s[1];
    gm[e] = gamma[c0 + e];
    sh[e] = beta[c0 + e] - s[0] * (s[1] * gm[e]);
    t0[e] = t[0];
    t1[e] = t[1];
  }
#pragma unroll 4
  for (uint32_t r = r0 + (uint32_t)rt; r < r1; r += (uint32_t)R) {
    float4 xv, g;
    gnb_load<uint32_t>(x, dy, b, V, r, Z, C, c0, tokens, xv, g);
    if (dres) *(float4*)(dres + (b * V + (long)r) * C + c0) = g;
    const float xx[4] = {xv.x, xv.y, xv.z, xv.w};
    const float gg[4] = {g.x, g.y, g.z, g.w};
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float xh = (xx[e] - mean[e]) * rstd[e];
      const float ge = (relu && fmaf(xx[e], rstd[e] * gm[e], sh[e]) <= 0.f) ? 0.f : gg[e];
      o[e] = rstd[e] * (ge * gm[e] - t0[e] - xh * t1[e]);
    }
    *(float4*)(dx + (b * V + (long)r) * C + c0) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

#define GNB_ROWS 256
// rows per workgroup of the partial-sum pass: 256 at the full-resolution volumes, fewer where 256-row blocks would
// leave the chip empty -- a [1 250, 1 024] GroupNorm was 5 workgroups of 256 SERIAL row steps per thread (56 us), a
// [10 000, 512] one 40 workgroups: the 80 calls of a training step averaged 47 us (r06v kernel statistics), most of it
// latency.  Aim at ~1 000 workgroups, at least four row steps per thread.
static int gnb_rows(int B, long V, int C) {
  const int R = 256 / (C / 4) > 0 ? 256 / (C / 4) : 1;         // rows a workgroup covers per step
  long rows = occf_cdiv(V * B, 1024L);
  rows = occf_cdiv(rows, (long)R) * R;
  if (rows < 4L * R) rows = 4L * R;
  return (int)(rows > GNB_ROWS ? GNB_ROWS : rows);
}
extern "C" long occf_groupnorm_bwd_workspace(int B, long V, int C, int G) {
  if (C <= 0 || C % 4) return 0;
  return (long)B * occf_cdiv(V, (long)gnb_rows(B, V, C)) * C * 2 + (long)B * C * 2 + (long)B * G * 2;
}

extern "C" int occf_groupnorm_bwd(const float* x, const float* stats, const float* gamma, const float* beta,
                                  const float* dy, float* dx, float* dgamma, float* dbeta, float* dresidual,
                                  float* workspace, int B, long P, int Z, int C, int G, int relu, int tokens,
                                  void* stream) {
  if (B <= 0 || P <= 0 || Z <= 0 || C % 4 != 0 || C > 1024 || G <= 0 || C % G != 0) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long V = P * Z;
  const int rows = gnb_rows(B, V, C);
  const int nblk = (int)occf_cdiv(V, (long)rows);
  const int nblk256 = (int)occf_cdiv(V, (long)GNB_ROWS);        // (the row-walking apply pass keeps 256-row blocks)
  float* partial = workspace;
  float* chan = partial + (long)B * nblk * C * 2;
  float* gs = chan + (long)B * C * 2;
  const bool idx32 = (long)B * V * (C / 4) < 2147483647L - 256 && V + rows < 2147483647L;
  if (idx32)
    hipLaunchKernelGGL(gn_bwd_partial_kernel<uint32_t>, dim3(nblk, B), dim3(256), 0, st, x, stats, gamma, beta, dy, partial,
                       V, Z, C, G, relu, tokens, rows);
  else
    hipLaunchKernelGGL(gn_bwd_partial_kernel<long>, dim3(nblk, B), dim3(256), 0, st, x, stats, gamma, beta, dy, partial, V,
                       Z, C, G, relu, tokens, rows);
  occf_reduce_partials(partial, chan, (long)nblk, 2 * C, (long)C * 2, 1, 0, st, B, (long)nblk * C * 2, (long)C * 2);
  const int tmax = B * G > C ? B * G : C;
  hipLaunchKernelGGL(gn_bwd_groups_kernel, dim3(occf_cdiv(tmax, 256)), dim3(256), 0, st, chan, gamma, gs, dgamma, dbeta,
                     B, C, G, (double)V * (C / G));
  const char* e_rows = getenv("OCCF_GNB_APPLY_ROWS");     // 0: the thread-per-float4 form (read per call: A/B probes)
  const int rows_form = e_rows ? atoi(e_rows) : 1;
  // (the row-walking form needs enough 256-row blocks to fill the chip -- r05o: taken for every shape it cost the
  // mid-size calls, [80 000, 256] and smaller, more than it gave the full-resolution ones)
  if (idx32 && rows_form && C <= 1024 && 256 / (C / 4) >= 1 && (nblk256 >= 1024 || rows_form == 2))  // (2: forced, tests)
    hipLaunchKernelGGL(gn_bwd_apply_rows_kernel, dim3(nblk256, B), dim3(256), 0, st, x, stats, gamma, beta, dy, gs, dx,
                       dresidual, V, Z, C, G, relu, tokens, GNB_ROWS);
  else if (idx32)
    hipLaunchKernelGGL(gn_bwd_apply_kernel<uint32_t>, dim3(occf_cdiv((long)B * V * (C / 4), 256)), dim3(256), 0, st, x,
                       stats, gamma, beta, dy, gs, dx, dresidual, B, V, Z, C, G, relu, tokens);
  else
    hipLaunchKernelGGL(gn_bwd_apply_kernel<long>, dim3(occf_cdiv((long)B * V * (C / 4), 256)), dim3(256), 0, st, x, stats,
                       gamma, beta, dy, gs, dx, dresidual, B, V, Z, C, G, relu, tokens);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= activations
__device__ __forceinline__ float be_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float be_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// act 1 = ReLU, 2 = exact GELU (F.gelu).  fwd: y = act(x); bwd: dx = dy * act'(x)  (ReLU accepts y for x)
__global__ void __launch_bounds__(256) act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = ((const float4*)x)[i];
  float4 o;
  if (act == 1) o = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
  else o = make_float4(be_gelu(v.x), be_gelu(v.y), be_gelu(v.z), be_gelu(v.w));
  ((float4*)y)[i] = o;
}
__global__ void __launch_bounds__(256) act_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ dx, long n4, int act) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 v = ((const float4*)x)[i];
  const float4 g = ((const float4*)dy)[i];
  float4 o;
  if (act == 1) o = make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f);
  else o = make_float4(g.x * be_gelu_grad(v.x), g.y * be_gelu_grad(v.y), g.z * be_gelu_grad(v.z), g.w * be_gelu_grad(v.w));
  ((float4*)dx)[i] = o;
}
extern "C" int occf_act_fwd(const float* x, float* y, long n, int act, void* stream) {
  if (n <= 0 || n % 4 != 0 || (act != 1 && act != 2)) return OCCF_EINVAL;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(occf_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n / 4, act);
  OCCF_LAUNCH_CHECK();
}
extern "C" int occf_act_bwd(const float* x, const float* dy, float* dx, long n, int act, void* stream) {
  if (n <= 0 || n % 4 != 0 || (act != 1 && act != 2)) return OCCF_EINVAL;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(occf_cdiv(n / 4, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n / 4,
                     act);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= stochastic depth
// DropPath of the shared SwinBlock (window_attention.py:311,332; mmcv DropPath: one Bernoulli draw per sample,
// survivors scaled by 1 / keep).  The block's "samples" are the slices of the token buffer: token row
// ((b*X + x)*Y + y)*S + s belongs to sample b*S + s.  out = identity + branch * scale[sample] (identity may be
// NULL: out = branch * scale, which is also the backward of the branch).
// IDX = the integer type of the element / row arithmetic: uint32_t whenever the tensor has < 2^31 float4 pieces (every
// shape of the path) -- three 64-bit divisions per float4 made this pass VALU-bound (r05: a 64-bit divide is ~100 VALU
// instructions, the pass moves 48 bytes per thread)
template <typename IDX>
__global__ void __launch_bounds__(256) droppath_kernel(const float* __restrict__ identity, const float* __restrict__ branch,
                                                       const float* __restrict__ scale, float* __restrict__ out, long rows,
                                                       int Q, long XY, int S) {
  const IDX gid = (IDX)blockIdx.x * (IDX)blockDim.x + threadIdx.x;
  if (gid >= (IDX)(rows * Q)) return;
  const IDX row = gid / (IDX)Q;
  const IDX b = row / (IDX)(XY * S);
  const IDX s = row % (IDX)S;
  const float sc = scale[b * (IDX)S + s];
  const float4 v = ((const float4*)branch)[gid];
  float4 o = make_float4(v.x * sc, v.y * sc, v.z * sc, v.w * sc);
  if (identity) {
    const float4 id = ((const float4*)identity)[gid];
    o.x += id.x; o.y += id.y; o.z += id.z; o.w += id.w;
  }
  ((float4*)out)[gid] = o;
}
extern "C" int occf_droppath(const float* identity, const float* branch, const float* scale, float* out, long rows,
                             int C, long XY, int S, void* stream) {
  if (rows <= 0 || C % 4 != 0 || XY <= 0 || S <= 0 || rows % (XY * S) != 0) return OCCF_EINVAL;
  const long n4 = rows * (C / 4);
  if (n4 < 2147483647L - 256)
    hipLaunchKernelGGL(droppath_kernel<uint32_t>, dim3(occf_cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, identity,
                       branch, scale, out, rows, C / 4, XY, S);
  else
    hipLaunchKernelGGL(droppath_kernel<long>, dim3(occf_cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, identity,
                       branch, scale, out, rows, C / 4, XY, S);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= dual-path fusion backward
// forward: out[bp, z] = tok[bp, z] + sigma(<tok[bp, z], w> + b) * bev[bp] + identity[bp, z]
//   t = <dout[bp, z], bev[bp]> * sigma * (1 - sigma)
//   dtok[bp, z] = dout[bp, z] + t * w;  dtok[bp, Z] = 0 (the BEV slot is not read by the fusion)
//   dbev[bp] = sum_z sigma_z * dout[bp, z];  dw = sum t * tok;  db = sum t;  d_identity = dout (no kernel)
// One wave per (b, p) walks its Z rows, so dbev needs no atomics; dw / db as workgroup partials.
__global__ void __launch_bounds__(256) dualpath_bwd_kernel(const float* __restrict__ tok, const float* __restrict__ bev,
                                                           const float* __restrict__ w, const float* __restrict__ bias_p,
                                                           const float* __restrict__ dout, float* __restrict__ dtok,
                                                           float* __restrict__ dbev, float* __restrict__ partial, long BP,
                                                           int Z, int C, int cols_per_wave) {
  __shared__ float red[4][1025];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int Q = C / 4;
  float4 wv[BE_MAXV];
  float dw[BE_MAXV][4];
  float dbias = 0.f;
#pragma unroll
  for (int i = 0; i < BE_MAXV; ++i) {
    const int q = lane + i * 64;
    wv[i] = q < Q ? *(const float4*)(w + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) dw[i][e] = 0.f;
  }
  const float bias = bias_p ? bias_p[0] : 0.f;
  const long c_begin = ((long)blockIdx.x * 4 + wave) * cols_per_wave;
  for (int cc = 0; cc < cols_per_wave; ++cc) {
    const long bp = c_begin + cc;
    if (bp >= BP) break;
    float4 bv[BE_MAXV], acc[BE_MAXV];
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      const int q = lane + i * 64;
      bv[i] = q < Q ? *(const float4*)(bev + bp * C + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int z = 0; z < Z; ++z) {
      const float* t = tok + (bp * (Z + 1) + z) * C;
      const float* g = dout + (bp * Z + z) * C;
      float4 tv[BE_MAXV], gv[BE_MAXV];
      float d = 0.f, gb = 0.f;
#pragma unroll
      for (int i = 0; i < BE_MAXV; ++i) {
        const int q = lane + i * 64;
        if (q < Q) {
          tv[i] = *(const float4*)(t + q * 4);
          gv[i] = *(const float4*)(g + q * 4);
          d += (tv[i].x * wv[i].x + tv[i].y * wv[i].y) + (tv[i].z * wv[i].z + tv[i].w * wv[i].w);
          gb += (gv[i].x * bv[i].x + gv[i].y * bv[i].y) + (gv[i].z * bv[i].z + gv[i].w * bv[i].w);
        } else {
          tv[i] = gv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        d += __shfl_xor(d, o);
        gb += __shfl_xor(gb, o);
      }
      const float sg = 1.0f / (1.0f + expf(-(d + bias)));
      const float tt = gb * sg * (1.0f - sg);
      dbias += tt;
#pragma unroll
      for (int i = 0; i < BE_MAXV; ++i) {
        const int q = lane + i * 64;
        if (q < Q) {
          *(float4*)(dtok + (bp * (Z + 1) + z) * C + q * 4) =
              make_float4(fmaf(tt, wv[i].x, gv[i].x), fmaf(tt, wv[i].y, gv[i].y), fmaf(tt, wv[i].z, gv[i].z),
                          fmaf(tt, wv[i].w, gv[i].w));
          acc[i].x = fmaf(sg, gv[i].x, acc[i].x); acc[i].y = fmaf(sg, gv[i].y, acc[i].y);
          acc[i].z = fmaf(sg, gv[i].z, acc[i].z); acc[i].w = fmaf(sg, gv[i].w, acc[i].w);
          dw[i][0] = fmaf(tt, tv[i].x, dw[i][0]); dw[i][1] = fmaf(tt, tv[i].y, dw[i][1]);
          dw[i][2] = fmaf(tt, tv[i].z, dw[i][2]); dw[i][3] = fmaf(tt, tv[i].w, dw[i][3]);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < BE_MAXV; ++i) {
      const int q = lane + i * 64;
      if (q < Q) {
        *(float4*)(dbev + bp * C + q * 4) = acc[i];
        *(float4*)(dtok + (bp * (Z + 1) + Z) * C + q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BE_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][q * 4 + e] = dw[i][e];
  }
  if (lane == 0) red[wave][1024] = dbias;      // identical on every lane of the wave
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256)
    partial[(long)blockIdx.x * (C + 1) + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
  if (threadIdx.x == 0)
    partial[(long)blockIdx.x * (C + 1) + C] = (red[0][1024] + red[1][1024]) + (red[2][1024] + red[3][1024]);
}

static int occf_dp_cols_per_wave(long BP) {
  long r = (BP + 4 * 1024 - 1) / (4 * 1024);
  return r < 1 ? 1 : (int)r;
}
extern "C" long occf_dualpath_combine_bwd_workspace(long BP, int C) {
  return (long)occf_cdiv(BP, 4L * occf_dp_cols_per_wave(BP)) * (C + 1);
}
extern "C" int occf_dualpath_combine_bwd(const float* tokens, const float* bev, const float* coeff_weight,
                                         const float* coeff_bias, const float* dout, float* dtokens, float* dbev,
                                         float* dweight, float* dbias, float* workspace, long BP, int Z, int C,
                                         void* stream) {
  if (BP <= 0 || Z <= 0 || C % 4 != 0 || C > 1024) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int cpw = occf_dp_cols_per_wave(BP);
  const int nblk = occf_cdiv(BP, 4L * cpw);
  hipLaunchKernelGGL(dualpath_bwd_kernel, dim3(nblk), dim3(256), 0, st, tokens, bev, coeff_weight, coeff_bias, dout,
                     dtokens, dbev, workspace, BP, Z, C, cpw);
  occf_reduce_partials(workspace, dweight, (long)nblk, C, (long)C + 1, 1, 0, st);
  occf_reduce_partials(workspace, dbias, (long)nblk, 1, (long)C + 1, 1, C, st);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= FPN upsample+add backward
// forward: out = lateral + trilinear(coarse -> lateral's size, align_corners=False); d_lateral = dout (no kernel).
// d_coarse in GATHER form (no atomics): a coarse cell collects from every fine cell whose two taps along each
// axis include it; the candidate range per axis is recomputed from the forward's own index arithmetic.
__device__ __forceinline__ void ua_axis(int x, int X, int X2, int& lo, int& hi) {
  // fine indices x2 with floor(f) in {x - 1, x}, f = max(0, (X / X2) * (x2 + 0.5) - 0.5): generous range, the
  // weights of non-contributing candidates evaluate to zero
  const float inv = (float)X2 / (float)X;
  lo = (int)floorf(((float)x - 1.0f + 0.5f) * inv - 0.5f) - 1;
  hi = (int)ceilf(((float)x + 1.0f + 0.5f) * inv - 0.5f) + 1;
  if (lo < 0) lo = 0;
  if (hi > X2 - 1) hi = X2 - 1;
}
__device__ __forceinline__ float ua_weight(int x, int x2, int X, int X2) {
  float f = ((float)X / (float)X2) * ((float)x2 + 0.5f) - 0.5f;
  f = f < 0.f ? 0.f : f;
  const int x0 = (int)f;
  const int x1 = x0 + (x0 < X - 1);
  const float t = f - x0;
  return (x0 == x ? 1.f - t : 0.f) + (x1 == x ? t : 0.f);
}
__global__ void __launch_bounds__(256) upsample_add_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dcoarse,
                                                               int B, int X, int Y, int Z, int X2, int Y2, int Z2, int C) {
  const int Q = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * X * Y * Z * Q) return;
  const int cq = (int)(gid % Q);
  long v = gid / Q;
  const int z = (int)(v % Z);
  v /= Z;
  const int y = (int)(v % Y);
  v /= Y;
  const int x = (int)(v % X);
  const int b = (int)(v / X);
  int xl, xh, yl, yh, zl, zh;
  ua_axis(x, X, X2, xl, xh);
  ua_axis(y, Y, Y2, yl, yh);
  ua_axis(z, Z, Z2, zl, zh);
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* base = dout + (long)b * X2 * Y2 * Z2 * C + cq * 4;
  for (int x2 = xl; x2 <= xh; ++x2) {
    const float wx = ua_weight(x, x2, X, X2);
    if (wx == 0.f) continue;
    for (int y2 = yl; y2 <= yh; ++y2) {
      const float wy = wx * ua_weight(y, y2, Y, Y2);
      if (wy == 0.f) continue;
      for (int z2 = zl; z2 <= zh; ++z2) {
        const float wz = wy * ua_weight(z, z2, Z, Z2);
        if (wz == 0.f) continue;
        const float4 g = *(const float4*)(base + (((long)x2 * Y2 + y2) * Z2 + z2) * C);
        acc[0] = fmaf(wz, g.x, acc[0]); acc[1] = fmaf(wz, g.y, acc[1]);
        acc[2] = fmaf(wz, g.z, acc[2]); acc[3] = fmaf(wz, g.w, acc[3]);
      }
    }
  }
  *(float4*)(dcoarse + ((((long)b * X + x) * Y + y) * Z + z) * C + cq * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
}
extern "C" int occf_upsample_add_bwd(const float* dout, float* dcoarse, int B, int X, int Y, int Z, int X2, int Y2,
                                     int Z2, int C, void* stream) {
  if (B <= 0 || C % 4 != 0 || X <= 0 || Y <= 0 || Z <= 0 || X2 <= 0 || Y2 <= 0 || Z2 <= 0) return OCCF_ESHAPE;
  hipLaunchKernelGGL(upsample_add_bwd_kernel, dim3(occf_cdiv((long)B * X * Y * Z * (C / 4), 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, dcoarse, B, X, Y, Z, X2, Y2, Z2, C);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= point sampling backward
// d_vol[n, c, corner] += w_corner * dout[n, c, p]  (F.grid_sample's backward w.r.t. the input; the points carry
// no gradient in the reference: they come from no_grad sampling).  d_vol must be zero-filled by the caller.
// FX: dvol is a zero-filled int64 buffer of the same shape, the contributions enter in fixed point (occf_scatter_add):
// the sums do not depend on the arrival order
template <bool FX>
__global__ void __launch_bounds__(256) point_sample_3d_bwd_kernel(const float* __restrict__ dout,
                                                                  const float* __restrict__ pts, void* __restrict__ dvol,
                                                                  int N, int C, int X, int Y, int Z, long P, int shared_pts,
                                                                  int align_corners, int border, long voxel_major_ld,
                                                                  int cgroups, const uint32_t* __restrict__ slot) {
  float fx_inv = 1.f;
  const float fx_scale = FX ? occf_fx_scale(slot[0], fx_inv) : 1.f;
  // thread = (n, channel group, point), like the forward
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)N * cgroups * P) return;
  const long pi = gid % P;
  const int cg = (int)((gid / P) % cgroups);
  const int n = (int)(gid / (P * cgroups));
  const float* pt = pts + ((shared_pts ? 0 : (long)n * P) + pi) * 3;
  const int dims[3] = {Z, Y, X};
  int i0[3], i1[3];
  float t[3];
  bool ok0[3], ok1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = pt[a] * 2.0f - 1.0f;
    float f = align_corners ? (g + 1.f) * 0.5f * (float)(dims[a] - 1) : ((g + 1.f) * (float)dims[a] - 1.f) * 0.5f;
    if (border) f = fminf(fmaxf(f, 0.f), (float)(dims[a] - 1));
    const float fl = floorf(f);
    i0[a] = (int)fl;
    i1[a] = i0[a] + 1;
    t[a] = f - fl;
    ok0[a] = i0[a] >= 0 && i0[a] < dims[a];
    ok1[a] = i1[a] >= 0 && i1[a] < dims[a];
  }
  const long V = (long)X * Y * Z;
  // voxel_major_ld > 0: dvol is [V, ld] with column n*C + c (the layout the mask-logit contraction's backward
  // consumes as a row-major [voxels, rows] operand); otherwise [N, C, V]
  const long vstride = voxel_major_ld > 0 ? voxel_major_ld : 1;
  for (int c = cg; c < C; c += cgroups) {
    const long v0 = voxel_major_ld > 0 ? ((long)n * C + c) : ((long)n * C + c) * V;
    const float g = dout[((long)n * C + c) * P + pi];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bz = k & 1, by = (k >> 1) & 1, bx = k >> 2;
      const bool ok = (bz ? ok1[0] : ok0[0]) && (by ? ok1[1] : ok0[1]) && (bx ? ok1[2] : ok0[2]);
      if (!ok) continue;
      const int zz = bz ? i1[0] : i0[0], yy = by ? i1[1] : i0[1], xx = bx ? i1[2] : i0[2];
      const float w = (bz ? t[0] : 1.f - t[0]) * (by ? t[1] : 1.f - t[1]) * (bx ? t[2] : 1.f - t[2]);
      if (w != 0.f) occf_scatter_add<FX>(dvol, v0 + (((long)xx * Y + yy) * Z + zz) * vstride, w * g, fx_scale);
    }
  }
}
extern "C" int occf_point_sample_3d_bwd(const float* dout, const float* pts, float* dvol, int N, int C, int X, int Y,
                                        int Z, long P, int shared_pts, int align_corners, int border_padding,
                                        long voxel_major_ld, void* stream) {
  if (N <= 0 || C <= 0 || X <= 0 || Y <= 0 || Z <= 0 || P < 0) return OCCF_EINVAL;
  if (voxel_major_ld != 0 && voxel_major_ld < (long)N * C) return OCCF_EINVAL;
  if (P == 0) return 0;
  const int cgroups = occf_sample_cgroups(N, C, P);
  hipLaunchKernelGGL(point_sample_3d_bwd_kernel<false>, dim3(occf_cdiv((long)N * cgroups * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, pts, (void*)dvol, N, C, X, Y, Z, P, shared_pts, align_corners,
                     border_padding, voxel_major_ld, cgroups, (const uint32_t*)nullptr);
  OCCF_LAUNCH_CHECK();
}

// the same scatter in 64-bit fixed point (reproducible): acc = zero-filled int64 buffer shaped like dvol; slot = scale
// slot holding max |dout| (occf_absmax_f32 / occf_absmax_flat on the same stream); occf_fx_to_f32 converts afterwards
extern "C" int occf_point_sample_3d_bwd_fx(const float* dout, const float* pts, long long* acc, const uint32_t* slot,
                                           int N, int C, int X, int Y, int Z, long P, int shared_pts, int align_corners,
                                           int border_padding, long voxel_major_ld, void* stream) {
  if (N <= 0 || C <= 0 || X <= 0 || Y <= 0 || Z <= 0 || P < 0 || !acc || !slot) return OCCF_EINVAL;
  if (voxel_major_ld != 0 && voxel_major_ld < (long)N * C) return OCCF_EINVAL;
  if (P == 0) return 0;
  const int cgroups = occf_sample_cgroups(N, C, P);
  hipLaunchKernelGGL(point_sample_3d_bwd_kernel<true>, dim3(occf_cdiv((long)N * cgroups * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, dout, pts, (void*)acc, N, C, X, Y, Z, P, shared_pts, align_corners,
                     border_padding, voxel_major_ld, cgroups, slot);
  OCCF_LAUNCH_CHECK();
}

__global__ void __launch_bounds__(256) fx_to_f32_kernel(const long long* __restrict__ acc, float* __restrict__ out, long n,
                                                        const uint32_t* __restrict__ slot) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float inv;
  (void)occf_fx_scale(slot[0], inv);
  out[i] = (float)((double)acc[i] * (double)inv);
}
extern "C" int occf_fx_to_f32(const long long* acc, float* out, long n, const uint32_t* slot, void* stream) {
  if (n <= 0 || !acc || !out || !slot) return OCCF_EINVAL;
  hipLaunchKernelGGL(fx_to_f32_kernel, dim3(occf_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, acc, out, n, slot);
  OCCF_LAUNCH_CHECK();
}

// ======================================================================================= point-loss rows backward
// forward (occf_point_loss_rows_fwd): out[r] = { sum BCE(x, t), sum s*t, sum s, sum t },  s = sigmoid(x)
//   dx = g0 * (s - t) + (g1 * t + g2) * s * (1 - s)      with g = d(loss) / d(out[r, 0..2])
__global__ void __launch_bounds__(256) point_loss_rows_bwd_kernel(const float* __restrict__ logits,
                                                                  const float* __restrict__ targets,
                                                                  const float* __restrict__ grows, float* __restrict__ dx,
                                                                  int R, long P) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)R * P) return;
  const int r = (int)(gid / P);
  const float x = logits[gid], t = targets[gid];
  const float s = 1.0f / (1.0f + expf(-x));
  const float* g = grows + (long)r * 4;
  dx[gid] = g[0] * (s - t) + (g[1] * t + g[2]) * s * (1.0f - s);
}
extern "C" int occf_point_loss_rows_bwd(const float* logits, const float* targets, const float* grad_rows, float* dlogits,
                                        int R, long P, void* stream) {
  if (R <= 0 || P <= 0) return OCCF_EINVAL;
  hipLaunchKernelGGL(point_loss_rows_bwd_kernel, dim3(occf_cdiv((long)R * P, 256)), dim3(256), 0, (hipStream_t)stream,
                     logits, targets, grad_rows, dlogits, R, P);
  OCCF_LAUNCH_CHECK();
}
