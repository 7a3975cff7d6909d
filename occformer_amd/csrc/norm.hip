// Normalisation / elementwise kernels of the dual-path encoder on channels-last voxel
// tensors: GroupNorm statistics + apply (+ReLU, + build of the Z+1 slice token buffer with the
// BEV mean slice), row LayerNorm, and the soft-gated dual-path fusion.
//
// Reference: projects/mmdet3d_plugin/occformer/backbones/dualpath_block.py:43-48 (conv-GN-ReLU),
//   :70-76 (mean over Z, rearrange, cat), :79-82 (sigmoid gate, fusion, skip);
//   window_attention.py:352-361 (norm1/norm2).  All HBM-bound, channel-contiguous accesses.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// ---------------------------------------------------------------------------------------
// GroupNorm statistics over x[B, V, C] (channels-last).  Deterministic two-stage reduction:
// stage 1 writes per-block partial (sum, sumsq) per group, stage 2 adds them in fixed order
// in double precision and emits mean / rstd.
#define GN_MAXG 64

__global__ void __launch_bounds__(256) gn_partial_kernel(const float* __restrict__ x,
                                                         float* __restrict__ partial, long V, int C, int G,
                                                         int rows_per_block) {
  __shared__ float s_sum[256][4], s_sq[256][4];
  __shared__ float c_sum[1024], c_sq[1024];
  const int Q = C / 4;                  // channel quads (<= 256)
  const int R = 256 / Q;                // row-threads per quad
  const int tid = threadIdx.x;
  const int cq = tid % Q, rt = tid / Q;
  const int b = blockIdx.y;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > V) r1 = V;
  float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  if (rt < R) {
    const float* base = x + ((long)b * V) * C + cq * 4;
#pragma unroll 8
    for (long r = r0 + rt; r < r1; r += R) {      // independent 16-B loads: keep 8 in flight
      const float4 v = *(const float4*)(base + r * C);
      sm[0] += v.x; sm[1] += v.y; sm[2] += v.z; sm[3] += v.w;
      sq[0] = fmaf(v.x, v.x, sq[0]); sq[1] = fmaf(v.y, v.y, sq[1]);
      sq[2] = fmaf(v.z, v.z, sq[2]); sq[3] = fmaf(v.w, v.w, sq[3]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) { s_sum[tid][i] = sm[i]; s_sq[tid][i] = sq[i]; }
  __syncthreads();
  if (tid < Q) {
    float a[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] += s_sum[r * Q + tid][i]; q[i] += s_sq[r * Q + tid][i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { c_sum[tid * 4 + i] = a[i]; c_sq[tid * 4 + i] = q[i]; }
  }
  __syncthreads();
  if (tid < G) {
    const int cg = C / G;
    float a = 0.f, q = 0.f;
    for (int c = tid * cg; c < (tid + 1) * cg; ++c) { a += c_sum[c]; q += c_sq[c]; }
    float* o = partial + (((long)b * gridDim.x + blockIdx.x) * G + tid) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// one 64-lane wave per (batch, group): lanes stride over the per-block partials (fixed
// assignment -> deterministic), then a butterfly reduction in double
__global__ void __launch_bounds__(256) gn_finalize_kernel(const float* __restrict__ partial,
                                                          float* __restrict__ stats, int nblk, int G, int BG,
                                                          double count, float eps) {
  const int lane = threadIdx.x & 63;
  const int bg = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (bg >= BG) return;
  const int b = bg / G, g = bg % G;
  double s = 0.0, q = 0.0;
  for (int k = lane; k < nblk; k += 64) {
    const float* p = partial + (((long)b * nblk + k) * G + g) * 2;
    s += (double)p[0];
    q += (double)p[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) {
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(long)bg * 2 + 0] = (float)mean;
    stats[(long)bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

extern "C" long occf_groupnorm_workspace(int B, long V, int C, int G) {
  const int rows = 256;
  return (long)B * occf_cdiv(V, rows) * G * 2;
}

extern "C" int occf_groupnorm_stats(const float* x, float* stats, float* workspace, int B, long V, int C,
                                    int G, float eps, void* stream) {
  if (B <= 0 || V <= 0 || C % 4 != 0 || C > 1024 || G <= 0 || G > GN_MAXG || C % G != 0) return OCCF_ESHAPE;
  const int rows = 256;
  const int nblk = occf_cdiv(V, rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_partial_kernel, dim3(nblk, B), dim3(256), 0, st, x, workspace, V, C, G, rows);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(occf_cdiv((long)B * G * 64, 256)), dim3(256), 0, st, workspace,
                     stats, nblk, G, B * G, (double)V * (C / G), eps);
  OCCF_LAUNCH_CHECK();
}

// Second stage for per-CHANNEL partial sums written by the convolution / GEMM epilogues:
// partial[B][nblk][C][2] -> stats[B][G][2].  One 256-thread workgroup per (batch, group); threads stride
// over the tiles with independent (unrolled) loads, double accumulation, fixed reduction order.
__global__ void __launch_bounds__(256) gn_finalize_channels_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ stats, long nblk, int C, int G,
                                                                   int BG, double count, float eps) {
  __shared__ double red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bg = blockIdx.x;
  const int b = bg / G, g = bg % G, cg = C / G;
  const float* base = partial + ((long)b * nblk * C + (long)g * cg) * 2;
  double s = 0.0, q = 0.0;
  long k = threadIdx.x;
  for (; k + 768 < nblk; k += 1024) {          // four tiles per trip: their loads are independent
    float s4[4] = {0.f, 0.f, 0.f, 0.f}, q4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float2* p = reinterpret_cast<const float2*>(base + (k + 256 * u) * C * 2);
      for (int c = 0; c < cg; ++c) {
        const float2 v = p[c];
        s4[u] += v.x;
        q4[u] += v.y;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      s += (double)s4[u];
      q += (double)q4[u];
    }
  }
  for (; k < nblk; k += 256) {
    const float2* p = reinterpret_cast<const float2*>(base + k * C * 2);
    float s1 = 0.f, q1 = 0.f;
    for (int c = 0; c < cg; ++c) {
      const float2 v = p[c];
      s1 += v.x;
      q1 += v.y;
    }
    s += (double)s1;
    q += (double)q1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o);
    q += __shfl_xor(q, o);
  }
  if (lane == 0) {
    red[0][wave] = s;
    red[1][wave] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    q = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[(long)bg * 2 + 0] = (float)mean;
    stats[(long)bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

extern "C" int occf_groupnorm_finalize(const float* partial, float* stats, int B, long nblk, int C, int G,
                                       double count, float eps, void* stream) {
  if (B <= 0 || nblk <= 0 || G <= 0 || C % G != 0 || count <= 0) return OCCF_EINVAL;
  hipLaunchKernelGGL(gn_finalize_channels_kernel, dim3(B * G), dim3(256), 0, (hipStream_t)stream, partial, stats,
                     nblk, C, G, B * G, count, eps);
  OCCF_LAUNCH_CHECK();
}

// GroupNorm apply.  x[B, P, Z, C] -> out[B, P, Zs, C] (Zs = Z, or Z + 1 in token mode where
// slot Z receives the mean over Z of the normalised values = the BEV slice of the dual-path
// block).  Optional ReLU and residual (same layout as x).  Thread = (b, p, channel quad).
__global__ void __launch_bounds__(256) gn_apply_kernel(
    const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ residual, float* __restrict__ out, int B,
    long P, int Z, int C, int G, int relu, int tokens) {
  const int Q = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * P * Q) return;
  const int cq = (int)(gid % Q);
  const long bp = gid / Q;
  const int b = (int)(bp / P);
  const int cg = C / G;
  float sc[4], sh[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cq * 4 + i;
    const float* s = stats + ((long)b * G + c / cg) * 2;
    sc[i] = s[1] * gamma[c];
    sh[i] = beta[c] - s[0] * sc[i];
  }
  const int Zs = tokens ? Z + 1 : Z;
  const float* xi = x + bp * Z * C + cq * 4;
  const float* ri = residual ? residual + bp * Z * C + cq * 4 : nullptr;
  float* oi = out + bp * Zs * C + cq * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
  for (int z = 0; z < Z; ++z) {
    const float4 v = *(const float4*)(xi + (long)z * C);
    float y[4] = {fmaf(v.x, sc[0], sh[0]), fmaf(v.y, sc[1], sh[1]), fmaf(v.z, sc[2], sh[2]),
                  fmaf(v.w, sc[3], sh[3])};
    if (relu) {
#pragma unroll
      for (int i = 0; i < 4; ++i) y[i] = fmaxf(y[i], 0.f);
    }
    if (ri) {
      const float4 r = *(const float4*)(ri + (long)z * C);
      y[0] += r.x; y[1] += r.y; y[2] += r.z; y[3] += r.w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += y[i];
    *(float4*)(oi + (long)z * C) = make_float4(y[0], y[1], y[2], y[3]);
  }
  if (tokens) {
    const float inv = 1.0f / (float)Z;
    *(float4*)(oi + (long)Z * C) = make_float4(acc[0] * inv, acc[1] * inv, acc[2] * inv, acc[3] * inv);
  }
}

extern "C" int occf_groupnorm_apply(const float* x, const float* stats, const float* gamma,
                                    const float* beta, const float* residual, float* out, int B, long P,
                                    int Z, int C, int G, int relu, int tokens, void* stream) {
  if (B <= 0 || P <= 0 || Z <= 0 || C % 4 != 0 || G <= 0 || C % G != 0) return OCCF_ESHAPE;
  const long total = (long)B * P * (C / 4);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, stats,
                     gamma, beta, residual, out, B, P, Z, C, G, relu, tokens);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Row LayerNorm, x[M, C] -> out[M, C]; one 64-lane wave per row, two-pass (mean, then
// centred variance) in registers.  C <= 1024, C % 4 == 0.
#define LN_MAXV 4     // float4 per lane: 64 lanes * 4 * 4 = 1024 channels

__global__ void __launch_bounds__(256) layernorm_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ out,
                                                        long M, int C, float eps) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const int Q = C / 4;
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q) {
      v[i] = *(const float4*)(x + row * C + q * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)C;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    if (lane + i * 64 < Q) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q2 += (a * a + b * b) + (c * c + d * d);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
  const float rstd = 1.0f / sqrtf(q2 / (float)C + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q) {
      const float4 g = *(const float4*)(gamma + q * 4);
      const float4 bb = *(const float4*)(beta + q * 4);
      *(float4*)(out + row * C + q * 4) =
          make_float4((v[i].x - mean) * rstd * g.x + bb.x, (v[i].y - mean) * rstd * g.y + bb.y,
                      (v[i].z - mean) * rstd * g.z + bb.z, (v[i].w - mean) * rstd * g.w + bb.w);
    }
  }
}

// C = 64 * NV (128 / 192 / 256): 16 lanes per row, NV float4 per lane, 4 rows per wave -- every lane is busy
// (one wave per row leaves half of them idle at C = 128) and a lane's NV loads are in flight together
template <int NV>
__global__ void __launch_bounds__(256) layernorm16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ out,
                                                          long M, float eps) {
  constexpr int C = 64 * NV;
  const int sub = threadIdx.x & 15;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long rc = row < M ? row : M - 1;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    v[j] = *(const float4*)(x + rc * C + (sub + 16 * j) * 4);
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
  if (row >= M) return;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int c0 = (sub + 16 * j) * 4;
    const float4 g = *(const float4*)(gamma + c0);
    const float4 bb = *(const float4*)(beta + c0);
    *(float4*)(out + row * C + c0) =
        make_float4((v[j].x - mean) * rstd * g.x + bb.x, (v[j].y - mean) * rstd * g.y + bb.y,
                    (v[j].z - mean) * rstd * g.z + bb.z, (v[j].w - mean) * rstd * g.w + bb.w);
  }
}

extern "C" int occf_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* out,
                                  long M, int C, float eps, void* stream) {
  if (M <= 0 || C % 4 != 0 || C > 64 * 4 * LN_MAXV) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g16(occf_cdiv(M * 16, 256));
  if (C == 128) hipLaunchKernelGGL(layernorm16_kernel<2>, g16, dim3(256), 0, st, x, gamma, beta, out, M, eps);
  else if (C == 192) hipLaunchKernelGGL(layernorm16_kernel<3>, g16, dim3(256), 0, st, x, gamma, beta, out, M, eps);
  else if (C == 256) hipLaunchKernelGGL(layernorm16_kernel<4>, g16, dim3(256), 0, st, x, gamma, beta, out, M, eps);
  else
    hipLaunchKernelGGL(layernorm_kernel, dim3(occf_cdiv(M * 64, 256)), dim3(256), 0, st, x, gamma, beta, out, M, C,
                       eps);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Dual-path fusion (dualpath_block.py:79-82):
//   coeff = sigmoid(<tok[b,p,z,:], w> + bias);  out[b,p,z,:] = tok[b,p,z,:] + coeff * bev[b,p,:]
//                                                              + identity[b,p,z,:]
// tok has Zs = Z + 1 slots per (b, p) (the token buffer), out/identity have Z.  One wave per
// voxel row, lanes over channels.
__global__ void __launch_bounds__(256) dualpath_combine_kernel(
    const float* __restrict__ tok, const float* __restrict__ bev, const float* __restrict__ w,
    const float* __restrict__ bias_p, const float* __restrict__ identity, float* __restrict__ out, long BP, int Z, int C) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;     // (bp, z)
  if (row >= BP * Z) return;
  const long bp = row / Z;
  const int z = (int)(row % Z);
  const int Q = C / 4;
  const float* t = tok + (bp * (Z + 1) + z) * C;
  float4 v[LN_MAXV];
  float d = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q) {
      v[i] = *(const float4*)(t + q * 4);
      const float4 ww = *(const float4*)(w + q * 4);
      d += (v[i].x * ww.x + v[i].y * ww.y) + (v[i].z * ww.z + v[i].w * ww.w);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
  const float bias = bias_p ? bias_p[0] : 0.f;
  const float coeff = 1.0f / (1.0f + expf(-(d + bias)));
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int q = lane + i * 64;
    if (q < Q) {
      const float4 bv = *(const float4*)(bev + bp * C + q * 4);
      const float4 id = *(const float4*)(identity + row * C + q * 4);
      *(float4*)(out + row * C + q * 4) =
          make_float4(v[i].x + coeff * bv.x + id.x, v[i].y + coeff * bv.y + id.y,
                      v[i].z + coeff * bv.z + id.z, v[i].w + coeff * bv.w + id.w);
    }
  }
}

extern "C" int occf_dualpath_combine(const float* tokens, const float* bev, const float* coeff_weight,
                                     const float* coeff_bias, const float* identity, float* out, long BP, int Z,
                                     int C, void* stream) {
  if (BP <= 0 || Z <= 0 || C % 4 != 0 || C > 64 * 4 * LN_MAXV) return OCCF_ESHAPE;
  hipLaunchKernelGGL(dualpath_combine_kernel, dim3(occf_cdiv(BP * Z * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, tokens, bev, coeff_weight, coeff_bias, identity, out, BP, Z, C);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// FPN top-down step of the pixel decoder (multiscale_deformattn_3d.py:233-243):
//   out = lateral + F.interpolate(coarse, size=lateral.shape, mode='trilinear', align_corners=False)
// on channels-last tensors; thread = (output voxel, channel quad).
__global__ void __launch_bounds__(256) upsample_add_kernel(const float* __restrict__ coarse,
                                                           const float* __restrict__ lateral,
                                                           float* __restrict__ out, int B, int X, int Y, int Z,
                                                           int X2, int Y2, int Z2, int C) {
  const int Q = C / 4;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * X2 * Y2 * Z2 * Q;
  if (gid >= total) return;
  const int cq = (int)(gid % Q);
  long v = gid / Q;
  const int z2 = (int)(v % Z2);
  v /= Z2;
  const int y2 = (int)(v % Y2);
  v /= Y2;
  const int x2 = (int)(v % X2);
  const int b = (int)(v / X2);
  const float sx = (float)X / (float)X2, sy = (float)Y / (float)Y2, sz = (float)Z / (float)Z2;
  float fx = sx * ((float)x2 + 0.5f) - 0.5f, fy = sy * ((float)y2 + 0.5f) - 0.5f,
        fz = sz * ((float)z2 + 0.5f) - 0.5f;
  fx = fx < 0.f ? 0.f : fx;
  fy = fy < 0.f ? 0.f : fy;
  fz = fz < 0.f ? 0.f : fz;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int x1 = x0 + (x0 < X - 1), y1 = y0 + (y0 < Y - 1), z1 = z0 + (z0 < Z - 1);
  const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
  const float* cb = coarse + (long)b * X * Y * Z * C + cq * 4;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int xx = (c >> 2) ? x1 : x0, yy = ((c >> 1) & 1) ? y1 : y0, zz = (c & 1) ? z1 : z0;
    const float w = ((c >> 2) ? tx : 1.f - tx) * (((c >> 1) & 1) ? ty : 1.f - ty) * ((c & 1) ? tz : 1.f - tz);
    const float4 t = *(const float4*)(cb + (((long)xx * Y + yy) * Z + zz) * C);
    acc[0] = fmaf(w, t.x, acc[0]); acc[1] = fmaf(w, t.y, acc[1]);
    acc[2] = fmaf(w, t.z, acc[2]); acc[3] = fmaf(w, t.w, acc[3]);
  }
  const long o = ((((long)b * X2 + x2) * Y2 + y2) * Z2 + z2) * C + cq * 4;
  const float4 l = *(const float4*)(lateral + o);
  *(float4*)(out + o) = make_float4(l.x + acc[0], l.y + acc[1], l.z + acc[2], l.w + acc[3]);
}

extern "C" int occf_upsample_add(const float* coarse, const float* lateral, float* out, int B, int X, int Y,
                                 int Z, int X2, int Y2, int Z2, int C, void* stream) {
  if (B <= 0 || C % 4 != 0 || X <= 0 || Y <= 0 || Z <= 0 || X2 <= 0 || Y2 <= 0 || Z2 <= 0) return OCCF_ESHAPE;
  const long total = (long)B * X2 * Y2 * Z2 * (C / 4);
  hipLaunchKernelGGL(upsample_add_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, coarse,
                     lateral, out, B, X, Y, Z, X2, Y2, Z2, C);
  OCCF_LAUNCH_CHECK();
}
