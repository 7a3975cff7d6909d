// Windowed multi-head self-attention of the dual-path encoder's shared SwinBlock
// (7x7 windows over every Z-slice and the BEV slice of a channels-last voxel tensor).
//
// Reference: projects/mmdet3d_plugin/occformer/backbones/modules/window_attention.py
//   WindowMSA.forward :69-107, ShiftWindowMSA.forward :168-242 (pad after LN, cyclic shift,
//   9-region -100 mask, window partition/reverse, crop).
//
// The pad / roll / partition / reverse / un-roll / crop copies of the reference (each one a
// full pass over the slice tensor) do not exist here: they are index arithmetic on the way
// in and out of the kernel.  Input is the fused qkv projection [n_tok, 3C] of the
// layer-normed tokens in channels-last token order tok = ((b*X + x)*Y + y)*S + s  (S = Z+1
// slices: Z height slices + the BEV mean slice); padded positions take qkv = qkv_bias
// (the reference zero-pads AFTER LayerNorm, so the linear layer maps them to its bias).
//
// One 64-lane wave per (window, head): lane t < 49 owns query row t (scores, softmax and
// the PV row live in its registers); K and V of the window sit in LDS and are read as
// wave-wide broadcasts.  4 waves (4 heads) per workgroup.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define WA_WS 7
#define WA_T 49
#define WA_HD 32

__global__ void __launch_bounds__(256) window_attn_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias,
    const float* __restrict__ bias_table, float* __restrict__ out, int B, int X, int Y, int S,
    int C, int heads, int shift, float scale) {
  __shared__ __attribute__((aligned(16))) float lds_k[4][WA_T * WA_HD];
  __shared__ __attribute__((aligned(16))) float lds_v[4][WA_T * WA_HD];
  __shared__ float lds_bias[4][(2 * WA_WS - 1) * (2 * WA_WS - 1)];
  __shared__ int lds_tok[4][WA_T];

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int nwx = (X + WA_WS - 1) / WA_WS, nwy = (Y + WA_WS - 1) / WA_WS;
  const int Xp = nwx * WA_WS, Yp = nwy * WA_WS;
  const int head_groups = (heads + 3) >> 2;
  // blockIdx.x -> (b, s, wx, wy, head group); heads fastest so a window's q/k/v rows are
  // touched by neighbouring workgroups (L2 reuse of the 3C-wide token rows)
  long bid = blockIdx.x;
  const int hg = (int)(bid % head_groups);
  bid /= head_groups;
  const int wy = (int)(bid % nwy);
  bid /= nwy;
  const int wx = (int)(bid % nwx);
  bid /= nwx;
  const int s = (int)(bid % S);
  const int b = (int)(bid / S);
  const int head = hg * 4 + wave;
  const bool active = head < heads;   // whole wave uniform
  const int C3 = 3 * C;

  // token bookkeeping: rolled-frame position -> source token (or -1 = padding)
  int my_tok = -1, my_region = 0;
  if (lane < WA_T) {
    const int i = lane / WA_WS, j = lane % WA_WS;
    const int px = wx * WA_WS + i, py = wy * WA_WS + j;          // rolled frame
    int sx = px + shift, sy = py + shift;                           // torch.roll(-shift)
    if (sx >= Xp) sx -= Xp;
    if (sy >= Yp) sy -= Yp;
    if (sx < X && sy < Y) my_tok = (int)((((long)b * X + sx) * Y + sy) * S + s);
    if (shift > 0) {
      const int rx = px < Xp - WA_WS ? 0 : (px < Xp - shift ? 1 : 2);
      const int ry = py < Yp - WA_WS ? 0 : (py < Yp - shift ? 1 : 2);
      my_region = rx * 3 + ry;
    }
    lds_tok[wave][lane] = my_tok;
  }
  if (active) {
    for (int t = lane; t < (2 * WA_WS - 1) * (2 * WA_WS - 1); t += 64)
      lds_bias[wave][t] = bias_table[(long)t * heads + head];
  }
  __syncthreads();

  // stage K and V of this (window, head): 49 rows x 32 floats, 8 lanes per row (16 B each)
  if (active) {
    for (int idx = lane; idx < WA_T * (WA_HD / 4); idx += 64) {
      const int t = idx >> 3, q4 = (idx & 7) * 4;
      const int tok = lds_tok[wave][t];
      const float* src = tok >= 0 ? qkv + (long)tok * C3 : qkv_bias;
      const float4 kk = *(const float4*)(src + C + head * WA_HD + q4);
      const float4 vv = *(const float4*)(src + 2 * C + head * WA_HD + q4);
      *(float4*)(&lds_k[wave][t * WA_HD + q4]) = kk;
      *(float4*)(&lds_v[wave][t * WA_HD + q4]) = vv;
    }
  }
  __syncthreads();
  if (!active || lane >= WA_T || my_tok < 0) return;   // padded / idle query rows are cropped

  // own query row, pre-scaled (window_attention.py:82)
  float q[WA_HD];
  {
    const float* src = qkv + (long)my_tok * C3;
#pragma unroll
    for (int d = 0; d < WA_HD; d += 4) {
      const float4 t = *(const float4*)(src + head * WA_HD + d);
      q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale;
    }
  }
  const int qi = lane / WA_WS, qj = lane % WA_WS;
  float sc[WA_T];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < WA_T; ++j) {
    const float* kr = &lds_k[wave][j * WA_HD];
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < WA_HD; ++d) a = fmaf(q[d], kr[d], a);
    const int ki = j / WA_WS, kj = j % WA_WS;
    a += lds_bias[wave][(qi - ki + WA_WS - 1) * (2 * WA_WS - 1) + (qj - kj + WA_WS - 1)];
    if (shift > 0) {
      const int px = wx * WA_WS + ki, py = wy * WA_WS + kj;
      const int rx = px < Xp - WA_WS ? 0 : (px < Xp - shift ? 1 : 2);
      const int ry = py < Yp - WA_WS ? 0 : (py < Yp - shift ? 1 : 2);
      if (rx * 3 + ry != my_region) a += -100.0f;
    }
    sc[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < WA_T; ++j) {
    sc[j] = expf(sc[j] - mx);
    sum += sc[j];
  }
  const float inv = 1.0f / sum;
  float o[WA_HD];
#pragma unroll
  for (int d = 0; d < WA_HD; ++d) o[d] = 0.f;
#pragma unroll
  for (int j = 0; j < WA_T; ++j) {
    const float p = sc[j] * inv;
    const float* vr = &lds_v[wave][j * WA_HD];
#pragma unroll
    for (int d = 0; d < WA_HD; ++d) o[d] = fmaf(p, vr[d], o[d]);
  }
  float* dst = out + (long)my_tok * C + head * WA_HD;
#pragma unroll
  for (int d = 0; d < WA_HD; d += 4) *(float4*)(dst + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
}

extern "C" int occf_window_attn_fwd(const float* qkv, const float* qkv_bias, const float* bias_table,
                                    float* out, int B, int X, int Y, int S, int C, int heads,
                                    int shift, void* stream) {
  if (B <= 0 || X <= 0 || Y <= 0 || S <= 0 || heads <= 0 || C != heads * WA_HD) return OCCF_ESHAPE;
  if (shift < 0 || shift >= WA_WS) return OCCF_EINVAL;
  const int nwx = (X + WA_WS - 1) / WA_WS, nwy = (Y + WA_WS - 1) / WA_WS;
  const long blocks = (long)B * S * nwx * nwy * ((heads + 3) / 4);
  if (blocks >= 2147483647L) return OCCF_ESHAPE;
  const float scale = (float)(1.0 / sqrt((double)WA_HD));   // python: head_dim ** -0.5
  hipLaunchKernelGGL(window_attn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     qkv, qkv_bias, bias_table, out, B, X, Y, S, C, heads, shift, scale);
  OCCF_LAUNCH_CHECK();
}
