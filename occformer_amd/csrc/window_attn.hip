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
  const int s = (int)(bid % S);          // slice fastest: the S slices of a spatial window are adjacent token rows
  bid /= S;
  const int wy = (int)(bid % nwy);
  bid /= nwy;
  const int wx = (int)(bid % nwx);
  const int b = (int)(bid / nwx);
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

// ---------------------------------------------------------------------------------------------
// Matrix-core variant.  Same mapping (one wave per (window, head), 4 heads per workgroup) and the same index
// arithmetic for pad / roll / region mask, but the 49 x 49 x 32 contractions run on v_mfma_f32_32x32x16_bf16
// (3-term bf16 split) in the transposed, register-chained form of xattn_mfma.hip:
//     St[64 keys x 64 queries] = K[64 x 32] . Qt[32 x 64]        (keys / queries 49..63 are padding)
//     Ot[32 d x 64 queries]    = V^T[32 x 64 keys] . Pt[64 x 64]
// 48 MFMAs per (window, head) instead of ~3100 VALU FMA instructions per lane: the scalar kernel was VALU
// bound (~0.6 ms per stage-0 call at 58 TF fp32); this one is bound by the HBM stream of the qkv rows.
typedef uint32_t wa_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t wa_bits(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
#else
  return __float_as_uint(x);
#endif
}
__device__ __forceinline__ float wa_from_bits(uint32_t u) {
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}
__device__ __forceinline__ uint32_t wa_bf16(float x) {
  const uint32_t u = wa_bits(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void wa_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) occf_bf16_split2(v[2 * e], v[2 * e + 1], h[e], l[e]);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    hi[2 * e] = (short)(h[e] & 0xFFFFu);
    hi[2 * e + 1] = (short)(h[e] >> 16);
    lo[2 * e] = (short)(l[e] & 0xFFFFu);
    lo[2 * e + 1] = (short)(l[e] >> 16);
  }
}

// per-wave LDS images (bytes): K hi/lo [2 ks][64 keys][2 x 16 B] = 4096 each; V^T hi/lo [4 kstep][32 d][2 x 16 B]
#define WM_IMG 4096

__global__ void __launch_bounds__(256) window_attn_mfma_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    float* __restrict__ out, int B, int X, int Y, int S, int C, int heads, int shift, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char img[4][4 * WM_IMG];     // [wave][Kh | Kl | Vh | Vl]
  __shared__ float lds_bias[4][(2 * WA_WS - 1) * (2 * WA_WS - 1)];
  __shared__ int lds_tok[4][64];
  __shared__ int lds_reg[4][64];

  const int wave = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int nwx = (X + WA_WS - 1) / WA_WS, nwy = (Y + WA_WS - 1) / WA_WS;
  const int Xp = nwx * WA_WS, Yp = nwy * WA_WS;
  const int head_groups = (heads + 3) >> 2;
  long bid = blockIdx.x;
  const int hg = (int)(bid % head_groups);
  bid /= head_groups;
  const int s = (int)(bid % S);          // slice fastest: the S slices of a spatial window are adjacent token rows
  bid /= S;
  const int wy = (int)(bid % nwy);
  bid /= nwy;
  const int wx = (int)(bid % nwx);
  const int b = (int)(bid / nwx);
  const int head = hg * 4 + wave;
  const bool active = head < heads;   // whole wave uniform
  const int hd = active ? head : heads - 1;
  const int C3 = 3 * C;

  // token bookkeeping: rolled-frame position -> source token (or -1 = padding), shift-mask region
  {
    int tok = -1, region = 0;
    if (lane < WA_T) {
      const int i = lane / WA_WS, j = lane % WA_WS;
      const int px = wx * WA_WS + i, py = wy * WA_WS + j;          // rolled frame
      int sx = px + shift, sy = py + shift;                           // torch.roll(-shift)
      if (sx >= Xp) sx -= Xp;
      if (sy >= Yp) sy -= Yp;
      if (sx < X && sy < Y) tok = (int)((((long)b * X + sx) * Y + sy) * S + s);
      if (shift > 0) {
        const int rx = px < Xp - WA_WS ? 0 : (px < Xp - shift ? 1 : 2);
        const int ry = py < Yp - WA_WS ? 0 : (py < Yp - shift ? 1 : 2);
        region = rx * 3 + ry;
      }
    }
    lds_tok[wave][lane] = tok;
    lds_reg[wave][lane] = region;
  }
  for (int t = lane; t < (2 * WA_WS - 1) * (2 * WA_WS - 1); t += 64)
    lds_bias[wave][t] = bias_table[(long)t * heads + hd];
  __syncthreads();

  // ---- stage K and V^T of this (window, head): 64 rows x 8 float4 per operand, all loads first
  // (padded positions take the bias row -- the reference zero-pads AFTER LayerNorm; rows 49..63 are zero)
  unsigned char* base = img[wave];
  {
    float4 rk[8], rv[8];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + i * 64;
      const int t = idx >> 3, q4 = (idx & 7) * 4;
      const int tok = t < WA_T ? lds_tok[wave][t] : -1;
      const float* src = tok >= 0 ? qkv + (long)tok * C3 : qkv_bias;
      rk[i] = *(const float4*)(src + C + hd * WA_HD + q4);
      rv[i] = *(const float4*)(src + 2 * C + hd * WA_HD + q4);
      if (t >= WA_T) { rk[i] = z4; rv[i] = z4; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int idx = lane + i * 64;
      const int t = idx >> 3, sc = (idx & 7) * 4;
      {   // K image: [ks = sc >> 4][key t][slot (sc >> 3) & 1][4 consecutive d at sc & 7]
        const float f[4] = {rk[i].x, rk[i].y, rk[i].z, rk[i].w};
        uint32_t hb[4], lb[4];
        {
          uint32_t h0, l0, h1, l1;
          occf_bf16_split2(f[0], f[1], h0, l0);
          occf_bf16_split2(f[2], f[3], h1, l1);
          hb[0] = h0 & 0xFFFFu; hb[1] = h0 >> 16; hb[2] = h1 & 0xFFFFu; hb[3] = h1 >> 16;
          lb[0] = l0 & 0xFFFFu; lb[1] = l0 >> 16; lb[2] = l1 & 0xFFFFu; lb[3] = l1 >> 16;
        }
        const int off = (sc >> 4) * 2048 + t * 32 + ((sc >> 3) & 1) * 16 + (sc & 7) * 2;
        const wa_u2 ph = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
        const wa_u2 pl = {lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
        *(wa_u2*)(base + off) = ph;
        *(wa_u2*)(base + WM_IMG + off) = pl;
      }
      {   // V^T image: [k-step t >> 4][row d][slot][e]: key (t & 15) = 8 (e>>2) + 4 slot + (e&3)
        const float f[4] = {rv[i].x, rv[i].y, rv[i].z, rv[i].w};
        const int kst = t >> 4, kk = t & 15;
        const int e = ((kk >> 3) << 2) | (kk & 3), slot = (kk >> 2) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          uint32_t hb, lb;
          occf_bf16_split2(f[j], 0.f, hb, lb);
          hb &= 0xFFFFu;
          lb &= 0xFFFFu;
          const int off = kst * 1024 + (sc + j) * 32 + slot * 16 + e * 2;
          *(uint16_t*)(base + 2 * WM_IMG + off) = (uint16_t)hb;
          *(uint16_t*)(base + 3 * WM_IMG + off) = (uint16_t)lb;
        }
      }
    }
  }
  // ---- queries of this lane: column li of query tile qt -> query 32 qt + li (B operand, pre-scaled)
  bf16x8 qh[2][2], ql[2][2];
  int qtok[2], qreg[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = qt * 32 + li;
    qtok[qt] = qi < WA_T ? lds_tok[wave][qi] : -1;
    qreg[qt] = lds_reg[wave][qi];
    const float* src = (qtok[qt] >= 0 ? qkv + (long)qtok[qt] * C3 : qkv_bias) + hd * WA_HD + lk * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 a = *(const float4*)(src + ks * 16), c = *(const float4*)(src + ks * 16 + 4);
      const float f[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale,
                          c.x * scale, c.y * scale, c.z * scale, c.w * scale};
      wa_split8(f, qh[qt][ks], ql[qt][ks]);
    }
  }
  __syncthreads();                                   // the images of all four waves are complete
  if (!active) return;

#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = qt * 32 + li;
    const int qrow = (qi * 37) >> 8, qcol = qi - qrow * WA_WS;        // qi / 7, qi % 7 for qi < 64
    // ---- GEMM1: St[key tile kt] = K . Qt
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int off = ks * 2048 + (kt * 32 + li) * 32 + lk * 16;
        const bf16x8 ah = *(const bf16x8*)(base + off);
        const bf16x8 al = *(const bf16x8*)(base + WM_IMG + off);
        st[kt] = occf_mfma_bf16_32x32x16(al, qh[qt][ks], st[kt]);
        st[kt] = occf_mfma_bf16_32x32x16(ah, ql[qt][ks], st[kt]);
        st[kt] = occf_mfma_bf16_32x32x16(ah, qh[qt][ks], st[kt]);
      }
    }
    // ---- relative position bias, shift mask, softmax over the 49 keys (a query = lanes li, li + 32)
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int krow = (key * 37) >> 8, kcol = key - krow * WA_WS;
        float a = st[kt][r];
        if (key < WA_T) {
          const int bi = (qrow - krow + WA_WS - 1) * (2 * WA_WS - 1) + (qcol - kcol + WA_WS - 1);
          a += lds_bias[wave][qi < WA_T ? bi : 0];
          if (shift > 0 && lds_reg[wave][key] != qreg[qt]) a += -100.0f;
        } else {
          a = -INFINITY;
        }
        st[kt][r] = a;
        mx = fmaxf(mx, a);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
    bf16x8 ph[4], pl[4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = expf(st[kt][s2 * 8 + e] - mx);      // padding keys: exp(-inf) = 0
          sum += pv[e];
        }
        wa_split8(pv, ph[kt * 2 + s2], pl[kt * 2 + s2]);
      }
    sum += __shfl_xor(sum, 32);
    // ---- GEMM2: Ot = V^T . Pt
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int kst = 0; kst < 4; ++kst) {
      const int off = kst * 1024 + li * 32 + lk * 16;
      const bf16x8 ah = *(const bf16x8*)(base + 2 * WM_IMG + off);
      const bf16x8 al = *(const bf16x8*)(base + 3 * WM_IMG + off);
      ot = occf_mfma_bf16_32x32x16(al, ph[kst], ot);
      ot = occf_mfma_bf16_32x32x16(ah, pl[kst], ot);
      ot = occf_mfma_bf16_32x32x16(ah, ph[kst], ot);
    }
    if (qtok[qt] >= 0) {                              // padded / idle query columns are cropped
      const float inv = 1.0f / sum;
      float* dst = out + (long)qtok[qt] * C + head * WA_HD + 4 * lk;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *(float4*)(dst + 8 * g) = make_float4(ot[g * 4 + 0] * inv, ot[g * 4 + 1] * inv, ot[g * 4 + 2] * inv,
                                              ot[g * 4 + 3] * inv);
    }
  }
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
  static const bool mfma = [] {
    const char* e = getenv("OCCF_WATTN_MFMA");
    return e ? atoi(e) != 0 : true;
  }();
  if (mfma)
    hipLaunchKernelGGL(window_attn_mfma_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, qkv,
                       qkv_bias, bias_table, out, B, X, Y, S, C, heads, shift, scale);
  else
    hipLaunchKernelGGL(window_attn_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       qkv, qkv_bias, bias_table, out, B, X, Y, S, C, heads, shift, scale);
  OCCF_LAUNCH_CHECK();
}
