// Chained token MLP for C = 128 / 192:   out = LNpost?( x + W2 . act( W1 . LNpre?(x) + b1 ) + b2 )
// (same contract as mlp_fused.hip; occf_mlp_fused_fwd dispatches here).
//
// Register-chained formulation: a wave owns 32 tokens for the whole kernel and works on the TRANSPOSED
// problem, so that the result of the first GEMM is already laid out as an operand of the second:
//
//   GEMM1   Ht[32 hidden x 32 tokens]  = W1g[32 x C] . Xt[C x 32 tokens]       A = W1 rows (LDS), B = Xt (registers)
//   GEMM2   Ot[C x 32 tokens]         += W2g[C x 32 hidden] . Ht               A = W2 rows (LDS), B = Ht (registers)
//
// v_mfma_f32_32x32x16_bf16 returns D with lane -> column (token) and registers -> rows (hidden units
// (r&3) + 8(r>>2) + 4(lane>>5)); that is exactly a B operand (lane -> column, 8 values along k) once the k
// index of GEMM2 is PERMUTED to follow the register order -- a free choice, paid for by staging the W2 tile
// in that order.  Consequences: the [tokens, hidden] activation never leaves registers (no LDS round trip,
// no barrier between the GEMMs), the token activations Xt live in registers for the whole kernel (read
// from HBM once, LayerNorm'ed across the two lanes that share a token), and LDS holds only the two weight
// tiles of the current 32-unit hidden group, shared by the 4 waves (128 tokens) of the workgroup.
// One wave per SIMD, up to 512 registers: accumulators in AGPRs, no spills.
//
// Reference: mmcv FFN inside SwinBlock (P/occformer/backbones/modules/window_attention.py:356-361) and
// BaseTransformerLayer ('ffn', 'norm') of the pixel decoder (P/occformer/necks/multiscale_deformattn_3d.py:81).
#include <stdlib.h>
#include "occf_common.h"
#include "../../include/occformer_hip.h"

struct MlpChainArgs {
  const float* x;
  const float* gamma;
  const float* beta;
  const uint16_t* W1h;
  const uint16_t* W1l;
  const float* b1;
  const uint16_t* W2h;
  const uint16_t* W2l;
  const float* b2;
  float* out;
  long M;
  int H, act, ln_mode;        // ln_mode: 0 none, 1 pre-LN (on the MLP input only), 2 post-LN
  float eps;
};

typedef uint32_t mc_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t mc_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mc_bits(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
#else
  return __float_as_uint(x);
#endif
}
__device__ __forceinline__ float mc_from_bits(uint32_t u) {
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}
__device__ __forceinline__ uint32_t mc_bf16_rne(float x) {
  const uint32_t u = mc_bits(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
// 8 fp32 -> bf16x8 hi and lo
__device__ __forceinline__ void mc_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
__device__ __forceinline__ float mc_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int TC, int TERMS>     // C = 64 * TC
__global__ void __launch_bounds__(256) mlp_chain_kernel(MlpChainArgs p) {
  constexpr int C = 64 * TC;
  constexpr int KC = C / 16;                 // k-steps of GEMM1
  constexpr int CT = C / 32;                 // 32-row output tiles of GEMM2
  constexpr int WB = 64 * C;                 // bytes of one weight image (hi or lo) of a 32-unit hidden group
  constexpr int NP = 2 * (WB / 16) / 256;    // 16-byte pieces per thread per weight tile (hi + lo): 4 (C=128), 6 (C=192)
  OCCF_DYN_SMEM(smem);
  unsigned char* W1i = (unsigned char*)smem;           // [hi | lo] images: [KC][32 rows][2 x 16 B]
  unsigned char* W2i = W1i + 2 * WB;                   // [hi | lo] images: [CT][2][32 rows][2 x 16 B]
  float* b1s = (float*)(W2i + 2 * WB);                 // [32] biases of the current hidden group

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long t0 = ((long)blockIdx.x * 4 + wave) * 32;
  const long tok = t0 + li;
  const long tokc = tok < p.M ? tok : p.M - 1;
  const bool three = TERMS == 3;

  // ---- token activations -> registers (B operand of GEMM1: lane = token li, 8 channels 16 ks + 8 lk ..)
  bf16x8 xh[KC], xl[KC];
  {
    float xf[KC][8];
    const float* xr = p.x + tokc * C + lk * 8;
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      const float4 a = *(const float4*)(xr + ks * 16), b = *(const float4*)(xr + ks * 16 + 4);
      xf[ks][0] = a.x; xf[ks][1] = a.y; xf[ks][2] = a.z; xf[ks][3] = a.w;
      xf[ks][4] = b.x; xf[ks][5] = b.y; xf[ks][6] = b.z; xf[ks][7] = b.w;
    }
    if (p.ln_mode == 1) {       // LayerNorm over the C channels of a token = this lane + its partner (lane ^ 32)
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KC; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xf[ks][e];
      s += __shfl_xor(s, 32);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KC; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xf[ks][e] - mean;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const int c0 = ks * 16 + lk * 8;
        const float4 g0 = *(const float4*)(p.gamma + c0), g1 = *(const float4*)(p.gamma + c0 + 4);
        const float4 e0 = *(const float4*)(p.beta + c0), e1 = *(const float4*)(p.beta + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[ks][e] = (xf[ks][e] - mean) * rstd * gg[e] + bb[e];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) mc_split8(xf[ks], xh[ks], xl[ks]);
  }

  f32x16 acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;

  // ---- weight tile staging.  Piece index i in [0, 2*WB/16): array (hi/lo) = i / (WB/16), the rest addresses one
  // 16-byte destination slot.  W1: slot = (ks, row, lk) <- W1[g*32 + row][16 ks + 8 lk .. +7] (16 contiguous bytes).
  // W2: slot = (ct, s2, row, lk) <- { W2[ct*32 + row][g*32 + 16 s2 + 4 lk + (0..3)],
  //                                   W2[ct*32 + row][g*32 + 16 s2 + 8 + 4 lk + (0..3)] }   (the permuted k order)
  const int n_groups = p.H / 32;
  mc_u4 w1r[NP];
  mc_u2 w2r[NP][2];
  auto fetch_w1 = [&](int g) __attribute__((always_inline)) {
    const int gc = g < n_groups ? g : n_groups - 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256;
      const int arr = idx / (WB / 16), s = idx % (WB / 16);
      const int ks = s >> 6, row = (s >> 1) & 31, k2 = s & 1;
      const uint16_t* W = (arr && three) ? p.W1l : p.W1h;
      w1r[i] = *(const mc_u4*)(W + ((long)gc * 32 + row) * C + ks * 16 + k2 * 8);
    }
  };
  auto fetch_w2 = [&](int g) __attribute__((always_inline)) {
    const int gc = g < n_groups ? g : n_groups - 1;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256;
      const int arr = idx / (WB / 16), s = idx % (WB / 16);
      const int ct = s >> 7, s2 = (s >> 6) & 1, row = (s >> 1) & 31, k2 = s & 1;
      const uint16_t* W = (arr && three) ? p.W2l : p.W2h;
      const uint16_t* src = W + ((long)ct * 32 + row) * p.H + (long)gc * 32 + s2 * 16 + k2 * 4;
      w2r[i][0] = *(const mc_u2*)src;
      w2r[i][1] = *(const mc_u2*)(src + 8);
    }
  };
  auto commit_w1 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256;
      if (three || idx < WB / 16) *(mc_u4*)(W1i + idx * 16) = w1r[i];
    }
  };
  auto commit_w2 = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int idx = tid + i * 256;
      if (three || idx < WB / 16) {
        const mc_u4 v = {w2r[i][0][0], w2r[i][0][1], w2r[i][1][0], w2r[i][1][1]};
        *(mc_u4*)(W2i + idx * 16) = v;
      }
    }
  };

  fetch_w1(0);
  fetch_w2(0);
  commit_w1();
  commit_w2();
  if (tid < 32) b1s[tid] = p.b1 ? p.b1[tid] : 0.f;
  __syncthreads();

  for (int g = 0; g < n_groups; ++g) {
    // weights of group g+1 travel to registers while group g multiplies
    fetch_w1(g + 1);
    fetch_w2(g + 1);
    // ---- GEMM1: Ht = W1g . Xt.  Three independent accumulator chains (one per split term; even/odd k-steps
    // in plain bf16): with one wave per SIMD a single dependent MFMA chain would run at half rate.
    f32x16 ht, hu, hv;
#pragma unroll
    for (int r = 0; r < 16; ++r) ht[r] = hu[r] = hv[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      const int off = ks * 1024 + li * 32 + lk * 16;
      const bf16x8 ah = *(const bf16x8*)(W1i + off);
      if (three) {
        const bf16x8 al = *(const bf16x8*)(W1i + WB + off);
        hu = occf_mfma_bf16_32x32x16(al, xh[ks], hu);
        hv = occf_mfma_bf16_32x32x16(ah, xl[ks], hv);
        ht = occf_mfma_bf16_32x32x16(ah, xh[ks], ht);
      } else if (ks & 1) {
        hu = occf_mfma_bf16_32x32x16(ah, xh[ks], hu);
      } else {
        ht = occf_mfma_bf16_32x32x16(ah, xh[ks], ht);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) ht[r] = (hu[r] + hv[r]) + ht[r];
    // ---- bias + activation in registers; registers r = 8 s2 + e are the k values of GEMM2's step s2
    bf16x8 hh[2], hl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = s2 * 8 + e;
        const float t = ht[r] + b1s[(r & 3) + 8 * (r >> 2) + 4 * lk];
        v[e] = p.act == 2 ? mc_gelu(t) : (p.act == 1 ? fmaxf(t, 0.f) : t);
      }
      mc_split8(v, hh[s2], hl[s2]);
    }
    // ---- GEMM2: Ot += W2g . Ht   (consecutive MFMAs go to different output tiles)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 ah[CT], al[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int off = (ct * 2 + s2) * 1024 + li * 32 + lk * 16;
        ah[ct] = *(const bf16x8*)(W2i + off);
        if (three) al[ct] = *(const bf16x8*)(W2i + WB + off);
      }
      if (three) {
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(al[ct], hh[s2], acc[ct]);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(ah[ct], hl[s2], acc[ct]);
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(ah[ct], hh[s2], acc[ct]);
    }
    __syncthreads();                                  // every wave is done with both weight tiles and b1s
    commit_w1();
    commit_w2();
    if (tid < 32 && g + 1 < n_groups) b1s[tid] = p.b1 ? p.b1[(g + 1) * 32 + tid] : 0.f;
    __syncthreads();
  }

  // ---- epilogue: lane = token li; register r of tile ct = channel ct*32 + (r&3) + 8(r>>2) + 4 lk, i.e. four
  // consecutive channels per r>>2 -> one 16-byte piece of the token's row
  float o[CT][16];
  float s = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int c0 = ct * 32 + 8 * q4 + 4 * lk;
      const float4 xr = *(const float4*)(p.x + tokc * C + c0);
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.b2) bv = *(const float4*)(p.b2 + c0);
      o[ct][q4 * 4 + 0] = acc[ct][q4 * 4 + 0] + bv.x + xr.x;
      o[ct][q4 * 4 + 1] = acc[ct][q4 * 4 + 1] + bv.y + xr.y;
      o[ct][q4 * 4 + 2] = acc[ct][q4 * 4 + 2] + bv.z + xr.z;
      o[ct][q4 * 4 + 3] = acc[ct][q4 * 4 + 3] + bv.w + xr.w;
#pragma unroll
      for (int e = 0; e < 4; ++e) s += o[ct][q4 * 4 + e];
    }
  if (p.ln_mode == 2) {
    s += __shfl_xor(s, 32);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = o[ct][r] - mean;
        q = fmaf(d, d, q);
      }
    q += __shfl_xor(q, 32);
    const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = ct * 32 + 8 * q4 + 4 * lk;
        const float4 gg = *(const float4*)(p.gamma + c0), bb = *(const float4*)(p.beta + c0);
        o[ct][q4 * 4 + 0] = (o[ct][q4 * 4 + 0] - mean) * rstd * gg.x + bb.x;
        o[ct][q4 * 4 + 1] = (o[ct][q4 * 4 + 1] - mean) * rstd * gg.y + bb.y;
        o[ct][q4 * 4 + 2] = (o[ct][q4 * 4 + 2] - mean) * rstd * gg.z + bb.z;
        o[ct][q4 * 4 + 3] = (o[ct][q4 * 4 + 3] - mean) * rstd * gg.w + bb.w;
      }
  }
  if (tok < p.M) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = ct * 32 + 8 * q4 + 4 * lk;
        *(float4*)(p.out + tok * C + c0) =
            make_float4(o[ct][q4 * 4 + 0], o[ct][q4 * 4 + 1], o[ct][q4 * 4 + 2], o[ct][q4 * 4 + 3]);
      }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-RESIDENT variant for C = H = 128 (the stage-0 Swin FFN: 680 000 tokens at the 200-grid, two calls per
// forward).  The kernel above re-stages both weight tiles of every 32-unit hidden group for every 128 tokens behind
// two barriers per group (8 barriers, 128 KB of L2 -> LDS per 128 tokens): at C = 128 it measured 0.53 ms per call
// where the token traffic (696 MB) is 0.2 ms and the matrix work 0.1 ms.  Here all four groups' tiles (hi + lo of W1
// and W2: 128 KB) are staged ONCE per workgroup, workgroups are persistent (one per CU) and walk the token tiles;
// the four waves never meet again after the staging barrier, and the next tile's rows are fetched under the current
// tile's MFMAs (one wave per SIMD: nothing else would hide the HBM latency).
// NW = waves per workgroup: 4 (one per SIMD: the next tile's rows are prefetched, three accumulator chains in GEMM1) or
// 8 (two per SIMD sharing the resident weights: a wave's LayerNorm / GELU / split VALU overlaps its partner's MFMAs, the
// partner also hides the row latency -- no prefetch, one accumulator chain, <= 256 registers)
template <int TERMS, int NW>
__global__ void __launch_bounds__(NW * 64) mlp_chain_res_kernel(MlpChainArgs p, long n_tiles) {
  constexpr int NT = NW * 64;
  constexpr int C = 128, KC = C / 16, CT = C / 32, NG = 4;
  constexpr int WB = 64 * C;                 // bytes of one weight image (hi or lo) of a 32-unit hidden group
  constexpr int NPC = (WB / 16) / NT;        // 16-byte pieces per thread per image
  OCCF_DYN_SMEM(smem);
  unsigned char* Wall = (unsigned char*)smem;          // group g: W1 [hi | lo] at g * 4 WB, W2 [hi | lo] at g * 4 WB + 2 WB
  float* b1s = (float*)(Wall + NG * 4 * WB);           // [128]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const bool three = TERMS == 3;

  // ---- all weight tiles -> LDS (same slot order as mlp_chain_kernel's fetch_w1 / fetch_w2)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int arr = 0; arr < (TERMS == 3 ? 2 : 1); ++arr) {
      const uint16_t* W1 = arr ? p.W1l : p.W1h;
      const uint16_t* W2 = arr ? p.W2l : p.W2h;
#pragma unroll
      for (int i = 0; i < NPC; ++i) {
        const int s = tid + i * NT;                     // slot within the image
        {
          const int ks = s >> 6, row = (s >> 1) & 31, k2 = s & 1;
          const mc_u4 v = *(const mc_u4*)(W1 + ((long)g * 32 + row) * C + ks * 16 + k2 * 8);
          *(mc_u4*)(Wall + g * 4 * WB + arr * WB + s * 16) = v;
        }
        {
          const int ct = s >> 7, s2 = (s >> 6) & 1, row = (s >> 1) & 31, k2 = s & 1;
          const uint16_t* src = W2 + ((long)ct * 32 + row) * p.H + (long)g * 32 + s2 * 16 + k2 * 4;
          const mc_u2 a = *(const mc_u2*)src, b = *(const mc_u2*)(src + 8);
          const mc_u4 v = {a[0], a[1], b[0], b[1]};
          *(mc_u4*)(Wall + g * 4 * WB + 2 * WB + arr * WB + s * 16) = v;
        }
      }
    }
  }
  if (tid < 128) b1s[tid] = p.b1 ? p.b1[tid] : 0.f;
  __syncthreads();

  float xf[KC][8];
  auto load_x = [&](long tile) __attribute__((always_inline)) {
    const long tok = (tile * NW + wave) * 32 + li;
    const float* xr = p.x + (tok < p.M ? tok : p.M - 1) * C + lk * 8;
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      const float4 a = *(const float4*)(xr + ks * 16), b = *(const float4*)(xr + ks * 16 + 4);
      xf[ks][0] = a.x; xf[ks][1] = a.y; xf[ks][2] = a.z; xf[ks][3] = a.w;
      xf[ks][4] = b.x; xf[ks][5] = b.y; xf[ks][6] = b.z; xf[ks][7] = b.w;
    }
  };
  long tile = blockIdx.x;
  if (NW == 4 && tile < n_tiles) load_x(tile);
  for (; tile < n_tiles; tile += gridDim.x) {
    if (NW != 4) load_x(tile);
    const long tok = (tile * NW + wave) * 32 + li;
    const long tokc = tok < p.M ? tok : p.M - 1;
    bf16x8 xh[KC], xl[KC];
    if (p.ln_mode == 1) {       // LayerNorm over the C channels of a token = this lane + its partner (lane ^ 32)
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KC; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) s += xf[ks][e];
      s += __shfl_xor(s, 32);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KC; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float d = xf[ks][e] - mean;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const int c0 = ks * 16 + lk * 8;
        const float4 g0 = *(const float4*)(p.gamma + c0), g1 = *(const float4*)(p.gamma + c0 + 4);
        const float4 e0 = *(const float4*)(p.beta + c0), e1 = *(const float4*)(p.beta + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[ks][e] = (xf[ks][e] - mean) * rstd * gg[e] + bb[e];
      }
    }
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) mc_split8(xf[ks], xh[ks], xl[ks]);
    // the next tile's rows travel while this one multiplies (the last tile re-reads itself: no branch around loads)
    if (NW == 4) {
      OCCF_SCHED_FENCE();
      load_x(tile + gridDim.x < n_tiles ? tile + gridDim.x : tile);
      OCCF_SCHED_FENCE();
    }

    f32x16 acc[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const unsigned char* W1i = Wall + g * 4 * WB;
      const unsigned char* W2i = W1i + 2 * WB;
      f32x16 ht, hu, hv;
#pragma unroll
      for (int r = 0; r < 16; ++r) ht[r] = hu[r] = hv[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KC; ++ks) {
        const int off = ks * 1024 + li * 32 + lk * 16;
        const bf16x8 ah = *(const bf16x8*)(W1i + off);
        if (three && NW != 4) {
          const bf16x8 al = *(const bf16x8*)(W1i + WB + off);
          ht = occf_mfma_bf16_32x32x16(al, xh[ks], ht);
          ht = occf_mfma_bf16_32x32x16(ah, xl[ks], ht);
          ht = occf_mfma_bf16_32x32x16(ah, xh[ks], ht);
        } else if (three) {
          const bf16x8 al = *(const bf16x8*)(W1i + WB + off);
          hu = occf_mfma_bf16_32x32x16(al, xh[ks], hu);
          hv = occf_mfma_bf16_32x32x16(ah, xl[ks], hv);
          ht = occf_mfma_bf16_32x32x16(ah, xh[ks], ht);
        } else if (ks & 1) {
          hu = occf_mfma_bf16_32x32x16(ah, xh[ks], hu);
        } else {
          ht = occf_mfma_bf16_32x32x16(ah, xh[ks], ht);
        }
        if (NW != 4 && (ks & 1)) OCCF_SCHED_FENCE();      // (256-register budget: no batch of all 16 fragment reads)
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) ht[r] = (hu[r] + hv[r]) + ht[r];
      bf16x8 hh[2], hl[2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = s2 * 8 + e;
          const float t = ht[r] + b1s[g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk];
          v[e] = p.act == 2 ? mc_gelu(t) : (p.act == 1 ? fmaxf(t, 0.f) : t);
        }
        mc_split8(v, hh[s2], hl[s2]);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 ah[CT], al[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          const int off = (ct * 2 + s2) * 1024 + li * 32 + lk * 16;
          ah[ct] = *(const bf16x8*)(W2i + off);
          if (three) al[ct] = *(const bf16x8*)(W2i + WB + off);
        }
        if (three) {
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(al[ct], hh[s2], acc[ct]);
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(ah[ct], hl[s2], acc[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = occf_mfma_bf16_32x32x16(ah[ct], hh[s2], acc[ct]);
        if (NW != 4) OCCF_SCHED_FENCE();
      }
    }

    // ---- epilogue (as mlp_chain_kernel): residual, optional post-LN, 16-byte stores
    float o[CT][16];
    float s = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int c0 = ct * 32 + 8 * q4 + 4 * lk;
        const float4 xr = *(const float4*)(p.x + tokc * C + c0);
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.b2) bv = *(const float4*)(p.b2 + c0);
        o[ct][q4 * 4 + 0] = acc[ct][q4 * 4 + 0] + bv.x + xr.x;
        o[ct][q4 * 4 + 1] = acc[ct][q4 * 4 + 1] + bv.y + xr.y;
        o[ct][q4 * 4 + 2] = acc[ct][q4 * 4 + 2] + bv.z + xr.z;
        o[ct][q4 * 4 + 3] = acc[ct][q4 * 4 + 3] + bv.w + xr.w;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += o[ct][q4 * 4 + e];
      }
    if (p.ln_mode == 2) {
      s += __shfl_xor(s, 32);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = o[ct][r] - mean;
          q = fmaf(d, d, q);
        }
      q += __shfl_xor(q, 32);
      const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c0 = ct * 32 + 8 * q4 + 4 * lk;
          const float4 gg = *(const float4*)(p.gamma + c0), bb = *(const float4*)(p.beta + c0);
          o[ct][q4 * 4 + 0] = (o[ct][q4 * 4 + 0] - mean) * rstd * gg.x + bb.x;
          o[ct][q4 * 4 + 1] = (o[ct][q4 * 4 + 1] - mean) * rstd * gg.y + bb.y;
          o[ct][q4 * 4 + 2] = (o[ct][q4 * 4 + 2] - mean) * rstd * gg.z + bb.z;
          o[ct][q4 * 4 + 3] = (o[ct][q4 * 4 + 3] - mean) * rstd * gg.w + bb.w;
        }
    }
    if (tok < p.M) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int c0 = ct * 32 + 8 * q4 + 4 * lk;
          *(float4*)(p.out + tok * C + c0) =
              make_float4(o[ct][q4 * 4 + 0], o[ct][q4 * 4 + 1], o[ct][q4 * 4 + 2], o[ct][q4 * 4 + 3]);
        }
    }
  }
}

// returns OCCF_ESHAPE when the shape is outside this kernel's envelope (the caller then uses mlp_fused.hip)
int occf_mlp_chain_launch(const float* x, const float* ln_gamma, const float* ln_beta, const uint16_t* w1_hi,
                          const uint16_t* w1_lo, const float* b1, const uint16_t* w2_hi, const uint16_t* w2_lo,
                          const float* b2, float* out, long M, int C, int H, int act, int ln_mode, float eps, int terms,
                          hipStream_t st) {
  if ((C != 128 && C != 192) || H % 32 != 0 || H <= 0 || M <= 0) return OCCF_ESHAPE;
  MlpChainArgs a = {x, ln_gamma, ln_beta, w1_hi, w1_lo, b1, w2_hi, w2_lo, b2, out, M, H, act, ln_mode, eps};
  if (C == 128 && H == 128) {
    // weight-resident persistent kernel: one workgroup per CU (OCCF_MLP_RES_WGS caps the grid: tests exercise the
    // tile loop with a handful of workgroups)
    const char* e = getenv("OCCF_MLP_RES_WGS");
    const long cap = e && atoi(e) > 0 ? atoi(e) : 256;
    const char* ew = getenv("OCCF_MLP_RES_WAVES");               // 4 or 8 (default) waves per workgroup
    const int nw = ew && atoi(ew) == 4 ? 4 : 8;
    const long n_tiles = occf_cdiv(M, 32 * nw);
    const unsigned grid = (unsigned)(n_tiles < cap ? n_tiles : cap);
    const size_t lds = (size_t)4 * 4 * 64 * 128 + 512;
    typedef void (*fn_t)(MlpChainArgs, long);
    const fn_t fn = nw == 4 ? (terms == 3 ? (fn_t)mlp_chain_res_kernel<3, 4> : (fn_t)mlp_chain_res_kernel<1, 4>)
                            : (terms == 3 ? (fn_t)mlp_chain_res_kernel<3, 8> : (fn_t)mlp_chain_res_kernel<1, 8>);
#ifndef OCCF_EMU
    static bool done[2][2] = {{false, false}, {false, false}};
    if (!done[nw == 8][terms == 3]) {
      hipError_t err = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (err != hipSuccess) return (int)err;
      done[nw == 8][terms == 3] = true;
    }
#endif
    hipLaunchKernelGGL(fn, dim3(grid), dim3(nw * 64), lds, st, a, n_tiles);
    return (int)hipGetLastError();
  }
  const size_t lds = (size_t)4 * 64 * C + 128;
  const unsigned grid = (unsigned)occf_cdiv(M, 128);
#ifndef OCCF_EMU
#define OCCF_MC_ATTR(TC_, T_)                                                                                 \
  do {                                                                                                        \
    static bool done = false;                                                                                 \
    if (!done) {                                                                                              \
      hipError_t e = hipFuncSetAttribute((const void*)mlp_chain_kernel<TC_, T_>,                             \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);             \
      if (e != hipSuccess) return (int)e;                                                                     \
      done = true;                                                                                            \
    }                                                                                                         \
  } while (0)
#else
#define OCCF_MC_ATTR(TC_, T_) do { } while (0)
#endif
#define OCCF_MC_LAUNCH(TC_, T_)                                                                    \
  do {                                                                                             \
    OCCF_MC_ATTR(TC_, T_);                                                                         \
    hipLaunchKernelGGL((mlp_chain_kernel<TC_, T_>), dim3(grid), dim3(256), lds, st, a);            \
  } while (0)
  if (C == 128) { if (terms == 3) OCCF_MC_LAUNCH(2, 3); else OCCF_MC_LAUNCH(2, 1); }
  else { if (terms == 3) OCCF_MC_LAUNCH(3, 3); else OCCF_MC_LAUNCH(3, 1); }
#undef OCCF_MC_LAUNCH
#undef OCCF_MC_ATTR
  return (int)hipGetLastError();
}
