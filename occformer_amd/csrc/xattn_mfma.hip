// Masked multi-head cross-attention partials on the matrix cores (head_dim = 32, <= 128 queries).
// Same contract and partial format as masked_xattn_partial_kernel (mask_head.hip): one workgroup per
// (key chunk, head, batch) writes per-query (max, sum, 32-wide weighted value sum); masked_xattn_merge_kernel
// combines the chunks.  Reference: nn.MultiheadAttention core inside mmcv MultiheadAttention as used by
// P/occformer/mask2former/mask2former_nusc_occ.py:652-667 (boolean attn_mask, all-masked rows re-opened :652-653).
//
// Register-chained like mlp_chain.hip: a wave owns 32 queries and works on the transposed problem
//     St[32 keys x 32 queries] = Kt[32 x 32 d] . Qt[32 d x 32 q]          A = key rows (LDS), B = Qt (registers)
//     Ot[32 d x 32 queries]   += Vt^T[32 d x 32 keys] . Pt[32 keys x 32 q]  A = V^T rows (LDS), B = Pt (registers)
// The MFMA result layout (lane -> query column, registers -> key rows (r&3) + 8(r>>2) + 4(lane>>5)) IS a B
// operand once GEMM2's k index follows the register order, so the probabilities never leave registers; the
// softmax statistics of a query live in its two lanes (lane, lane ^ 32).  Products: 3-term bf16 split.
// The scalar kernel spends 64 VALU FMAs per (query, key, head); here a 32 x 32 tile costs 12 MFMAs.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define XM_HD 32

typedef uint32_t xm_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t xm_bits(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
#else
  return __float_as_uint(x);
#endif
}
__device__ __forceinline__ float xm_from_bits(uint32_t u) {
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}
__device__ __forceinline__ uint32_t xm_bf16(float x) {
  const uint32_t u = xm_bits(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void xm_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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

// LDS images of one 32-key tile (bytes): K hi/lo [2 ks][32 keys][2 x 16 B], V^T hi/lo [2 s2][32 d][2 x 16 B]
#define XM_IMG 2048

__global__ void __launch_bounds__(256) masked_xattn_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ blocked, const int* __restrict__ row_open, float* __restrict__ part_o,
    float* __restrict__ part_ml, int B, int Q, int L, int E, int heads, int chunk, int n_chunks, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char img[2][4 * XM_IMG];   // [buffer][Kh | Kl | Vh | Vl]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int k0 = ck * chunk;
  const int k1 = (k0 + chunk < L) ? k0 + chunk : L;
  const int n_tiles = (k1 - k0 + 31) / 32;

  // ---- this lane's query (column li of the wave's 32): B operand of GEMM1, pre-scaled
  const int qi = wave * 32 + li;
  const bool qvalid = qi < Q;
  const int qc = qvalid ? qi : Q - 1;
  bf16x8 qh[2], ql[2];
  {
    const float* qp = q + ((long)b * Q + qc) * E + h * XM_HD + lk * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const float4 a = *(const float4*)(qp + ks * 16), c = *(const float4*)(qp + ks * 16 + 4);
      const float f[8] = {a.x * scale, a.y * scale, a.z * scale, a.w * scale,
                          c.x * scale, c.y * scale, c.z * scale, c.w * scale};
      xm_split8(f, qh[ks], ql[ks]);
    }
  }
  const bool use_mask = blocked != nullptr && row_open[b * Q + qc] != 0;
  const uint8_t* brow = blocked != nullptr ? blocked + ((long)b * Q + qc) * L : nullptr;

  // ---- tile staging: thread -> (key row r = tid >> 3, 4 channels c4 = (tid & 7) * 4) of K and of V
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  float4 rk, rv;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    int key = k0 + t * 32 + sr;
    key = key < k1 ? key : k1 - 1;                   // rows past the chunk re-read its last key (masked below)
    const long src = ((long)b * L + key) * E + h * XM_HD + sc;
    rk = *(const float4*)(k + src);
    rv = *(const float4*)(v + src);
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
    unsigned char* base = img[buf];
    // K image: row = key sr, k-step ks = sc >> 4, slot lk' = (sc >> 3) & 1, 4 consecutive d at (sc & 7)
    {
      const float f[4] = {rk.x, rk.y, rk.z, rk.w};
      uint32_t hb[4], lb[4];
      {
        uint32_t h0, l0, h1, l1;
        occf_bf16_split2(f[0], f[1], h0, l0);
        occf_bf16_split2(f[2], f[3], h1, l1);
        hb[0] = h0 & 0xFFFFu; hb[1] = h0 >> 16; hb[2] = h1 & 0xFFFFu; hb[3] = h1 >> 16;
        lb[0] = l0 & 0xFFFFu; lb[1] = l0 >> 16; lb[2] = l1 & 0xFFFFu; lb[3] = l1 >> 16;
      }
      const int off = (sc >> 4) * 1024 + sr * 32 + ((sc >> 3) & 1) * 16 + (sc & 7) * 2;
      const xm_u2 ph = {hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)};
      const xm_u2 pl = {lb[0] | (lb[1] << 16), lb[2] | (lb[3] << 16)};
      *(xm_u2*)(base + off) = ph;
      *(xm_u2*)(base + XM_IMG + off) = pl;
    }
    // V^T image: row = channel d, the key sr sits at the permuted k position: key = 16 s2 + 8 (e>>2) + 4 lk' + (e&3)
    {
      const float f[4] = {rv.x, rv.y, rv.z, rv.w};
      const int s2 = sr >> 4, kk = sr & 15;
      const int e = ((kk >> 3) << 2) | (kk & 3), lkp = (kk >> 2) & 1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        uint32_t hb, lb;
        occf_bf16_split2(f[j], 0.f, hb, lb);
        hb &= 0xFFFFu;
        lb &= 0xFFFFu;
        const int off = s2 * 1024 + (sc + j) * 32 + lkp * 16 + e * 2;
        *(uint16_t*)(base + 2 * XM_IMG + off) = (uint16_t)hb;
        *(uint16_t*)(base + 3 * XM_IMG + off) = (uint16_t)lb;
      }
    }
  };

  f32x16 ot;
#pragma unroll
  for (int r = 0; r < 16; ++r) ot[r] = 0.f;
  float m = -INFINITY, l = 0.f;

  fetch(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    fetch(t + 1 < n_tiles ? t + 1 : t);
    // mask bytes of this lane's query for its 16 keys of the tile (key = t0 + 8 (r>>2) + 4 lk + (r&3))
    const int t0 = k0 + t * 32;
    bool open[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t0 + 8 * g + 4 * lk + e;
        const int kc = key < k1 ? key : k1 - 1;
        const bool masked = use_mask && brow[kc] != 0;
        open[g * 4 + e] = key < k1 && !masked;
      }
    // ---- GEMM1: St = Kt . Qt
    const unsigned char* base = img[buf];
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ks * 1024 + li * 32 + lk * 16;
      const bf16x8 ah = *(const bf16x8*)(base + off);
      const bf16x8 al = *(const bf16x8*)(base + XM_IMG + off);
      st = occf_mfma_bf16_32x32x16(al, qh[ks], st);
      st = occf_mfma_bf16_32x32x16(ah, ql[ks], st);
      st = occf_mfma_bf16_32x32x16(ah, qh[ks], st);
    }
    // ---- online softmax over the keys of this tile (a query = lanes li and li + 32)
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = open[r] ? st[r] : -INFINITY;
      mt = fmaxf(mt, st[r]);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32));
    const float m_new = fmaxf(m, mt);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float c = expf(m - m_use);                  // m = -inf -> 0
    l *= c;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] *= c;
    bf16x8 ph[2], pl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float pv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        pv[e] = expf(st[s2 * 8 + e] - m_use);         // masked: exp(-inf) = 0
        l += pv[e];
      }
      xm_split8(pv, ph[s2], pl[s2]);
    }
    m = m_new;
    // ---- GEMM2: Ot += V^T . Pt
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int off = s2 * 1024 + li * 32 + lk * 16;
      const bf16x8 ah = *(const bf16x8*)(base + 2 * XM_IMG + off);
      const bf16x8 al = *(const bf16x8*)(base + 3 * XM_IMG + off);
      ot = occf_mfma_bf16_32x32x16(al, ph[s2], ot);
      ot = occf_mfma_bf16_32x32x16(ah, pl[s2], ot);
      ot = occf_mfma_bf16_32x32x16(ah, ph[s2], ot);
    }
    // next tile -> the other buffer (its last readers finished before the previous barrier)
    if (t + 1 < n_tiles) commit(buf ^ 1);
    __syncthreads();
  }

  // ---- partials: lane (query li, lk) holds d = (r&3) + 8(r>>2) + 4 lk; the pair's l adds up
  l += __shfl_xor(l, 32);
  if (!qvalid) return;
  const long slot = (((long)b * heads + h) * Q + qi) * n_chunks + ck;
  if (lk == 0) {
    part_ml[slot * 2 + 0] = m;
    part_ml[slot * 2 + 1] = l;
  }
  float* po = part_o + slot * XM_HD;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *(float4*)(po + 8 * g + 4 * lk) = make_float4(ot[g * 4 + 0], ot[g * 4 + 1], ot[g * 4 + 2], ot[g * 4 + 3]);
}

// launcher used by occf_masked_xattn_fwd (mask_head.hip)
void occf_xattn_mfma_launch(const float* q, const float* k, const float* v, const uint8_t* blocked,
                            const int* row_open, float* part_o, float* part_ml, int B, int Q, int L, int E, int heads,
                            int chunk, int n_chunks, float scale, hipStream_t st) {
  hipLaunchKernelGGL(masked_xattn_mfma_kernel, dim3(n_chunks, heads, B), dim3(256), 0, st, q, k, v, blocked,
                     row_open, part_o, part_ml, B, Q, L, E, heads, chunk, n_chunks, scale);
}
