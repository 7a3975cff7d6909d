// Backward of the masked multi-head attention of the occupancy decoder on the matrix cores (head_dim 32, <= 128
// queries): ATen autograd through nn.MultiheadAttention in the reference (P/occformer/mask2former/
// mask2former_nusc_occ.py:652-667).  With P = softmax(mask(S)), S = (q * scale) k^T, D_q = <dO_q, O_q>:
//     dV = P^T dO          dS = P o (dO V^T - D)          dK = dS^T (q * scale)          dQ = scale * dS K
// One kernel per operand orientation (as csrc/attn_bwd.hip does for the window attention), so that every softmax
// quantity sits where the next MFMA wants it and P / dS never leave registers:
//   * xattn_bwd_q_mfma_kernel  -- a wave owns 32 QUERIES (lane -> query column) and walks the 32-key tiles of its key
//     chunk on the TRANSPOSED problem, exactly like the forward (xattn_mfma.hip): St = K Qt, dPt = V dOt (A = key rows
//     from LDS, B = query fragments in registers), dSt in the accumulator layout = the B operand of dQt += Kt dSt.
//     lse / D are per-lane scalars.  dQ leaves as one partial per key chunk (reduced in fixed order afterwards).
//   * xattn_bwd_kv_mfma_kernel -- a wave owns 32 KEYS (lane -> key column; K / V fragments live in registers as B
//     operands, read straight from global memory) and walks the <= 4 query groups: S = Q K^T, dP = dO V^T (A = query
//     rows), P / dS in the accumulator layout = the B operands of dVt += dOt P and dKt += Qt dS (A = transposed query
//     images, staged in LDS once per workgroup with the k order of the accumulator registers).  Every dK / dV element
//     is written exactly once: no atomics, no partials.
// Products: 3-term bf16 split, fp32 accumulate.  The scalar kernel this replaces (xattn_bwd_kernel) spent 0.35 ms per
// call on VALU FMAs (6.3 ms per training step, 11x the forward).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define XG_HD 32
#define XG_IMG 2048            // one 32 x 32 bf16 operand image: [2 k-steps][32 rows][2 x 16 B]

typedef uint32_t xg_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void xg_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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

// 8 consecutive fp32 of a row -> (hi, lo) fragment, optionally scaled; `ok` = false gives zeros
__device__ __forceinline__ void xg_row_frag(const float* p, bool ok, float scale, bf16x8& hi, bf16x8& lo) {
  float f[8];
  if (ok) {
    const float4 a = *(const float4*)p, c = *(const float4*)(p + 4);
    f[0] = a.x * scale; f[1] = a.y * scale; f[2] = a.z * scale; f[3] = a.w * scale;
    f[4] = c.x * scale; f[5] = c.y * scale; f[6] = c.z * scale; f[7] = c.w * scale;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = 0.f;
  }
  xg_split8(f, hi, lo);
}

// row-image store (rows = keys): row r, 4 consecutive d at c4 -> k-step c4 >> 4, slot (c4 >> 3) & 1, offset (c4 & 7)
__device__ __forceinline__ void xg_store_row_image(unsigned char* hi_img, unsigned char* lo_img, int r, int c4, float4 v) {
  uint32_t h0, l0, h1, l1;
  occf_bf16_split2(v.x, v.y, h0, l0);
  occf_bf16_split2(v.z, v.w, h1, l1);
  const int off = (c4 >> 4) * 1024 + r * 32 + ((c4 >> 3) & 1) * 16 + (c4 & 7) * 2;
  const xg_u2 ph = {h0, h1}, pl = {l0, l1};
  *(xg_u2*)(hi_img + off) = ph;
  *(xg_u2*)(lo_img + off) = pl;
}

// transposed-image store (rows = d, k = the 32 rows of the source tile in ACCUMULATOR-register order: source row
// rr = 16 s2 + 8 (e >> 2) + 4 lk' + (e & 3) sits at k-step s2, slot lk', position e)
__device__ __forceinline__ void xg_store_t_image(unsigned char* hi_img, unsigned char* lo_img, int rr, int c4, float4 v) {
  const int s2 = rr >> 4, kk = rr & 15;
  const int e = ((kk >> 3) << 2) | (kk & 3), lkp = (kk >> 2) & 1;
  const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    uint32_t hb, lb;
    occf_bf16_split2(f[j], 0.f, hb, lb);
    const int off = s2 * 1024 + (c4 + j) * 32 + lkp * 16 + e * 2;
    *(uint16_t*)(hi_img + off) = (uint16_t)(hb & 0xFFFFu);
    *(uint16_t*)(lo_img + off) = (uint16_t)(lb & 0xFFFFu);
  }
}

#define XG_MMA3(acc, ah, al, bh, bl)                \
  do {                                              \
    acc = occf_mfma_bf16_32x32x16(al, bh, acc);     \
    acc = occf_mfma_bf16_32x32x16(ah, bl, acc);     \
    acc = occf_mfma_bf16_32x32x16(ah, bh, acc);     \
  } while (0)

// ------------------------------------------------------------------------------------------------ dQ (query-major)
__global__ void __launch_bounds__(256) xattn_bwd_q_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ blocked, const int* __restrict__ row_open, const float* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ Dv, float* __restrict__ dq_part, int B, int Q, int L, int E,
    int heads, int chunk, int n_chunks, float scale) {
  // per buffer: K hi | K lo | V hi | V lo | K^T hi | K^T lo
  __shared__ __attribute__((aligned(16))) unsigned char img[2][6 * XG_IMG];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int k0 = ck * chunk;
  const int k1 = (k0 + chunk < L) ? k0 + chunk : L;
  const int n_tiles = (k1 - k0 + 31) / 32;

  const int qi = wave * 32 + li;
  const bool qvalid = qi < Q;
  const int qc = qvalid ? qi : Q - 1;
  bf16x8 qh[2], ql[2], gh[2], gl[2];
  {
    const float* qp = q + ((long)b * Q + qc) * E + h * XG_HD + lk * 8;
    const float* gp = dout + ((long)b * Q + qc) * E + h * XG_HD + lk * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      xg_row_frag(qp + ks * 16, true, scale, qh[ks], ql[ks]);
      xg_row_frag(gp + ks * 16, qvalid, 1.0f, gh[ks], gl[ks]);
    }
  }
  const long sidx = ((long)b * heads + h) * Q + qc;
  const float my_lse = lse[sidx], my_D = Dv[sidx];
  const bool use_mask = blocked != nullptr && row_open[b * Q + qc] != 0;
  const uint8_t* brow = blocked != nullptr ? blocked + ((long)b * Q + qc) * L : nullptr;

  const int sr = tid >> 3, sc = (tid & 7) * 4;
  float4 rk, rv;
  auto fetch = [&](int t) __attribute__((always_inline)) {
    int key = k0 + t * 32 + sr;
    key = key < k1 ? key : k1 - 1;
    const long src = ((long)b * L + key) * E + h * XG_HD + sc;
    rk = *(const float4*)(k + src);
    rv = *(const float4*)(v + src);
  };
  auto commit = [&](int buf) __attribute__((always_inline)) {
    unsigned char* base = img[buf];
    xg_store_row_image(base, base + XG_IMG, sr, sc, rk);
    xg_store_row_image(base + 2 * XG_IMG, base + 3 * XG_IMG, sr, sc, rv);
    xg_store_t_image(base + 4 * XG_IMG, base + 5 * XG_IMG, sr, sc, rk);
  };

  f32x16 dqt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dqt[r] = 0.f;

  fetch(0);
  commit(0);
  __syncthreads();
  for (int t = 0; t < n_tiles; ++t) {
    const int buf = t & 1;
    fetch(t + 1 < n_tiles ? t + 1 : t);
    const int t0 = k0 + t * 32;
    bool open[16];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = t0 + 8 * g + 4 * lk + e;
        const int kc = key < k1 ? key : k1 - 1;
        const bool masked = use_mask && brow[kc] != 0;
        open[g * 4 + e] = qvalid && key < k1 && !masked;
      }
    const unsigned char* base = img[buf];
    f32x16 st, dpt;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int off = ks * 1024 + li * 32 + lk * 16;
      const bf16x8 kh = *(const bf16x8*)(base + off), kl = *(const bf16x8*)(base + XG_IMG + off);
      const bf16x8 vh = *(const bf16x8*)(base + 2 * XG_IMG + off), vl = *(const bf16x8*)(base + 3 * XG_IMG + off);
      XG_MMA3(st, kh, kl, qh[ks], ql[ks]);
      XG_MMA3(dpt, vh, vl, gh[ks], gl[ks]);
    }
    bf16x8 sh[2], sl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float ds[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = s2 * 8 + e;
        const float p = open[r] ? expf(st[r] - my_lse) : 0.f;
        ds[e] = p * (dpt[r] - my_D);
      }
      xg_split8(ds, sh[s2], sl[s2]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int off = s2 * 1024 + li * 32 + lk * 16;
      const bf16x8 th = *(const bf16x8*)(base + 4 * XG_IMG + off), tl = *(const bf16x8*)(base + 5 * XG_IMG + off);
      XG_MMA3(dqt, th, tl, sh[s2], sl[s2]);
    }
    if (t + 1 < n_tiles) commit(buf ^ 1);
    __syncthreads();
  }
  if (!qvalid) return;
  // lane (query li, lk) holds d = (r & 3) + 8 (r >> 2) + 4 lk
  float* po = dq_part + ((((long)b * heads + h) * Q + qi) * n_chunks + ck) * XG_HD;
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *(float4*)(po + 8 * g + 4 * lk) = make_float4(dqt[g * 4 + 0] * scale, dqt[g * 4 + 1] * scale, dqt[g * 4 + 2] * scale,
                                                  dqt[g * 4 + 3] * scale);
}

// ------------------------------------------------------------------------------------------------ dK, dV (key-major)
__global__ void __launch_bounds__(256) xattn_bwd_kv_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ blocked, const int* __restrict__ row_open, const float* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ Dv, float* __restrict__ dk, float* __restrict__ dv, int B,
    int Q, int L, int E, int heads, float scale) {
  // per query group: (q * scale)^T hi | lo | dO^T hi | lo ; then lse[128], D[128], mask-in-use flags[128]
  __shared__ __attribute__((aligned(16))) unsigned char timg[4][4 * XG_IMG];
  __shared__ __attribute__((aligned(16))) float s_lse[128], s_D[128];
  __shared__ int s_use[128];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int n_groups = (Q + 31) / 32;

  // ---- stage the transposed query images (zeros beyond Q) and the row statistics
  for (int idx = tid; idx < 128 * 8; idx += 256) {
    const int qq = idx >> 3, c4 = (idx & 7) * 4;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), g = a;
    if (qq < Q) {
      const long src = ((long)b * Q + qq) * E + h * XG_HD + c4;
      a = *(const float4*)(q + src);
      a = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
      g = *(const float4*)(dout + src);
    }
    unsigned char* base = timg[qq >> 5];
    xg_store_t_image(base, base + XG_IMG, qq & 31, c4, a);
    xg_store_t_image(base + 2 * XG_IMG, base + 3 * XG_IMG, qq & 31, c4, g);
  }
  if (tid < 128) {
    const bool ok = tid < Q;
    const long sidx = ((long)b * heads + h) * Q + (ok ? tid : 0);
    s_lse[tid] = ok ? lse[sidx] : 0.f;
    s_D[tid] = ok ? Dv[sidx] : 0.f;
    s_use[tid] = ok && blocked != nullptr && row_open[b * Q + tid] != 0;
  }

  // ---- this lane's key (column li of the wave's 32): K and V fragments as B operands
  const int key = (blockIdx.x * 4 + wave) * 32 + li;
  const bool kvalid = key < L;
  const int kc = kvalid ? key : L - 1;
  bf16x8 kh[2], kl[2], vh[2], vl[2];
  {
    const long src = ((long)b * L + kc) * E + h * XG_HD + lk * 8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      xg_row_frag(k + src + ks * 16, true, 1.0f, kh[ks], kl[ks]);
      xg_row_frag(v + src + ks * 16, true, 1.0f, vh[ks], vl[ks]);
    }
  }
  f32x16 dvt, dkt;
#pragma unroll
  for (int r = 0; r < 16; ++r) dvt[r] = dkt[r] = 0.f;
  __syncthreads();

  for (int g = 0; g < n_groups; ++g) {
    // A operands: rows = the group's queries (lane li -> query g * 32 + li, 8 consecutive d at lk * 8)
    const int qa = g * 32 + li;
    const bool qa_ok = qa < Q;
    const long qsrc = ((long)b * Q + (qa_ok ? qa : Q - 1)) * E + h * XG_HD + lk * 8;
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 ah, al, bh, bl;
      xg_row_frag(q + qsrc + ks * 16, qa_ok, scale, ah, al);
      xg_row_frag(dout + qsrc + ks * 16, qa_ok, 1.0f, bh, bl);
      XG_MMA3(s, ah, al, kh[ks], kl[ks]);
      XG_MMA3(dp, bh, bl, vh[ks], vl[ks]);
    }
    // accumulator: column = key li, register r -> query row g * 32 + (r & 3) + 8 (r >> 2) + 4 lk
    bf16x8 ph[2], pl[2], sh[2], sl[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float pv[8], ds[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int r = s2 * 8 + e;
        const int qq = g * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        bool open = kvalid && qq < Q;
        if (open && s_use[qq]) open = blocked[((long)b * Q + qq) * L + key] == 0;
        const float p = open ? expf(s[r] - s_lse[qq]) : 0.f;
        pv[e] = p;
        ds[e] = p * (dp[r] - s_D[qq]);
      }
      xg_split8(pv, ph[s2], pl[s2]);
      xg_split8(ds, sh[s2], sl[s2]);
    }
    const unsigned char* base = timg[g];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      const int off = s2 * 1024 + li * 32 + lk * 16;
      const bf16x8 qth = *(const bf16x8*)(base + off), qtl = *(const bf16x8*)(base + XG_IMG + off);
      const bf16x8 gth = *(const bf16x8*)(base + 2 * XG_IMG + off), gtl = *(const bf16x8*)(base + 3 * XG_IMG + off);
      XG_MMA3(dvt, gth, gtl, ph[s2], pl[s2]);
      XG_MMA3(dkt, qth, qtl, sh[s2], sl[s2]);
    }
  }
  if (!kvalid) return;
  const long dst = ((long)b * L + key) * E + h * XG_HD;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    *(float4*)(dv + dst + 8 * g + 4 * lk) = make_float4(dvt[g * 4 + 0], dvt[g * 4 + 1], dvt[g * 4 + 2], dvt[g * 4 + 3]);
    *(float4*)(dk + dst + 8 * g + 4 * lk) = make_float4(dkt[g * 4 + 0], dkt[g * 4 + 1], dkt[g * 4 + 2], dkt[g * 4 + 3]);
  }
}

// launchers used by occf_masked_xattn_bwd (attn_bwd.hip)
void occf_xattn_bwd_mfma_launch(const float* q, const float* k, const float* v, const uint8_t* blocked,
                                const int* row_open, const float* dout, const float* lse, const float* Dv,
                                float* dq_part, float* dk, float* dv, int B, int Q, int L, int E, int heads, int chunk,
                                int n_chunks, float scale, hipStream_t st) {
  hipLaunchKernelGGL(xattn_bwd_q_mfma_kernel, dim3(n_chunks, heads, B), dim3(256), 0, st, q, k, v, blocked, row_open,
                     dout, lse, Dv, dq_part, B, Q, L, E, heads, chunk, n_chunks, scale);
  hipLaunchKernelGGL(xattn_bwd_kv_mfma_kernel, dim3(occf_cdiv(L, 128), heads, B), dim3(256), 0, st, q, k, v, blocked,
                     row_open, dout, lse, Dv, dk, dv, B, Q, L, E, heads, scale);
}
