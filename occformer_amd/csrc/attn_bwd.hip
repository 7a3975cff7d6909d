// Backward of the two attention cores of the path (training step; ATen autograd through
// WindowMSA.forward / nn.MultiheadAttention in the reference: window_attention.py:69-107,168-242;
// mask2former_nusc_occ.py:652-667):
//   * occf_window_attn_bwd   -- 7x7 shifted-window attention of the shared SwinBlock
//   * occf_masked_xattn_bwd  -- masked multi-head cross / self attention of the occupancy decoder
// Both recompute the probabilities from q, k (flash-style: nothing of size [rows, keys] is saved by the forward)
// and use D_i = <dO_i, O_i> for the softmax Jacobian.  fp32 VALU arithmetic; parameter gradients (relative
// position table, the qkv bias seen by zero-padded window tokens) as per-workgroup partials reduced in fixed
// order, except the few padded-token bias rows (float atomics).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define WB_WS 7
#define WB_T 49
#define WB_HD 32
#define WB_NB 169          // (2*7-1)^2 relative positions
#define WB_HPB 2           // heads (= waves) per workgroup

// --------------------------------------------------------------------------------------------- window attention
// One wave per (window, head); lane t < 49 first acts as QUERY row t (P, dS rows in registers, dQ), then as KEY
// row t (dK, dV as sums over the queries, read from the LDS copies of P / dS).  A workgroup walks `wpb`
// consecutive windows and keeps the relative-position-bias gradient of its heads in registers.
__global__ void __launch_bounds__(64 * WB_HPB) window_attn_bwd_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    const float* __restrict__ attn_out, const float* __restrict__ dout, float* __restrict__ dqkv,
    float* __restrict__ dqkv_bias, float* __restrict__ dtable_partial, int B, int X, int Y, int S, int C, int heads,
    int shift, float scale, int wpb, long n_windows) {
  __shared__ __attribute__((aligned(16))) float lds_k[WB_HPB][WB_T * WB_HD];
  __shared__ __attribute__((aligned(16))) float lds_v[WB_HPB][WB_T * WB_HD];
  __shared__ float lds_m[WB_HPB][WB_T * WB_T];
  __shared__ float lds_bias[WB_HPB][WB_NB];
  __shared__ int lds_tok[WB_HPB][WB_T];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int nwx = (X + WB_WS - 1) / WB_WS, nwy = (Y + WB_WS - 1) / WB_WS;
  const int Xp = nwx * WB_WS, Yp = nwy * WB_WS;
  const int head = blockIdx.y * WB_HPB + wave;
  const bool active = head < heads;
  const int C3 = 3 * C;
  if (active)
    for (int t = lane; t < WB_NB; t += 64) lds_bias[wave][t] = bias_table[(long)t * heads + head];
  float dtab[3] = {0.f, 0.f, 0.f};                           // bins lane, lane + 64, lane + 128

  for (int wi = 0; wi < wpb; ++wi) {
    const long win = (long)blockIdx.x * wpb + wi;
    if (win >= n_windows) break;                               // uniform over the workgroup
    // slice index fastest: the S slices of one spatial window are S ADJACENT token rows per window position (26 KB
    // contiguous at 17 x 384 floats), so the windows a workgroup walks reuse the same pages and cache lines -- with
    // y fastest every window touched 49 fresh 4 KB pages per tensor and the kernel sat in s_waitcnt (61 % of the
    // wave cycles, PMC r02 probe)
    long w = win;
    const int s = (int)(w % S);
    w /= S;
    const int wy = (int)(w % nwy);
    w /= nwy;
    const int wx = (int)(w % nwx);
    const int b = (int)(w / nwx);

    int my_tok = -1, my_region = 0;
    if (lane < WB_T) {
      const int i = lane / WB_WS, j = lane % WB_WS;
      const int px = wx * WB_WS + i, py = wy * WB_WS + j;
      int sx = px + shift, sy = py + shift;
      if (sx >= Xp) sx -= Xp;
      if (sy >= Yp) sy -= Yp;
      if (sx < X && sy < Y) my_tok = (int)((((long)b * X + sx) * Y + sy) * S + s);
      if (shift > 0) {
        const int rx = px < Xp - WB_WS ? 0 : (px < Xp - shift ? 1 : 2);
        const int ry = py < Yp - WB_WS ? 0 : (py < Yp - shift ? 1 : 2);
        my_region = rx * 3 + ry;
      }
    }
    __syncthreads();                                           // previous window's LDS reads are done
    if (lane < WB_T) lds_tok[wave][lane] = my_tok;
    __syncthreads();
    if (active) {
      for (int idx = lane; idx < WB_T * (WB_HD / 4); idx += 64) {
        const int t = idx >> 3, q4 = (idx & 7) * 4;
        const int tok = lds_tok[wave][t];
        const float* src = tok >= 0 ? qkv + (long)tok * C3 : qkv_bias;
        *(float4*)(&lds_k[wave][t * WB_HD + q4]) = *(const float4*)(src + C + head * WB_HD + q4);
        *(float4*)(&lds_v[wave][t * WB_HD + q4]) = *(const float4*)(src + 2 * C + head * WB_HD + q4);
      }
    }
    __syncthreads();

    // ---- query role.  Padded query rows are cropped by the forward: P = dS = 0, no dQ.
    const bool qrow = active && lane < WB_T && my_tok >= 0;
    float q[WB_HD], go[WB_HD], dq[WB_HD];
    float sc[WB_T];
    float Dq = 0.f;
#pragma unroll
    for (int d = 0; d < WB_HD; ++d) q[d] = go[d] = dq[d] = 0.f;
    if (qrow) {
      const float* src = qkv + (long)my_tok * C3 + head * WB_HD;
      const float* gsrc = dout + (long)my_tok * C + head * WB_HD;
      const float* osrc = attn_out + (long)my_tok * C + head * WB_HD;
#pragma unroll
      for (int d = 0; d < WB_HD; d += 4) {
        const float4 t = *(const float4*)(src + d);
        const float4 g = *(const float4*)(gsrc + d);
        const float4 o = *(const float4*)(osrc + d);
        q[d] = t.x * scale; q[d + 1] = t.y * scale; q[d + 2] = t.z * scale; q[d + 3] = t.w * scale;
        go[d] = g.x; go[d + 1] = g.y; go[d + 2] = g.z; go[d + 3] = g.w;
        Dq += (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
      }
    }
    const int qi = lane / WB_WS, qj = lane % WB_WS;
    if (lane < WB_T) {
      float mx = -3.0e38f;
#pragma unroll
      for (int j = 0; j < WB_T; ++j) {
        const float* kr = &lds_k[wave][j * WB_HD];
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;           // four independent chains (FMA latency)
#pragma unroll
        for (int d = 0; d < WB_HD; d += 4) {
          a0 = fmaf(q[d], kr[d], a0);
          a1 = fmaf(q[d + 1], kr[d + 1], a1);
          a2 = fmaf(q[d + 2], kr[d + 2], a2);
          a3 = fmaf(q[d + 3], kr[d + 3], a3);
        }
        float a = (a0 + a1) + (a2 + a3);
        const int ki = j / WB_WS, kj = j % WB_WS;
        a += lds_bias[wave][(qi - ki + WB_WS - 1) * (2 * WB_WS - 1) + (qj - kj + WB_WS - 1)];
        if (shift > 0) {
          const int px = wx * WB_WS + ki, py = wy * WB_WS + kj;
          const int rx = px < Xp - WB_WS ? 0 : (px < Xp - shift ? 1 : 2);
          const int ry = py < Yp - WB_WS ? 0 : (py < Yp - shift ? 1 : 2);
          if (rx * 3 + ry != my_region) a += -100.0f;
        }
        sc[j] = a;
        mx = fmaxf(mx, a);
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < WB_T; ++j) {
        sc[j] = expf(sc[j] - mx);
        sum += sc[j];
      }
      const float inv = qrow ? 1.0f / sum : 0.f;
#pragma unroll
      for (int j = 0; j < WB_T; ++j) {
        sc[j] *= inv;                                          // P[i][j]
        lds_m[wave][lane * WB_T + j] = sc[j];
      }
      // dS = P * (dO . V_j - D);  dQ = scale * sum_j dS K_j
#pragma unroll
      for (int j = 0; j < WB_T; ++j) {
        const float* vr = &lds_v[wave][j * WB_HD];
        const float* kr = &lds_k[wave][j * WB_HD];
        float p0 = 0.f, p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
        for (int d = 0; d < WB_HD; d += 4) {
          p0 = fmaf(go[d], vr[d], p0);
          p1 = fmaf(go[d + 1], vr[d + 1], p1);
          p2 = fmaf(go[d + 2], vr[d + 2], p2);
          p3 = fmaf(go[d + 3], vr[d + 3], p3);
        }
        const float dp = (p0 + p1) + (p2 + p3);
        const float ds = sc[j] * (dp - Dq);
        sc[j] = ds;
#pragma unroll
        for (int d = 0; d < WB_HD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
      }
      if (qrow) {
        float* dst = dqkv + (long)my_tok * C3 + head * WB_HD;
#pragma unroll
        for (int d = 0; d < WB_HD; d += 4)
          *(float4*)(dst + d) = make_float4(dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale);
      }
    }
    __syncthreads();                                           // every lane is done with K / V
    // scaled queries -> K buffer, dO -> V buffer
    if (lane < WB_T) {
#pragma unroll
      for (int d = 0; d < WB_HD; d += 4) {
        *(float4*)(&lds_k[wave][lane * WB_HD + d]) = make_float4(q[d], q[d + 1], q[d + 2], q[d + 3]);
        *(float4*)(&lds_v[wave][lane * WB_HD + d]) = make_float4(go[d], go[d + 1], go[d + 2], go[d + 3]);
      }
    }
    __syncthreads();
    // ---- key role: dV_j = sum_i P[i][j] dO_i
    float acc[WB_HD];
    if (active && lane < WB_T) {
#pragma unroll
      for (int d = 0; d < WB_HD; ++d) acc[d] = 0.f;
      for (int i = 0; i < WB_T; ++i) {
        const float pij = lds_m[wave][i * WB_T + lane];
        const float* gr = &lds_v[wave][i * WB_HD];
#pragma unroll
        for (int d = 0; d < WB_HD; ++d) acc[d] = fmaf(pij, gr[d], acc[d]);
      }
      if (my_tok >= 0) {
        float* dst = dqkv + (long)my_tok * C3 + 2 * C + head * WB_HD;
#pragma unroll
        for (int d = 0; d < WB_HD; d += 4) *(float4*)(dst + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
      } else {
#pragma unroll
        for (int d = 0; d < WB_HD; ++d) atomicAdd(dqkv_bias + 2 * C + head * WB_HD + d, acc[d]);
      }
    }
    __syncthreads();                                           // P has been consumed
    if (lane < WB_T) {
#pragma unroll
      for (int j = 0; j < WB_T; ++j) lds_m[wave][lane * WB_T + j] = sc[j];    // dS[i][j]
    }
    __syncthreads();
    if (active && lane < WB_T) {
#pragma unroll
      for (int d = 0; d < WB_HD; ++d) acc[d] = 0.f;
      for (int i = 0; i < WB_T; ++i) {
        const float dij = lds_m[wave][i * WB_T + lane];
        const float* qr = &lds_k[wave][i * WB_HD];
#pragma unroll
        for (int d = 0; d < WB_HD; ++d) acc[d] = fmaf(dij, qr[d], acc[d]);
      }
      if (my_tok >= 0) {
        float* dst = dqkv + (long)my_tok * C3 + C + head * WB_HD;
#pragma unroll
        for (int d = 0; d < WB_HD; d += 4) *(float4*)(dst + d) = make_float4(acc[d], acc[d + 1], acc[d + 2], acc[d + 3]);
      } else {
#pragma unroll
        for (int d = 0; d < WB_HD; ++d) atomicAdd(dqkv_bias + C + head * WB_HD + d, acc[d]);
      }
    }
    // ---- relative-position bias: bin r = (di + 6) * 13 + (dj + 6) collects dS[(ki+di, kj+dj)][(ki, kj)]
    if (active) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int r = lane + u * 64;
        if (r < WB_NB) {
          const int di = r / (2 * WB_WS - 1) - (WB_WS - 1), dj = r % (2 * WB_WS - 1) - (WB_WS - 1);
          float a = 0.f;
          for (int ki = 0; ki < WB_WS; ++ki) {
            const int qi2 = ki + di;
            if (qi2 < 0 || qi2 >= WB_WS) continue;
            for (int kj = 0; kj < WB_WS; ++kj) {
              const int qj2 = kj + dj;
              if (qj2 < 0 || qj2 >= WB_WS) continue;
              a += lds_m[wave][(qi2 * WB_WS + qj2) * WB_T + ki * WB_WS + kj];
            }
          }
          dtab[u] += a;
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int r = lane + u * 64;
      if (r < WB_NB) dtable_partial[((long)blockIdx.x * WB_NB + r) * heads + head] = dtab[u];
    }
  }
}


// --------------------------------------------------------------------------------------------- matrix-core variant
// Same contract as window_attn_bwd_kernel; the five 64 x 64 x 32 contractions per (window, head) run on
// v_mfma_f32_32x32x16_bf16 (3-term bf16 split, fp32 accumulate).  With D = A . B returning lane -> column,
// registers -> rows (r&3) + 8 (r>>2) + 4 (lane>>5), and an operand lane supplying 8 consecutive k, the transposed
// products need P / dS with the lane on the QUERY for dQ and on the KEY for dK, dV.  Two kernels, one per orientation:
//   window_attn_bwd_q_kernel  (lane = query):  St = K . Q^T,  dPt = V . dO^T  -> softmax statistics of a query in ONE
//        lane pair -> lse, D = <dO, O> (stored for the second kernel), dS;  dQt[d][query] = K^T[d][key] . dSt[key][query];
//        relative-position-bias gradient from an LDS copy of dS
//   window_attn_bwd_kv_kernel (lane = key):    S' = Q . K^T,  dP' = dO . V^T,  P' = exp(S' - lse), dS' = P' (dP' - D);
//        dVt[d][key] = dO^T[d][query] . P'[query][key],  dKt[d][key] = Qs^T[d][query] . dS'[query][key]
// Every operand comes straight from global memory: row operands (8 consecutive d of one token) as two 16-byte loads,
// transposed operands (8 tokens of one d) as 8 dword loads -- the 32 d-lanes of a half wave read one 128-byte row
// segment, fully coalesced; the rows were just read by the other loads (L1 / L2 hits).  The first version of this
// kernel (r02c..r02f) did both orientations in one wave with three transposed LDS images: 478 VGPRs and 38 KB of LDS
// per wave = ONE wave per SIMD, every LDS / memory latency exposed (61 % of the wave cycles in s_waitcnt, 3.85 ms
// per call, 31 ms per training step).  Split, each kernel stays below 256 registers and 14 KB of LDS per wave.
typedef uint32_t wbm_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void wbm_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
// 8 consecutive floats of a row, scaled, as hi / lo operand
__device__ __forceinline__ void wbm_row8(const float* __restrict__ p, float m, bf16x8& hi, bf16x8& lo) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  const float f[8] = {a.x * m, a.y * m, a.z * m, a.w * m, b.x * m, b.y * m, b.z * m, b.w * m};
  wbm_split8(f, hi, lo);
}
// token of element e of 16-token k-step kst for the lane half lk: the order in which a C-layout register file hands
// its rows on as B operand (t & 15 = 8 (e >> 2) + 4 lk + (e & 3))
__device__ __forceinline__ int wbm_ktok(int kst, int e, int lk) { return kst * 16 + 8 * (e >> 2) + 4 * lk + (e & 3); }

// window decode shared by both kernels: token index (or -1) and shift-mask region of window position `lane`
__device__ __forceinline__ void wbm_window_tokens(long win, int lane, int B, int X, int Y, int S, int shift, int& tok,
                                                  int& region) {
  const int nwx = (X + WB_WS - 1) / WB_WS, nwy = (Y + WB_WS - 1) / WB_WS;
  const int Xp = nwx * WB_WS, Yp = nwy * WB_WS;
  // slice index fastest: the S slices of one spatial window are S adjacent token rows per window position
  long w = win;
  const int s = (int)(w % S);
  w /= S;
  const int wy = (int)(w % nwy);
  w /= nwy;
  const int wx = (int)(w % nwx);
  const int b = (int)(w / nwx);
  tok = -1;
  region = 0;
  if (lane < WB_T) {
    const int i = lane / WB_WS, j = lane % WB_WS;
    const int px = wx * WB_WS + i, py = wy * WB_WS + j;
    int sx = px + shift, sy = py + shift;
    if (sx >= Xp) sx -= Xp;
    if (sy >= Yp) sy -= Yp;
    if (sx < X && sy < Y) tok = (int)((((long)b * X + sx) * Y + sy) * S + s);
    if (shift > 0) {
      const int rx = px < Xp - WB_WS ? 0 : (px < Xp - shift ? 1 : 2);
      const int ry = py < Yp - WB_WS ? 0 : (py < Yp - shift ? 1 : 2);
      region = rx * 3 + ry;
    }
  }
}

#define WBM_MSTR 64                       // row stride of the dS copy [key][query]
#define WBM_MPAD 48                       // |query - key| offsets of a bin reach +-48: guard floats on both sides
#define WBM_MSIZE (WBM_MPAD + WB_T * WBM_MSTR + WBM_MSTR + WBM_MPAD)

// workgroup = the two 32-query tiles of ONE (window, head): wave = query tile (K, V, K^T operands are loaded by
// both waves; the second read hits the L1), the dS copy and the 169 bins are shared
__global__ void __launch_bounds__(128, 2) window_attn_bwd_q_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    const float* __restrict__ attn_out, const float* __restrict__ dout, float* __restrict__ dqkv,
    float* __restrict__ dtable_partial, float* __restrict__ ws_stats, int B, int X, int Y, int S, int C, int heads,
    int shift, float scale, int wpb, long n_windows) {
  __shared__ float lds_bias[WB_NB];
  __shared__ float lds_m[WBM_MSIZE];                  // dS[key][query] of the current window (bias-table gradient)
  __shared__ int lds_tok[64], lds_reg[64];

  const int qt = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int head = blockIdx.y;
  const int C3 = 3 * C;
  for (int t = threadIdx.x; t < WB_NB; t += 128) lds_bias[t] = bias_table[(long)t * heads + head];
  for (int t = threadIdx.x; t < WBM_MSIZE; t += 128) lds_m[t] = 0.f;
  // bins tid, tid + 128 of the table gradient: offset query - key of the bin and the keys it is valid for
  float dtab[2] = {0.f, 0.f};
  int bin_off[2];
  unsigned long long bin_keys[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int r = (int)threadIdx.x + u * 128;
    bin_off[u] = 0;
    bin_keys[u] = 0ull;
    if (r < WB_NB) {
      const int di = r / (2 * WB_WS - 1) - (WB_WS - 1), dj = r % (2 * WB_WS - 1) - (WB_WS - 1);
      bin_off[u] = di * WB_WS + dj;
      for (int ki = 0; ki < WB_WS; ++ki)
        for (int kj = 0; kj < WB_WS; ++kj)
          if (ki + di >= 0 && ki + di < WB_WS && kj + dj >= 0 && kj + dj < WB_WS)
            bin_keys[u] |= 1ull << (ki * WB_WS + kj);
    }
  }

  for (int wi = 0; wi < wpb; ++wi) {
    const long win = (long)blockIdx.x * wpb + wi;
    if (win >= n_windows) break;
    __syncthreads();                                           // previous window's LDS contents are consumed
    {
      int tok, region;
      wbm_window_tokens(win, lane, B, X, Y, S, shift, tok, region);
      if (qt == 0) {
        lds_tok[lane] = tok;
        lds_reg[lane] = region;
      }
    }
    __syncthreads();

    // ---- A operands: K, V rows of this lane's two key tokens (tile kt: token 32 kt + li), d = 16 ks + 8 lk ..+8.
    // rows 49..63 are zero; padded window positions (t < 49, no token) take the bias row (the forward pads x with
    // zeros BEFORE the qkv projection)
    bf16x8 Kh[2][2], Kl[2][2], Vh[2][2], Vl[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int t = kt * 32 + li;
      const int tok = lds_tok[t];
      const float km = t < WB_T ? 1.f : 0.f;
      const float* src = (tok >= 0 ? qkv + (long)tok * C3 : qkv_bias) + head * WB_HD + lk * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        wbm_row8(src + C + ks * 16, km, Kh[kt][ks], Kl[kt][ks]);
        wbm_row8(src + 2 * C + ks * 16, km, Vh[kt][ks], Vl[kt][ks]);
      }
    }
    {
      const int qi = qt * 32 + li;
      const int qtok = lds_tok[qi];
      const bool qreal = qi < WB_T && qtok >= 0;
      const float qm = qreal ? 1.f : 0.f;
      const int qrow = (qi * 37) >> 8, qcol = qi - qrow * WB_WS;
      const int qreg = lds_reg[qi];
      // B operands: scaled q and dO rows of the query token; D = <dO, O>
      bf16x8 Qh[2], Ql[2], Gh[2], Gl[2];
      float Dq = 0.f;
      {
        const long row = qtok >= 0 ? qtok : 0;
        const float* qsrc = qkv + row * C3 + head * WB_HD + lk * 8;
        const float* gsrc = dout + row * C + head * WB_HD + lk * 8;
        const float* osrc = attn_out + row * C + head * WB_HD + lk * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          wbm_row8(qsrc + ks * 16, scale * qm, Qh[ks], Ql[ks]);
          const float4 g0 = *(const float4*)(gsrc + ks * 16), g1 = *(const float4*)(gsrc + ks * 16 + 4);
          const float4 o0 = *(const float4*)(osrc + ks * 16), o1 = *(const float4*)(osrc + ks * 16 + 4);
          const float fg[8] = {g0.x * qm, g0.y * qm, g0.z * qm, g0.w * qm, g1.x * qm, g1.y * qm, g1.z * qm, g1.w * qm};
          wbm_split8(fg, Gh[ks], Gl[ks]);
          Dq += qm * ((g0.x * o0.x + g0.y * o0.y) + (g0.z * o0.z + g0.w * o0.w) +
                      (g1.x * o1.x + g1.y * o1.y) + (g1.z * o1.z + g1.w * o1.w));
        }
        Dq += __shfl_xor(Dq, 32);
      }
      f32x16 st[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = dp[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          st[kt] = occf_mfma_bf16_32x32x16(Kl[kt][ks], Qh[ks], st[kt]);
          st[kt] = occf_mfma_bf16_32x32x16(Kh[kt][ks], Ql[ks], st[kt]);
          st[kt] = occf_mfma_bf16_32x32x16(Kh[kt][ks], Qh[ks], st[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vl[kt][ks], Gh[ks], dp[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vh[kt][ks], Gl[ks], dp[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vh[kt][ks], Gh[ks], dp[kt]);
        }
      }
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          const int krow = (key * 37) >> 8, kcol = key - krow * WB_WS;
          float a = st[kt][r];
          if (key < WB_T) {
            const int bi = (qrow - krow + WB_WS - 1) * (2 * WB_WS - 1) + (qcol - kcol + WB_WS - 1);
            a += lds_bias[qi < WB_T ? bi : 0];
            if (shift > 0 && lds_reg[key] != qreg) a += -100.0f;
          } else {
            a = -INFINITY;
          }
          st[kt][r] = a;
          mx = fmaxf(mx, a);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          st[kt][r] = expf(st[kt][r] - mx);
          sum += st[kt][r];
        }
      sum += __shfl_xor(sum, 32);
      const float inv = qreal ? 1.0f / sum : 0.f;
      if (lk == 0) {
        float* wsp = ws_stats + ((win * heads + head) * 2) * 64;
        wsp[qi] = qreal ? mx + logf(sum) : INFINITY;                  // exp(x - inf) = 0: padded queries drop out
        wsp[64 + qi] = Dq;
      }
      // dS = P (dP - D); copy for the relative-position-bias gradient; dSt as B operand of dQt = K^T . dSt
      bf16x8 sh[4], sl[4];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          float dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = s2 * 8 + e;
            const float ds = st[kt][r] * inv * (dp[kt][r] - Dq);
            dv[e] = ds;
            const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (key < WB_T) lds_m[WBM_MPAD + key * WBM_MSTR + qi] = ds;
          }
          wbm_split8(dv, sh[kt * 2 + s2], sl[kt * 2 + s2]);
        }
      // A operand of dQt = K^T . dSt: lane = channel li, the 8 key tokens of a k-step (loaded only now: the score
      // accumulators are dead, the kernel stays below 256 registers = two waves per SIMD)
      OCCF_SCHED_FENCE();
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int kst = 0; kst < 4; ++kst) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int t = wbm_ktok(kst, e, lk);
          const int tok = lds_tok[t];
          const float x = (tok >= 0 ? qkv + (long)tok * C3 : qkv_bias)[C + head * WB_HD + li];
          v[e] = t < WB_T ? x : 0.f;
        }
        bf16x8 Th, Tl;
        wbm_split8(v, Th, Tl);
        dq = occf_mfma_bf16_32x32x16(Tl, sh[kst], dq);
        dq = occf_mfma_bf16_32x32x16(Th, sl[kst], dq);
        dq = occf_mfma_bf16_32x32x16(Th, sh[kst], dq);
      }
      if (qreal) {
        float* dst = dqkv + (long)qtok * C3 + head * WB_HD + 4 * lk;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(dst + 8 * g) = make_float4(dq[g * 4 + 0] * scale, dq[g * 4 + 1] * scale, dq[g * 4 + 2] * scale,
                                                dq[g * 4 + 3] * scale);
      }
    }
    __syncthreads();                                           // dS of every query is in LDS
    // relative-position-bias gradient: bin (di, dj) collects dS[key + (di, dj)][key] over the keys it is valid for;
    // the key loop is uniform (constant LDS offsets), invalid lanes read a guard / neighbouring float and drop it
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const float* mp = &lds_m[WBM_MPAD + bin_off[u]];
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int key = 0; key < WB_T; ++key) {
        const float x = mp[key * (WBM_MSTR + 1)];
        const float y = (bin_keys[u] >> key) & 1ull ? x : 0.f;
        if (key & 1) a1 += y; else a0 += y;
      }
      dtab[u] += a0 + a1;
    }
  }
  {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int r = (int)threadIdx.x + u * 128;
      if (r < WB_NB) dtable_partial[((long)blockIdx.x * WB_NB + r) * heads + head] = dtab[u];
    }
  }
}

__global__ void __launch_bounds__(64 * WB_HPB) window_attn_bwd_kv_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    const float* __restrict__ dout, float* __restrict__ dqkv, float* __restrict__ dbias_partial,
    const float* __restrict__ ws_stats, int B, int X, int Y, int S, int C, int heads, int shift, float scale, int wpb,
    long n_windows) {
  __shared__ float lds_bias[WB_HPB][WB_NB];
  __shared__ float lds_lse[WB_HPB][64], lds_D[WB_HPB][64];
  __shared__ float lds_bacc[WB_HPB][64];              // d(qkv bias) of this head through padded key positions: k | v
  __shared__ int lds_tok[WB_HPB][64], lds_reg[WB_HPB][64];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int head_raw = blockIdx.y * WB_HPB + wave;
  const bool active = head_raw < heads;
  const int head = active ? head_raw : heads - 1;
  const int C3 = 3 * C;
  for (int t = lane; t < WB_NB; t += 64) lds_bias[wave][t] = bias_table[(long)t * heads + head];
  lds_bacc[wave][lane] = 0.f;

  for (int wi = 0; wi < wpb; ++wi) {
    const long win = (long)blockIdx.x * wpb + wi;
    if (win >= n_windows) break;
    __syncthreads();
    {
      int tok, region;
      wbm_window_tokens(win, lane, B, X, Y, S, shift, tok, region);
      lds_tok[wave][lane] = tok;
      lds_reg[wave][lane] = region;
      const float* wsp = ws_stats + ((win * heads + head) * 2) * 64;
      lds_lse[wave][lane] = wsp[lane];
      lds_D[wave][lane] = wsp[64 + lane];
    }
    __syncthreads();

#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int ki = kt * 32 + li;
      const int ktok = lds_tok[wave][ki];
      const bool kreal = ki < WB_T;
      const int krow = (ki * 37) >> 8, kcol = ki - krow * WB_WS;
      const int kreg = lds_reg[wave][ki];
      // B operands: K, V rows of the key token
      bf16x8 Kh[2], Kl[2], Vh[2], Vl[2];
      {
        const float km = kreal ? 1.f : 0.f;
        const float* src = (ktok >= 0 ? qkv + (long)ktok * C3 : qkv_bias) + head * WB_HD + lk * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          wbm_row8(src + C + ks * 16, km, Kh[ks], Kl[ks]);
          wbm_row8(src + 2 * C + ks * 16, km, Vh[ks], Vl[ks]);
        }
      }
      f32x16 dvt, dkt;
#pragma unroll
      for (int r = 0; r < 16; ++r) dvt[r] = dkt[r] = 0.f;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        // A operands: scaled q and dO rows of query token 32 qt + li
        bf16x8 Qh[2], Ql[2], Gh[2], Gl[2];
        {
          const int t = qt * 32 + li;
          const int tok = lds_tok[wave][t];
          const float qm = (t < WB_T && tok >= 0) ? 1.f : 0.f;
          const long row = tok >= 0 ? tok : 0;
          const float* qsrc = qkv + row * C3 + head * WB_HD + lk * 8;
          const float* gsrc = dout + row * C + head * WB_HD + lk * 8;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            wbm_row8(qsrc + ks * 16, scale * qm, Qh[ks], Ql[ks]);
            wbm_row8(gsrc + ks * 16, qm, Gh[ks], Gl[ks]);
          }
        }
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          st = occf_mfma_bf16_32x32x16(Ql[ks], Kh[ks], st);
          st = occf_mfma_bf16_32x32x16(Qh[ks], Kl[ks], st);
          st = occf_mfma_bf16_32x32x16(Qh[ks], Kh[ks], st);
          dp = occf_mfma_bf16_32x32x16(Gl[ks], Vh[ks], dp);
          dp = occf_mfma_bf16_32x32x16(Gh[ks], Vl[ks], dp);
          dp = occf_mfma_bf16_32x32x16(Gh[ks], Vh[ks], dp);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int kst = qt * 2 + s2;
          // A operands of dVt / dKt: dO^T and (scaled q)^T, lane = channel li, the 8 query tokens of this k-step
          float gt[8], qtv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int t = wbm_ktok(kst, e, lk);
            const int tok = lds_tok[wave][t];
            const bool ok = t < WB_T && tok >= 0;
            const long row = tok >= 0 ? tok : 0;
            const float g = dout[row * C + head * WB_HD + li];
            const float q = qkv[row * C3 + head * WB_HD + li];
            gt[e] = ok ? g : 0.f;
            qtv[e] = ok ? q * scale : 0.f;
          }
          float pv[8], dv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int r = s2 * 8 + e;
            const int qi = qt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int qrow = (qi * 37) >> 8, qcol = qi - qrow * WB_WS;
            float a = st[r];
            float pp = 0.f;
            if (kreal) {
              const int bi = (qrow - krow + WB_WS - 1) * (2 * WB_WS - 1) + (qcol - kcol + WB_WS - 1);
              a += lds_bias[wave][qi < WB_T ? bi : 0];
              if (shift > 0 && lds_reg[wave][qi] != kreg) a += -100.0f;
              pp = expf(a - lds_lse[wave][qi]);
            }
            pv[e] = pp;
            dv[e] = pp * (dp[r] - lds_D[wave][qi]);
          }
          bf16x8 ph, pl, sh, sl, gh, gl, qh, ql;
          wbm_split8(pv, ph, pl);
          wbm_split8(dv, sh, sl);
          wbm_split8(gt, gh, gl);
          wbm_split8(qtv, qh, ql);
          dvt = occf_mfma_bf16_32x32x16(gl, ph, dvt);
          dvt = occf_mfma_bf16_32x32x16(gh, pl, dvt);
          dvt = occf_mfma_bf16_32x32x16(gh, ph, dvt);
          dkt = occf_mfma_bf16_32x32x16(ql, sh, dkt);
          dkt = occf_mfma_bf16_32x32x16(qh, sl, dkt);
          dkt = occf_mfma_bf16_32x32x16(qh, sh, dkt);
        }
      }
      if (active && kreal && ktok >= 0) {
        float* dk = dqkv + (long)ktok * C3 + C + head * WB_HD + 4 * lk;
        float* dvp = dk + C;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          *(float4*)(dk + 8 * g) = make_float4(dkt[g * 4 + 0], dkt[g * 4 + 1], dkt[g * 4 + 2], dkt[g * 4 + 3]);
          *(float4*)(dvp + 8 * g) = make_float4(dvt[g * 4 + 0], dvt[g * 4 + 1], dvt[g * 4 + 2], dvt[g * 4 + 3]);
        }
      }
      // padded window positions (edge windows only) have no token: their k / v are the bias row, so their dK / dV
      // rows belong to d(qkv bias).  Summed over the key lanes here and kept per wave in LDS -- float atomics on the
      // 2 x 32 bias entries of a head serialised ~5 M same-line atomics per call at the L2 (most of the 8.7 ms of
      // the r02c..r02f kernel)
      const bool pad = kreal && ktok < 0;
      if (__ballot(pad) != 0ull) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float a = pad ? dkt[r] : 0.f, b = pad ? dvt[r] : 0.f;
#pragma unroll
          for (int o = 16; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            b += __shfl_xor(b, o);
          }
          if (li == 0) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lk;
            lds_bacc[wave][d] += a;
            lds_bacc[wave][32 + d] += b;
          }
        }
      }
    }
  }
  // per-workgroup partial of d(qkv bias)[C .. 3C): column kv * C + head * 32 + d
  __syncthreads();
  if (active) dbias_partial[(long)blockIdx.x * 2 * C + (lane >> 5) * C + head * WB_HD + (lane & 31)] = lds_bacc[wave][lane];
}

// dtable[r][head] = sum_x partial[x][r][head]: 64 columns x 16 row groups per workgroup, double accumulation
__global__ void __launch_bounds__(1024) window_table_reduce_kernel(const float* __restrict__ partial,
                                                                   float* __restrict__ dtable, long nblk, int C) {
  __shared__ double red[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + tx;
  double s = 0.0;
  if (c < C)
    for (long k = ty; k < nblk; k += 16) s += (double)partial[k * C + c];
  red[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && c < C) {
    double t = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][tx];
    dtable[c] = (float)t;
  }
}

static int wb_windows_per_block(long n_windows) {
  long w = (n_windows + 1023) / 1024;
  return w < 1 ? 1 : (w > 64 ? 64 : (int)w);
}
// floats: the per-workgroup partial bias-table gradients, the per-workgroup partial d(qkv bias) of the k / v thirds,
// then lse / D of every (window, head, query)
extern "C" long occf_window_attn_bwd_workspace(int B, int X, int Y, int S, int heads) {
  const long nwin = (long)B * S * ((X + 6) / 7) * ((Y + 6) / 7);
  const long nblk = occf_cdiv(nwin, wb_windows_per_block(nwin));
  return nblk * heads * WB_NB + nblk * 2 * heads * WB_HD + nwin * heads * 128;
}

extern "C" int occf_window_attn_bwd(const float* qkv, const float* qkv_bias, const float* bias_table,
                                    const float* attn_out, const float* dout, float* dqkv, float* dqkv_bias,
                                    float* dbias_table, float* workspace, int B, int X, int Y, int S, int C,
                                    int heads, int shift, void* stream) {
  if (B <= 0 || X <= 0 || Y <= 0 || S <= 0 || heads <= 0 || C != heads * WB_HD) return OCCF_ESHAPE;
  if (shift < 0 || shift >= WB_WS) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long nwin = (long)B * S * ((X + 6) / 7) * ((Y + 6) / 7);
  const int wpb = wb_windows_per_block(nwin);
  const int nblk = occf_cdiv(nwin, wpb);
  static const bool mfma = [] {
    const char* e = getenv("OCCF_WATTN_BWD_MFMA");
    return e ? atoi(e) != 0 : true;
  }();
  if (mfma) {
    float* bias_part = workspace + (long)nblk * heads * WB_NB;
    float* stats = bias_part + (long)nblk * 2 * C;
    const dim3 grid(nblk, occf_cdiv(heads, WB_HPB)), block(64 * WB_HPB);
    const float scale = 1.0f / sqrtf((float)WB_HD);
    hipLaunchKernelGGL(window_attn_bwd_q_kernel, dim3(nblk, heads), dim3(128), 0, st, qkv, qkv_bias, bias_table, attn_out, dout, dqkv,
                       workspace, stats, B, X, Y, S, C, heads, shift, scale, wpb, nwin);
    hipLaunchKernelGGL(window_attn_bwd_kv_kernel, grid, block, 0, st, qkv, qkv_bias, bias_table, dout, dqkv, bias_part,
                       stats, B, X, Y, S, C, heads, shift, scale, wpb, nwin);
    hipLaunchKernelGGL(window_table_reduce_kernel, dim3(occf_cdiv(2 * C, 64)), dim3(1024), 0, st, bias_part,
                       dqkv_bias + C, (long)nblk, 2 * C);
  } else
    hipLaunchKernelGGL(window_attn_bwd_kernel, dim3(nblk, occf_cdiv(heads, WB_HPB)), dim3(64 * WB_HPB), 0, st, qkv,
                       qkv_bias, bias_table, attn_out, dout, dqkv, dqkv_bias, workspace, B, X, Y, S, C, heads, shift,
                       1.0f / sqrtf((float)WB_HD), wpb, nwin);
  hipLaunchKernelGGL(window_table_reduce_kernel, dim3(occf_cdiv(WB_NB * heads, 64)), dim3(1024), 0, st, workspace,
                     dbias_table, (long)nblk, WB_NB * heads);
  OCCF_LAUNCH_CHECK();
}

// --------------------------------------------------------------------------------------------- masked attention
#define XB_HD 32
#define XB_QMAX 128
#define XB_TILE 32

// lse[b, h, q] and D[b, h, q] = <dO, O>:  partial (m, l) per key chunk, then merged
__global__ void __launch_bounds__(XB_QMAX) xattn_stats_partial_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const uint8_t* __restrict__ blocked,
    const int* __restrict__ row_open, float* __restrict__ part_ml, int B, int Q, int L, int E, int heads, int chunk,
    int n_chunks, float scale) {
  __shared__ __attribute__((aligned(16))) float lds_k[XB_TILE * XB_HD];
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int qi = threadIdx.x;
  const bool valid = qi < Q;
  float qr[XB_HD];
  bool use_mask = false;
  if (valid) {
    const float* qp = q + ((long)b * Q + qi) * E + h * XB_HD;
#pragma unroll
    for (int d = 0; d < XB_HD; ++d) qr[d] = qp[d] * scale;
    use_mask = blocked != nullptr && row_open[b * Q + qi] != 0;
  } else {
#pragma unroll
    for (int d = 0; d < XB_HD; ++d) qr[d] = 0.f;
  }
  const int k0 = ck * chunk;
  const int k1 = (k0 + chunk < L) ? k0 + chunk : L;
  const uint8_t* brow = valid && use_mask ? blocked + ((long)b * Q + qi) * L : nullptr;
  float m = -INFINITY, l = 0.f;
  for (int t0 = k0; t0 < k1; t0 += XB_TILE) {
    const int nt = (k1 - t0 < XB_TILE) ? k1 - t0 : XB_TILE;
    __syncthreads();
    for (int idx = threadIdx.x; idx < XB_TILE * (XB_HD / 4); idx += XB_QMAX) {
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      int key = t0 + r;
      key = key < k1 ? key : k1 - 1;
      *(float4*)(&lds_k[r * XB_HD + c4]) = *(const float4*)(k + ((long)b * L + key) * E + h * XB_HD + c4);
    }
    __syncthreads();
    if (!valid) continue;
    for (int r = 0; r < nt; ++r) {
      if (brow != nullptr && brow[t0 + r]) continue;
      const float* kr = &lds_k[r * XB_HD];
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < XB_HD; ++d) a = fmaf(qr[d], kr[d], a);
      const float mn = fmaxf(m, a);
      l = l * expf(m - mn) + expf(a - mn);
      m = mn;
    }
  }
  if (!valid) return;
  const long slot = (((long)b * heads + h) * Q + qi) * n_chunks + ck;
  part_ml[slot * 2] = m;
  part_ml[slot * 2 + 1] = l;
}

__global__ void __launch_bounds__(256) xattn_stats_merge_kernel(const float* __restrict__ part_ml,
                                                                const float* __restrict__ out,
                                                                const float* __restrict__ dout, float* __restrict__ lse,
                                                                float* __restrict__ Dv, int B, int Q, int E, int heads,
                                                                int n_chunks) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * heads * Q) return;
  const int qi = (int)(gid % Q);
  const int h = (int)((gid / Q) % heads);
  const int b = (int)(gid / ((long)Q * heads));
  float m = -INFINITY;
  for (int c = 0; c < n_chunks; ++c) m = fmaxf(m, part_ml[(gid * n_chunks + c) * 2]);
  float l = 0.f;
  for (int c = 0; c < n_chunks; ++c) {
    const float mc = part_ml[(gid * n_chunks + c) * 2];
    if (mc != -INFINITY) l += part_ml[(gid * n_chunks + c) * 2 + 1] * expf(mc - m);
  }
  lse[gid] = m + logf(l);
  const float* o = out + ((long)b * Q + qi) * E + h * XB_HD;
  const float* g = dout + ((long)b * Q + qi) * E + h * XB_HD;
  float d = 0.f;
#pragma unroll
  for (int e = 0; e < XB_HD; ++e) d = fmaf(o[e], g[e], d);
  Dv[gid] = d;
}

// Main pass: a workgroup owns a contiguous key chunk of one (batch, head).  Threads first act as QUERIES
// (P, dS of the current 32-key tile -> LDS; dQ accumulated in registers over the chunk), then as (key, 8-dim
// slice) pairs for dK / dV.  dQ leaves as one partial per chunk (reduced afterwards in fixed order).
__global__ void __launch_bounds__(XB_QMAX) xattn_bwd_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ blocked, const int* __restrict__ row_open, const float* __restrict__ dout,
    const float* __restrict__ lse, const float* __restrict__ Dv, float* __restrict__ dq_part, float* __restrict__ dk,
    float* __restrict__ dv, int B, int Q, int L, int E, int heads, int chunk, int n_chunks, float scale) {
  __shared__ __attribute__((aligned(16))) float lds_k[XB_TILE * XB_HD];
  __shared__ __attribute__((aligned(16))) float lds_v[XB_TILE * XB_HD];
  __shared__ __attribute__((aligned(16))) float lds_q[XB_QMAX * XB_HD];      // scaled queries
  __shared__ __attribute__((aligned(16))) float lds_g[XB_QMAX * XB_HD];      // dO
  __shared__ float lds_p[XB_QMAX * (XB_TILE + 1)];
  __shared__ float lds_s[XB_QMAX * (XB_TILE + 1)];
  const int ck = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const int qi = threadIdx.x;
  const bool valid = qi < Q;
  float qr[XB_HD], go[XB_HD], dq[XB_HD];
  float my_lse = 0.f, my_D = 0.f;
  bool use_mask = false;
#pragma unroll
  for (int d = 0; d < XB_HD; ++d) qr[d] = go[d] = dq[d] = 0.f;
  if (valid) {
    const float* qp = q + ((long)b * Q + qi) * E + h * XB_HD;
    const float* gp = dout + ((long)b * Q + qi) * E + h * XB_HD;
#pragma unroll
    for (int d = 0; d < XB_HD; ++d) {
      qr[d] = qp[d] * scale;
      go[d] = gp[d];
    }
    const long sidx = ((long)b * heads + h) * Q + qi;
    my_lse = lse[sidx];
    my_D = Dv[sidx];
    use_mask = blocked != nullptr && row_open[b * Q + qi] != 0;
  }
#pragma unroll
  for (int d = 0; d < XB_HD; d += 4) {
    *(float4*)(&lds_q[qi * XB_HD + d]) = make_float4(qr[d], qr[d + 1], qr[d + 2], qr[d + 3]);
    *(float4*)(&lds_g[qi * XB_HD + d]) = make_float4(go[d], go[d + 1], go[d + 2], go[d + 3]);
  }
  const int k0 = ck * chunk;
  const int k1 = (k0 + chunk < L) ? k0 + chunk : L;
  const uint8_t* brow = valid && use_mask ? blocked + ((long)b * Q + qi) * L : nullptr;
  const int kj = threadIdx.x & 31, dpart = (threadIdx.x >> 5) * 8;          // key role: key kj, dims dpart..+7
  for (int t0 = k0; t0 < k1; t0 += XB_TILE) {
    const int nt = (k1 - t0 < XB_TILE) ? k1 - t0 : XB_TILE;
    __syncthreads();
    for (int idx = threadIdx.x; idx < XB_TILE * (XB_HD / 4); idx += XB_QMAX) {
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      int key = t0 + r;
      key = key < k1 ? key : k1 - 1;
      const long src = ((long)b * L + key) * E + h * XB_HD + c4;
      *(float4*)(&lds_k[r * XB_HD + c4]) = *(const float4*)(k + src);
      *(float4*)(&lds_v[r * XB_HD + c4]) = *(const float4*)(v + src);
    }
    __syncthreads();
    for (int r = 0; r < XB_TILE; ++r) {
      float p = 0.f, ds = 0.f;
      if (valid && r < nt && !(brow != nullptr && brow[t0 + r])) {
        const float* kr = &lds_k[r * XB_HD];
        const float* vr = &lds_v[r * XB_HD];
        float a = 0.f, dp = 0.f;
#pragma unroll
        for (int d = 0; d < XB_HD; ++d) {
          a = fmaf(qr[d], kr[d], a);
          dp = fmaf(go[d], vr[d], dp);
        }
        p = expf(a - my_lse);
        ds = p * (dp - my_D);
#pragma unroll
        for (int d = 0; d < XB_HD; ++d) dq[d] = fmaf(ds, kr[d], dq[d]);
      }
      lds_p[qi * (XB_TILE + 1) + r] = p;
      lds_s[qi * (XB_TILE + 1) + r] = ds;
    }
    __syncthreads();
    // key role: dV[kj][dpart..] = sum_q P[q][kj] dO[q][..];  dK[kj][..] = sum_q dS[q][kj] (q * scale)[..]
    if (kj < nt) {
      float av[8], ak[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) av[e] = ak[e] = 0.f;
      for (int qq = 0; qq < Q; ++qq) {
        const float p = lds_p[qq * (XB_TILE + 1) + kj];
        const float ds = lds_s[qq * (XB_TILE + 1) + kj];
        const float* gr = &lds_g[qq * XB_HD + dpart];
        const float* qs = &lds_q[qq * XB_HD + dpart];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          av[e] = fmaf(p, gr[e], av[e]);
          ak[e] = fmaf(ds, qs[e], ak[e]);
        }
      }
      const long dst = ((long)b * L + t0 + kj) * E + h * XB_HD + dpart;
      *(float4*)(dv + dst) = make_float4(av[0], av[1], av[2], av[3]);
      *(float4*)(dv + dst + 4) = make_float4(av[4], av[5], av[6], av[7]);
      *(float4*)(dk + dst) = make_float4(ak[0], ak[1], ak[2], ak[3]);
      *(float4*)(dk + dst + 4) = make_float4(ak[4], ak[5], ak[6], ak[7]);
    }
  }
  if (!valid) return;
  float* po = dq_part + ((((long)b * heads + h) * Q + qi) * n_chunks + ck) * XB_HD;
#pragma unroll
  for (int d = 0; d < XB_HD; d += 4)
    *(float4*)(po + d) = make_float4(dq[d] * scale, dq[d + 1] * scale, dq[d + 2] * scale, dq[d + 3] * scale);
}

__global__ void __launch_bounds__(256) xattn_dq_reduce_kernel(const float* __restrict__ dq_part, float* __restrict__ dq,
                                                              int B, int Q, int E, int heads, int n_chunks) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * heads * Q * XB_HD) return;
  const int d = (int)(gid % XB_HD);
  const long row = gid / XB_HD;                       // (b, h, q)
  const int qi = (int)(row % Q);
  const int h = (int)((row / Q) % heads);
  const int b = (int)(row / ((long)Q * heads));
  float s = 0.f;
  for (int c = 0; c < n_chunks; ++c) s += dq_part[(row * n_chunks + c) * XB_HD + d];
  dq[((long)b * Q + qi) * E + h * XB_HD + d] = s;
}

// matrix-core kernels (xattn_mfma.hip: forward partials = the softmax statistics; xattn_bwd_mfma.hip: dQ / dK / dV)
void occf_xattn_mfma_launch(const float* q, const float* k, const float* v, const uint8_t* blocked, const int* row_open,
                            float* part_o, float* part_ml, int B, int Q, int L, int E, int heads, int chunk, int n_chunks,
                            float scale, hipStream_t st);
void occf_xattn_bwd_mfma_launch(const float* q, const float* k, const float* v, const uint8_t* blocked,
                                const int* row_open, const float* dout, const float* lse, const float* Dv,
                                float* dq_part, float* dk, float* dv, int B, int Q, int L, int E, int heads, int chunk,
                                int n_chunks, float scale, hipStream_t st);

static void xb_chunks(int B, int L, int heads, int& chunk, int& n_chunks) {
  // enough workgroups to fill the chip, at least 4 tiles per chunk
  long want = 1024 / ((long)B * heads > 0 ? (long)B * heads : 1);
  if (want < 1) want = 1;
  long c = (L + want - 1) / want;
  c = (c + XB_TILE - 1) / XB_TILE * XB_TILE;
  if (c < 4 * XB_TILE) c = 4 * XB_TILE;
  chunk = (int)c;
  n_chunks = (L + chunk - 1) / chunk;
}

extern "C" long occf_masked_xattn_bwd_workspace(int B, int Q, int L, int heads) {
  int chunk, nc;
  xb_chunks(B, L, heads, chunk, nc);
  const long rows = (long)B * heads * Q;
  return rows * nc * 2 + rows * 2 + rows * nc * XB_HD;
}

extern "C" int occf_masked_xattn_bwd(const float* q, const float* k, const float* v, const uint8_t* blocked,
                                     const int32_t* row_open, const float* out, const float* dout, float* dq, float* dk,
                                     float* dv, float* workspace, int B, int Q, int L, int E, int heads, void* stream) {
  if (B <= 0 || Q <= 0 || Q > XB_QMAX || L <= 0 || E != heads * XB_HD) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  int chunk, nc;
  xb_chunks(B, L, heads, chunk, nc);
  const long rows = (long)B * heads * Q;
  float* part_ml = workspace;
  float* lse = part_ml + rows * nc * 2;
  float* Dv = lse + rows;
  float* dq_part = Dv + rows;
  const float scale = 1.0f / sqrtf((float)XB_HD);
  static const int mfma_env = [] {
    const char* e = getenv("OCCF_XATTN_BWD_MFMA");          // diagnostics: 0 = the scalar kernels of round 2
    return e ? atoi(e) : 1;
  }();
  if (mfma_env) {
    // statistics from the forward's own matrix-core partial kernel ((max, sum) per key chunk; its weighted value sums
    // land in the dQ partial buffer, which is rewritten below)
    occf_xattn_mfma_launch(q, k, v, blocked, (const int*)row_open, dq_part, part_ml, B, Q, L, E, heads, chunk, nc, scale,
                           st);
    hipLaunchKernelGGL(xattn_stats_merge_kernel, dim3(occf_cdiv(rows, 256)), dim3(256), 0, st, part_ml, out, dout, lse,
                       Dv, B, Q, E, heads, nc);
    occf_xattn_bwd_mfma_launch(q, k, v, blocked, (const int*)row_open, dout, lse, Dv, dq_part, dk, dv, B, Q, L, E, heads,
                               chunk, nc, scale, st);
  } else {
    hipLaunchKernelGGL(xattn_stats_partial_kernel, dim3(nc, heads, B), dim3(XB_QMAX), 0, st, q, k, blocked,
                       (const int*)row_open, part_ml, B, Q, L, E, heads, chunk, nc, scale);
    hipLaunchKernelGGL(xattn_stats_merge_kernel, dim3(occf_cdiv(rows, 256)), dim3(256), 0, st, part_ml, out, dout, lse,
                       Dv, B, Q, E, heads, nc);
    hipLaunchKernelGGL(xattn_bwd_kernel, dim3(nc, heads, B), dim3(XB_QMAX), 0, st, q, k, v, blocked,
                       (const int*)row_open, dout, lse, Dv, dq_part, dk, dv, B, Q, L, E, heads, chunk, nc, scale);
  }
  hipLaunchKernelGGL(xattn_dq_reduce_kernel, dim3(occf_cdiv(rows * XB_HD, 256)), dim3(256), 0, st, dq_part, dq, B, Q,
                     E, heads, nc);
  OCCF_LAUNCH_CHECK();
}
