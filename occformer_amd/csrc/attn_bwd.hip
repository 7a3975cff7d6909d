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
// v_mfma_f32_32x32x16_bf16 (3-term bf16 split, fp32 accumulate) instead of ~7 800 VALU FMAs per lane (the VALU kernel
// ran at 6 % of the vector rate: 12.8 ms per stage-0 call, 50 ms per training step).  With D = A . B returning
// lane -> column, registers -> rows (r&3) + 8 (r>>2) + 4 (lane>>5), and an operand lane supplying 8 consecutive k:
//   orientation 1 (lane = query):  St  = K . Q^T,  dPt = V . dO^T   (A = key rows, B = query rows; k = d)
//        -> P, dS per query column in registers (the softmax statistics of a query sit in ONE lane pair)
//        dQt[d][query] = K^T[d][key] . dSt[key][query]               (A = K^T image in LDS, B = dSt from registers)
//   orientation 2 (lane = key):    S'  = Q . K^T,  dP' = dO . V^T   (the same row operands with the roles swapped)
//        -> P', dS' per key column, using the lse / D of orientation 1 (LDS)
//        dVt[d][key] = dO^T[d][query] . P'[query][key],  dKt[d][key] = Qs^T[d][query] . dS'[query][key]
// Row operands (K, V, scaled Q, dO: 8 consecutive d of one token) come straight from global memory; only the three
// transposed images (K^T, dO^T, Qs^T: [4 k-steps][32 d][2 slots][8 tokens], hi / lo) live in LDS: 24 KB per wave.
typedef uint32_t wbm_u2 __attribute__((ext_vector_type(2)));
#define WBM_IMG 4096            // bytes of one transposed image half (hi or lo)

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
// token t, head channel d -> byte offset of its bf16 in a transposed image (token order inside a 16-token k-step:
// t & 15 = 8 (e >> 2) + 4 slot + (e & 3), the order in which a C-layout register file hands its rows on as B operand)
__device__ __forceinline__ int wbm_toff(int t, int d) {
  const int kk = t & 15;
  const int e = ((kk >> 3) << 2) | (kk & 3), slot = (kk >> 2) & 1;
  return (t >> 4) * 1024 + d * 32 + slot * 16 + e * 2;
}

__global__ void __launch_bounds__(64 * WB_HPB) window_attn_bwd_mfma_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_bias, const float* __restrict__ bias_table,
    const float* __restrict__ attn_out, const float* __restrict__ dout, float* __restrict__ dqkv,
    float* __restrict__ dqkv_bias, float* __restrict__ dtable_partial, int B, int X, int Y, int S, int C, int heads,
    int shift, float scale, int wpb, long n_windows) {
  __shared__ __attribute__((aligned(16))) unsigned char img[WB_HPB][6 * WBM_IMG];   // K^T h|l, dO^T h|l, Qs^T h|l
  __shared__ float lds_bias[WB_HPB][WB_NB];
  __shared__ float lds_m[WB_HPB][WB_T * 64];          // dS[key][query] of the current window (bias-table gradient)
  __shared__ float lds_lse[WB_HPB][64], lds_D[WB_HPB][64];
  __shared__ int lds_tok[WB_HPB][64], lds_reg[WB_HPB][64];

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int nwx = (X + WB_WS - 1) / WB_WS, nwy = (Y + WB_WS - 1) / WB_WS;
  const int Xp = nwx * WB_WS, Yp = nwy * WB_WS;
  const int head_raw = blockIdx.y * WB_HPB + wave;
  const bool active = head_raw < heads;
  const int head = active ? head_raw : heads - 1;
  const int C3 = 3 * C;
  unsigned char* base = img[wave];
  for (int t = lane; t < WB_NB; t += 64) lds_bias[wave][t] = bias_table[(long)t * heads + head];
  float dtab[3] = {0.f, 0.f, 0.f};                    // bins lane, lane + 64, lane + 128 (LDS float atomics are slow)

  for (int wi = 0; wi < wpb; ++wi) {
    const long win = (long)blockIdx.x * wpb + wi;
    if (win >= n_windows) break;
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
    __syncthreads();                                           // previous window's LDS contents are consumed
    {
      int tok = -1, region = 0;
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
      lds_tok[wave][lane] = tok;
      lds_reg[wave][lane] = region;
    }
    __syncthreads();

    // ---- row operands of this lane's two tokens (tile tt: token 32 tt + li), 8 consecutive d at lk*8 + 16 ks.
    // rows 49..63 are zero; padded window positions (t < 49, no token) take the bias row for k / v and have no
    // query / dO (the forward crops them)
    bf16x8 Kh[2][2], Kl[2][2], Vh[2][2], Vl[2][2], Qh[2][2], Ql[2][2], Gh[2][2], Gl[2][2];
    int tokt[2];
    float Dpart[2] = {0.f, 0.f};
    // all 40 row loads of the window are issued before the first conversion (one memory round trip instead of one
    // per operand group: at one wave per SIMD nothing else hides the latency)
    float4 ld[2][2][10];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = tt * 32 + li;
      const int tok = lds_tok[wave][t];
      tokt[tt] = tok;
      const float* src = (tok >= 0 ? qkv + (long)tok * C3 : qkv_bias) + head * WB_HD + lk * 8;
      const float* gsrc = dout + (long)(tok >= 0 ? tok : 0) * C + head * WB_HD + lk * 8;
      const float* osrc = attn_out + (long)(tok >= 0 ? tok : 0) * C + head * WB_HD + lk * 8;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        ld[tt][ks][0] = *(const float4*)(src + ks * 16);
        ld[tt][ks][1] = *(const float4*)(src + ks * 16 + 4);
        ld[tt][ks][2] = *(const float4*)(src + C + ks * 16);
        ld[tt][ks][3] = *(const float4*)(src + C + ks * 16 + 4);
        ld[tt][ks][4] = *(const float4*)(src + 2 * C + ks * 16);
        ld[tt][ks][5] = *(const float4*)(src + 2 * C + ks * 16 + 4);
        ld[tt][ks][6] = *(const float4*)(gsrc + ks * 16);
        ld[tt][ks][7] = *(const float4*)(gsrc + ks * 16 + 4);
        ld[tt][ks][8] = *(const float4*)(osrc + ks * 16);
        ld[tt][ks][9] = *(const float4*)(osrc + ks * 16 + 4);
      }
    }
    OCCF_SCHED_FENCE();
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int t = tt * 32 + li;
      const int tok = tokt[tt];
      const bool real = t < WB_T;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float4 q0 = ld[tt][ks][0], q1 = ld[tt][ks][1], k0 = ld[tt][ks][2], k1 = ld[tt][ks][3];
        const float4 v0 = ld[tt][ks][4], v1 = ld[tt][ks][5], g0 = ld[tt][ks][6], g1 = ld[tt][ks][7];
        const float4 o0 = ld[tt][ks][8], o1 = ld[tt][ks][9];
        const float km = real ? 1.f : 0.f, qm = (real && tok >= 0) ? 1.f : 0.f;
        const float fk[8] = {k0.x * km, k0.y * km, k0.z * km, k0.w * km, k1.x * km, k1.y * km, k1.z * km, k1.w * km};
        const float fv[8] = {v0.x * km, v0.y * km, v0.z * km, v0.w * km, v1.x * km, v1.y * km, v1.z * km, v1.w * km};
        const float fq[8] = {q0.x * scale * qm, q0.y * scale * qm, q0.z * scale * qm, q0.w * scale * qm,
                             q1.x * scale * qm, q1.y * scale * qm, q1.z * scale * qm, q1.w * scale * qm};
        const float fg[8] = {g0.x * qm, g0.y * qm, g0.z * qm, g0.w * qm, g1.x * qm, g1.y * qm, g1.z * qm, g1.w * qm};
        Dpart[tt] += qm * ((g0.x * o0.x + g0.y * o0.y) + (g0.z * o0.z + g0.w * o0.w) +
                           (g1.x * o1.x + g1.y * o1.y) + (g1.z * o1.z + g1.w * o1.w));
        wbm_split8(fk, Kh[tt][ks], Kl[tt][ks]);
        wbm_split8(fv, Vh[tt][ks], Vl[tt][ks]);
        wbm_split8(fq, Qh[tt][ks], Ql[tt][ks]);
        wbm_split8(fg, Gh[tt][ks], Gl[tt][ks]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int d = ks * 16 + lk * 8 + e;
          const int off = wbm_toff(t, d);
          *(uint16_t*)(base + 0 * WBM_IMG + off) = (uint16_t)Kh[tt][ks][e];
          *(uint16_t*)(base + 1 * WBM_IMG + off) = (uint16_t)Kl[tt][ks][e];
          *(uint16_t*)(base + 2 * WBM_IMG + off) = (uint16_t)Gh[tt][ks][e];
          *(uint16_t*)(base + 3 * WBM_IMG + off) = (uint16_t)Gl[tt][ks][e];
          *(uint16_t*)(base + 4 * WBM_IMG + off) = (uint16_t)Qh[tt][ks][e];
          *(uint16_t*)(base + 5 * WBM_IMG + off) = (uint16_t)Ql[tt][ks][e];
        }
      }
      Dpart[tt] += __shfl_xor(Dpart[tt], 32);
    }
    __syncthreads();                                           // images complete

    // ================= orientation 1: lane = query column qi
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
      const int qi = qt * 32 + li;
      const int qrow = (qi * 37) >> 8, qcol = qi - qrow * WB_WS;
      const bool qreal = qi < WB_T && tokt[qt] >= 0;
      const int qreg = lds_reg[wave][qi];
      f32x16 st[2], dp[2];
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kt][r] = dp[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          st[kt] = occf_mfma_bf16_32x32x16(Kl[kt][ks], Qh[qt][ks], st[kt]);
          st[kt] = occf_mfma_bf16_32x32x16(Kh[kt][ks], Ql[qt][ks], st[kt]);
          st[kt] = occf_mfma_bf16_32x32x16(Kh[kt][ks], Qh[qt][ks], st[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vl[kt][ks], Gh[qt][ks], dp[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vh[kt][ks], Gl[qt][ks], dp[kt]);
          dp[kt] = occf_mfma_bf16_32x32x16(Vh[kt][ks], Gh[qt][ks], dp[kt]);
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
            a += lds_bias[wave][qi < WB_T ? bi : 0];
            if (shift > 0 && lds_reg[wave][key] != qreg) a += -100.0f;
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
      const float Dq = Dpart[qt];
      if (lk == 0) {
        lds_lse[wave][qi] = qreal ? mx + logf(sum) : INFINITY;          // exp(x - inf) = 0: padded queries drop out
        lds_D[wave][qi] = Dq;
      }
      // dS = P (dP - D); relative-position-bias gradient; dSt as B operand of dQt = K^T . dSt
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
            if (key < WB_T) lds_m[wave][key * 64 + qi] = qreal ? ds : 0.f;
          }
          wbm_split8(dv, sh[kt * 2 + s2], sl[kt * 2 + s2]);
        }
      f32x16 dq;
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
      for (int kst = 0; kst < 4; ++kst) {
        const int off = kst * 1024 + li * 32 + lk * 16;
        const bf16x8 ah = *(const bf16x8*)(base + 0 * WBM_IMG + off), al = *(const bf16x8*)(base + 1 * WBM_IMG + off);
        dq = occf_mfma_bf16_32x32x16(al, sh[kst], dq);
        dq = occf_mfma_bf16_32x32x16(ah, sl[kst], dq);
        dq = occf_mfma_bf16_32x32x16(ah, sh[kst], dq);
      }
      if (active && qreal) {
        float* dst = dqkv + (long)tokt[qt] * C3 + head * WB_HD + 4 * lk;
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *(float4*)(dst + 8 * g) = make_float4(dq[g * 4 + 0] * scale, dq[g * 4 + 1] * scale, dq[g * 4 + 2] * scale,
                                                dq[g * 4 + 3] * scale);
      }
    }
    __syncthreads();                                           // lse / D / dS of every query are in LDS
    // relative-position-bias gradient: bin r = (di + 6) * 13 + (dj + 6) collects dS[(ki+di, kj+dj)][(ki, kj)]
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
              a += lds_m[wave][(ki * WB_WS + kj) * 64 + qi2 * WB_WS + qj2];
            }
          }
          dtab[u] += a;
        }
      }
    }

    // ================= orientation 2: lane = key column ki
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      const int ki = kt * 32 + li;
      const int krow = (ki * 37) >> 8, kcol = ki - krow * WB_WS;
      const bool kreal = ki < WB_T;
      const int kreg = lds_reg[wave][ki];
      bf16x8 ph[4], pl[4], sh[4], sl[4];
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        f32x16 st, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          st = occf_mfma_bf16_32x32x16(Ql[qt][ks], Kh[kt][ks], st);
          st = occf_mfma_bf16_32x32x16(Qh[qt][ks], Kl[kt][ks], st);
          st = occf_mfma_bf16_32x32x16(Qh[qt][ks], Kh[kt][ks], st);
          dp = occf_mfma_bf16_32x32x16(Gl[qt][ks], Vh[kt][ks], dp);
          dp = occf_mfma_bf16_32x32x16(Gh[qt][ks], Vl[kt][ks], dp);
          dp = occf_mfma_bf16_32x32x16(Gh[qt][ks], Vh[kt][ks], dp);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
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
          wbm_split8(pv, ph[qt * 2 + s2], pl[qt * 2 + s2]);
          wbm_split8(dv, sh[qt * 2 + s2], sl[qt * 2 + s2]);
        }
      }
      f32x16 dvt, dkt;
#pragma unroll
      for (int r = 0; r < 16; ++r) dvt[r] = dkt[r] = 0.f;
#pragma unroll
      for (int kst = 0; kst < 4; ++kst) {
        const int off = kst * 1024 + li * 32 + lk * 16;
        const bf16x8 gh = *(const bf16x8*)(base + 2 * WBM_IMG + off), gl = *(const bf16x8*)(base + 3 * WBM_IMG + off);
        const bf16x8 qh = *(const bf16x8*)(base + 4 * WBM_IMG + off), ql = *(const bf16x8*)(base + 5 * WBM_IMG + off);
        dvt = occf_mfma_bf16_32x32x16(gl, ph[kst], dvt);
        dvt = occf_mfma_bf16_32x32x16(gh, pl[kst], dvt);
        dvt = occf_mfma_bf16_32x32x16(gh, ph[kst], dvt);
        dkt = occf_mfma_bf16_32x32x16(ql, sh[kst], dkt);
        dkt = occf_mfma_bf16_32x32x16(qh, sl[kst], dkt);
        dkt = occf_mfma_bf16_32x32x16(qh, sh[kst], dkt);
      }
      if (active && kreal) {
        if (tokt[kt] >= 0) {
          float* dk = dqkv + (long)tokt[kt] * C3 + C + head * WB_HD + 4 * lk;
          float* dvp = dk + C;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            *(float4*)(dk + 8 * g) = make_float4(dkt[g * 4 + 0], dkt[g * 4 + 1], dkt[g * 4 + 2], dkt[g * 4 + 3]);
            *(float4*)(dvp + 8 * g) = make_float4(dvt[g * 4 + 0], dvt[g * 4 + 1], dvt[g * 4 + 2], dvt[g * 4 + 3]);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = (r & 3) + 8 * (r >> 2) + 4 * lk;
            atomicAdd(dqkv_bias + C + head * WB_HD + d, dkt[r]);
            atomicAdd(dqkv_bias + 2 * C + head * WB_HD + d, dvt[r]);
          }
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
extern "C" long occf_window_attn_bwd_workspace(int B, int X, int Y, int S, int heads) {
  const long nwin = (long)B * S * ((X + 6) / 7) * ((Y + 6) / 7);
  return (long)occf_cdiv(nwin, wb_windows_per_block(nwin)) * heads * WB_NB;
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
  if (mfma)
    hipLaunchKernelGGL(window_attn_bwd_mfma_kernel, dim3(nblk, occf_cdiv(heads, WB_HPB)), dim3(64 * WB_HPB), 0, st, qkv,
                       qkv_bias, bias_table, attn_out, dout, dqkv, dqkv_bias, workspace, B, X, Y, S, C, heads, shift,
                       1.0f / sqrtf((float)WB_HD), wpb, nwin);
  else
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
  hipLaunchKernelGGL(xattn_stats_partial_kernel, dim3(nc, heads, B), dim3(XB_QMAX), 0, st, q, k, blocked,
                     (const int*)row_open, part_ml, B, Q, L, E, heads, chunk, nc, scale);
  hipLaunchKernelGGL(xattn_stats_merge_kernel, dim3(occf_cdiv(rows, 256)), dim3(256), 0, st, part_ml, out, dout, lse,
                     Dv, B, Q, E, heads, nc);
  hipLaunchKernelGGL(xattn_bwd_kernel, dim3(nc, heads, B), dim3(XB_QMAX), 0, st, q, k, v, blocked,
                     (const int*)row_open, dout, lse, Dv, dq_part, dk, dv, B, Q, L, E, heads, chunk, nc, scale);
  hipLaunchKernelGGL(xattn_dq_reduce_kernel, dim3(occf_cdiv(rows * XB_HD, 256)), dim3(256), 0, st, dq_part, dq, B, Q,
                     E, heads, nc);
  OCCF_LAUNCH_CHECK();
}
