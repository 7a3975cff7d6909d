// Attention half of the shared SwinBlock, fused for embed_dims = 128 (4 heads, the 200-grid stage that dominates):
//     out = x + proj( WindowMSA( LayerNorm(x) ) )                     window_attention.py:346-372 (first residual),
//           WindowMSA.forward :69-107, ShiftWindowMSA.forward :168-242
// One workgroup per 7x7 window of one slice (4 waves = 4 heads).  Unfused, this is LayerNorm, the qkv GEMM, the window
// attention kernel and the proj GEMM: 13 passes over [tokens, 128..384] tensors (4.5 GB at the 200-grid); here the
// window's 49 token rows are read once and written once and everything in between stays on chip.
//
// Register chaining (see mlp_chain.hip / xattn_mfma.hip): with D = A . B on v_mfma_f32_32x32x16_bf16 returning
// lane -> column, registers -> rows (r&3) + 8(r>>2) + 4(lane>>5):
//   Qt = Wq . Xn^T,  Kt = Wk . Xn^T   (lane = token, registers = head channel d)
//   V  = Xn . Wv^T                    (lane = head channel d, registers = token)
//   St = Kt^T(as A: row = key = lane) . Qt(as B: column = query = lane)       k = d, register order on both sides
//   Ot = V(as A: row = d = lane) . Pt(as B: column = query = lane)             k = key, register order on both sides
// so q, k, v, the scores and the probabilities never leave registers.  LDS holds the layer-normed window (the B / A
// operand of the projections) and, after the attention, the four heads' outputs (the B operand of proj, each wave
// then producing 32 output channels for all tokens -- no cross-wave reduction).  3-term bf16 split products.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define SF_WS 7
#define SF_T 49
#define SF_C 128
#define SF_HD 32

struct SwinAttnArgs {
  const float* x;
  const float* gamma;
  const float* beta;
  const uint16_t* Wqkv_h;     // [3C, C] bf16 hi / lo
  const uint16_t* Wqkv_l;
  const float* bqkv;          // [3C]
  const float* bias_table;    // [(2*7-1)^2, heads]
  const uint16_t* Wp_h;       // [C, C]
  const uint16_t* Wp_l;
  const float* bp;            // [C]
  float* out;
  int B, X, Y, S, shift;
  float eps, scale;
  int packed;                 // weights in fragment order (occf_swin_attn_pack): [row / 32][k-step][lane][8]
};

typedef uint32_t sf_u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t sf_bits(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
#else
  return __float_as_uint(x);
#endif
}
__device__ __forceinline__ float sf_from_bits(uint32_t u) {
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}
__device__ __forceinline__ uint32_t sf_bf16(float x) {
  const uint32_t u = sf_bits(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void sf_split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
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
// 4 consecutive channels (c .. c+3) of token row t -> operand image [ks = c>>4][row t][slot (c>>3)&1][e = c&7]
__device__ __forceinline__ void sf_put4(unsigned char* hi, unsigned char* lo, int t, int c, float a, float b, float cc,
                                        float d) {
  uint32_t h0, l0, h1, l1;
  occf_bf16_split2(a, b, h0, l0);
  occf_bf16_split2(cc, d, h1, l1);
  const int off = (c >> 4) * 2048 + t * 32 + ((c >> 3) & 1) * 16 + (c & 7) * 2;
  const sf_u2 ph = {h0, h1};
  const sf_u2 pl = {l0, l1};
  *(sf_u2*)(hi + off) = ph;
  *(sf_u2*)(lo + off) = pl;
}

#ifdef OCCF_EMU
#define SF_WAVES2
#else
#define SF_WAVES2 __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#define SF_IMG 16384     // bytes of one [8 ks][64 rows][2 x 16 B] bf16 image (hi or lo)

__global__ void __launch_bounds__(256) SF_WAVES2 swin_attn_fused_kernel(SwinAttnArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char img_h[SF_IMG], img_l[SF_IMG];   // Xn, later the heads' outputs
  __shared__ float lds_bias[4][256];                 // 169 entries used (index 255 = padding key, value unused)
  __shared__ int lds_tok[64];
  __shared__ int lds_reg[64];
  // [key][query] -> relative-position-bias index (bits 0..7), shift-mask flag (bit 8), 0xFFFF = padding key;
  // built once per window by all threads (the index arithmetic used to run per score element in every head)
  __shared__ uint16_t lds_rel[64 * 64];

  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int li = lane & 31, lk = lane >> 5;
  const int X = p.X, Y = p.Y, S = p.S, shift = p.shift;
  const int nwx = (X + SF_WS - 1) / SF_WS, nwy = (Y + SF_WS - 1) / SF_WS;
  const int Xp = nwx * SF_WS, Yp = nwy * SF_WS;
  // window y fastest, then x, then slice.  (Slice-fastest -- neighbouring workgroups on the S adjacent token rows of
  // one window position -- measured the same: 2.09 vs 2.08 ms per forward, probes 25 / 31.)
  long bid = blockIdx.x;
  const int wy = (int)(bid % nwy);
  bid /= nwy;
  const int wx = (int)(bid % nwx);
  bid /= nwx;
  const int s = (int)(bid % S);
  const int b = (int)(bid / S);
  const int head = wave;
  constexpr int C = SF_C;

  // ---- rolled-frame position -> source token (-1 = padding) and shift-mask region
  if (tid < 64) {
    int tok = -1, region = 0;
    if (tid < SF_T) {
      const int i = tid / SF_WS, j = tid % SF_WS;
      const int px = wx * SF_WS + i, py = wy * SF_WS + j;
      int sx = px + shift, sy = py + shift;                         // torch.roll(-shift)
      if (sx >= Xp) sx -= Xp;
      if (sy >= Yp) sy -= Yp;
      if (sx < X && sy < Y) tok = (int)((((long)b * X + sx) * Y + sy) * S + s);
      if (shift > 0) {
        const int rx = px < Xp - SF_WS ? 0 : (px < Xp - shift ? 1 : 2);
        const int ry = py < Yp - SF_WS ? 0 : (py < Yp - shift ? 1 : 2);
        region = rx * 3 + ry;
      }
    }
    lds_tok[tid] = tok;
    lds_reg[tid] = region;
  }
  for (int t = lane; t < (2 * SF_WS - 1) * (2 * SF_WS - 1); t += 64)
    lds_bias[wave][t] = p.bias_table[(long)t * 4 + head];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int idx = tid + i * 256;
    const int key = idx >> 6, qi = idx & 63;
    const int krow = (key * 37) >> 8, kcol = key - krow * SF_WS;      // key / 7, key % 7 for key < 64
    const int qrow = (qi * 37) >> 8, qcol = qi - qrow * SF_WS;
    int v = 0xFFFF;
    if (key < SF_T) {
      v = qi < SF_T ? (qrow - krow + SF_WS - 1) * (2 * SF_WS - 1) + (qcol - kcol + SF_WS - 1) : 0;
      if (shift > 0 && lds_reg[key] != lds_reg[qi]) v |= 0x100;
    }
    lds_rel[idx] = (uint16_t)v;
  }

  // ---- LayerNorm of the window's rows -> operand image (16 lanes per row; rows without a token are zero:
  // the reference pads AFTER the norm, so their q / k / v are the projection biases)
  {
    const int sub = tid & 15, rloc = tid >> 4;
    float4 v[4][2];
    int toks[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int t = pass * 16 + rloc;
      toks[pass] = lds_tok[t];
      const long row = toks[pass] >= 0 ? toks[pass] : 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) v[pass][j] = *(const float4*)(p.x + row * C + (sub + 16 * j) * 4);
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int t = pass * 16 + rloc;
      float sm = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) sm += (v[pass][j].x + v[pass][j].y) + (v[pass][j].z + v[pass][j].w);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
      const float mean = sm / (float)C;
      float q2 = 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a = v[pass][j].x - mean, bb = v[pass][j].y - mean, c = v[pass][j].z - mean, d = v[pass][j].w - mean;
        q2 += (a * a + bb * bb) + (c * c + d * d);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) q2 += __shfl_xor(q2, o);
      const float rstd = 1.0f / sqrtf(q2 / (float)C + p.eps);
      const bool live = toks[pass] >= 0;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c0 = (sub + 16 * j) * 4;
        const float4 g = *(const float4*)(p.gamma + c0), be = *(const float4*)(p.beta + c0);
        const float y0 = live ? (v[pass][j].x - mean) * rstd * g.x + be.x : 0.f;
        const float y1 = live ? (v[pass][j].y - mean) * rstd * g.y + be.y : 0.f;
        const float y2 = live ? (v[pass][j].z - mean) * rstd * g.z + be.z : 0.f;
        const float y3 = live ? (v[pass][j].w - mean) * rstd * g.w + be.w : 0.f;
        sf_put4(img_h, img_l, t, c0, y0, y1, y2, y3);
      }
    }
  }
  __syncthreads();

  // ---- projections of this head, one matrix at a time (keeps the accumulator footprint at two tiles):
  // Qt / Kt (lane = token, registers = channel d) and V (lane = channel d, registers = token), two 32-token tiles
  // The weight fragments come straight from global memory (L2) in operand layout; a matrix' 16 fragments are
  // fetched as ONE batch, one matrix ahead of the MFMAs that use them (one wave per SIMD: nothing else would
  // hide 24 dependent L2 round trips).  Matrix 3 = proj, consumed after the attention.
  bf16x8 qfh[2][2], qfl[2][2], kfh[2][2], kfl[2][2], vfh[2][2], vfl[2][2];
  bf16x8 wfh[2][8], wfl[2][8];
  auto fetch_w = [&](int mat, int buf) __attribute__((always_inline)) {
    if (p.packed) {
      // fragment order: the 64 lanes of a (32-row group, k-step) read 1 KB contiguous -- 8 cache lines per load
      // instruction instead of the 32 a row-major weight costs (each lane's 16 bytes in a different row)
      const int grp = mat < 3 ? mat * 4 + head : wave;
      const uint16_t* bh = (mat < 3 ? p.Wqkv_h : p.Wp_h) + (long)grp * 8 * 512 + lane * 8;
      const uint16_t* bl = (mat < 3 ? p.Wqkv_l : p.Wp_l) + (long)grp * 8 * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        wfh[buf][ks] = *(const bf16x8*)(bh + ks * 512);
        wfl[buf][ks] = *(const bf16x8*)(bl + ks * 512);
      }
      return;
    }
    const uint16_t* bh = mat < 3 ? p.Wqkv_h + ((long)(mat * C + head * SF_HD + li)) * C : p.Wp_h + ((long)(wave * 32 + li)) * C;
    const uint16_t* bl = mat < 3 ? p.Wqkv_l + ((long)(mat * C + head * SF_HD + li)) * C : p.Wp_l + ((long)(wave * 32 + li)) * C;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      wfh[buf][ks] = *(const bf16x8*)(bh + lk * 8 + ks * 16);
      wfl[buf][ks] = *(const bf16x8*)(bl + lk * 8 + ks * 16);
    }
  };
  // (the fences pin the fetch order: left alone, the compiler hoists all three matrices' loads to the top -- 192
  // registers of fragments in flight, 65 spilled: the 0.9 GB of scratch writes per launch PMC showed as "output")
  fetch_w(0, 0);
#pragma unroll
  for (int mat = 0; mat < 3; ++mat) {
    OCCF_SCHED_FENCE();
    if (mat < 2) fetch_w(mat + 1, (mat + 1) & 1);      // next matrix' fragments travel under this matrix' MFMAs
    OCCF_SCHED_FENCE();
    f32x16 acc[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const bf16x8 wh = wfh[mat & 1][ks], wl = wfl[mat & 1][ks];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int off = ks * 2048 + (tt * 32 + li) * 32 + lk * 16;
        const bf16x8 xh = *(const bf16x8*)(img_h + off), xl = *(const bf16x8*)(img_l + off);
        if (mat < 2) {        // W . Xn^T
          acc[tt] = occf_mfma_bf16_32x32x16(wl, xh, acc[tt]);
          acc[tt] = occf_mfma_bf16_32x32x16(wh, xl, acc[tt]);
          acc[tt] = occf_mfma_bf16_32x32x16(wh, xh, acc[tt]);
        } else {              // Xn . W^T
          acc[tt] = occf_mfma_bf16_32x32x16(xl, wh, acc[tt]);
          acc[tt] = occf_mfma_bf16_32x32x16(xh, wl, acc[tt]);
          acc[tt] = occf_mfma_bf16_32x32x16(xh, wh, acc[tt]);
        }
      }
    }
    // bias (q, k: per register row d = (r&3) + 8(r>>2) + 4 lk; v: per lane column d = li); q scaled after its bias
    float br[16];
    if (mat < 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 a4 = *(const float4*)(p.bqkv + mat * C + head * SF_HD + 8 * g + 4 * lk);
        br[g * 4 + 0] = a4.x; br[g * 4 + 1] = a4.y; br[g * 4 + 2] = a4.z; br[g * 4 + 3] = a4.w;
      }
    } else {
      const float bv = p.bqkv[2 * C + head * SF_HD + li];
#pragma unroll
      for (int r = 0; r < 16; ++r) br[r] = bv;
    }
    const float mul = mat == 0 ? p.scale : 1.0f;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (acc[tt][s2 * 8 + e] + br[s2 * 8 + e]) * mul;
        if (mat == 0) sf_split8(f, qfh[tt][s2], qfl[tt][s2]);
        else if (mat == 1) sf_split8(f, kfh[tt][s2], kfl[tt][s2]);
        else sf_split8(f, vfh[tt][s2], vfl[tt][s2]);
      }
  }
  __syncthreads();                                   // every wave is done reading the Xn image
  OCCF_SCHED_FENCE();
  fetch_w(3, 1);                                     // proj fragments: in flight during the attention
  OCCF_SCHED_FENCE();

  // ---- attention of this head; the result goes to the (reused) image as the B operand of proj
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int qi = qt * 32 + li;
    f32x16 st[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) st[kt][r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        st[kt] = occf_mfma_bf16_32x32x16(kfl[kt][s2], qfh[qt][s2], st[kt]);
        st[kt] = occf_mfma_bf16_32x32x16(kfh[kt][s2], qfl[qt][s2], st[kt]);
        st[kt] = occf_mfma_bf16_32x32x16(kfh[kt][s2], qfh[qt][s2], st[kt]);
      }
    }
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int tv = lds_rel[key * 64 + qi];
        float a = st[kt][r] + lds_bias[wave][tv & 0xFF];
        if (tv & 0x100) a += -100.0f;
        a = tv == 0xFFFF ? -INFINITY : a;
        st[kt][r] = a;
        mx = fmaxf(mx, a);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float pv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          pv[e] = __expf(st[kt][s2 * 8 + e] - mx);    // padding keys: exp(-inf) = 0
          sum += pv[e];
        }
        bf16x8 ph, pl;
        sf_split8(pv, ph, pl);
        ot = occf_mfma_bf16_32x32x16(vfl[kt][s2], ph, ot);
        ot = occf_mfma_bf16_32x32x16(vfh[kt][s2], pl, ot);
        ot = occf_mfma_bf16_32x32x16(vfh[kt][s2], ph, ot);
      }
    sum += __shfl_xor(sum, 32);
    const float inv = 1.0f / sum;
    // Ot (lane = token qi, registers = d) -> image [ks = 2 head + (r>>3)][row qi][slot (r>>2)&1][e = 4 lk + (r&3)]
#pragma unroll
    for (int g = 0; g < 4; ++g)
      sf_put4(img_h, img_l, qi, head * SF_HD + 8 * g + 4 * lk, ot[g * 4 + 0] * inv, ot[g * 4 + 1] * inv,
              ot[g * 4 + 2] * inv, ot[g * 4 + 3] * inv);
  }
  __syncthreads();

  // ---- proj + bias + residual: this wave produces output channels 32 wave .. +31 of every token
  f32x16 ao[2];
#pragma unroll
  for (int tt = 0; tt < 2; ++tt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ao[tt][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) {
    const bf16x8 wh = wfh[1][ks], wl = wfl[1][ks];      // proj fragments
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int off = ks * 2048 + (tt * 32 + li) * 32 + lk * 16;
      const bf16x8 oh = *(const bf16x8*)(img_h + off), ol = *(const bf16x8*)(img_l + off);
      ao[tt] = occf_mfma_bf16_32x32x16(wl, oh, ao[tt]);
      ao[tt] = occf_mfma_bf16_32x32x16(wh, ol, ao[tt]);
      ao[tt] = occf_mfma_bf16_32x32x16(wh, oh, ao[tt]);
    }
  }
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int t = tt * 32 + li;
    const int tok = t < SF_T ? lds_tok[t] : -1;
    if (tok < 0) continue;                           // padded / idle token columns are cropped
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c0 = wave * 32 + 8 * g + 4 * lk;
      const float4 bb = *(const float4*)(p.bp + c0);
      const float4 xr = *(const float4*)(p.x + (long)tok * C + c0);
      *(float4*)(p.out + (long)tok * C + c0) =
          make_float4(ao[tt][g * 4 + 0] + bb.x + xr.x, ao[tt][g * 4 + 1] + bb.y + xr.y,
                      ao[tt][g * 4 + 2] + bb.z + xr.z, ao[tt][g * 4 + 3] + bb.w + xr.w);
    }
  }
}

// w[R][128] (bf16) -> [R / 32][8 k-steps][64 lanes][8]: element e of lane (lk, li) = w[g * 32 + li][ks * 16 + lk * 8 + e]
__global__ void __launch_bounds__(256) swin_pack_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ f, int R) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)R * 16) return;                     // 16-byte groups: R * 128 / 8
  const int lane = (int)(gid & 63);
  const int ks = (int)((gid >> 6) & 7);
  const int g = (int)(gid >> 9);
  const int li = lane & 31, lk = lane >> 5;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  *(u4*)(f + gid * 8) = *(const u4*)(w + (long)(g * 32 + li) * SF_C + ks * 16 + lk * 8);
}

extern "C" int occf_swin_attn_pack(const uint16_t* w_hi, const uint16_t* w_lo, uint16_t* f_hi, uint16_t* f_lo, int rows,
                                   int C, void* stream) {
  if (C != SF_C || rows <= 0 || rows % 32 || !w_hi || !w_lo || !f_hi || !f_lo) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(swin_pack_kernel, dim3(occf_cdiv((long)rows * 16, 256)), dim3(256), 0, st, w_hi, f_hi, rows);
  hipLaunchKernelGGL(swin_pack_kernel, dim3(occf_cdiv((long)rows * 16, 256)), dim3(256), 0, st, w_lo, f_lo, rows);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_swin_attn_fused_fwd(const float* x, const float* ln_gamma, const float* ln_beta, float eps,
                                        const uint16_t* wqkv_hi, const uint16_t* wqkv_lo, const float* bqkv,
                                        const float* bias_table, const uint16_t* wproj_hi, const uint16_t* wproj_lo,
                                        const float* bproj, float* out, int B, int X, int Y, int S, int C, int heads,
                                        int shift, int weights_packed, void* stream) {
  if (C != SF_C || heads != 4) return OCCF_ESHAPE;
  if (B <= 0 || X <= 0 || Y <= 0 || S <= 0 || shift < 0 || shift >= SF_WS) return OCCF_EINVAL;
  if (!wqkv_lo || !wproj_lo || !bqkv || !bproj) return OCCF_EINVAL;
  if (x == out) return OCCF_EINVAL;                  // windows read their (shifted) neighbours' rows
  const int nwx = (X + SF_WS - 1) / SF_WS, nwy = (Y + SF_WS - 1) / SF_WS;
  const long blocks = (long)B * S * nwx * nwy;
  if (blocks >= 2147483647L) return OCCF_ESHAPE;
  SwinAttnArgs a = {x, ln_gamma, ln_beta, wqkv_hi, wqkv_lo, bqkv, bias_table, wproj_hi, wproj_lo, bproj, out,
                    B, X, Y, S, shift, eps, (float)(1.0 / sqrt((double)SF_HD)), weights_packed != 0};
  hipLaunchKernelGGL(swin_attn_fused_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  OCCF_LAUNCH_CHECK();
}
