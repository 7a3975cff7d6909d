// Fused token MLP:   out = LNpost?( x + W2 . act( W1 . LNpre?(x) + b1 ) + b2 )
//
// One kernel for the feed-forward halves of the hot path's transformer blocks:
//   * SwinBlock FFN of the dual-path encoder (pre-LN, GELU, hidden = C):
//     projects/mmdet3d_plugin/occformer/backbones/modules/window_attention.py:356-361 + mmcv FFN
//   * pixel-decoder encoder-layer FFN (ReLU, hidden = 4C, LayerNorm AFTER the residual):
//     mmcv BaseTransformerLayer ('ffn', 'norm'), multiscale_deformattn_3d.py:81
// The reference (and our unfused path) makes 5-7 passes over the token tensor (LN, fc1, act, fc2,
// residual, LN) and round-trips the [tokens, hidden] activation through HBM; here a workgroup keeps
// its 64 tokens in LDS, streams the weights (L2-resident, pre-split bf16 hi/lo) and touches HBM
// once to read x and once to write out.
//
// 256 threads = 4 waves (2 x 2); 64-token tile; hidden processed in chunks of 128:
//   GEMM1  h[64 x 128] = act(Xn[64 x C] . W1c^T + b1c)      (per wave 32 x 64)
//   GEMM2  acc[64 x C] += h[64 x 128] . W2c^T               (per wave 32 x C/2)
// on v_mfma_f32_32x32x16_bf16 with the 3-term bf16 split (or plain bf16).  Operand tiles use the
// 64-B row / XOR-swizzled k-slot LDS layout of gemm_bf16.hip, one 64-row slab per 32-wide k-tile.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

struct MlpArgs {
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
  int C, H, act, ln_mode;     // ln_mode: 0 none, 1 pre-LN (on the MLP input only), 2 post-LN
  float eps;
};

__device__ __forceinline__ uint32_t ml_bf16_rne(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
#else
  const uint32_t u = __float_as_uint(x);
#endif
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float ml_bf16_up(uint32_t h) {
#ifdef OCCF_EMU
  uint32_t u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(h << 16);
#endif
}
__device__ __forceinline__ void ml_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  occf_bf16_split2(a, b, hi, lo);
}
__device__ __forceinline__ int ml_slot(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float ml_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
typedef uint32_t ml_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t ml_u2 __attribute__((ext_vector_type(2)));

// store 4 consecutive-k fp32 values of operand row `row` at k index `k` (multiple of 4) into a
// [k-tile][rows][64 B] hi/lo image with `rows` rows per k-tile
__device__ __forceinline__ void ml_put4(unsigned char* hi, unsigned char* lo, int rows, int row, int k, float a,
                                        float b, float c, float d, bool three) {
  uint32_t h0, l0, h1, l1;
  ml_split2(a, b, h0, l0);
  ml_split2(c, d, h1, l1);
  const ml_u2 h = {h0, h1}, l = {l0, l1};
  const int off = (k >> 5) * rows * 64 + ml_slot(row, (k & 31) >> 3) + ((k >> 2) & 1) * 8;
  *(ml_u2*)(hi + off) = h;
  if (three) *(ml_u2*)(lo + off) = l;
}

template <int TN, int TERMS>     // C = 64 * TN
__global__ void __launch_bounds__(256) mlp_fused_kernel(MlpArgs p) {
  constexpr int C = 64 * TN;
  constexpr int KT_C = C / 32;            // k-tiles of GEMM1
  constexpr int WROWS = C > 128 ? C : 128;
  OCCF_DYN_SMEM(smem);
  // LDS carve-up (bytes): Xh/Xl [KT_C][64][64], Hh/Hl [4][64][64], Wh/Wl [WROWS][64]
  unsigned char* Xh = (unsigned char*)smem;
  unsigned char* Xl = Xh + KT_C * 4096;
  unsigned char* Hh = Xl + KT_C * 4096;
  unsigned char* Hl = Hh + 4 * 4096;
  unsigned char* Wh = Hl + 4 * 4096;
  unsigned char* Wl = Wh + WROWS * 64;
  float* otile = (float*)smem;            // [64][C + 4] fp32, aliases X/H after the last chunk (post-LN)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  const long m0 = (long)occf_xcd_remap(blockIdx.x, gridDim.x) * 64;
  const bool three = TERMS == 3;

  // ---- stage X (optionally layer-normed): 16 lanes per row, 16 rows per pass
  {
    const int sub = tid & 15, rloc = tid >> 4;
    constexpr int NV = C / 64;            // float4 per lane per row
    // all 4 * NV row pieces of this thread are fetched first (rows >= M read row M-1 and are zeroed)
    float4 xv[4][NV];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const long m = m0 + pass * 16 + rloc;
      const long mc = m < p.M ? m : (long)p.M - 1;
#pragma unroll
      for (int j = 0; j < NV; ++j) xv[pass][j] = *(const float4*)(p.x + mc * C + (sub + 16 * j) * 4);
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 16 + rloc;
      const long m = m0 + row;
      float4 v[NV];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        v[j] = m < p.M ? xv[pass][j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
      if (p.ln_mode == 1) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
          q += (a * a + b * b) + (c * c + d * d);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int c0 = (sub + 16 * j) * 4;
          const float4 g = *(const float4*)(p.gamma + c0);
          const float4 bb = *(const float4*)(p.beta + c0);
          v[j] = make_float4((v[j].x - mean) * rstd * g.x + bb.x, (v[j].y - mean) * rstd * g.y + bb.y,
                             (v[j].z - mean) * rstd * g.z + bb.z, (v[j].w - mean) * rstd * g.w + bb.w);
        }
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) ml_put4(Xh, Xl, 64, row, (sub + 16 * j) * 4, v[j].x, v[j].y, v[j].z, v[j].w, three);
    }
  }

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // Weight tiles (`rows` rows x 32 k of a row-major bf16 matrix) form one flat stream over the chunks:
  // per chunk KT_C tiles of W1 (rows = 128) then 4 tiles of W2 (rows = C).  Tile f+1 is fetched into
  // registers (unconditional, batched loads) while tile f multiplies, and committed to LDS after it.
  constexpr int NWP = WROWS * 4 / 256;                  // 16-B pieces per thread per array
  constexpr int SPC = KT_C + 4;                          // steps (tiles) per chunk
  const int n_chunks = p.H / 128;
  const int n_steps = n_chunks * SPC;
  ml_u4 wrh[NWP], wrl[NWP];
  auto tile_rows = [&](int f) __attribute__((always_inline)) { return (f % SPC) < KT_C ? 128 : C; };
  auto fetch_w = [&](int f) __attribute__((always_inline)) {
    const int fc = f < n_steps ? f : n_steps - 1;
    const int ch = fc / SPC, r = fc - ch * SPC;
    const bool g1 = r < KT_C;
    const uint16_t* Wgh = g1 ? p.W1h : p.W2h;
    const uint16_t* Wgl = g1 ? p.W1l : p.W2l;
    const int rows = g1 ? 128 : C;
    const long row0 = g1 ? (long)ch * 128 : 0, ld = g1 ? (long)C : (long)p.H;
    const long k0 = g1 ? (long)r * 32 : (long)ch * 128 + (r - KT_C) * 32;
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      const int idc = idx < rows * 4 ? idx : rows * 4 - 1;
      const long o = (row0 + (idc >> 2)) * ld + k0 + (idc & 3) * 8;
      wrh[i] = *(const ml_u4*)(Wgh + o);
      if (three) wrl[i] = *(const ml_u4*)(Wgl + o);
    }
  };
  auto commit_w = [&](int rows) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
      const int idx = tid + i * 256;
      if (idx < rows * 4) {
        const int off = ml_slot(idx >> 2, idx & 3);
        *(ml_u4*)(Wh + off) = wrh[i];
        if (three) *(ml_u4*)(Wl + off) = wrl[i];
      }
    }
  };
  fetch_w(0);
  commit_w(tile_rows(0));
  __syncthreads();                                        // X image and the first weight tile are in LDS
  int f = 0;

  for (int ch = 0; ch < n_chunks; ++ch) {
    // ---------------- GEMM1: h = act(Xn . W1[ch*128 .. +128, :]^T + b1)
    f32x16 hacc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) hacc[j][r] = 0.f;
    for (int kt = 0; kt < KT_C; ++kt) {
      fetch_w(f + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kslot = s * 2 + lk;
        const int aoff = kt * 4096 + ml_slot(wm * 32 + li, kslot);
        const bf16x8 ah = *(const bf16x8*)(Xh + aoff);
        bf16x8 al;
        if (three) al = *(const bf16x8*)(Xl + aoff);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int boff = ml_slot(wn * 64 + j * 32 + li, kslot);
          const bf16x8 bh = *(const bf16x8*)(Wh + boff);
          if (three) {
            const bf16x8 bl = *(const bf16x8*)(Wl + boff);
            hacc[j] = occf_mfma_bf16_32x32x16(al, bh, hacc[j]);
            hacc[j] = occf_mfma_bf16_32x32x16(ah, bl, hacc[j]);
          }
          hacc[j] = occf_mfma_bf16_32x32x16(ah, bh, hacc[j]);
        }
      }
      if (kt + 1 < KT_C) {
        __syncthreads();                                  // every wave is done with this weight tile
        commit_w(tile_rows(f + 1));
        __syncthreads();
        ++f;
      }
    }
    // epilogue 1: bias + activation, re-laid as the A operand of GEMM2 (row = token, k = hidden unit)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int hcol = wn * 64 + j * 32 + li;               // hidden unit inside the chunk
      const float bv = p.b1 ? p.b1[ch * 128 + hcol] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        float v = hacc[j][r] + bv;
        v = p.act == 2 ? ml_gelu(v) : (p.act == 1 ? fmaxf(v, 0.f) : v);
        const uint32_t hb = ml_bf16_rne(v);
        const int off = (hcol >> 5) * 4096 + ml_slot(row, (hcol & 31) >> 3) + (hcol & 7) * 2;
        *(uint16_t*)(Hh + off) = (uint16_t)hb;
        if (three) *(uint16_t*)(Hl + off) = (uint16_t)ml_bf16_rne(v - ml_bf16_up(hb));
      }
    }
    __syncthreads();                                      // last W1 tile consumed, h written
    commit_w(tile_rows(f + 1));
    __syncthreads();
    ++f;
    // ---------------- GEMM2: acc += h . W2[:, ch*128 .. +128]^T
    for (int kt = 0; kt < 4; ++kt) {
      fetch_w(f + 1);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int kslot = s * 2 + lk;
        const int aoff = kt * 4096 + ml_slot(wm * 32 + li, kslot);
        const bf16x8 ah = *(const bf16x8*)(Hh + aoff);
        bf16x8 al;
        if (three) al = *(const bf16x8*)(Hl + aoff);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int boff = ml_slot(wn * (C / 2) + j * 32 + li, kslot);
          const bf16x8 bh = *(const bf16x8*)(Wh + boff);
          if (three) {
            const bf16x8 bl = *(const bf16x8*)(Wl + boff);
            acc[j] = occf_mfma_bf16_32x32x16(al, bh, acc[j]);
            acc[j] = occf_mfma_bf16_32x32x16(ah, bl, acc[j]);
          }
          acc[j] = occf_mfma_bf16_32x32x16(ah, bh, acc[j]);
        }
      }
      __syncthreads();                                    // every wave is done with this weight tile (and h)
      commit_w(tile_rows(f + 1));
      __syncthreads();
      ++f;
    }
  }

  // ---------------- epilogue 2: + b2 + residual x, optional post-LN
  if (p.ln_mode != 2) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = wn * (C / 2) + j * 32 + li;
      const float bv = p.b2 ? p.b2[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < p.M) p.out[m * C + n] = acc[j][r] + bv + p.x[m * C + n];
      }
    }
    return;
  }
  __syncthreads();                                        // all MFMA reads of X / H are done
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = wn * (C / 2) + j * 32 + li;
    const float bv = p.b2 ? p.b2[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      const long m = m0 + row;
      otile[row * (C + 4) + n] = acc[j][r] + bv + (m < p.M ? p.x[m * C + n] : 0.f);
    }
  }
  __syncthreads();
  {
    const int sub = tid & 15, rloc = tid >> 4;
    constexpr int NV = C / 64;
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 16 + rloc;
      const long m = m0 + row;
      float4 v[NV];
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        v[j] = *(const float4*)(otile + row * (C + 4) + (sub + 16 * j) * 4);
        s += (v[j].x + v[j].y) + (v[j].z + v[j].w);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o);
      const float mean = s / (float)C;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < NV; ++j) {
        const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) q += __shfl_xor(q, o);
      const float rstd = 1.0f / sqrtf(q / (float)C + p.eps);
      if (m < p.M) {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const int c0 = (sub + 16 * j) * 4;
          const float4 g = *(const float4*)(p.gamma + c0);
          const float4 bb = *(const float4*)(p.beta + c0);
          *(float4*)(p.out + m * C + c0) =
              make_float4((v[j].x - mean) * rstd * g.x + bb.x, (v[j].y - mean) * rstd * g.y + bb.y,
                          (v[j].z - mean) * rstd * g.z + bb.z, (v[j].w - mean) * rstd * g.w + bb.w);
        }
      }
    }
  }
}

static size_t mlp_lds_bytes(int C, int terms) {
  const size_t x = (size_t)(C / 32) * 4096 * 2, h = 4 * 4096 * 2, w = (size_t)(C > 128 ? C : 128) * 64 * 2;
  const size_t o = (size_t)64 * (C + 4) * 4;
  (void)terms;
  const size_t a = x + h + w;
  return a > o ? a : o;
}

// mlp_chain.hip: register-chained variant for C = 128 / 192
int occf_mlp_chain_launch(const float* x, const float* ln_gamma, const float* ln_beta, const uint16_t* w1_hi,
                          const uint16_t* w1_lo, const float* b1, const uint16_t* w2_hi, const uint16_t* w2_lo,
                          const float* b2, float* out, long M, int C, int H, int act, int ln_mode, float eps, int terms,
                          hipStream_t st);

extern "C" int occf_mlp_fused_fwd(const float* x, const float* ln_gamma, const float* ln_beta,
                                  const uint16_t* w1_hi, const uint16_t* w1_lo, const float* b1,
                                  const uint16_t* w2_hi, const uint16_t* w2_lo, const float* b2, float* out, long M,
                                  int C, int H, int act, int ln_mode, float eps, int terms, void* stream) {
  if (M <= 0 || (C != 128 && C != 192 && C != 256) || H <= 0 || H % 128 != 0) return OCCF_ESHAPE;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  if (terms == 3 && (w1_lo == nullptr || w2_lo == nullptr)) return OCCF_EINVAL;
  if (ln_mode != 0 && (ln_gamma == nullptr || ln_beta == nullptr)) return OCCF_EINVAL;
  // OCCF_MLP_CHAIN: 0 = never, 1 = the register-chained kernels: C = 192 (any H), C = H = 128 (weight-resident),
  // 2 = also C = 128 with H != 128
  static const int chain = [] {
    const char* e = getenv("OCCF_MLP_CHAIN");
    return e ? atoi(e) : 1;
  }();
  if (chain && (C == 192 || (C == 128 && (H == 128 || chain >= 2)))) {
    const int rc = occf_mlp_chain_launch(x, ln_gamma, ln_beta, w1_hi, w1_lo, b1, w2_hi, w2_lo, b2, out, M, C, H, act,
                                         ln_mode, eps, terms, (hipStream_t)stream);
    if (rc != OCCF_ESHAPE) return rc;
  }
  MlpArgs a = {x, ln_gamma, ln_beta, w1_hi, w1_lo, b1, w2_hi, w2_lo, b2, out, M, C, H, act, ln_mode, eps};
  const size_t lds = mlp_lds_bytes(C, terms);
  const unsigned grid = (unsigned)occf_cdiv(M, 64);
  hipStream_t st = (hipStream_t)stream;
#ifndef OCCF_EMU
#define OCCF_MLP_ATTR(TN_, T_)                                                                              \
  do {                                                                                                      \
    static bool done = false;                                                                               \
    if (!done) {                                                                                            \
      hipError_t e = hipFuncSetAttribute((const void*)mlp_fused_kernel<TN_, T_>,                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);           \
      if (e != hipSuccess) return (int)e;                                                                   \
      done = true;                                                                                          \
    }                                                                                                       \
  } while (0)
#else
#define OCCF_MLP_ATTR(TN_, T_) do { } while (0)
#endif
#define OCCF_MLP_LAUNCH(TN_, T_)                                                                      \
  do {                                                                                                \
    OCCF_MLP_ATTR(TN_, T_);                                                                           \
    hipLaunchKernelGGL((mlp_fused_kernel<TN_, T_>), dim3(grid), dim3(256), lds, st, a);               \
  } while (0)
  if (C == 128) { if (terms == 3) OCCF_MLP_LAUNCH(2, 3); else OCCF_MLP_LAUNCH(2, 1); }
  else if (C == 192) { if (terms == 3) OCCF_MLP_LAUNCH(3, 3); else OCCF_MLP_LAUNCH(3, 1); }
  else { if (terms == 3) OCCF_MLP_LAUNCH(4, 3); else OCCF_MLP_LAUNCH(4, 1); }
#undef OCCF_MLP_LAUNCH
#undef OCCF_MLP_ATTR
  OCCF_LAUNCH_CHECK();
}
