// Weight-resident persistent linear for the STREAMING shapes of the path (included by gemm_bf16.hip):
//
//     C[M, N] = act( A[M, K] . W[N, K]^T + bias ) (+ residual),   M >> N, K <= 256
//
// (the Swin / pixel-decoder / decoder-memory projections: [680 000, 128] x 128 / 384, [91 250, 192] x 192 / 384 / 768,
// [90 000, 256] x 256 ...: 14 of the 22 ms of `linear` per training step, 5 of the 6.4 ms per forward.)  These are
// HBM-bound by 1.5 ... 2.5x over the matrix work, and the generic tile kernel above runs them 2.7x over either floor:
// per 128 x BN tile it re-stages the weight through LDS, converts the activation tile into LDS behind two barriers per
// 32-wide k-tile, and nothing overlaps one tile's epilogue with the next tile's loads except a co-resident workgroup.
//
// Here a workgroup (8 waves, one per CU: two per SIMD) stages its block of W -- up to 192 output channels, hi and lo bf16
// images in MFMA-fragment order, <= 148 KB -- into LDS ONCE and then streams token tiles: each wave owns 32 tokens at
// a time and works on the TRANSPOSED problem (as mlp_chain.hip)
//
//     Ct[32 channels x 32 tokens] = Wtile[32 x K] . At[K x 32 tokens]        A operand = W rows (LDS), B = At (registers)
//
// so the activation rows go global -> registers -> split (hi, lo) without touching LDS, the result leaves the
// accumulators as four 16-byte stores per lane (lane = token, 4 consecutive channels), and there is no barrier after
// the staging one: the eight waves drift apart and one wave's row latency / epilogue hides under its SIMD partner's
// MFMAs.  HBM traffic = the rows once in, once out (N blocks > 1: the blocks of one token range run on the same XCD at
// the same time, so the re-reads are L2 hits).
//
// The k index of the contraction is free to permute as long as both operands agree; lane (token li, half lk) takes the 8
// consecutive input channels 16 ks + 8 lk .. of k-step ks, the W image is staged to match.
#pragma once

struct GemmStreamArgs {
  const float* A;
  const uint16_t* Wh;
  const uint16_t* Wl;
  const float* bias;
  const float* residual;
  float* C;
  long M;
  int N, K;
  long lda, ldc, ldr;
  int act;             // 0 none, 1 ReLU, 2 GELU, 3: multiply by GELU'(aux[row, col]) (aux = `residual`; no residual add)
  float* pre_out;      // optional: the pre-activation (acc + bias) as a second output [M, N] (row stride ldc)
  const float* row_scale;   // optional DropPath scale per sample: out = residual + row_scale[sample(row)] * (acc + bias),
  long xy_s;                // sample(row) = (row / xy_s) * s_slices + row % s_slices  (token rows ((b XY + xy) S + s))
  int s_slices;
  int ntb;             // 32-channel tiles per workgroup block (N block = 32 * ntb)
  int n_blocks;        // N / (32 * ntb)
  int streams;         // workgroups per N block (each walks the wave tiles q * 8 + wave, + streams * 8, ...)
};

#define GS_NW 8
// d/dx of the exact GELU (the formula of bwd_elem.hip's act_bwd_kernel)
__device__ __forceinline__ float gs_gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}
template <int V>
struct gs_int {
  static constexpr int value = V;
};

// PLAIN: bias only (act 0, no residual / aux, no second output, no row scale) -- the epilogue of most launches of a
// training step, without the general form's ~10 wave-uniform branches per 4-channel group (40 per 32-channel tile of
// 36 MFMAs; scripts/gemm_stream_ablation_probe.py: the kernel with every load and store removed ran 2.5x over its
// MFMA time)
template <int KS, int TERMS, bool PLAIN = false>     // K = 16 * KS
__global__ void __launch_bounds__(GS_NW * 64) gemm_stream_kernel(GemmStreamArgs p) {
  constexpr int K = 16 * KS;
  constexpr int IMG = KS * 1024;                   // bytes of one image (hi or lo) of a 32-channel tile
  constexpr int ARR = TERMS == 3 ? 2 : 1;
  // the NEXT tile's rows travel while this one multiplies (K / 2 more registers: up to K = 192 inside the 256-register
  // budget of two waves per SIMD; r05b without it: 4.2 TB/s on [680 000, 128] x 128 -- two of a CU's eight waves in their
  // load phase at a time is 32 KB in flight per CU, i.e. latency-bound)
  // K = 192 / 224 / 256: the first 8 / 6 / 4 k-steps' worth of the next rows (the rest is fetched at the top of its own iteration)
  constexpr int PFK = KS <= 10 ? KS : KS == 12 ? 8 : KS == 14 ? 6 : 4;
  OCCF_DYN_SMEM(smem);
  unsigned char* Wimg = (unsigned char*)smem;      // tile j: [hi | lo] at j * ARR * IMG; slot = ks * 64 + row * 2 + k2
  float* bias_s = (float*)(Wimg + (size_t)p.ntb * ARR * IMG);       // [32 * ntb]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lk = lane >> 5;
  // workgroup -> (XCD x, slot s); the N blocks of one stream sit on the same XCD
  const int b = blockIdx.x;
  const int x = b & 7, s = b >> 3;
  const int nb = s % p.n_blocks;
  const int q = x + 8 * (s / p.n_blocks);
  const int n0 = nb * 32 * p.ntb;

  // ---- this block's weight rows -> LDS in fragment order, as LDS-DMA (global -> LDS without registers): every wave
  // issues all of its 1 KB pieces back to back and waits once (r05b: a load -> store loop of 18 dependent L2 round
  // trips per thread, all 256 workgroups on the same 147 KB, cost half of the 52 us of the [91 250, 192] x 192 call)
  {
    const uint32_t wbytes = (uint32_t)(32 * p.ntb) * K * 2u;
    const int pieces = p.ntb * KS;                 // 64-slot (1 KB) pieces per array
    const int r = lane;                            // slot within the piece: row = (r >> 1) & 31, k2 = r & 1
    for (int arr = 0; arr < ARR; ++arr) {
      const occf_bbuf wb = occf_make_bbuf((arr ? p.Wl : p.Wh) + (long)n0 * K, wbytes);
      for (int pc = wave; pc < pieces; pc += GS_NW) {
        const int j = pc / KS, ks = pc - j * KS;
        const uint32_t voff = (uint32_t)(((j * 32 + (r >> 1)) * K + ks * 16 + (r & 1) * 8) * 2);
        occf_bbuf_load_lds_b128(wb, voff, Wimg + (size_t)j * ARR * IMG + arr * IMG + ks * 1024);
      }
    }
  }
  for (int i = tid; i < 32 * p.ntb; i += GS_NW * 64) bias_s[i] = p.bias ? p.bias[n0 + i] : 0.f;
  __syncthreads();                                 // (waits for this wave's LDS-DMA, then for everybody's)

  const long n_wtiles = (p.M + 31) / 32;
  const long wt_step = (long)p.streams * GS_NW;
  float4 ra[KS], rb[KS];
  auto load_rows = [&](long wt, auto first, auto last) __attribute__((always_inline)) {
    const long tok = wt * 32 + li;
    const float* xr = p.A + (tok < p.M ? tok : p.M - 1) * p.lda + lk * 8;
#pragma unroll
    for (int ks = decltype(first)::value; ks < decltype(last)::value; ++ks) {
      ra[ks] = *(const float4*)(xr + ks * 16);
      rb[ks] = *(const float4*)(xr + ks * 16 + 4);
    }
  };
  typedef gs_int<0> k_begin;
  typedef gs_int<PFK> k_pf;
  typedef gs_int<KS> k_end;
  long wt = (long)q * GS_NW + wave;
  if (PFK > 0 && wt < n_wtiles) load_rows(wt, k_begin(), k_pf());
  for (; wt < n_wtiles; wt += wt_step) {
    if (PFK < KS) load_rows(wt, k_pf(), k_end());
    const long tok = wt * 32 + li;
    const bool tok_ok = tok < p.M;
    const long tokc = tok_ok ? tok : p.M - 1;
    // ---- 32 token rows (B operand: lane = token li, 8 channels 16 ks + 8 lk ..) split into (hi, lo)
    bf16x8 xh[KS], xl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      uint32_t h[4], l[4];
      occf_bf16_split2(ra[ks].x, ra[ks].y, h[0], l[0]);
      occf_bf16_split2(ra[ks].z, ra[ks].w, h[1], l[1]);
      occf_bf16_split2(rb[ks].x, rb[ks].y, h[2], l[2]);
      occf_bf16_split2(rb[ks].z, rb[ks].w, h[3], l[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        xh[ks][2 * e] = (short)(h[e] & 0xFFFFu);
        xh[ks][2 * e + 1] = (short)(h[e] >> 16);
        xl[ks][2 * e] = (short)(l[e] & 0xFFFFu);
        xl[ks][2 * e + 1] = (short)(l[e] >> 16);
      }
    }
    if (PFK > 0) {       // (the last tile re-reads itself: no branch around loads)
      OCCF_SCHED_FENCE();
      load_rows(wt + wt_step < n_wtiles ? wt + wt_step : wt, k_begin(), k_pf());
      OCCF_SCHED_FENCE();
    }
    float* crow = p.C + tokc * p.ldc + n0 + lk * 4;
    float* prow = p.pre_out ? p.pre_out + tokc * p.ldc + n0 + lk * 4 : nullptr;
    const float* rrow = p.residual ? p.residual + tokc * p.ldr + n0 + lk * 4 : nullptr;
    const float rs = p.row_scale ? p.row_scale[(tokc / p.xy_s) * p.s_slices + tokc % p.s_slices] : 1.0f;
    for (int j = 0; j < p.ntb; ++j) {
      const unsigned char* Wt = Wimg + (size_t)j * ARR * IMG + li * 32 + lk * 16;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 wh = *(const bf16x8*)(Wt + ks * 1024);
        if (TERMS == 3) {
          const bf16x8 wl = *(const bf16x8*)(Wt + IMG + ks * 1024);
          acc = occf_mfma_bf16_32x32x16(wl, xh[ks], acc);
          acc = occf_mfma_bf16_32x32x16(wh, xl[ks], acc);
        }
        acc = occf_mfma_bf16_32x32x16(wh, xh[ks], acc);
      }
      // ---- epilogue of this 32-channel tile: registers r = 4 g + e <-> channel 8 g + 4 lk + e of token li
      // (the residual rows are fetched HERE, not ahead of the MFMAs: 16 registers the prefetched rows need at K = 192;
      // the SIMD partner covers the latency)
      if (PLAIN) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 b4 = *(const float4*)(bias_s + j * 32 + g * 8 + lk * 4);
          const float4 v = make_float4(acc[4 * g] + b4.x, acc[4 * g + 1] + b4.y, acc[4 * g + 2] + b4.z, acc[4 * g + 3] + b4.w);
          if (tok_ok) *(float4*)(crow + j * 32 + g * 8) = v;
        }
        continue;
      }
      // the general epilogue, one wave-uniform branch per STAGE of a 32-channel tile (not per 4-channel group: the
      // branches, not the arithmetic, were what the bias-only variant above saved)
      float4 res[4], v[4];
      OCCF_SCHED_FENCE();
      if (rrow) {
#pragma unroll
        for (int g = 0; g < 4; ++g) res[g] = *(const float4*)(rrow + j * 32 + g * 8);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *(const float4*)(bias_s + j * 32 + g * 8 + lk * 4);
        v[g] = make_float4(acc[4 * g] + b4.x, acc[4 * g + 1] + b4.y, acc[4 * g + 2] + b4.z, acc[4 * g + 3] + b4.w);
      }
      if (prow && tok_ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(prow + j * 32 + g * 8) = v[g];
      }
      if (p.act == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          v[g] = make_float4(fmaxf(v[g].x, 0.f), fmaxf(v[g].y, 0.f), fmaxf(v[g].z, 0.f), fmaxf(v[g].w, 0.f));
      } else if (p.act == 2) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          v[g] = make_float4(occf_gelu_b(v[g].x), occf_gelu_b(v[g].y), occf_gelu_b(v[g].z), occf_gelu_b(v[g].w));
      } else if (p.act == 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          v[g].x *= gs_gelu_grad(res[g].x); v[g].y *= gs_gelu_grad(res[g].y);
          v[g].z *= gs_gelu_grad(res[g].z); v[g].w *= gs_gelu_grad(res[g].w);
        }
      }
      if (p.row_scale) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { v[g].x *= rs; v[g].y *= rs; v[g].z *= rs; v[g].w *= rs; }
      }
      if (rrow && p.act != 3) {
#pragma unroll
        for (int g = 0; g < 4; ++g) { v[g].x += res[g].x; v[g].y += res[g].y; v[g].z += res[g].z; v[g].w += res[g].w; }
      }
      if (tok_ok) {
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(crow + j * 32 + g * 8) = v[g];
      }
    }
  }
}

// the N block of a (K, N) problem: the most 32-channel tiles whose two images fit the LDS next to the bias (0: none)
static inline int occf_gemm_stream_ntb(int N, int K, int terms) {
  if (N % 32 != 0 || K % 16 != 0) return 0;
  const int tiles = N / 32;
  const long per_tile = (long)(terms == 3 ? 2 : 1) * (K / 16) * 1024 + 128;
  int best = 0;
  for (int t = 1; t <= tiles && t <= 8; ++t)
    if (tiles % t == 0 && t * per_tile <= 156 * 1024) best = t;
  return best;
}

// OCCF_GEMM_STREAM: 0 = off; n > 1 = take the streaming kernel from n rows up (default 16 384: below that the tile
// kernel's 128-row workgroups fill the chip better than 256-token persistent ones).  Read per call (tests switch it).
static inline long occf_gemm_stream_min_rows() {
  const char* e = getenv("OCCF_GEMM_STREAM");
  if (!e) return 16384L;
  const long n = atol(e);
  return n <= 0 ? -1L : (n == 1 ? 16384L : n);
}

// OCCF_ESHAPE: outside this kernel's envelope (the caller then takes the tile kernel)
static int occf_gemm_stream_launch(const float* A, const uint16_t* Wh, const uint16_t* Wl, const float* bias,
                                   const float* residual, float* C, long M, int N, int K, long lda, long ldc, long ldr,
                                   int act, int terms, hipStream_t st, float* pre_out = nullptr,
                                   const float* row_scale = nullptr, long xy_s = 1, int s_slices = 1) {
  if (M < 64) return OCCF_ESHAPE;                       // (the decoder's 100-query linears: no getenv on their path)
  const long min_rows = occf_gemm_stream_min_rows();
  if (min_rows < 0 || M < min_rows) return OCCF_ESHAPE;
  if (K != 64 && K != 96 && K != 128 && K != 160 && K != 192 && K != 224 && K != 256) return OCCF_ESHAPE;
  if ((lda | ldc) % 4 != 0 || (residual && ldr % 4 != 0) || (terms != 1 && terms != 3)) return OCCF_ESHAPE;
  const int ntb = occf_gemm_stream_ntb(N, K, terms);
  if (ntb == 0) return OCCF_ESHAPE;
  if (act < 0 || act > 3 || (act == 3 && !residual) || (row_scale && (xy_s <= 0 || s_slices <= 0))) return OCCF_EINVAL;
  // (row_scale[(row / xy_s) * s_slices + row % s_slices]: whole samples only, or the last rows read past the vector)
  if (row_scale && M % xy_s != 0) return OCCF_EINVAL;
  GemmStreamArgs a = {A, Wh, Wl, bias, residual, C, M, N, K, lda, ldc, ldr, act, pre_out, row_scale, xy_s, s_slices,
                      ntb, N / (32 * ntb), 0};
  if (a.n_blocks > 32) return OCCF_ESHAPE;
  // one workgroup per CU: 8 XCDs x (32 / n_blocks) streams per N block (OCCF_GEMM_STREAM_WGS caps it: tests walk the
  // tile loop with a handful of workgroups)
  const char* ecap = getenv("OCCF_GEMM_STREAM_WGS");
  const int cap = ecap && atoi(ecap) > 0 ? atoi(ecap) : 256;
  int per_xcd = cap / 8 / a.n_blocks;                    // streams per XCD and N block
  if (per_xcd < 1) per_xcd = 1;
  const long n_wtiles = (M + 31) / 32;
  while (per_xcd > 1 && (long)(per_xcd - 1) * 8 * GS_NW >= n_wtiles) --per_xcd;     // (no idle workgroups on small M)
  a.streams = 8 * per_xcd;
  const unsigned grid = (unsigned)(8 * per_xcd * a.n_blocks);
  const size_t lds = (size_t)ntb * (terms == 3 ? 2 : 1) * (K / 16) * 1024 + (size_t)ntb * 128;
  typedef void (*fn_t)(GemmStreamArgs);
  fn_t fn = nullptr;
  const bool plain = act == 0 && !residual && !pre_out && !row_scale;
#define GS_PICK(KS_)                                                                                     \
  case KS_: fn = terms == 3 ? (plain ? (fn_t)gemm_stream_kernel<KS_, 3, true> : (fn_t)gemm_stream_kernel<KS_, 3>) \
                            : (fn_t)gemm_stream_kernel<KS_, 1>; break
  switch (K / 16) {
    GS_PICK(4); GS_PICK(6); GS_PICK(8); GS_PICK(10); GS_PICK(12); GS_PICK(14); GS_PICK(16);
    default: return OCCF_ESHAPE;
  }
#undef GS_PICK
#ifndef OCCF_EMU
  static bool done[17][3] = {};
  if (!done[K / 16][terms == 3 ? (plain ? 2 : 1) : 0]) {
    hipError_t err = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (err != hipSuccess) return (int)err;
    done[K / 16][terms == 3 ? (plain ? 2 : 1) : 0] = true;
  }
#endif
  hipLaunchKernelGGL(fn, dim3(grid), dim3(GS_NW * 64), lds, st, a);
  return (int)hipGetLastError();
}
