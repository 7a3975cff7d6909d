// MFMA GEMM / implicit-GEMM convolution for gfx950, fp32 in / fp32 accumulate.
//
//   C[m, n] = epilogue( sum_k A(m, k) * W[n, k] )
//
// W is a torch Linear weight [N, K] (row-major, K contiguous); for convolutions it is the
// conv weight re-laid as [Cout, taps*Cin] with k = tap*Cin + cin, tap = (dx*kY + dy)*kZ + dz.
// A(m, k) comes from a loader:
//   DENSE  : A[m*lda + k]
//   CONV3D : m = ((b*Xo + xo)*Yo + yo)*Zo + zo over a channels-last input
//            [B, Xi, Yi, Zi, Cin] addressed through explicit strides, zero outside
//            (im2col is never materialised).  2-D convolutions are the Zi = Zo = kZ = 1 case.
// Epilogue: + bias[n], ReLU / exact GELU, + residual[m, n], row-major store.
//
// Used for every dense contraction of the hot path: the 3^3 / 1^3 voxel convolutions of
// DualpathTransformerBlock and the pixel decoder (dualpath_block.py:36-48,
// multiscale_deformattn_3d.py:70-116), the qkv / proj / FFN linears of the SwinBlock
// (window_attention.py:63-66,336-344), the deformable-attention projections, the decoder
// MLPs and the mask_embed x mask_feature contraction (mask2former_nusc_occ.py:448-455).
//
// Matrix core: v_mfma_f32_32x32x2_f32 -- exact fp32 (bitwise an fmaf chain), 64 FLOP/clk/SIMD
// (157 TF chip peak).  Per wave-instruction lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; result reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31].
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN (128x128, 128x64 or
// 64x64), BK = 16.  A and W tiles are staged k-major in LDS ([BK][BM]) so that the 32 lanes
// of a half-wave read 32 consecutive floats (conflict-free ds_read_b32) and each operand
// dword feeds a 64-cycle MFMA.  Global loads of tile t+1 are issued before the MFMAs of
// tile t (register prefetch); a float4 along K per thread.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define G_BK 16

struct ConvGeom {
  int Xo, Yo, Zo;            // output grid
  int Xi, Yi, Zi;            // input grid
  int kX, kY, kZ;            // kernel taps
  int stride, dil, pad_x, pad_y, pad_z;
  int Cin;
  long sb, sx, sy, sz;       // input element strides (channel stride = 1)
};

struct GemmArgs {
  const float* A;
  const float* W;
  const float* bias;
  const float* residual;
  float* C;
  int M, N, K;
  long lda, ldc, ldr;
  int act;                   // 0 none, 1 relu, 2 gelu (erf)
  ConvGeom g;
};

__device__ __forceinline__ float occf_gelu(float x) {
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

template <int BM, int BN, bool CONV>
__global__ void __launch_bounds__(256) gemm_f32_kernel(GemmArgs p) {
  constexpr int WM = BM / 2, WN = BN / 2;      // per-wave tile
  constexpr int TM = WM / 32, TN = WN / 32;    // 32x32 MFMA tiles per wave
  constexpr int NA = BM * 4 / 256, NB = BN * 4 / 256;   // float4 loads per thread per tile
  __shared__ __attribute__((aligned(16))) float As[G_BK][BM];
  __shared__ __attribute__((aligned(16))) float Bs[G_BK][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // blockIdx.x walks N-tiles fastest: neighbouring workgroups share the same A rows (L2)
  const int n_tiles = (p.N + BN - 1) / BN;
  const unsigned wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const long m0 = (long)(wg / n_tiles) * BM;
  const int n0 = (wg % n_tiles) * BN;

  // ---- per-thread load bookkeeping (rows are fixed over the K loop)
  long a_base[NA];            // DENSE: element offset of the row; CONV: offset of (b, 0,0,0)
  int a_x[NA], a_y[NA], a_z[NA];
  bool a_ok[NA];
  int a_kq[NA], a_m[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + i * 256;
    a_m[i] = idx % BM;
    a_kq[i] = idx / BM;
    const long m = m0 + a_m[i];
    a_ok[i] = m < p.M;
    if (CONV) {
      const long mm = a_ok[i] ? m : 0;
      const int zo = (int)(mm % p.g.Zo);
      const int yo = (int)((mm / p.g.Zo) % p.g.Yo);
      const int xo = (int)((mm / ((long)p.g.Zo * p.g.Yo)) % p.g.Xo);
      const long b = mm / ((long)p.g.Zo * p.g.Yo * p.g.Xo);
      a_base[i] = b * p.g.sb;
      a_x[i] = xo * p.g.stride - p.g.pad_x;
      a_y[i] = yo * p.g.stride - p.g.pad_y;
      a_z[i] = zo * p.g.stride - p.g.pad_z;
    } else {
      a_base[i] = (a_ok[i] ? m : 0) * p.lda;
      a_x[i] = a_y[i] = a_z[i] = 0;
    }
  }
  long b_base[NB];
  bool b_ok[NB];
  int b_kq[NB], b_n[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + i * 256;
    b_n[i] = idx % BN;
    b_kq[i] = idx / BN;
    const int n = n0 + b_n[i];
    b_ok[i] = n < p.N;
    b_base[i] = (long)(b_ok[i] ? n : 0) * p.K;
  }

  float4 ra[NA], rb[NB];
  auto load_tile = [&](int kt) {
    const int k0 = kt * G_BK;
    if (CONV) {
      // Cin % 16 == 0: a BK-wide k-slab lies inside one tap and the decode is block-uniform;
      // otherwise (Cin % 4 == 0) every float4 piece decodes its own tap.
      const bool uniform = (p.g.Cin % G_BK) == 0;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int k = k0 + (uniform ? 0 : a_kq[i] * 4);
        const int tap = k / p.g.Cin;
        const int c0 = k - tap * p.g.Cin + (uniform ? a_kq[i] * 4 : 0);
        const int dz = tap % p.g.kZ, dy = (tap / p.g.kZ) % p.g.kY, dx = tap / (p.g.kZ * p.g.kY);
        const int xi = a_x[i] + dx * p.g.dil, yi = a_y[i] + dy * p.g.dil, zi = a_z[i] + dz * p.g.dil;
        const bool ok = a_ok[i] && k0 + a_kq[i] * 4 < p.K && xi >= 0 && xi < p.g.Xi && yi >= 0 &&
                        yi < p.g.Yi && zi >= 0 && zi < p.g.Zi;
        if (ok) {
          ra[i] = *(const float4*)(p.A + a_base[i] + xi * p.g.sx + yi * p.g.sy + zi * p.g.sz + c0);
        } else {
          ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        if (a_ok[i] && k0 + a_kq[i] * 4 < p.K) {
          ra[i] = *(const float4*)(p.A + a_base[i] + k0 + a_kq[i] * 4);
        } else {
          ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      if (b_ok[i] && k0 + b_kq[i] * 4 < p.K) {
        rb[i] = *(const float4*)(p.W + b_base[i] + k0 + b_kq[i] * 4);
      } else {
        rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      const int k = a_kq[i] * 4, m = a_m[i];
      As[k + 0][m] = ra[i].x; As[k + 1][m] = ra[i].y; As[k + 2][m] = ra[i].z; As[k + 3][m] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int k = b_kq[i] * 4, n = b_n[i];
      Bs[k + 0][n] = rb[i].x; Bs[k + 1][n] = rb[i].y; Bs[k + 2][n] = rb[i].z; Bs[k + 3][n] = rb[i].w;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (p.K + G_BK - 1) / G_BK;
  load_tile(0);
  store_tile();
  __syncthreads();
  const int li = lane & 31, lk = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
    for (int kp = 0; kp < G_BK / 2; ++kp) {
      const int k = kp * 2 + lk;
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[k][wm * WM + i * 32 + li];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[k][wn * WN + j * 32 + li];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_f32_32x32x2(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
    if (kt + 1 < nk) {
      store_tile();
      __syncthreads();
    }
  }

  // ---- epilogue
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 32 + li;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bv;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = occf_gelu(v);
        if (p.residual) v += p.residual[m * p.ldr + n];
        p.C[m * p.ldc + n] = v;
      }
    }
  }
}

template <bool CONV>
static int launch_gemm(const GemmArgs& a, hipStream_t st) {
  if (a.M <= 0 || a.N <= 0 || a.K <= 0 || a.K % 4 != 0) return OCCF_ESHAPE;
  // tile choice: widest tile that still yields >= ~2 waves of workgroups on 256 CUs
  const long t128 = (long)occf_cdiv(a.M, 128) * occf_cdiv(a.N, 128);
  const long t12864 = (long)occf_cdiv(a.M, 128) * occf_cdiv(a.N, 64);
  if (a.N > 64 && t128 >= 512) {
    hipLaunchKernelGGL((gemm_f32_kernel<128, 128, CONV>), dim3((unsigned)t128), dim3(256), 0, st, a);
  } else if (t12864 >= 512 || a.M >= 4096) {
    hipLaunchKernelGGL((gemm_f32_kernel<128, 64, CONV>), dim3((unsigned)t12864), dim3(256), 0, st, a);
  } else {
    const long t64 = (long)occf_cdiv(a.M, 64) * occf_cdiv(a.N, 64);
    hipLaunchKernelGGL((gemm_f32_kernel<64, 64, CONV>), dim3((unsigned)t64), dim3(256), 0, st, a);
  }
  return (int)hipGetLastError();
}

extern "C" int occf_linear_fwd(const float* x, const float* weight, const float* bias,
                               const float* residual, float* out, long M, int N, int K, long ldx,
                               long ldo, long ldr, int act, void* stream) {
  if (M >= 2147483647L) return OCCF_ESHAPE;
  if (ldx % 4 != 0 || K % 4 != 0) return OCCF_ESHAPE;     // float4 loads along K
  GemmArgs a = {};
  a.A = x; a.W = weight; a.bias = bias; a.residual = residual; a.C = out;
  a.M = (int)M; a.N = N; a.K = K; a.lda = ldx; a.ldc = ldo; a.ldr = ldr; a.act = act;
  return launch_gemm<false>(a, (hipStream_t)stream);
}

extern "C" int occf_conv3d_fwd(const float* x, const float* weight_tapmajor, const float* bias,
                               const float* residual, float* out, int B, int Xi, int Yi, int Zi, int Cin,
                               int Cout, int kX, int kY, int kZ, int stride, int dil, int pad_x,
                               int pad_y, int pad_z, long in_sb, long in_sx, long in_sy, long in_sz,
                               int act, void* stream) {
  if (B <= 0 || Cin % 4 != 0 || stride <= 0 || dil <= 0) return OCCF_ESHAPE;
  if (in_sb % 4 || in_sx % 4 || in_sy % 4 || in_sz % 4) return OCCF_ESHAPE;
  GemmArgs a = {};
  ConvGeom& g = a.g;
  g.Xi = Xi; g.Yi = Yi; g.Zi = Zi; g.kX = kX; g.kY = kY; g.kZ = kZ;
  g.stride = stride; g.dil = dil; g.pad_x = pad_x; g.pad_y = pad_y; g.pad_z = pad_z; g.Cin = Cin;
  g.Xo = (Xi + 2 * pad_x - dil * (kX - 1) - 1) / stride + 1;
  g.Yo = (Yi + 2 * pad_y - dil * (kY - 1) - 1) / stride + 1;
  g.Zo = (Zi + 2 * pad_z - dil * (kZ - 1) - 1) / stride + 1;
  g.sb = in_sb; g.sx = in_sx; g.sy = in_sy; g.sz = in_sz;
  const long M = (long)B * g.Xo * g.Yo * g.Zo;
  if (g.Xo <= 0 || g.Yo <= 0 || g.Zo <= 0 || M >= 2147483647L) return OCCF_ESHAPE;
  a.A = x; a.W = weight_tapmajor; a.bias = bias; a.residual = residual; a.C = out;
  a.M = (int)M; a.N = Cout; a.K = kX * kY * kZ * Cin; a.lda = 0; a.ldc = Cout; a.ldr = Cout; a.act = act;
  return launch_gemm<true>(a, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------
// Tiny contractions (the 100-query decoder: M <= 128 rows, N*M <= 256k): a 128-wide MFMA tile
// would be >75 % padding and the kernel is pure latency, so one thread computes one output
// with 16-byte loads along K (x row broadcast across the wave, W rows L1-resident).  Exact fp32.
__global__ void __launch_bounds__(256) linear_small_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ residual,
                                                           float* __restrict__ out, int M, int N, int K, long ldx,
                                                           long ldo, long ldr, int act) {
  // 8 lanes per output: each lane walks K in 16-byte pieces 128 B apart (the 8 lanes read one
  // contiguous 128-B line of the x row and of the W row), then a 3-step butterfly.
  const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const bool valid = gid < (long)M * N;
  const int n = valid ? (int)(gid % N) : 0, m = valid ? (int)(gid / N) : 0;
  const float* xr = x + m * ldx;
  const float* wr = w + (long)n * K;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 4
  for (int k = sub * 4; k < K; k += 32) {
    const float4 xv = *(const float4*)(xr + k);
    const float4 wv = *(const float4*)(wr + k);
    a0 = fmaf(xv.x, wv.x, a0);
    a1 = fmaf(xv.y, wv.y, a1);
    a2 = fmaf(xv.z, wv.z, a2);
    a3 = fmaf(xv.w, wv.w, a3);
  }
  float v = (a0 + a1) + (a2 + a3);
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  if (!valid || sub != 0) return;
  if (bias) v += bias[n];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = occf_gelu(v);
  if (residual) v += residual[m * ldr + n];
  out[m * ldo + n] = v;
}

// any K / row stride (scalar loads): the data gradient of a classifier head has K = number of classes
__global__ void __launch_bounds__(256) linear_small_scalar_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ w,
                                                                  const float* __restrict__ bias,
                                                                  const float* __restrict__ residual,
                                                                  float* __restrict__ out, int M, int N, int K, long ldx,
                                                                  long ldo, long ldr, int act) {
  const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const bool valid = gid < (long)M * N;
  const int n = valid ? (int)(gid % N) : 0, m = valid ? (int)(gid / N) : 0;
  const float* xr = x + m * ldx;
  const float* wr = w + (long)n * K;
  float v = 0.f;
  for (int k = sub; k < K; k += 8) v = fmaf(xr[k], wr[k], v);
  v += __shfl_xor(v, 1);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 4);
  if (!valid || sub != 0) return;
  if (bias) v += bias[n];
  if (act == 1) v = fmaxf(v, 0.f);
  else if (act == 2) v = occf_gelu(v);
  if (residual) v += residual[m * ldr + n];
  out[m * ldo + n] = v;
}

extern "C" int occf_linear_small_fwd(const float* x, const float* weight, const float* bias,
                                     const float* residual, float* out, int M, int N, int K, long ldx,
                                     long ldo, long ldr, int act, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return OCCF_ESHAPE;
  const long total = (long)M * N * 8;
  if (K % 4 != 0 || ldx % 4 != 0) {
    hipLaunchKernelGGL(linear_small_scalar_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       weight, bias, residual, out, M, N, K, ldx, ldo, ldr, act);
    OCCF_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(linear_small_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     weight, bias, residual, out, M, N, K, ldx, ldo, ldr, act);
  OCCF_LAUNCH_CHECK();
}
