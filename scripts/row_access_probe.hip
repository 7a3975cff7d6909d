// micro-probe (r05): what does the "one lane = one token row" access pattern of the register-direct GEMM kernels
// (csrc/gemm_stream.h, mlp_chain.hip, swin_attn_fused.hip: a wave instruction touches 32 rows x 2 x 16 B) cost against
// fully coalesced 16-byte accesses (a wave instruction = 1 KB contiguous)?  M x K fp32 matrix in, same out; 256
// persistent workgroups x 8 waves, a wave walks 32-row tiles.
//   mode 0: coalesced loads, coalesced stores            (the copy roof of this launch shape)
//   mode 1: ROW loads (lane (li, lk): row li, floats 16 ks + 8 lk + {0..3 | 4..7}), coalesced stores
//   mode 2: coalesced loads, ROW stores (lane: row li, floats 32 j + 8 g + 4 lk .. -- the MFMA C layout)
//   mode 3: ROW loads, ROW stores                        (what gemm_stream does, without the arithmetic)
//   mode 4: ROW loads with 32 contiguous bytes per lane PAIR per instruction (lk picks the 16-byte half), coalesced stores
//   mode 5: coalesced loads, DWORD stores that cover whole lines (lane = channel, the token-major MFMA C layout)
//   mode 6: ROW loads, the C-layout tile transposed through a wave-private LDS region (4.6 KB), float4 whole-line stores
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/row_access_probe.hip -o /tmp/rap && /tmp/rap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int KS, int MODE>
__global__ void __launch_bounds__(512) probe(const float* __restrict__ in, float* __restrict__ out, long M, int streams) {
  constexpr int K = 16 * KS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long n_tiles = (M + 31) / 32;
  for (long t = (long)blockIdx.x * 8 + wave; t < n_tiles; t += (long)streams * 8) {
    const long row0 = t * 32;
    float4 v[2 * KS];
    if (MODE == 0 || MODE == 2 || MODE == 5) {
      const float4* src = (const float4*)(in + row0 * K);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) v[i] = src[i * 64 + lane];
    } else if (MODE == 4) {
      const float* xr = in + (row0 + li) * K + lk * 4;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        v[2 * ks] = *(const float4*)(xr + ks * 16);
        v[2 * ks + 1] = *(const float4*)(xr + ks * 16 + 8);
      }
    } else {
      const float* xr = in + (row0 + li) * K + lk * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        v[2 * ks] = *(const float4*)(xr + ks * 16);
        v[2 * ks + 1] = *(const float4*)(xr + ks * 16 + 4);
      }
    }
    if (MODE == 5) {
      // value v[i] component c -> token (row) 2 * (4 i + c) / ... : any bijection serves; 16 dword stores per 32 channels
      float* cb = out + row0 * K + (lane & 31);
      const int lk4 = (lane >> 5) * 4;
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) {
        const float f[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = (i & 3) * 4 + c, j = i >> 2;               // n-tile j, accumulator register r
          cb[(long)((r & 3) + 8 * (r >> 2) + lk4) * K + j * 32] = f[c];
        }
      }
    } else if (MODE == 6) {
      __shared__ __attribute__((aligned(16))) float stage[8][32 * 36];
      float* st = stage[wave];
#pragma unroll
      for (int j = 0; j < KS / 2; ++j) {                           // 32-channel tiles
#pragma unroll
        for (int g = 0; g < 4; ++g) *(float4*)(st + li * 36 + g * 8 + lk * 4) = v[j * 4 + g];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = i * 8 + (lane >> 3), ch = (lane & 7) * 4;
          const float4 o = *(const float4*)(st + row * 36 + ch);
          *(float4*)(out + (row0 + row) * K + j * 32 + ch) = o;
        }
      }
    } else if (MODE == 0 || MODE == 1 || MODE == 4) {
      float4* dst = (float4*)(out + row0 * K);
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) dst[i * 64 + lane] = v[i];
    } else {
      float* cr = out + (row0 + li) * K + lk * 4;
#pragma unroll
      for (int i = 0; i < 2 * KS; ++i) *(float4*)(cr + (i >> 2) * 32 + (i & 3) * 8) = v[i];
    }
  }
}
template <int KS>
void run(long M) {
  const int K = 16 * KS;
  float *in, *out;
  const int NB = 3;                                    // rotate buffers: nothing is served from the MALL
  hipMalloc(&in, (size_t)M * K * 4 * NB);
  hipMalloc(&out, (size_t)M * K * 4 * NB);
  hipMemset(in, 0, (size_t)M * K * 4 * NB);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 7; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 8; ++rep) {
      const float* a = in + (size_t)(rep % NB) * M * K;
      float* b = out + (size_t)(rep % NB) * M * K;
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL((probe<KS, 0>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        case 1: hipLaunchKernelGGL((probe<KS, 1>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        case 2: hipLaunchKernelGGL((probe<KS, 2>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        case 3: hipLaunchKernelGGL((probe<KS, 3>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        case 5: hipLaunchKernelGGL((probe<KS, 5>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        case 6: hipLaunchKernelGGL((probe<KS, 6>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
        default: hipLaunchKernelGGL((probe<KS, 4>), dim3(256), dim3(512), 0, 0, a, b, M, 256); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep >= 2 && ms < best) best = ms;
    }
    printf("M=%ld K=%d mode %d: %8.1f us  %7.0f GB/s (read + write)\n", M, K, mode, best * 1e3,
           2.0 * M * K * 4 / (best * 1e-3) / 1e9);
  }
  hipFree(in);
  hipFree(out);
}
int main() {
  run<8>(680000);
  run<12>(640000);
  run<12>(91264);
  return 0;
}
