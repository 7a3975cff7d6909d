// Per-tensor absolute maximum for the fp16-piece operands (two-product contractions): included by wgrad.hip and
// conv_wino.hip.
#pragma once
#include "occf_common.h"

// max |x| of a [M][N] tensor (row stride ld) as an fp32 BIT PATTERN (non-negative floats order like their bit patterns):
// <= WG_ABSMAX_BLOCKS workgroups leave one partial maximum each, a one-workgroup second kernel reduces them into
// slot[0].  No atomics: the first version issued one atomicMax per wave on ONE address -- 8 192 same-line atomics per
// call serialise at the L2 (r06a: linear_wgrad 11.6 -> 18.8 ms per step).  Feeds the power-of-two scale of the fp16
// weight-gradient operands (occf_f16_scale_bits).
#define WG_ABSMAX_BLOCKS 1024
static __global__ void __launch_bounds__(256) wg_absmax_kernel(const float* __restrict__ x, long M, int N4, long ld,
                                                        uint32_t* __restrict__ partial) {
  __shared__ uint32_t wmax[4];
  const long total = M * N4;
  uint32_t m = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = ld == (long)N4 * 4 ? 0 : i / N4;
    const float4 v = *(const float4*)(ld == (long)N4 * 4 ? x + i * 4 : x + r * ld + (i - r * N4) * 4);
    const uint32_t a = occf_f2u(v.x) & 0x7FFFFFFFu, b = occf_f2u(v.y) & 0x7FFFFFFFu, c = occf_f2u(v.z) & 0x7FFFFFFFu,
                   d = occf_f2u(v.w) & 0x7FFFFFFFu;
    const uint32_t ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
    m = q > m ? q : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    partial[blockIdx.x] = a > b ? a : b;
  }
}
static __global__ void __launch_bounds__(256) wg_absmax_finish_kernel(const uint32_t* __restrict__ partial, int n,
                                                               uint32_t* __restrict__ slot) {
  __shared__ uint32_t wmax[4];
  uint32_t m = 0u;
  for (int i = threadIdx.x; i < n; i += 256) m = partial[i] > m ? partial[i] : m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    slot[0] = a > b ? a : b;
  }
}
// slot: WG_SCALE_SLOT floats -- [0] the maximum, [1] 2^-k (written by the split pass), [16 ...) the partial maxima
static inline void wg_absmax(const float* x, long M, int N, long ld, uint32_t* slot, hipStream_t st) {
  const long total = M * (N / 4);
  long blocks = (total + 256 * 8 - 1) / (256 * 8);
  blocks = blocks < 1 ? 1 : (blocks > WG_ABSMAX_BLOCKS ? WG_ABSMAX_BLOCKS : blocks);
  hipLaunchKernelGGL(wg_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, M, N / 4, ld, slot + 16);
  hipLaunchKernelGGL(wg_absmax_finish_kernel, dim3(1), dim3(256), 0, st, slot + 16, (int)blocks, slot);
}


// the same for a flat array of n floats (any alignment / length): scalar loads
static __global__ void __launch_bounds__(256) wg_absmax_flat_kernel(const float* __restrict__ x, long n,
                                                                    uint32_t* __restrict__ partial) {
  __shared__ uint32_t wmax[4];
  uint32_t m = 0u;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint32_t a = occf_f2u(x[i]) & 0x7FFFFFFFu;
    m = a > m ? a : m;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = (uint32_t)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t a = wmax[0] > wmax[1] ? wmax[0] : wmax[1], b = wmax[2] > wmax[3] ? wmax[2] : wmax[3];
    partial[blockIdx.x] = a > b ? a : b;
  }
}
static inline void wg_absmax_flat(const float* x, long n, uint32_t* slot, hipStream_t st) {
  long blocks = (n + 256 * 16 - 1) / (256 * 16);
  blocks = blocks < 1 ? 1 : (blocks > WG_ABSMAX_BLOCKS ? WG_ABSMAX_BLOCKS : blocks);
  hipLaunchKernelGGL(wg_absmax_flat_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n, slot + 16);
  hipLaunchKernelGGL(wg_absmax_finish_kernel, dim3(1), dim3(256), 0, st, slot + 16, (int)blocks, slot);
}

#define OCCF_ABSMAX_SLOT (16 + WG_ABSMAX_BLOCKS)   // uint32 words of a scale slot: [0] max bits, [1] 2^-k, [16 ...) partials
