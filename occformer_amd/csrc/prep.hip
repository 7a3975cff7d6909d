// "Prepare weights" in ONE pass per training step: every layout the kernels consume is derived from a
// parameter in the reference's checkpoint layout -- the bf16 (hi, lo) split for the matrix cores, the tap-major
// [Cout, taps*Cin] form of a convolution, W^T of a linear for its data gradient, the tap-flipped / channel-swapped
// [Cin, taps*Cout] form for the stride-1 data gradient, the unflipped one for the strided data gradient.  After an
// optimizer step all of them are stale; rebuilding them one by one cost ~1 200 launches per step (permute + copy +
// split per weight: 4.4 ms of split kernels + ~3 ms of ATen copies, r03c).  Here a table of strided-gather descriptors
// is walked by one launch: out[i0..i4] (contiguous) = in[base + sum_k i_k * stride_k], written as fp32 (optional) and
// as the bf16 pair (hi = rne(x), lo = rne(x - hi)) -- the same split as occf_split_bf16.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// one row of the descriptor table (all int64): in, f32, hi, lo, pair0, d0..d4, s0..s4, mode  (16 words; row n = sentinel
// whose pair0 is the total span).  Pairs = two consecutive outputs along the LAST dimension (d4 even).  A workgroup
// (256 threads) owns a SLOT of 512 pairs; every row's span [pair0, next pair0) is a whole number of slots, so one lookup
// per workgroup finds its row (round 5: the per-THREAD binary search was ~9 dependent table loads per pair).
//   mode 0: slot s of the row = output pairs [512 s, 512 s + 512) in output order (the input's fastest dimension is the
//           output's: 'split', 'tap')
//   mode 1: the output is a transpose of the input -- its last dimension d4 is the input's SLOWEST ('wt', 'dg', 'flip':
//           out [M = d0 d1 d2 d3][N = d4]) -- and a slot is one 32 x 32 tile staged through LDS: rows of the input are
//           read along m (contiguous, or reversed for the flipped taps), the output is written along n.  The strided
//           form read 4 scattered bytes per lane (1 TB/s over the 2.4 GB the pass moves, r04).
#define PREP_WORDS 16
#define PREP_SLOT 512

__device__ __forceinline__ void prep_emit(float a, float b, long j, float* f32, uint32_t* hi, uint32_t* lo) {
  uint32_t h, l;
  occf_bf16_split2(a, b, h, l);
  hi[j] = h;
  lo[j] = l;
  if (f32) {
    f32[2 * j] = a;
    f32[2 * j + 1] = b;
  }
}

__global__ void __launch_bounds__(256) prep_weights_kernel(const long* __restrict__ table, int n) {
  __shared__ int s_row;
  __shared__ float tile[32][33];
  const int tid = threadIdx.x;
  const long slot0 = (long)blockIdx.x * PREP_SLOT;
  if (tid == 0) {
    int lo_i = 0, hi_i = n;                       // last row with pair0 <= slot0
    while (hi_i - lo_i > 1) {
      const int mid = (lo_i + hi_i) >> 1;
      if (table[(long)mid * PREP_WORDS + 4] <= slot0) lo_i = mid; else hi_i = mid;
    }
    s_row = lo_i;
  }
  __syncthreads();
  const long* d = table + (long)s_row * PREP_WORDS;
  const float* in = (const float*)d[0];
  float* f32 = (float*)d[1];
  uint32_t* hi = (uint32_t*)d[2];
  uint32_t* lo = (uint32_t*)d[3];
  const unsigned d0 = (unsigned)d[5], d1 = (unsigned)d[6], d2 = (unsigned)d[7], d3 = (unsigned)d[8], d4 = (unsigned)d[9];
  const long s0 = d[10], s1 = d[11], s2 = d[12], s3 = d[13], s4 = d[14];
  const unsigned slot = (unsigned)((slot0 - d[4]) / PREP_SLOT);
  if (d[15] == 0) {
    const unsigned pairs = (unsigned)(((unsigned long)d0 * d1 * d2 * d3 * d4) >> 1);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned j = slot * PREP_SLOT + tid + k * 256;
      if (j >= pairs) continue;
      // index decomposition in 32 bits (a layout has < 2^31 elements)
      unsigned e = 2 * j;
      const unsigned i4 = e % d4;
      e /= d4;
      const unsigned i3 = e % d3;
      e /= d3;
      const unsigned i2 = e % d2;
      e /= d2;
      const unsigned i1 = e % d1;
      const unsigned i0 = e / d1;
      const long off = (long)i0 * s0 + (long)i1 * s1 + (long)i2 * s2 + (long)i3 * s3 + (long)i4 * s4;
      prep_emit(in[off], in[off + s4], j, f32, hi, lo);
    }
    return;
  }
  // ---- mode 1: 32 (m) x 32 (n) tile of out [M][N]
  const unsigned M = d0 * d1 * d2 * d3, N = d4;
  const unsigned tiles_n = (N + 31) / 32;
  const unsigned m0 = (slot / tiles_n) * 32, n0 = (slot % tiles_n) * 32;
  {
    const unsigned tm = tid & 31, tn = tid >> 5;
    const unsigned m = m0 + tm;
    if (m < M) {
      unsigned e = m;
      const unsigned i3 = e % d3;
      e /= d3;
      const unsigned i2 = e % d2;
      e /= d2;
      const unsigned i1 = e % d1;
      const unsigned i0 = e / d1;
      const long base = (long)i0 * s0 + (long)i1 * s1 + (long)i2 * s2 + (long)i3 * s3;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned nn = n0 + tn + 8 * k;
        if (nn < N) tile[tn + 8 * k][tm] = in[base + (long)nn * s4];
      }
    }
  }
  __syncthreads();
  {
    const unsigned pn = tid & 15, rm = tid >> 4;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned ml = rm + 16 * k, m = m0 + ml, nn = n0 + 2 * pn;
      if (m < M && nn < N) prep_emit(tile[2 * pn][ml], tile[2 * pn + 1][ml], ((long)m * N + nn) >> 1, f32, hi, lo);
    }
  }
}

extern "C" int occf_prep_weights(const int64_t* table, int n, long total_pairs, void* stream) {
  if (n <= 0 || total_pairs <= 0 || total_pairs % PREP_SLOT != 0) return OCCF_EINVAL;
  hipLaunchKernelGGL(prep_weights_kernel, dim3((unsigned)(total_pairs / PREP_SLOT)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)table, n);
  OCCF_LAUNCH_CHECK();
}
