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

// one row of the descriptor table (all int64): in, f32, hi, lo, pair0, d0..d4, s0..s4  (15 words; row n = sentinel
// whose pair0 is the total pair count).  Pairs = two consecutive outputs along the LAST dimension (d4 even).
#define PREP_WORDS 15

__global__ void __launch_bounds__(256) prep_weights_kernel(const long* __restrict__ table, int n, long total_pairs) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total_pairs) return;
  int lo_i = 0, hi_i = n;                       // last row with pair0 <= i
  while (hi_i - lo_i > 1) {
    const int mid = (lo_i + hi_i) >> 1;
    if (table[(long)mid * PREP_WORDS + 4] <= i) lo_i = mid; else hi_i = mid;
  }
  const long* d = table + (long)lo_i * PREP_WORDS;
  const float* in = (const float*)d[0];
  float* f32 = (float*)d[1];
  uint32_t* hi = (uint32_t*)d[2];
  uint32_t* lo = (uint32_t*)d[3];
  const long j = i - d[4];
  // index decomposition in 32 bits (a layout has < 2^31 elements; 64-bit divisions were ~500 instructions per thread)
  unsigned e = (unsigned)(2 * j);
  const unsigned d4 = (unsigned)d[9], d3 = (unsigned)d[8], d2 = (unsigned)d[7], d1 = (unsigned)d[6];
  const unsigned i4 = e % d4;
  e /= d4;
  const unsigned i3 = e % d3;
  e /= d3;
  const unsigned i2 = e % d2;
  e /= d2;
  const unsigned i1 = e % d1;
  const unsigned i0 = e / d1;
  const long off = (long)i0 * d[10] + (long)i1 * d[11] + (long)i2 * d[12] + (long)i3 * d[13] + (long)i4 * d[14];
  const float a = in[off], b = in[off + d[14]];
  uint32_t h, l;
  occf_bf16_split2(a, b, h, l);
  hi[j] = h;
  lo[j] = l;
  if (f32) {
    f32[2 * j] = a;
    f32[2 * j + 1] = b;
  }
}

extern "C" int occf_prep_weights(const int64_t* table, int n, long total_pairs, void* stream) {
  if (n <= 0 || total_pairs <= 0) return OCCF_EINVAL;
  hipLaunchKernelGGL(prep_weights_kernel, dim3(occf_cdiv(total_pairs, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const long*)table, n, total_pairs);
  OCCF_LAUNCH_CHECK();
}
