// Inference epilogue of the 2-D image branch (ResNet / SECONDFPN on MIOpen):
//     y[p, c] = act( y[p, c] * scale[c] + shift[c] (+ residual[p, c]) )          in place, channels-last rows
// = eval-mode BatchNorm (scale = gamma / sqrt(var + eps), shift = beta - mean * scale) + the Bottleneck's identity add +
// ReLU in ONE pass over the convolution output.  The reference runs them as separate modules
// (mmdet ResNet Bottleneck: bn, relu, bn, relu, bn, +identity, relu = 7 elementwise passes per block; here 3).
// Memory-bound; 16-byte accesses, fp32 or bf16 feature maps (bench.py --image-dtype).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

typedef uint32_t ie_u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ie_lo(uint32_t u) { return occf_u2f(u << 16); }
__device__ __forceinline__ float ie_hi(uint32_t u) { return occf_u2f(u & 0xFFFF0000u); }

template <bool BF16>
__global__ void __launch_bounds__(256) scale_shift_act_kernel(void* __restrict__ y, const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const void* __restrict__ residual, long n_vec, int C,
                                                              int relu) {
  constexpr int E = BF16 ? 8 : 4;                 // elements per 16-byte vector
  const long stride = (long)gridDim.x * blockDim.x;
  const unsigned vpr = (unsigned)(C / E);         // vectors per row (C % E == 0: a vector never straddles a row)
  const long first = (long)blockIdx.x * blockDim.x + threadIdx.x;
  // the launch makes the thread count a multiple of vpr whenever it can (every ResNet / FPN width): a thread then stays
  // on ONE channel group and keeps its scale / shift in registers; otherwise the group is re-derived per vector
  // (r05d: a 64-bit modulo per vector made this pass VALU-bound, 17.6 us per call against 6.5 for MIOpen's BatchNorm)
  const bool fixed = (stride % vpr) == 0;
  float sc[E], sh[E];
  unsigned c0 = (unsigned)(first % vpr) * E;
#pragma unroll
  for (int e = 0; e < E; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
  for (long v = first; v < n_vec; v += stride) {
    if (!fixed) {
      c0 = (unsigned)(v % vpr) * E;
#pragma unroll
      for (int e = 0; e < E; ++e) { sc[e] = scale[c0 + e]; sh[e] = shift[c0 + e]; }
    }
    float x[E], r[E];
    if (BF16) {
      const ie_u4 q = ((const ie_u4*)y)[v];
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[2 * e] = ie_lo(q[e]); x[2 * e + 1] = ie_hi(q[e]); }
      if (residual) {
        const ie_u4 z = ((const ie_u4*)residual)[v];
#pragma unroll
        for (int e = 0; e < 4; ++e) { r[2 * e] = ie_lo(z[e]); r[2 * e + 1] = ie_hi(z[e]); }
      }
    } else {
      const float4 q = ((const float4*)y)[v];
      x[0] = q.x; x[1] = q.y; x[2] = q.z; x[3] = q.w;
      if (residual) {
        const float4 z = ((const float4*)residual)[v];
        r[0] = z.x; r[1] = z.y; r[2] = z.z; r[3] = z.w;
      }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
      float t = fmaf(x[e], sc[e], sh[e]);
      if (residual) t += r[e];
      x[e] = relu ? fmaxf(t, 0.f) : t;
    }
    if (BF16) {
      ie_u4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = occf_bf16_pack2(x[2 * e], x[2 * e + 1]);
      ((ie_u4*)y)[v] = o;
    } else {
      ((float4*)y)[v] = make_float4(x[0], x[1], x[2], x[3]);
    }
  }
}

extern "C" int occf_scale_shift_act(void* y, const float* scale, const float* shift, const void* residual, long rows,
                                    int C, int relu, int bf16, void* stream) {
  const int E = bf16 ? 8 : 4;
  if (rows <= 0 || C <= 0 || C % E != 0) return OCCF_ESHAPE;
  const long n_vec = rows * C / E;
  long blocks = (n_vec + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  const long vpr = C / E;                          // keep blocks * 256 a multiple of the vectors per row when possible
  if (blocks * 256 >= vpr && (blocks * 256) % vpr != 0 && vpr <= 256 * 64) {
    long b2 = blocks;
    while (b2 > 1 && (b2 * 256) % vpr != 0) --b2;
    if ((b2 * 256) % vpr == 0) blocks = b2;
  }
  if (bf16)
    hipLaunchKernelGGL(scale_shift_act_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y, scale,
                       shift, residual, n_vec, C, relu);
  else
    hipLaunchKernelGGL(scale_shift_act_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, y,
                       scale, shift, residual, n_vec, C, relu);
  OCCF_LAUNCH_CHECK();
}
