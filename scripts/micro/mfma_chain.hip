// Microbenchmark: how fast does a wave issue v_mfma_f32_32x32x16_bf16 when consecutive MFMAs (a) extend ONE accumulator,
// (b) alternate between two, with the A operand (c) in registers or (d) read from LDS one step ahead -- 8 waves per
// workgroup (two per SIMD), one workgroup per CU.  hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k(float* out, int iters) {
  extern __shared__ unsigned char smem[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 24 * 1024 / 4; i += blockDim.x) ((float*)smem)[i] = 0.001f * i;
  __syncthreads();
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 b[12];
  for (int ks = 0; ks < 12; ++ks)
    for (int e = 0; e < 8; ++e) b[ks][e] = (short)(0x3F80 + lane + ks + e);
  bf16x8 a0;
  for (int e = 0; e < 8; ++e) a0[e] = (short)(0x3F00 + lane + e);
  const unsigned char* Wt = smem + lane * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
      bf16x8 wh = a0, wl = a0;
      if (LDS) {
        wh = *(const bf16x8*)(Wt + ks * 1024);
        wl = *(const bf16x8*)(Wt + 12288 + ks * 1024);
      }
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b[ks], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b[(ks + 1) % 12], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b[ks], acc[t], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS, int WAVES>
static void run(const char* name, float* out) {
  const int iters = 2000 / NACC;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)k<NACC, LDS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS, WAVES>), dim3(256), dim3(WAVES * 64), 64 * 1024, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 256.0 * WAVES * iters * 36.0 * NACC;
  const double flops = mfmas * 2.0 * 32 * 32 * 16;
  printf("%-44s %8.3f ms  %7.1f TF/s  (%.1f cycles per MFMA per SIMD at 2.1 GHz)\n", name, ms, flops / ms / 1e9,
         ms * 1e-3 * 2.1e9 / (mfmas / 1024.0));
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 1024 * 4);
  run<1, false, 8>("1 acc, operands in registers, 8 waves", out);
  run<2, false, 8>("2 acc, operands in registers, 8 waves", out);
  run<1, true, 8>("1 acc, A from LDS, 8 waves", out);
  run<2, true, 8>("2 acc, A from LDS, 8 waves", out);
  run<1, false, 4>("1 acc, registers, 4 waves (one per SIMD)", out);
  run<2, false, 4>("2 acc, registers, 4 waves (one per SIMD)", out);
  run<1, false, 16>("1 acc, registers, 16 waves (four per SIMD)", out);
  run<1, true, 16>("1 acc, A from LDS, 16 waves", out);
  return 0;
}
