// micro-probe: cost of LDS atomics on gfx950 (float add vs integer add, 32 / 64 bit).
//   part 1: distinct addresses per lane, 4 waves per CU (latency view: ticks per dependent loop iteration)
//   part 2: ds_add_u64 THROUGHPUT per CU vs waves per CU (4 / 8 / 16) and vs same-address sharing inside a wave
//           (1 = all lanes distinct, 4 / 16 = groups of lanes on one address)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>
__global__ void __launch_bounds__(256) probe(float* out, int iters, int stride) {
  __shared__ float f[8192];
  __shared__ unsigned long long u64[4096];
  unsigned* u = (unsigned*)f;
  for (int i = threadIdx.x; i < 8192; i += 256) f[i] = 0.f;
  for (int i = threadIdx.x; i < 4096; i += 256) u64[i] = 0;
  __syncthreads();
  const int a = (threadIdx.x * stride) & 4095;
  long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    const int idx = (a + i * 17) & 4095;
    if (MODE == 0) atomicAdd(&f[idx], 1.0f);
    if (MODE == 1) atomicAdd(&u[idx], 1u);
    if (MODE == 2) atomicAdd(&u64[idx], 1ull);
    if (MODE == 3) f[idx] += 1.0f;          // plain RMW (racy), baseline
  }
  __syncthreads();
  long t1 = clock64();
  if (threadIdx.x == 0) out[blockIdx.x] = (float)(t1 - t0) / iters;
  if (threadIdx.x == 1) out[1024 + blockIdx.x] = f[a] + (float)u64[a];
}
// 8 independent atomics per iteration (no address dependence between them): throughput
__global__ void __launch_bounds__(1024) tput(float* out, int iters, int share) {
  __shared__ unsigned long long u64[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) u64[i] = 0;
  __syncthreads();
  const int a = (threadIdx.x / share) * 3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&u64[(a + i * 17 + j * 1021) & 8191], 1ull);
  }
  __syncthreads();
  if (threadIdx.x == 1) out[blockIdx.x] = (float)u64[a & 8191];
}
int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  float h[8];
  const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "plain rmw"};
  for (int stride : {1, 3, 32}) for (int m = 0; m < 4; ++m) {
    const int iters = 2000;
    if (m == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, d, iters, stride);
    if (m == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, d, iters, stride);
    if (m == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, d, iters, stride);
    if (m == 3) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(256), 0, 0, d, iters, stride);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("stride %2d  %-12s  %.1f clock64 ticks per loop iteration (4 waves / workgroup)\n", stride, names[m], h[0]);
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512, 1024}) for (int share : {1, 4, 16}) {
    const int iters = 4000;
    hipLaunchKernelGGL(tput, dim3(256), dim3(threads), 0, 0, d, 10, share);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(tput, dim3(256), dim3(threads), 0, 0, d, iters, share);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_cu = (double)iters * 8 * (threads / 64);
    printf("ds_add_u64 throughput: %4d threads/CU, %2d lanes per address: %.2f ms, %.1f ns per wave instruction per CU (%.1f cycles at 2.4 GHz)\n",
           threads, share, ms, ms * 1e6 / wave_instr_per_cu, ms * 1e6 / wave_instr_per_cu * 2.4);
  }
  return 0;
}
