// micro-probe: cost of LDS atomics on gfx950 (float add vs integer add, 32 / 64 bit), distinct addresses per lane
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
  return 0;
}
