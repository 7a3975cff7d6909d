/* ORACLE (test infrastructure only -- never linked into the product).
 *
 * Plain-C restatement of the reference's only native hot-path op,
 *   mmdetection3d/mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:20-42 (forward)
 *   mmdetection3d/mmdet3d/ops/bev_pool/src/bev_pool_cuda.cu:61-84 (backward)
 * with the host wrapper's zero-fill (bev_pool.cpp:43, :80).  One (interval, channel) pair
 * per CUDA thread becomes two nested loops; the per-interval accumulation order (k = 0 ..
 * len-1, fp32, no fma) is the reference's.  The real op needs nvcc + a CUDA torch and cannot
 * be built in this image (stated in DESIGN.md), so this restatement is pinned against the
 * golden output of the reference's own Python wrapper (tests/golden/bev_pool.npz).
 */
#include <stdint.h>
#include <string.h>

void bev_pool_ref_forward(int b, int d, int h, int w, int n, int c, int n_intervals, const float* x,
                          const int32_t* geom_feats, const int32_t* interval_starts,
                          const int32_t* interval_lengths, float* out) {
  memset(out, 0, sizeof(float) * (size_t)b * d * h * w * c);
  for (int index = 0; index < n_intervals; ++index) {
    const int start = interval_starts[index];
    const int len = interval_lengths[index];
    const int32_t* g = geom_feats + (size_t)start * 4;
    float* cur_out = out + (size_t)g[3] * d * h * w * c + (size_t)g[2] * h * w * c +
                     (size_t)g[0] * w * c + (size_t)g[1] * c;
    for (int cur_c = 0; cur_c < c; ++cur_c) {
      volatile float psum = 0.f; /* volatile: forbid contraction / reassociation */
      for (int i = 0; i < len; ++i) psum = psum + x[((size_t)start + i) * c + cur_c];
      cur_out[cur_c] = psum;
    }
  }
  (void)n;
}

void bev_pool_ref_backward(int b, int d, int h, int w, int n, int c, int n_intervals,
                           const float* out_grad, const int32_t* geom_feats,
                           const int32_t* interval_starts, const int32_t* interval_lengths,
                           float* x_grad) {
  memset(x_grad, 0, sizeof(float) * (size_t)n * c);
  for (int index = 0; index < n_intervals; ++index) {
    const int start = interval_starts[index];
    const int len = interval_lengths[index];
    const int32_t* g = geom_feats + (size_t)start * 4;
    const float* cur = out_grad + (size_t)g[3] * d * h * w * c + (size_t)g[2] * h * w * c +
                       (size_t)g[0] * w * c + (size_t)g[1] * c;
    for (int i = 0; i < len; ++i)
      for (int cur_c = 0; cur_c < c; ++cur_c) x_grad[((size_t)start + i) * c + cur_c] = cur[cur_c];
  }
  (void)b;
}
