// Hungarian matching of the occupancy head ("GPU Hungarian to kill the
// host syncs per iteration").
//
// Reference: projects/mmdet3d_plugin/occformer/mask2former/assigners/mask_hungarian_assigner.py:104-126
//   cost = cls_cost + mask_cost + dice_cost           [num_queries, num_gt] on the GPU
//   matched_row_inds, matched_col_inds = scipy.optimize.linear_sum_assignment(cost.detach().cpu())
//   assigned_gt_inds[:] = 0;  assigned_gt_inds[matched_row_inds] = matched_col_inds + 1
// scipy 1.x implements linear_sum_assignment as the rectangular shortest-augmenting-path algorithm of
// Crouse (2016, "On implementing 2D rectangular assignment algorithms"): transpose so that rows <=
// columns, then for every row grow a shortest-path tree over the columns (Dijkstra on reduced costs
// c[i][j] - u[i] - v[j]) until an unassigned column is reached, update the duals and flip the path.
// This file restates that algorithm for the GPU: one 64-lane wave per problem, lanes own the columns
// (strided), the arg-min of every Dijkstra step is a wave reduction, all arithmetic in double like scipy.
//
// Ties between equal path costs: scipy takes the candidate found first in its internal visiting order and
// replaces it only by an equally cheap UNASSIGNED column; here: unassigned first, then the lowest column
// index.  Both rules return an optimal assignment; they can only differ when the optimum is not unique.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define HG_MAX 1024          // max(num_queries, num_gt) supported by the LDS working set

struct HgBest {
  double val;
  int key;                   // (assigned ? 1 : 0) << 20 | column  -- smaller is better
};

__device__ __forceinline__ HgBest hg_min(HgBest a, HgBest b) {
  if (b.val < a.val || (b.val == a.val && b.key < a.key)) return b;
  return a;
}

__global__ void __launch_bounds__(64) hungarian_kernel(const float* __restrict__ cost, int* __restrict__ match_gt,
                                                       int* __restrict__ assigned, int Q, int G) {
  __shared__ double u[HG_MAX], v[HG_MAX], sp[HG_MAX];     // row / column duals, shortest path cost per column
  __shared__ int path[HG_MAX], col4row[HG_MAX], row4col[HG_MAX];
  __shared__ unsigned char SR[HG_MAX], SC[HG_MAX];
  __shared__ int s_i, s_sink;
  __shared__ double s_min;
  const int lane = threadIdx.x;
  const float* C = cost + (long)blockIdx.x * Q * G;
  const bool tr = G > Q;                  // rows = the smaller side (scipy transposes the same way)
  const int n = tr ? Q : G, m = tr ? G : Q;
  // cost of (row i, column j) in the caller's [Q, G] layout
  auto cst = [&](int i, int j) -> double { return (double)(tr ? C[(long)i * G + j] : C[(long)j * G + i]); };
  const double INF = __builtin_huge_val();

  for (int j = lane; j < m; j += 64) {
    v[j] = 0.0;
    row4col[j] = -1;
  }
  for (int i = lane; i < n; i += 64) {
    u[i] = 0.0;
    col4row[i] = -1;
  }
  __syncthreads();

  for (int cur = 0; cur < n; ++cur) {
    for (int j = lane; j < m; j += 64) {
      sp[j] = INF;
      SC[j] = 0;
    }
    for (int i = lane; i < n; i += 64) SR[i] = 0;
    if (lane == 0) {
      s_i = cur;
      s_sink = -1;
      s_min = 0.0;
    }
    __syncthreads();
    // ---- shortest augmenting path from row `cur`
    while (true) {
      const int i = s_i;
      const double minv = s_min, ui = u[i];
      if (lane == 0) SR[i] = 1;
      HgBest best{INF, 0x7fffffff};
      for (int j = lane; j < m; j += 64) {
        if (SC[j]) continue;
        const double r = minv + cst(i, j) - ui - v[j];
        double s = sp[j];
        if (r < s) {
          path[j] = i;
          sp[j] = r;
          s = r;
        }
        best = hg_min(best, HgBest{s, ((row4col[j] != -1) << 20) | j});
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        HgBest other;
        other.val = __shfl_xor(best.val, o);
        other.key = __shfl_xor(best.key, o);
        best = hg_min(best, other);
      }
      __syncthreads();                         // every lane has read s_i / s_min of this step
      if (lane == 0) {
        if (best.val == INF) {                 // infeasible (a row of +inf): leave the row unassigned
          s_sink = -2;
        } else {
          const int j = best.key & 0xFFFFF;
          s_min = best.val;
          SC[j] = 1;
          if (row4col[j] == -1) s_sink = j;
          else s_i = row4col[j];
        }
      }
      __syncthreads();
      if (s_sink != -1) break;
    }
    const int sink = s_sink;
    if (sink >= 0) {
      // ---- dual update (before the assignment changes: col4row of the rows in SR is still the old one)
      const double minv = s_min;
      for (int i = lane; i < n; i += 64) {
        if (i == cur) u[i] += minv;
        else if (SR[i]) u[i] += minv - sp[col4row[i]];
      }
      for (int j = lane; j < m; j += 64)
        if (SC[j]) v[j] -= minv - sp[j];
      __syncthreads();
      // ---- flip the path
      if (lane == 0) {
        int j = sink;
        while (true) {
          const int i = path[j];
          row4col[j] = i;
          const int t = col4row[i];
          col4row[i] = j;
          j = t;
          if (i == cur) break;
        }
      }
    }
    __syncthreads();
  }

  // ---- results in the reference's terms
  int* mg = match_gt + (long)blockIdx.x * G;
  int* as = assigned + (long)blockIdx.x * Q;
  if (!tr) {          // rows = GTs, columns = queries
    for (int g = lane; g < G; g += 64) mg[g] = col4row[g];
    for (int q = lane; q < Q; q += 64) as[q] = row4col[q] + 1;       // -1 -> 0 = background
  } else {            // rows = queries, columns = GTs
    for (int g = lane; g < G; g += 64) mg[g] = row4col[g];
    for (int q = lane; q < Q; q += 64) as[q] = col4row[q] + 1;
  }
}

extern "C" int occf_hungarian_fwd(const float* cost, int* match_gt, int* assigned_gt, int P, int Q, int G,
                                  void* stream) {
  if (P < 0 || Q <= 0 || G <= 0 || Q > HG_MAX || G > HG_MAX) return OCCF_ESHAPE;
  if (P == 0) return 0;
  hipLaunchKernelGGL(hungarian_kernel, dim3(P), dim3(64), 0, (hipStream_t)stream, cost, match_gt, assigned_gt, Q, G);
  OCCF_LAUNCH_CHECK();
}
