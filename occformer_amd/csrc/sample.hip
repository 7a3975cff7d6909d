// Training-time sampling kernels of the occupancy head:
//   * point_sample_3d: trilinear gather of per-query / per-GT volumes at normalised points
//   * class-guided sampling WITHOUT replacement over ~2 M voxels (torch.multinomial semantics):
//     exponential-race keys  key_i = w_i / e_i,  e_i ~ Exp(1)  and a radix-select of the k largest
//     keys, compacted with wavefront ballots / prefix popcounts.
//
// Reference: projects/mmdet3d_plugin/occformer/mask2former/base/mmdet_utils.py
//   point_sample_3d :21-47 (F.grid_sample wrapper), sample_valid_coords_with_frequencies :91-108,
//   batch_sample_valid_coords_with_frequencies :110-136 (torch.multinomial(w, k, replacement=False);
//   ATen implements that as topk(w / q, k), q ~ Exp(1)), get_uncertain_point_coords_3d_with_frequency
//   :179-246 (torch.topk of -|logit|).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// ---------------------------------------------------------------------------------------
// vol[N, C, X, Y, Z]; pts[N or 1, P, 3] in [0, 1], last dim ordered like grid_sample's grid for a
// [.., X, Y, Z] tensor: pts[..,0] -> Z, pts[..,1] -> Y, pts[..,2] -> X.  out[N, C, P].
__global__ void __launch_bounds__(256) point_sample_3d_kernel(const float* __restrict__ vol,
                                                              const float* __restrict__ pts,
                                                              float* __restrict__ out, int N, int C, int X, int Y,
                                                              int Z, long P, int shared_pts, int align_corners,
                                                              int border, int cgroups,
                                                              const long* __restrict__ rows) {
  // thread = (n, channel group, point): a thread per point alone left 12 544-point calls with 49 workgroups
  // walking 100 channels x 8 dependent gathers each (127 us per call)
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)N * cgroups * P) return;
  const long pi = gid % P;
  const int cg = (int)((gid / P) % cgroups);
  const int n = (int)(gid / (P * cgroups));
  const float* pt = pts + ((shared_pts ? 0 : (long)n * P) + pi) * 3;
  const int dims[3] = {Z, Y, X};
  int i0[3], i1[3];
  float t[3];
  bool ok0[3], ok1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = pt[a] * 2.0f - 1.0f;
    float f = align_corners ? (g + 1.f) * 0.5f * (float)(dims[a] - 1) : ((g + 1.f) * (float)dims[a] - 1.f) * 0.5f;
    if (border) f = fminf(fmaxf(f, 0.f), (float)(dims[a] - 1));
    const float fl = floorf(f);
    i0[a] = (int)fl;
    i1[a] = i0[a] + 1;
    t[a] = f - fl;
    ok0[a] = i0[a] >= 0 && i0[a] < dims[a];
    ok1[a] = i1[a] >= 0 && i1[a] < dims[a];
  }
  const long V = (long)X * Y * Z;
  for (int c = cg; c < C; c += cgroups) {
    const float* v = vol + ((rows ? rows[n] : (long)n) * C + c) * V;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int bz = k & 1, by = (k >> 1) & 1, bx = k >> 2;
      const bool ok = (bz ? ok1[0] : ok0[0]) && (by ? ok1[1] : ok0[1]) && (bx ? ok1[2] : ok0[2]);
      if (!ok) continue;
      const int zz = bz ? i1[0] : i0[0], yy = by ? i1[1] : i0[1], xx = bx ? i1[2] : i0[2];
      const float w = (bz ? t[0] : 1.f - t[0]) * (by ? t[1] : 1.f - t[1]) * (bx ? t[2] : 1.f - t[2]);
      acc = fmaf(w, v[((long)xx * Y + yy) * Z + zz], acc);
    }
    out[((long)n * C + c) * P + pi] = acc;
  }
}

extern "C" int occf_point_sample_3d_fwd(const float* vol, const float* pts, float* out, int N, int C, int X, int Y,
                                        int Z, long P, int shared_pts, int align_corners, int border_padding,
                                        void* stream) {
  if (N <= 0 || C <= 0 || X <= 0 || Y <= 0 || Z <= 0 || P < 0) return OCCF_EINVAL;
  if (P == 0) return 0;
  // one thread per point walks ALL channels (cgroups = 1): the threads of a launch then move through the channel
  // volumes together and one 2.5 MB volume at a time stays L2-resident; spreading the channels over more threads was
  // measured slower (160 vs 128 us average per call at the 200-grid: six volumes in flight thrash the 4 MB L2s)
  const int cgroups = 1;
  hipLaunchKernelGGL(point_sample_3d_kernel, dim3(occf_cdiv((long)N * cgroups * P, 256)), dim3(256), 0,
                     (hipStream_t)stream, vol, pts, out, N, C, X, Y, Z, P, shared_pts, align_corners, border_padding,
                     cgroups, (const long*)nullptr);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_point_sample_3d_rows_fwd(const float* vol, const int64_t* rows, const float* pts, float* out, int N,
                                             int X, int Y, int Z, long P, int shared_pts, int align_corners,
                                             int border_padding, void* stream) {
  if (N <= 0 || X <= 0 || Y <= 0 || Z <= 0 || P < 0 || !rows) return OCCF_EINVAL;
  if (P == 0) return 0;
  hipLaunchKernelGGL(point_sample_3d_kernel, dim3(occf_cdiv((long)N * P, 256)), dim3(256), 0, (hipStream_t)stream, vol,
                     pts, out, N, 1, X, Y, Z, P, shared_pts, align_corners, border_padding, 1, (const long*)rows);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// channels-last variant: tok[V, ld] rows of C floats; thread = (point, 4 channels); out[P, C]
__global__ void __launch_bounds__(256) point_sample_tokens_kernel(const float* __restrict__ tok,
                                                                  const float* __restrict__ pts, float* __restrict__ out,
                                                                  int X, int Y, int Z, int C, long ld, long P,
                                                                  int align_corners, int border) {
  const int cq = C >> 2;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= P * cq) return;
  const long pi = gid / cq;
  const int c4 = (int)(gid - pi * cq) * 4;
  const float* pt = pts + pi * 3;
  const int dims[3] = {Z, Y, X};
  int i0[3], i1[3];
  float t[3];
  bool ok0[3], ok1[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float g = pt[a] * 2.0f - 1.0f;
    float f = align_corners ? (g + 1.f) * 0.5f * (float)(dims[a] - 1) : ((g + 1.f) * (float)dims[a] - 1.f) * 0.5f;
    if (border) f = fminf(fmaxf(f, 0.f), (float)(dims[a] - 1));
    const float fl = floorf(f);
    i0[a] = (int)fl;
    i1[a] = i0[a] + 1;
    t[a] = f - fl;
    ok0[a] = i0[a] >= 0 && i0[a] < dims[a];
    ok1[a] = i1[a] >= 0 && i1[a] < dims[a];
  }
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int bz = k & 1, by = (k >> 1) & 1, bx = k >> 2;
    const bool ok = (bz ? ok1[0] : ok0[0]) && (by ? ok1[1] : ok0[1]) && (bx ? ok1[2] : ok0[2]);
    const int zz = occf_clampi(bz ? i1[0] : i0[0], Z - 1), yy = occf_clampi(by ? i1[1] : i0[1], Y - 1),
              xx = occf_clampi(bx ? i1[2] : i0[2], X - 1);
    const float w = ok ? (bz ? t[0] : 1.f - t[0]) * (by ? t[1] : 1.f - t[1]) * (bx ? t[2] : 1.f - t[2]) : 0.f;
    const float4 v = *(const float4*)(tok + (((long)xx * Y + yy) * Z + zz) * ld + c4);
    // (same accumulation order over the corners as point_sample_3d_kernel)
    acc.x = fmaf(w, v.x, acc.x);
    acc.y = fmaf(w, v.y, acc.y);
    acc.z = fmaf(w, v.z, acc.z);
    acc.w = fmaf(w, v.w, acc.w);
  }
  *(float4*)(out + pi * C + c4) = acc;
}

extern "C" int occf_point_sample_tokens_fwd(const float* tok, const float* pts, float* out, int X, int Y, int Z, int C,
                                            long ld, long P, int align_corners, int border_padding, void* stream) {
  if (X <= 0 || Y <= 0 || Z <= 0 || C <= 0 || C % 4 || ld < C || ld % 4 || P < 0) return OCCF_EINVAL;
  if (P == 0) return 0;
  hipLaunchKernelGGL(point_sample_tokens_kernel, dim3(occf_cdiv(P * (C / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     tok, pts, out, X, Y, Z, C, ld, P, align_corners, border_padding);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Weighted sampling without replacement: k indices per row out of V with probability ~ weights.
// keys[r, i] = w[i] / (-log u[r, i])  (u uniform in (0,1]; w >= 0; w == 0 -> key 0, never drawn
// before a positive weight).  Selection = the k largest keys per row:
//   1. key kernel (one pass, also the level-0 histogram of the top 11 key bits),
//   2. two more histogram passes (11 + 10 bits) narrow the k-th largest key to its exact value,
//   3. compaction: every element above the threshold, plus elements equal to it until k slots
//      are filled; slots are allocated per wave from a ballot + prefix popcount (one atomic per
//      wave), so the output is an unordered set (torch returns it sorted by key; the consumers
//      -- gather + top-k by uncertainty -- only use it as a set).
#define TK_B0 11
#define TK_B1 11
#define TK_B2 10

__device__ __forceinline__ unsigned tk_bits(float f) {
#ifdef OCCF_EMU
  unsigned u;
  memcpy(&u, &f, 4);
  return u;
#else
  return __float_as_uint(f);
#endif
}
__device__ __forceinline__ float tk_from_bits(unsigned u) {
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(u);
#endif
}

__global__ void __launch_bounds__(256) tk_keys_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                      float* __restrict__ keys, unsigned* __restrict__ hist0,
                                                      long V, int w_shared, int mode) {
  __shared__ unsigned h[1 << TK_B0];
  const int r = blockIdx.y;
  for (int i = threadIdx.x; i < (1 << TK_B0); i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
    const float wi = w[(w_shared ? 0 : (long)r * V) + i];
    float key;
    if (mode == 0) {
      const float e = -logf(u[(long)r * V + i]);
      key = wi > 0.f ? wi / fmaxf(e, 1e-38f) : 0.f;
    } else if (mode == 2) {  // the noise already is Exp(1) (ATen's multinomial: key = w / q)
      key = wi > 0.f ? wi / u[(long)r * V + i] : 0.f;
    } else {  // smallest |v| first: order-reversing bit pattern of |v|
      key = tk_from_bits(0x7F800000u - (tk_bits(wi) & 0x7FFFFFFFu));
    }
    keys[(long)r * V + i] = key;
    atomicAdd(&h[tk_bits(key) >> (32 - TK_B0)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << TK_B0); i += blockDim.x)
    if (h[i]) atomicAdd(&hist0[(long)r * (1 << TK_B0) + i], h[i]);
}

// histogram of the next `bits` bits among the keys whose already-fixed prefix equals state.prefix
// state[r] = {prefix, prefix_bits, remaining_k, _}
__global__ void __launch_bounds__(256) tk_hist_kernel(const float* __restrict__ keys, const unsigned* __restrict__ state,
                                                      unsigned* __restrict__ hist, long V, int bits) {
  __shared__ unsigned h[1 << TK_B0];
  const int r = blockIdx.y;
  const unsigned prefix = state[r * 4 + 0], pbits = state[r * 4 + 1];
  for (int i = threadIdx.x; i < (1 << bits); i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (long)gridDim.x * blockDim.x) {
    const unsigned b = tk_bits(keys[(long)r * V + i]);
    if ((b >> (32 - pbits)) == prefix) atomicAdd(&h[(b >> (32 - pbits - bits)) & ((1u << bits) - 1)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << bits); i += blockDim.x)
    if (h[i]) atomicAdd(&hist[(long)r * (1 << TK_B0) + i], h[i]);
}

// one workgroup per row: walk the histogram from the largest bin down, find the bin that holds the
// remaining_k-th key; serial scan by wave 0 with a wavefront prefix sum
__global__ void __launch_bounds__(64) tk_scan_kernel(unsigned* __restrict__ hist, unsigned* __restrict__ state,
                                                     int bits, int first, unsigned k) {
  const int r = blockIdx.x, lane = threadIdx.x;
  unsigned* h = hist + (long)r * (1 << TK_B0);
  unsigned remaining = first ? k : state[r * 4 + 2];
  const unsigned prefix = first ? 0u : state[r * 4 + 0], pbits = first ? 0u : state[r * 4 + 1];
  const int nb = 1 << bits;
  int found = -1;
  unsigned above = 0;
  for (int base = nb - 64; base >= 0 && found < 0; base -= 64) {
    const int bin = base + 63 - lane;                      // lane 0 = largest bin of this group
    unsigned c = h[bin], incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {                     // inclusive wavefront scan
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const unsigned total = __shfl(incl, 63);
    const bool hit = above + incl >= remaining && above + incl - c < remaining;
    const unsigned long long m = __ballot(hit);
    if (m) {
      const int l = __ffsll((long long)m) - 1;
      found = base + 63 - l;
      above += __shfl(incl - c, l);
    } else {
      above += total;
    }
  }
  for (int i = lane; i < nb; i += 64) h[i] = 0;            // ready for the next level
  if (lane == 0) {
    state[r * 4 + 0] = (prefix << bits) | (unsigned)(found < 0 ? 0 : found);
    state[r * 4 + 1] = pbits + bits;
    state[r * 4 + 2] = remaining - above;                  // still to take inside the chosen bin
    state[r * 4 + 3] = 0;                                  // compaction counters
  }
}

// One returning atomic per WORKGROUP and counter: a workgroup owns a contiguous chunk, counts its hits first (second
// read of the keys comes from the L2) and places them behind one reservation.  One atomic per wave iteration
// (~10 000 same-address returning atomics per row at V = 640 000) serialised at the L2: 420 us per call.
__global__ void __launch_bounds__(256) tk_compact_kernel(const float* __restrict__ keys,
                                                         unsigned* __restrict__ state, int* __restrict__ counters,
                                                         int64_t* __restrict__ out, long V, unsigned k) {
  __shared__ int wcnt[2][4], wbase[2][4];
  const int r = blockIdx.y;
  const unsigned thr = state[r * 4 + 0];                   // exact bit pattern of the k-th largest key
  const unsigned ties = state[r * 4 + 2];                  // how many keys == thr to take
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long per = (V + gridDim.x - 1) / gridDim.x;
  per = (per + 255) / 256 * 256;
  const long c0 = (long)blockIdx.x * per;
  const long c1 = c0 + per < V ? c0 + per : V;
  const float* row = keys + (long)r * V;
  int ng = 0, ne = 0;
  for (long i0 = c0 + wave * 64; i0 < c1; i0 += 256) {
    const long i = i0 + lane;
    const unsigned b = i < c1 ? tk_bits(row[i]) : 0u;
    ng += __popcll(__ballot(i < c1 && b > thr));
    ne += __popcll(__ballot(i < c1 && b == thr));
  }
  if (lane == 0) {
    wcnt[0][wave] = ng;
    wcnt[1][wave] = ne;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int c = threadIdx.x;
    const int tot = wcnt[c][0] + wcnt[c][1] + wcnt[c][2] + wcnt[c][3];
    int base = tot ? atomicAdd(&counters[r * 2 + c], tot) : 0;
    for (int w = 0; w < 4; ++w) {
      wbase[c][w] = base;
      base += wcnt[c][w];
    }
  }
  __syncthreads();
  int og = wbase[0][wave], oe = wbase[1][wave];
  for (long i0 = c0 + wave * 64; i0 < c1; i0 += 256) {
    const long i = i0 + lane;
    const unsigned b = i < c1 ? tk_bits(row[i]) : 0u;
    const bool gt = i < c1 && b > thr;
    const bool eq = i < c1 && b == thr;
    const unsigned long long mg = __ballot(gt), me = __ballot(eq);
    if (gt) out[(long)r * k + og + __popcll(mg & ((1ull << lane) - 1))] = i;
    og += __popcll(mg);
    const int slot = oe + __popcll(me & ((1ull << lane) - 1));
    if (eq && slot < (int)ties) out[(long)r * k + (k - ties) + slot] = i;
    oe += __popcll(me);
  }
}

// ---- V <= TK_SMALL_V: ONE workgroup per row does the whole selection in LDS (three histogram passes + compaction).
// The pipeline above is 8 launches per call and its compaction reserves output slots with same-address returning
// atomics at the L2 (one per workgroup and counter): 244 us per call for 13 rows x 37 632 importance-sampling logits,
// ten calls per training step.  Here the keys are recomputed from the inputs in every pass (they come from the L2) and
// slots are reserved with LDS atomics.
#define TK_SMALL_V 65536
__device__ __forceinline__ unsigned tk_key_bits(const float* __restrict__ w, const float* __restrict__ u, long row_off,
                                                long i, int w_shared, int mode) {
  const float wi = w[(w_shared ? 0 : row_off) + i];
  float key;
  if (mode == 0) {
    const float e = -logf(u[row_off + i]);
    key = wi > 0.f ? wi / fmaxf(e, 1e-38f) : 0.f;
  } else if (mode == 2) {
    key = wi > 0.f ? wi / u[row_off + i] : 0.f;
  } else {
    key = tk_from_bits(0x7F800000u - (tk_bits(wi) & 0x7FFFFFFFu));
  }
  return tk_bits(key);
}

__global__ void __launch_bounds__(1024) tk_small_kernel(const float* __restrict__ w, const float* __restrict__ u,
                                                        int64_t* __restrict__ out, long V, unsigned k, int w_shared,
                                                        int mode) {
  __shared__ unsigned h[1 << TK_B0];
  __shared__ unsigned st[3];
  __shared__ int cnt[2];
  const int r = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const long row_off = (long)r * V;
  unsigned prefix = 0u, pbits = 0u, remaining = k;
  const int pass_bits[3] = {TK_B0, TK_B1, TK_B2};
  for (int pass = 0; pass < 3; ++pass) {
    const int bits = pass_bits[pass], nb = 1 << bits;
    for (int i = tid; i < nb; i += 1024) h[i] = 0u;
    __syncthreads();
    for (long i = tid; i < V; i += 1024) {
      const unsigned b = tk_key_bits(w, u, row_off, i, w_shared, mode);
      if (pbits == 0u || (b >> (32 - pbits)) == prefix) atomicAdd(&h[(b >> (32 - pbits - bits)) & ((1u << bits) - 1)], 1u);
    }
    __syncthreads();
    if (tid < 64) {                                  // the scan of tk_scan_kernel, on the LDS histogram
      int found = -1;
      unsigned above = 0;
      for (int base = nb - 64; base >= 0 && found < 0; base -= 64) {
        const int bin = base + 63 - lane;
        unsigned c = h[bin], incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const unsigned t = __shfl_up(incl, o);
          if (lane >= o) incl += t;
        }
        const unsigned total = __shfl(incl, 63);
        const bool hit = above + incl >= remaining && above + incl - c < remaining;
        const unsigned long long m = __ballot(hit);
        if (m) {
          const int l = __ffsll((long long)m) - 1;
          found = base + 63 - l;
          above += __shfl(incl - c, l);
        } else {
          above += total;
        }
      }
      if (lane == 0) {
        st[0] = (prefix << bits) | (unsigned)(found < 0 ? 0 : found);
        st[1] = pbits + bits;
        st[2] = remaining - above;
        cnt[0] = cnt[1] = 0;
      }
    }
    __syncthreads();
    prefix = st[0];
    pbits = st[1];
    remaining = st[2];
    __syncthreads();
  }
  const unsigned thr = prefix, ties = remaining;
  for (long i0 = (long)(tid & ~63); i0 < V; i0 += 1024) {
    const long i = i0 + lane;
    const unsigned b = i < V ? tk_key_bits(w, u, row_off, i, w_shared, mode) : 0u;
    const bool gt = i < V && b > thr;
    const bool eq = i < V && b == thr;
    const unsigned long long mg = __ballot(gt), me = __ballot(eq);
    int bg = 0, be = 0;
    if (lane == 0) {
      if (mg) bg = atomicAdd(&cnt[0], __popcll(mg));
      if (me) be = atomicAdd(&cnt[1], __popcll(me));
    }
    bg = __shfl(bg, 0);
    be = __shfl(be, 0);
    if (gt) out[(long)r * k + bg + __popcll(mg & ((1ull << lane) - 1))] = i;
    const int slot = be + __popcll(me & ((1ull << lane) - 1));
    if (eq && slot < (int)ties) out[(long)r * k + (k - ties) + slot] = i;
  }
}

extern "C" long occf_sample_wor_workspace(int R, long V) {
  // keys [R*V] floats + hist [R * 2^11] + state [R*4] + counters [R*2]   (all 4-byte words)
  return (long)R * V + (long)R * (1 << TK_B0) + (long)R * 4 + (long)R * 2;
}

static int tk_select(const float* weights, const float* uniforms, int64_t* out_indices, float* workspace, int R,
                     long V, long k, int weights_shared, int mode, void* stream) {
  if (R <= 0 || V <= 0 || k <= 0 || k > V || V >= 2147483647L) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  static const int small_env = [] {
    const char* e = getenv("OCCF_TK_SMALL");               // 0: the multi-kernel pipeline for every size
    return e ? atoi(e) : 1;
  }();
  if (small_env && V <= TK_SMALL_V) {
    hipLaunchKernelGGL(tk_small_kernel, dim3(R), dim3(1024), 0, st, weights, uniforms, out_indices, V, (unsigned)k,
                       weights_shared, mode);
    OCCF_LAUNCH_CHECK();
  }
  float* keys = workspace;
  unsigned* hist = (unsigned*)(workspace + (long)R * V);
  unsigned* state = hist + (long)R * (1 << TK_B0);
  int* counters = (int*)(state + (long)R * 4);
  const size_t zero_bytes = sizeof(unsigned) * ((size_t)R * (1 << TK_B0) + (size_t)R * 4 + (size_t)R * 2);
#ifndef OCCF_EMU
  hipError_t e = hipMemsetAsync(hist, 0, zero_bytes, st);
  if (e != hipSuccess) return (int)e;
#else
  memset(hist, 0, zero_bytes);
#endif
  const dim3 grid((unsigned)(occf_cdiv(V, 256) < 1024 ? occf_cdiv(V, 256) : 1024), R);
  hipLaunchKernelGGL(tk_keys_kernel, grid, dim3(256), 0, st, weights, uniforms, keys, hist, V, weights_shared, mode);
  hipLaunchKernelGGL(tk_scan_kernel, dim3(R), dim3(64), 0, st, hist, state, TK_B0, 1, (unsigned)k);
  hipLaunchKernelGGL(tk_hist_kernel, grid, dim3(256), 0, st, keys, state, hist, V, TK_B1);
  hipLaunchKernelGGL(tk_scan_kernel, dim3(R), dim3(64), 0, st, hist, state, TK_B1, 0, (unsigned)k);
  hipLaunchKernelGGL(tk_hist_kernel, grid, dim3(256), 0, st, keys, state, hist, V, TK_B2);
  hipLaunchKernelGGL(tk_scan_kernel, dim3(R), dim3(64), 0, st, hist, state, TK_B2, 0, (unsigned)k);
  // compaction: <= 64 workgroups per row (one returning same-address atomic per workgroup and counter: 588 of them per
  // row serialised at the L2 -- 106 us per call for 13 rows x 150 528 importance-sampling candidates)
  const dim3 cgrid((unsigned)(grid.x < 64u ? grid.x : 64u), R);
  hipLaunchKernelGGL(tk_compact_kernel, cgrid, dim3(256), 0, st, keys, state, counters, out_indices, V,
                     (unsigned)k);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_sample_wor_fwd(const float* weights, const float* noise, int64_t* out_indices, float* workspace,
                                   int R, long V, long k, int weights_shared, int noise_is_exponential,
                                   void* stream) {
  return tk_select(weights, noise, out_indices, workspace, R, V, k, weights_shared, noise_is_exponential ? 2 : 0,
                   stream);
}

extern "C" int occf_topk_smallest_abs_fwd(const float* values, int64_t* out_indices, float* workspace, int R, long V,
                                          long k, void* stream) {
  return tk_select(values, values, out_indices, workspace, R, V, k, 0, 1, stream);
}

// ---------------------------------------------------------------------------------------------
// Row sums for the point-sampled mask losses: out[r] = { sum BCE-with-logits(x, t), sum sigmoid(x)*t,
// sum sigmoid(x), sum t } over the P sampled points of row r.
// ---------------------------------------------------------------------------------------------
// (1024 threads per row: the ~20 matched rows of a prediction set are all the parallelism there is, and 256 threads
// walking 50 176 points with two transcendentals each took 130 us per call)
__global__ void __launch_bounds__(1024) point_loss_rows_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                               float* __restrict__ out, long P) {
  __shared__ float red[16][4];
  const int r = blockIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (long i = threadIdx.x; i < P; i += blockDim.x) {
    const float xi = x[(long)r * P + i], ti = t[(long)r * P + i];
    // max(x,0) - x*t + log1p(exp(-|x|))   (the form F.binary_cross_entropy_with_logits uses)
    a0 += fmaxf(xi, 0.f) - xi * ti + log1pf(expf(-fabsf(xi)));
    const float s = 1.f / (1.f + expf(-xi));
    a1 += s * ti;
    a2 += s;
    a3 += ti;
  }
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_down(a0, o);
    a1 += __shfl_down(a1, o);
    a2 += __shfl_down(a2, o);
    a3 += __shfl_down(a3, o);
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wv][0] = a0; red[wv][1] = a1; red[wv][2] = a2; red[wv][3] = a3; }
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    const int nw = blockDim.x >> 6;
    for (int w = 0; w < nw; ++w) s += red[w][threadIdx.x];
    out[(long)r * 4 + threadIdx.x] = s;
  }
}

extern "C" int occf_point_loss_rows_fwd(const float* logits, const float* targets, float* out, int R, long P,
                                        void* stream) {
  if (R <= 0 || P <= 0) return OCCF_EINVAL;
  hipLaunchKernelGGL(point_loss_rows_kernel, dim3(R), dim3(P >= 8192 ? 1024 : 256), 0, (hipStream_t)stream, logits,
                     targets, out, P);
  OCCF_LAUNCH_CHECK();
}
