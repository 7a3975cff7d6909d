// Mask2Former-style occupancy decoder kernels:
//   * preserve-pooling  (adaptive 3-D max-pool of the mask logits -> attention mask)
//   * masked cross-attention  (100 queries x up to ~10^5 voxel keys, split over key chunks)
//   * fused trilinear upsample + sigmoid + class-weighted reduction (final occupancy volume)
//   * lidarseg point sampling of the class volume
//
// Reference: projects/mmdet3d_plugin/occformer/mask2former/mask2former_nusc_occ.py
//   forward_head :426-471, forward :589-689 (all-masked-row fix :652-653),
//   simple_test :698-745, format_results :691-696, forward_lidarseg :505-542.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

// ---------------------------------------------------------------------------------------
// Preserve-pooling: F.adaptive_max_pool3d(mask_pred, (ox,oy,oz)) then `.sigmoid() < 0.5`
// (== pooled logit < 0).  mask_pred [BQ, X, Y, Z] (z fastest).  Writes the pooled logits
// [BQ, L] (L = ox*oy*oz), the blocked bytes [BQ, L] (1 = key may NOT be attended) and ORs
// row_open[bq] = 1 when at least one key of the row is open (feeds the all-masked fix).
__device__ __forceinline__ float occf_nanmax(float m, float v) { return (v > m || v != v) ? v : m; }

// Workgroup = (bq, output x-cell).  Phase 1: the x-window is a contiguous run of
// (x1-x0)*Y*Z floats; every thread owns float4 columns (y, z..z+3) and folds the window's
// x-planes into them (fully coalesced 16-byte reads).  Phase 2: the Y*Z column maxima sit in
// LDS and one thread per output cell folds its (y, z) window.
__global__ void __launch_bounds__(256) mask_pool_kernel(
    const float* __restrict__ mask_pred, float* __restrict__ pooled, uint8_t* __restrict__ blocked,
    int* __restrict__ row_open, long BQ, int X, int Y, int Z, int ox, int oy, int oz) {
  OCCF_DYN_SMEM(smem);
  float* col = (float*)smem;                       // [Y*Z]
  const long bq = blockIdx.x / ox;
  const int cx = (int)(blockIdx.x % ox);
  const int x0 = (int)(((long)cx * X) / ox), x1 = (int)((((long)cx + 1) * X + ox - 1) / ox);
  const int YZ = Y * Z;
  const float* src = mask_pred + bq * (long)X * YZ;
  if ((YZ & 3) == 0) {
    for (int e = threadIdx.x * 4; e < YZ; e += blockDim.x * 4) {
      float4 m = *(const float4*)(src + (long)x0 * YZ + e);
      for (int x = x0 + 1; x < x1; ++x) {
        const float4 v = *(const float4*)(src + (long)x * YZ + e);
        m.x = occf_nanmax(m.x, v.x); m.y = occf_nanmax(m.y, v.y);
        m.z = occf_nanmax(m.z, v.z); m.w = occf_nanmax(m.w, v.w);
      }
      *(float4*)(col + e) = m;
    }
  } else {
    for (int e = threadIdx.x; e < YZ; e += blockDim.x) {
      float m = src[(long)x0 * YZ + e];
      for (int x = x0 + 1; x < x1; ++x) m = occf_nanmax(m, src[(long)x * YZ + e]);
      col[e] = m;
    }
  }
  __syncthreads();
  const long L = (long)ox * oy * oz;
  bool any_open = false;
  for (int c = threadIdx.x; c < oy * oz; c += blockDim.x) {
    const int cz = c % oz, cy = c / oz;
    const int y0 = (int)(((long)cy * Y) / oy), y1 = (int)((((long)cy + 1) * Y + oy - 1) / oy);
    const int z0 = (int)(((long)cz * Z) / oz), z1 = (int)((((long)cz + 1) * Z + oz - 1) / oz);
    float m = -INFINITY;
    for (int y = y0; y < y1; ++y)
      for (int z = z0; z < z1; ++z) m = occf_nanmax(m, col[y * Z + z]);
    const long o = bq * L + ((long)cx * oy + cy) * oz + cz;
    pooled[o] = m;
    // sigmoid(m) < 0.5, evaluated as the reference does (fp32 sigmoid, then compare)
    const float sg = 1.0f / (1.0f + expf(-m));
    const bool blk = sg < 0.5f;
    blocked[o] = blk ? 1 : 0;
    any_open |= !blk;
  }
  if (any_open) atomicOr((unsigned*)&row_open[bq], 1u);
}

extern "C" int occf_mask_pool_fwd(const float* mask_pred, float* pooled, uint8_t* blocked,
                                  int32_t* row_open, long BQ, int X, int Y, int Z, int ox, int oy,
                                  int oz, void* stream) {
  if (BQ <= 0 || X <= 0 || Y <= 0 || Z <= 0 || ox <= 0 || oy <= 0 || oz <= 0) return OCCF_EINVAL;
  if (ox > X || oy > Y || oz > Z) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
#ifndef OCCF_EMU
  hipError_t e = hipMemsetAsync(row_open, 0, sizeof(int32_t) * BQ, st);
  if (e != hipSuccess) return (int)e;
#else
  memset(row_open, 0, sizeof(int32_t) * BQ);
#endif
  if (BQ * ox >= 2147483647L || (long)Y * Z * 4 > 160 * 1024) return OCCF_ESHAPE;
  hipLaunchKernelGGL(mask_pool_kernel, dim3((unsigned)(BQ * ox)), dim3(256), (size_t)Y * Z * sizeof(float), st,
                     mask_pred, pooled, blocked, (int*)row_open, BQ, X, Y, Z, ox, oy, oz);
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Masked cross-attention, head_dim = 32.  q [B, Q, E] (already projected, includes the
// query positional embedding), k/v [B, L, E] (projected; k includes the key positional
// encoding), blocked [B, Q, L] bytes shared by all heads, row_open [B*Q].
// Phase 1: workgroup = (key chunk, head, batch); thread = one query row (q in registers,
// running max / sum / 32-wide output in registers); the chunk's K and V head slices are
// staged through LDS and consumed as broadcasts.  Phase 2 merges the chunks (log-sum-exp).
#define XA_HD 32
#define XA_TILE 64       // keys per LDS tile
#define XA_QPB 128       // query rows per workgroup (threads)

__global__ void __launch_bounds__(XA_QPB) masked_xattn_partial_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
    const uint8_t* __restrict__ blocked, const int* __restrict__ row_open, float* __restrict__ part_o,
    float* __restrict__ part_ml, int B, int Q, int L, int E, int heads, int chunk, int n_chunks,
    float scale) {
  __shared__ __attribute__((aligned(16))) float lds_k[XA_TILE * XA_HD];
  __shared__ __attribute__((aligned(16))) float lds_v[XA_TILE * XA_HD];
  const int ck = blockIdx.x % n_chunks;
  const int qb = blockIdx.x / n_chunks;      // query block
  const int h = blockIdx.y;
  const int b = blockIdx.z;
  const int qi = qb * XA_QPB + threadIdx.x;
  const bool valid = qi < Q;
  float qr[XA_HD], o[XA_HD];
  float m = -INFINITY, l = 0.f;
  bool use_mask = false;
  if (valid) {
    const float* qp = q + ((long)b * Q + qi) * E + h * XA_HD;
#pragma unroll
    for (int d = 0; d < XA_HD; ++d) qr[d] = qp[d] * scale;
    use_mask = blocked != nullptr && row_open[b * Q + qi] != 0;
  }
#pragma unroll
  for (int d = 0; d < XA_HD; ++d) o[d] = 0.f;
  const int k0 = ck * chunk;
  const int k1 = (k0 + chunk < L) ? k0 + chunk : L;
  const uint8_t* brow = valid && use_mask ? blocked + ((long)b * Q + qi) * L : nullptr;
  // K / V tiles: 4 + 4 float4 per thread, fetched as one unconditional batch (rows past the end of the chunk
  // read the last valid key) one tile AHEAD of the tile being multiplied
  float4 rk[4], rv[4];
  auto fetch_tile = [&](int t0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * XA_QPB;
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      int key = t0 + r;
      key = key < k1 ? key : k1 - 1;
      key = key < k0 ? k0 : key;
      const long src = ((long)b * L + key) * E + h * XA_HD + c4;
      rk[i] = *(const float4*)(k + src);
      rv[i] = *(const float4*)(v + src);
    }
  };
  auto commit_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = threadIdx.x + i * XA_QPB;
      const int r = idx >> 3, c4 = (idx & 7) * 4;
      *(float4*)(&lds_k[r * XA_HD + c4]) = rk[i];
      *(float4*)(&lds_v[r * XA_HD + c4]) = rv[i];
    }
  };
  fetch_tile(k0);
  for (int t0 = k0; t0 < k1; t0 += XA_TILE) {
    const int nt = (k1 - t0 < XA_TILE) ? k1 - t0 : XA_TILE;
    __syncthreads();                       // the previous tile has been consumed
    commit_tile();
    __syncthreads();
    fetch_tile(t0 + XA_TILE);              // next tile (clamped: a harmless re-read after the last one)
    if (!valid) continue;
    // groups of 8 keys, branch-free: masked / out-of-range keys get a score of -inf (weight exp(-inf) = 0),
    // one running-max update and one rescale of the accumulators per group
    for (int r0 = 0; r0 < nt; r0 += 8) {
      unsigned long long mbits = 0;                     // byte j != 0  <=>  key r0 + j is masked out
      if (brow != nullptr) {
        const uint8_t* bp = brow + t0 + r0;
        if ((((unsigned long long)bp) & 7ull) == 0 && r0 + 8 <= nt) {
          mbits = *(const unsigned long long*)bp;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (r0 + j < nt && bp[j]) mbits |= 0xFFull << (8 * j);
        }
      }
      float sc[8];
      float mt = -INFINITY;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + j < nt ? r0 + j : nt - 1;
        const float* kr = &lds_k[r * XA_HD];
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int d = 0; d < XA_HD; d += 2) {
          a0 = fmaf(qr[d], kr[d], a0);
          a1 = fmaf(qr[d + 1], kr[d + 1], a1);
        }
        const bool ok = r0 + j < nt && ((mbits >> (8 * j)) & 0xFFull) == 0;
        sc[j] = ok ? a0 + a1 : -INFINITY;
        mt = fmaxf(mt, sc[j]);
      }
      const float m_new = fmaxf(m, mt);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;    // nothing open yet: every weight is 0
      const float c = expf(m - m_use);                          // m = -inf -> 0
      l *= c;
#pragma unroll
      for (int d = 0; d < XA_HD; ++d) o[d] *= c;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int r = r0 + j < nt ? r0 + j : nt - 1;
        const float pw = expf(sc[j] - m_use);                  // -inf -> 0
        l += pw;
        const float* vr = &lds_v[r * XA_HD];
#pragma unroll
        for (int d = 0; d < XA_HD; ++d) o[d] = fmaf(pw, vr[d], o[d]);
      }
      m = m_new;
    }
  }
  if (!valid) return;
  const long slot = (((long)b * heads + h) * Q + qi) * n_chunks + ck;
  part_ml[slot * 2 + 0] = m;
  part_ml[slot * 2 + 1] = l;
  float* po = part_o + slot * XA_HD;
#pragma unroll
  for (int d = 0; d < XA_HD; d += 4) *(float4*)(po + d) = make_float4(o[d], o[d + 1], o[d + 2], o[d + 3]);
}

// one 64-lane wave per (b, h, q): lanes stride over the key chunks (log-sum-exp merge); the
// lane -> chunk assignment and the butterfly order are fixed, so the result is deterministic
__global__ void __launch_bounds__(256) masked_xattn_merge_kernel(
    const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out, int B,
    int Q, int E, int heads, int n_chunks) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;      // (b, h, q)
  if (row >= (long)B * heads * Q) return;
  const int qi = (int)(row % Q);
  const int h = (int)((row / Q) % heads);
  const int b = (int)(row / ((long)Q * heads));
  const long base = row * n_chunks;
  float M = -INFINITY;
  for (int c = lane; c < n_chunks; c += 64) M = fmaxf(M, part_ml[(base + c) * 2]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o));
  float l = 0.f, acc[XA_HD];
#pragma unroll
  for (int d = 0; d < XA_HD; ++d) acc[d] = 0.f;
  for (int c = lane; c < n_chunks; c += 64) {
    const float mc = part_ml[(base + c) * 2];
    if (mc == -INFINITY) continue;
    const float w = expf(mc - M);
    l = fmaf(part_ml[(base + c) * 2 + 1], w, l);
    const float* po = part_o + (base + c) * XA_HD;
#pragma unroll
    for (int d = 0; d < XA_HD; d += 4) {
      const float4 t = *(const float4*)(po + d);
      acc[d] = fmaf(t.x, w, acc[d]); acc[d + 1] = fmaf(t.y, w, acc[d + 1]);
      acc[d + 2] = fmaf(t.z, w, acc[d + 2]); acc[d + 3] = fmaf(t.w, w, acc[d + 3]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    l += __shfl_xor(l, o);
#pragma unroll
    for (int d = 0; d < XA_HD; ++d) acc[d] += __shfl_xor(acc[d], o);
  }
  if (lane < XA_HD) {
    float v = 0.f;
#pragma unroll
    for (int d = 0; d < XA_HD; ++d) v = (d == lane) ? acc[d] : v;
    out[((long)b * Q + qi) * E + h * XA_HD + lane] = v / l;
  }
}

// xattn_mfma.hip: the matrix-core variant of the partial kernel (<= 128 queries)
void occf_xattn_mfma_launch(const float* q, const float* k, const float* v, const uint8_t* blocked,
                            const int* row_open, float* part_o, float* part_ml, int B, int Q, int L, int E, int heads,
                            int chunk, int n_chunks, float scale, hipStream_t st);

static bool occf_xattn_use_mfma(int Q) {
  static const bool on = [] {
    const char* e = getenv("OCCF_XATTN_MFMA");
    return e ? atoi(e) != 0 : true;
  }();
  return on && Q <= 128;
}

static int occf_xattn_chunk(int B, int Q, int L, int heads) {
  if (occf_xattn_use_mfma(Q)) {
    // keys per workgroup (a multiple of the 32-key MFMA tile): long chunks amortise the query set-up,
    // but the grid should still cover the 256 CUs about twice
    int chunk = 1024;
    while (chunk > 32 && (long)occf_cdiv(L, chunk) * heads * B < 512) chunk >>= 1;
    return chunk;
  }
  // scalar kernel: short enough that the per-thread serial walk stays ~10 us, while the merge
  // (one wave per row, lanes over chunks) absorbs the chunk count
  int chunk = 256;
  while (chunk > XA_TILE && (long)occf_cdiv(L, chunk) * heads * B < 512) chunk >>= 1;
  return chunk;
}

extern "C" int occf_masked_xattn_fwd(const float* q, const float* k, const float* v,
                                     const uint8_t* blocked, const int32_t* row_open, float* out,
                                     float* workspace, long workspace_floats, int B, int Q, int L,
                                     int E, int heads, void* stream) {
  if (B <= 0 || Q <= 0 || L <= 0 || heads <= 0 || E != heads * XA_HD) return OCCF_ESHAPE;
  // chunking: enough workgroups to fill 256 CUs, chunks a multiple of the LDS tile
  const int chunk = occf_xattn_chunk(B, Q, L, heads);
  const int n_chunks = occf_cdiv(L, chunk);
  const int qblocks = occf_cdiv(Q, XA_QPB);
  const long need = (long)B * heads * Q * n_chunks * (XA_HD + 2);
  if (workspace == nullptr || workspace_floats < need) return OCCF_EINVAL;
  float* part_o = workspace;
  float* part_ml = workspace + (long)B * heads * Q * n_chunks * XA_HD;
  const float scale = (float)(1.0 / sqrt((double)XA_HD));
  hipStream_t st = (hipStream_t)stream;
  if (occf_xattn_use_mfma(Q))
    occf_xattn_mfma_launch(q, k, v, blocked, (const int*)row_open, part_o, part_ml, B, Q, L, E, heads, chunk,
                           n_chunks, scale, st);
  else
    hipLaunchKernelGGL(masked_xattn_partial_kernel, dim3(n_chunks * qblocks, heads, B), dim3(XA_QPB), 0, st,
                       q, k, v, blocked, (const int*)row_open, part_o, part_ml, B, Q, L, E, heads, chunk,
                       n_chunks, scale);
  const long total = (long)B * heads * Q * 64;
  hipLaunchKernelGGL(masked_xattn_merge_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, st, part_o,
                     part_ml, out, B, Q, E, heads, n_chunks);
  OCCF_LAUNCH_CHECK();
}

extern "C" long occf_masked_xattn_workspace(int B, int Q, int L, int heads) {
  const int chunk = occf_xattn_chunk(B, Q, L, heads);
  return (long)B * heads * Q * occf_cdiv(L, chunk) * (XA_HD + 2);
}

// ---------------------------------------------------------------------------------------
// Final occupancy volume: trilinear upsample (align_corners=True) of the last mask logits
// to occ_size, sigmoid, and the class-weighted sum over queries with softmax(cls)[..., :-1]
// -- one pass; the reference's [B,100,256,256,32] fp32 intermediate (839 MB) never exists.
// mask_pred [B, Q, X, Y, Z]; cls [B, Q, K+1]; out [B, K, X2, Y2, Z2].
#define UC_MAXK 24

// softmax(cls)[..., :K] of every query -> prob[B][Q][UC_MAXK] (read back by the main kernel through the
// SCALAR cache: the table index is wave-uniform, so the 17 class weights of a query cost s_loads instead
// of 17 LDS broadcasts per lane -- the LDS issue rate was this kernel's limiter)
__global__ void __launch_bounds__(128) class_prob_kernel(const float* __restrict__ cls, float* __restrict__ prob, int BQ,
                                                         int K) {
  const int qi = blockIdx.x * blockDim.x + threadIdx.x;
  if (qi >= BQ) return;
  const float* c = cls + (long)qi * (K + 1);
  float mx = c[0];
  for (int i = 1; i <= K; ++i) mx = fmaxf(mx, c[i]);
  float sum = 0.f;
  for (int i = 0; i <= K; ++i) sum += expf(c[i] - mx);
  for (int i = 0; i < UC_MAXK; ++i) prob[(long)qi * UC_MAXK + i] = i < K ? expf(c[i] - mx) / sum : 0.f;
}

// One output voxel per lane with z fastest: the 64 lanes of a wave cover 2 output rows, so each of the 8
// tap loads touches 2-4 cache lines (variants with 2 / 4 outputs per lane spread a wave over 4 / 8 rows and
// became bound by the lines per gather instead).  What is left is VALU work, kept to ~35 instructions per
// (voxel, query): corner weights precomputed (8 FMAs for the blend), hardware exp / rcp for the sigmoid, KB
// class FMAs with the class weights in SGPRs, buffer-addressed gathers (fixed 32-bit lane offsets, the
// query's plane offset is scalar) and the next query's taps in flight while this one is consumed.
template <int KB>
__global__ void __launch_bounds__(256) upsample_classify_kernel(
    const float* __restrict__ mask_pred, const float* __restrict__ prob, float* __restrict__ out, int B,
    int Q, int K, int X, int Y, int Z, int X2, int Y2, int Z2) {
  const int b = blockIdx.y;
  const long V2 = (long)X2 * Y2 * Z2;
  // x2-major workgroup order + XCD remap: the output planes that read one source plane share an L2
  const long wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  int x2, y2, z2;
  long vid;
  if (Z2 == 32 && (X2 & 1) == 0 && (Y2 & 3) == 0) {
    // workgroup = 2 x-planes x 4 y-rows x 32 z: both output planes blend the same two source planes and the
    // 4 rows need ~3 source rows -- 6 source lines per query instead of 10 for 8 rows of one plane (the
    // kernel is bound by the L2 -> L1 line traffic of its gathers)
    const int yb = Y2 >> 2;
    z2 = threadIdx.x & 31;
    y2 = (int)(wg % yb) * 4 + ((threadIdx.x >> 5) & 3);
    x2 = (int)(wg / yb) * 2 + (threadIdx.x >> 7);
    vid = ((long)x2 * Y2 + y2) * Z2 + z2;
  } else {
    vid = wg * blockDim.x + threadIdx.x;
    if (vid >= V2) return;
    z2 = (int)(vid % Z2);
    y2 = (int)((vid / Z2) % Y2);
    x2 = (int)(vid / ((long)Z2 * Y2));
  }
  // align_corners=True source coordinate: dst * (in-1)/(out-1)
  const float sx = X2 > 1 ? (float)(X - 1) / (float)(X2 - 1) : 0.f;
  const float sy = Y2 > 1 ? (float)(Y - 1) / (float)(Y2 - 1) : 0.f;
  const float sz = Z2 > 1 ? (float)(Z - 1) / (float)(Z2 - 1) : 0.f;
  const float fx = sx * x2, fy = sy * y2, fz = sz * z2;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int x1 = x0 + (x0 < X - 1), y1 = y0 + (y0 < Y - 1), z1 = z0 + (z0 < Z - 1);
  const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
  float w[8];
  uint32_t off[8];                   // byte offsets of the 8 taps inside one query's volume
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = ((c & 4) ? tx : 1.f - tx) * ((c & 2) ? ty : 1.f - ty) * ((c & 1) ? tz : 1.f - tz);
    off[c] = (uint32_t)((((c & 4) ? x1 : x0) * Y + ((c & 2) ? y1 : y0)) * Z + ((c & 1) ? z1 : z0)) * 4u;
  }
  float acc[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] = 0.f;
  const uint32_t Vb = (uint32_t)X * Y * Z * 4u;          // bytes per query volume (host checks Q * Vb < 2^32)
  const occf_buf mb = occf_make_buf(mask_pred + (long)b * Q * ((long)X * Y * Z));
  const float* pb = prob + (long)b * Q * UC_MAXK;
  float ma[8], mq[8];                // two tap sets: one in flight while the other is used
  auto fetch = [&](float* m, int q) {
    const uint32_t so = (uint32_t)(q < Q ? q : Q - 1) * Vb;     // past the end: re-read the last volume
#pragma unroll
    for (int c = 0; c < 8; ++c) m[c] = occf_buf_load_f32(mb, off[c], so);
  };
  auto consume = [&](const float* m, int q) {
    float val = w[0] * m[0];
#pragma unroll
    for (int c = 1; c < 8; ++c) val = fmaf(w[c], m[c], val);
    const float sg = occf_rcp_fast(1.0f + __expf(-val));
    const float* pr = pb + q * UC_MAXK;                  // wave-uniform address -> scalar loads
#pragma unroll
    for (int i = 0; i < KB; ++i) acc[i] = fmaf(pr[i], sg, acc[i]);
  };
  fetch(ma, 0);
  for (int qi = 0; qi < Q; qi += 2) {
    fetch(mq, qi + 1);
    consume(ma, qi);
    fetch(ma, qi + 2);
    if (qi + 1 < Q) consume(mq, qi + 1);
  }
#pragma unroll
  for (int i = 0; i < KB; ++i)
    if (i < K) out[((long)b * K + i) * V2 + vid] = acc[i];
}


// LDS-staged variant (2x upsampling, Z = 16 -> Z2 = 32).  PMC on the gather kernel above (profiles/
// r02e_upsample_classify_pmc_summary.txt): 64.6 M VMEM read instructions and 264 M L1 tag lookups per launch,
// SQ_WAIT_INST_ANY = 67 % of the wave cycles -- it is bound by the ISSUE of its 4-byte gathers (8 per voxel and
// query, 4.1 cache lines each), not by VALU or HBM.  Here a workgroup (2 output x-planes x 4 y-rows x 32 z) stages the
// <= 3 x 4 x 16 source cells it blends per query through LDS with ONE 16-byte load per lane (48 lanes per query, four
// queries per round, next round's loads in flight), and the 8 taps become ds_read2_b32 pairs: 1/170 of the vector
// memory instructions.
#define UCL_QG 4                 // queries per round (one per wave)
#define UCL_CELLS 192            // 3 x 4 x 16 floats per query
template <int KB>
__global__ void __launch_bounds__(256) upsample_classify_lds_kernel(
    const float* __restrict__ mask_pred, const float* __restrict__ prob, float* __restrict__ out, int B,
    int Q, int K, int X, int Y, int X2, int Y2) {
  constexpr int Z = 16, Z2 = 32;
  __shared__ __attribute__((aligned(16))) float tile[2][UCL_QG][UCL_CELLS];
  const int b = blockIdx.y;
  const long V2 = (long)X2 * Y2 * Z2;
  const long wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const int yb = Y2 >> 2;
  const int z2 = threadIdx.x & 31;
  const int y2 = (int)(wg % yb) * 4 + ((threadIdx.x >> 5) & 3);
  const int x2 = (int)(wg / yb) * 2 + (threadIdx.x >> 7);
  const long vid = ((long)x2 * Y2 + y2) * Z2 + z2;
  const float sx = X2 > 1 ? (float)(X - 1) / (float)(X2 - 1) : 0.f;
  const float sy = Y2 > 1 ? (float)(Y - 1) / (float)(Y2 - 1) : 0.f;
  const float sz = (float)(Z - 1) / (float)(Z2 - 1);
  const float fx = sx * x2, fy = sy * y2, fz = sz * z2;
  const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
  const int x1 = x0 + (x0 < X - 1), y1 = y0 + (y0 < Y - 1), z1 = z0 + (z0 < Z - 1);
  const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
  // region origin of this workgroup (first output plane / row): every tap lies in [xa, xa+2] x [ya, ya+3]
  const int xa = (int)(sx * (float)((int)(wg / yb) * 2)), ya = (int)(sy * (float)((int)(wg % yb) * 4));
  float w[8];
  int off[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    w[c] = ((c & 4) ? tx : 1.f - tx) * ((c & 2) ? ty : 1.f - ty) * ((c & 1) ? tz : 1.f - tz);
    off[c] = ((((c & 4) ? x1 : x0) - xa) * 4 + (((c & 2) ? y1 : y0) - ya)) * Z + ((c & 1) ? z1 : z0);
  }
  float acc[KB];
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i] = 0.f;
  // loader role: wave wv fetches query (round * 4 + wv); lane l < 48 -> (rx, ry, z quad)
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int lrx = ln >> 4, lry = (ln >> 2) & 3, lzq = ln & 3;
  const int lx = occf_clampi(xa + lrx, X - 1), ly = occf_clampi(ya + lry, Y - 1);
  const float* src0 = mask_pred + (long)b * Q * ((long)X * Y * Z) + ((long)lx * Y + ly) * Z + lzq * 4;
  const long qstride = (long)X * Y * Z;
  const float* pb = prob + (long)b * Q * UC_MAXK;
  float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int q) __attribute__((always_inline)) {
    const int qq = q < Q ? q : Q - 1;
    if (ln < 48) r = *(const float4*)(src0 + (long)qq * qstride);
  };
  fetch(wv);
  const int rounds = (Q + UCL_QG - 1) / UCL_QG;
  for (int rd = 0; rd < rounds; ++rd) {
    float* dst = &tile[rd & 1][wv][0];
    if (ln < 48) *(float4*)(dst + (lrx * 4 + lry) * Z + lzq * 4) = r;
    __syncthreads();                                     // round rd is staged (the other buffer is free again)
    fetch((rd + 1) * UCL_QG + wv);
#pragma unroll
    for (int u = 0; u < UCL_QG; ++u) {
      const int q = rd * UCL_QG + u;
      if (q < Q) {
        const float* m = &tile[rd & 1][u][0];
        float val = w[0] * m[off[0]];
#pragma unroll
        for (int c = 1; c < 8; ++c) val = fmaf(w[c], m[off[c]], val);
        const float sg = occf_rcp_fast(1.0f + __expf(-val));
        const float* pr = pb + q * UC_MAXK;                // wave-uniform address -> scalar loads
#pragma unroll
        for (int i = 0; i < KB; ++i) acc[i] = fmaf(pr[i], sg, acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < KB; ++i)
    if (i < K) out[((long)b * K + i) * V2 + vid] = acc[i];
}

// Matrix-core variant of the LDS-staged kernel: the class volume is a contraction over the queries,
//   out[class, voxel] = sum_q prob[q, class] * sigmoid(mask_q(voxel)),
// i.e. [classes x Q] . [Q x voxels].  The VALU version spends 18 of its ~45 instructions per (voxel, query) on the
// class FMAs; here they are one v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate) per 32 voxels and TWO
// queries: A = prob^T (row = class, k = query parity), B = sigmoid values (k = query parity, column = voxel), so lane
// l computes the sigmoid of voxel l & 31 for query 2t + (l >> 5) -- exactly its B element -- and the matrix pipe does
// the class sums while the other waves interpolate.  A wave owns two 32-voxel z-lines (two accumulators).
// (Measured 0.73 ms against 0.62 ms for the VALU variant; kept behind OCCF_CLASSIFY_MFMA=1.)
template <int QMAX>
__global__ void __launch_bounds__(256) upsample_classify_mfma_kernel(
    const float* __restrict__ mask_pred, const float* __restrict__ prob, float* __restrict__ out, int B,
    int Q, int K, int X, int Y, int X2, int Y2) {
  constexpr int Z = 16, Z2 = 32;
  __shared__ __attribute__((aligned(16))) float tile[2][UCL_QG][UCL_CELLS];
  __shared__ float ptab[QMAX * 32];                       // [query][class], zero beyond K / Q
  const int b = blockIdx.y;
  const long V2 = (long)X2 * Y2 * Z2;
  const long wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const int yb = Y2 >> 2;
  const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
  const int vj = ln & 31, par = ln >> 5;
  const float* pb = prob + (long)b * Q * UC_MAXK;
  for (int i = threadIdx.x; i < QMAX * 32; i += 256) {
    const int q = i >> 5, c = i & 31;
    ptab[i] = (q < Q && c < K) ? pb[q * UC_MAXK + c] : 0.f;
  }
  const float sx = X2 > 1 ? (float)(X - 1) / (float)(X2 - 1) : 0.f;
  const float sy = Y2 > 1 ? (float)(Y - 1) / (float)(Y2 - 1) : 0.f;
  const float sz = (float)(Z - 1) / (float)(Z2 - 1);
  const int x2 = (int)(wg / yb) * 2 + (wv >> 1);
  const int xa = (int)(sx * (float)((int)(wg / yb) * 2)), ya = (int)(sy * (float)((int)(wg % yb) * 4));
  float w[2][8];
  int off[2][8];
  long vid[2];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int y2 = (int)(wg % yb) * 4 + ((2 * wv) & 3) + g;
    vid[g] = ((long)x2 * Y2 + y2) * Z2 + vj;
    const float fx = sx * x2, fy = sy * y2, fz = sz * vj;
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    const int x1 = x0 + (x0 < X - 1), y1 = y0 + (y0 < Y - 1), z1 = z0 + (z0 < Z - 1);
    const float tx = fx - x0, ty = fy - y0, tz = fz - z0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      w[g][c] = ((c & 4) ? tx : 1.f - tx) * ((c & 2) ? ty : 1.f - ty) * ((c & 1) ? tz : 1.f - tz);
      off[g][c] = ((((c & 4) ? x1 : x0) - xa) * 4 + (((c & 2) ? y1 : y0) - ya)) * Z + ((c & 1) ? z1 : z0);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;
  // loader role (as in the VALU variant): wave wv fetches query (round * 4 + wv); lane l < 48 -> (rx, ry, z quad)
  const int lrx = ln >> 4, lry = (ln >> 2) & 3, lzq = ln & 3;
  const int lx = occf_clampi(xa + lrx, X - 1), ly = occf_clampi(ya + lry, Y - 1);
  const float* src0 = mask_pred + (long)b * Q * ((long)X * Y * Z) + ((long)lx * Y + ly) * Z + lzq * 4;
  const long qstride = (long)X * Y * Z;
  float4 r4 = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int q) __attribute__((always_inline)) {
    const int qq = q < Q ? q : Q - 1;
    if (ln < 48) r4 = *(const float4*)(src0 + (long)qq * qstride);
  };
  fetch(wv);
  const int rounds = (Q + UCL_QG - 1) / UCL_QG;
  for (int rd = 0; rd < rounds; ++rd) {
    float* dst = &tile[rd & 1][wv][0];
    if (ln < 48) *(float4*)(dst + (lrx * 4 + lry) * Z + lzq * 4) = r4;
    __syncthreads();                                     // round rd (and, the first time, ptab) is staged
    fetch((rd + 1) * UCL_QG + wv);
#pragma unroll
    for (int pr = 0; pr < UCL_QG / 2; ++pr) {
      const int q = rd * UCL_QG + 2 * pr + par;          // this lane half's query of the pair
      const float* m = &tile[rd & 1][2 * pr + par][0];
      const float a = q < Q ? ptab[q * 32 + vj] : 0.f;    // A element: class vj of query q
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float val = w[g][0] * m[off[g][0]];
#pragma unroll
        for (int c = 1; c < 8; ++c) val = fmaf(w[g][c], m[off[g][c]], val);
        const float sg = q < Q ? occf_rcp_fast(1.0f + __expf(-val)) : 0.f;
        acc[g] = occf_mfma_f32_32x32x2(a, sg, acc[g]);
      }
    }
  }
  // D[class][voxel]: lane (voxel vj, half par) holds classes (r & 3) + 8 (r >> 2) + 4 par
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int cls = (r & 3) + 8 * (r >> 2) + 4 * par;
      if (cls < K) out[((long)b * K + cls) * V2 + vid[g]] = acc[g][r];
    }
}

// Same-resolution case (X2 == X ...: the nuScenes head predicts masks at the output grid): the
// align_corners resample is the identity, so every output voxel needs ONE mask value per query instead
// of 8 taps.  A thread owns 2 consecutive voxels (float2 loads, fully coalesced), keeps UQ queries'
// loads in flight, and accumulates KB classes x 2 voxels in registers; q runs in ascending order as in
// the resampling kernel, so both produce the same bits.
template <int KB, int UQ>
__global__ void __launch_bounds__(256) classify_identity_kernel(const float* __restrict__ mask_pred,
                                                                const float* __restrict__ prob,
                                                                float* __restrict__ out, int Q, int K, long V) {
  const int b = blockIdx.y;
  const long v = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (v >= V) return;
  float acc[KB][2];
#pragma unroll
  for (int i = 0; i < KB; ++i) acc[i][0] = acc[i][1] = 0.f;
  const float* mp = mask_pred + (long)b * Q * V + v;
  const float* pb = prob + (long)b * Q * UC_MAXK;
  int q0 = 0;
  for (; q0 + UQ <= Q; q0 += UQ) {
    float2 m[UQ];
#pragma unroll
    for (int u = 0; u < UQ; ++u) m[u] = *reinterpret_cast<const float2*>(mp + (long)(q0 + u) * V);
#pragma unroll
    for (int u = 0; u < UQ; ++u) {
      const float s0 = 1.0f / (1.0f + expf(-m[u].x)), s1 = 1.0f / (1.0f + expf(-m[u].y));
      const float* pr = pb + (q0 + u) * UC_MAXK;         // wave-uniform -> scalar loads
#pragma unroll
      for (int i = 0; i < KB; ++i) {
        acc[i][0] = fmaf(pr[i], s0, acc[i][0]);
        acc[i][1] = fmaf(pr[i], s1, acc[i][1]);
      }
    }
  }
  for (; q0 < Q; ++q0) {
    const float2 m = *reinterpret_cast<const float2*>(mp + (long)q0 * V);
    const float s0 = 1.0f / (1.0f + expf(-m.x)), s1 = 1.0f / (1.0f + expf(-m.y));
    const float* pr = pb + q0 * UC_MAXK;
#pragma unroll
    for (int i = 0; i < KB; ++i) {
      acc[i][0] = fmaf(pr[i], s0, acc[i][0]);
      acc[i][1] = fmaf(pr[i], s1, acc[i][1]);
    }
  }
#pragma unroll
  for (int i = 0; i < KB; ++i)
    if (i < K) {
      float2 o;
      o.x = acc[i][0];
      o.y = acc[i][1];
      *reinterpret_cast<float2*>(out + ((long)b * K + i) * V + v) = o;
    }
}

extern "C" int occf_upsample_classify_fwd(const float* mask_pred, const float* cls, float* out, float* workspace,
                                          int B, int Q, int K, int X, int Y, int Z, int X2, int Y2, int Z2,
                                          void* stream) {
  if (B <= 0 || Q <= 0 || K <= 0 || K > UC_MAXK || workspace == nullptr) return OCCF_ESHAPE;
  const long V2 = (long)X2 * Y2 * Z2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(class_prob_kernel, dim3(occf_cdiv((long)B * Q, 128)), dim3(128), 0, st, cls, workspace, B * Q, K);
  static const bool ident_env = [] {
    const char* e = getenv("OCCF_CLASSIFY_IDENTITY");    // diagnostics: 0 forces the resampling kernel
    return e == nullptr || atoi(e) != 0;
  }();
  if (ident_env && X2 == X && Y2 == Y && Z2 == Z && V2 % 2 == 0) {
    const dim3 grid(occf_cdiv(V2 / 2, 256), B);
    if (K <= 18)
      hipLaunchKernelGGL((classify_identity_kernel<18, 8>), grid, dim3(256), 0, st, mask_pred,
                         (const float*)workspace, out, Q, K, V2);
    else
      hipLaunchKernelGGL((classify_identity_kernel<UC_MAXK, 8>), grid, dim3(256), 0, st, mask_pred,
                         (const float*)workspace, out, Q, K, V2);
  } else {
    if ((long)Q * X * Y * Z >= (1L << 30)) return OCCF_ESHAPE;      // 32-bit byte offsets inside the mask volume
    static const bool lds_env = [] {
      const char* e = getenv("OCCF_CLASSIFY_LDS");        // diagnostics: 0 forces the gather kernel
      return e == nullptr || atoi(e) != 0;
    }();
    // 2x (or coarser-to-finer) upsampling with 16 -> 32 slices: the LDS-staged kernel
    if (lds_env && Z == 16 && Z2 == 32 && (X2 & 1) == 0 && (Y2 & 3) == 0 && X2 >= 2 * X - 1 && Y2 >= 2 * Y - 1 &&
        K <= UC_MAXK) {
      const dim3 grid((unsigned)((long)(X2 / 2) * (Y2 / 4)), B);
      static const bool mfma_env = [] {
        // measured SLOWER than the VALU class sums (0.73 vs 0.62 ms at the 200-grid, probe 32: the fp32 MFMA issues
        // every 64 cycles and the two sigmoids that feed it sit in the same wave's instruction stream): opt-in only
        const char* e = getenv("OCCF_CLASSIFY_MFMA");
        return e != nullptr && atoi(e) != 0;
      }();
      if (mfma_env && Q <= 128) {
        hipLaunchKernelGGL((upsample_classify_mfma_kernel<128>), grid, dim3(256), 0, st, mask_pred,
                           (const float*)workspace, out, B, Q, K, X, Y, X2, Y2);
        OCCF_LAUNCH_CHECK();
      }
      if (K <= 18)
        hipLaunchKernelGGL((upsample_classify_lds_kernel<18>), grid, dim3(256), 0, st, mask_pred,
                           (const float*)workspace, out, B, Q, K, X, Y, X2, Y2);
      else
        hipLaunchKernelGGL((upsample_classify_lds_kernel<UC_MAXK>), grid, dim3(256), 0, st, mask_pred,
                           (const float*)workspace, out, B, Q, K, X, Y, X2, Y2);
      OCCF_LAUNCH_CHECK();
    }
    if (K <= 18)
      hipLaunchKernelGGL((upsample_classify_kernel<18>), dim3(occf_cdiv(V2, 256), B), dim3(256), 0, st, mask_pred,
                         (const float*)workspace, out, B, Q, K, X, Y, Z, X2, Y2, Z2);
    else
      hipLaunchKernelGGL((upsample_classify_kernel<UC_MAXK>), dim3(occf_cdiv(V2, 256), B), dim3(256), 0, st,
                         mask_pred, (const float*)workspace, out, B, Q, K, X, Y, Z, X2, Y2, Z2);
  }
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------
// Lidarseg: per point, trilinear sample (align_corners=True, border padding) of the
// low-resolution class volume  sum_q softmax(cls)[q, :K] * sigmoid(mask_pred[q, cell]),
// then a softmax over the K classes.  One 64-lane wave per point, lanes over queries.
// pts [P, 4] = (batch index, gx, gy, gz) with g in grid_sample's [-1, 1] convention,
// gx along X (first spatial dim), gz along Z.
__global__ void __launch_bounds__(256) lidarseg_sample_kernel(
    const float* __restrict__ mask_pred, const float* __restrict__ cls, const float* __restrict__ pts,
    float* __restrict__ out, int P, int Q, int K, int X, int Y, int Z) {
  const int lane = threadIdx.x & 63;
  const long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (p >= P) return;
  const int b = (int)pts[p * 4 + 0];
  float c3[3];
  int i0[3], i1[3];
  const int dims[3] = {X, Y, Z};
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float f = (pts[p * 4 + 1 + a] + 1.f) * 0.5f * (float)(dims[a] - 1);
    f = fminf(fmaxf(f, 0.f), (float)(dims[a] - 1));               // border padding
    const float fl = floorf(f);
    i0[a] = (int)fl;
    i1[a] = i0[a] + 1 < dims[a] ? i0[a] + 1 : i0[a];               // weight of the clamped tap is 0
    c3[a] = f - fl;
  }
  const long V = (long)X * Y * Z;
  float acc[UC_MAXK];
#pragma unroll
  for (int i = 0; i < UC_MAXK; ++i) acc[i] = 0.f;
  for (int qi = lane; qi < Q; qi += 64) {
    const float* mp = mask_pred + ((long)b * Q + qi) * V;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int xx = (c >> 2) ? i1[0] : i0[0], yy = ((c >> 1) & 1) ? i1[1] : i0[1], zz = (c & 1) ? i1[2] : i0[2];
      const float w = ((c >> 2) ? c3[0] : 1.f - c3[0]) * (((c >> 1) & 1) ? c3[1] : 1.f - c3[1]) *
                      ((c & 1) ? c3[2] : 1.f - c3[2]);
      const float m = mp[((long)xx * Y + yy) * Z + zz];
      s = fmaf(w, 1.0f / (1.0f + expf(-m)), s);
    }
    const float* cl = cls + ((long)b * Q + qi) * (K + 1);
    float mx = cl[0];
    for (int i = 1; i <= K; ++i) mx = fmaxf(mx, cl[i]);
    float sum = 0.f;
    for (int i = 0; i <= K; ++i) sum += expf(cl[i] - mx);
#pragma unroll
    for (int i = 0; i < UC_MAXK; ++i)
      if (i < K) acc[i] = fmaf(expf(cl[i] - mx) / sum, s, acc[i]);
  }
#pragma unroll
  for (int i = 0; i < UC_MAXK; ++i) {
    if (i < K) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
    }
  }
  if (lane == 0) {
    float mx = acc[0];
    for (int i = 1; i < K; ++i) mx = fmaxf(mx, acc[i]);
    float sum = 0.f;
    for (int i = 0; i < K; ++i) sum += expf(acc[i] - mx);
    for (int i = 0; i < K; ++i) out[p * K + i] = expf(acc[i] - mx) / sum;
  }
}

extern "C" int occf_lidarseg_sample_fwd(const float* mask_pred, const float* cls, const float* pts,
                                        float* out, int P, int B, int Q, int K, int X, int Y, int Z,
                                        void* stream) {
  if (P < 0 || B <= 0 || Q <= 0 || K <= 0 || K > UC_MAXK) return OCCF_ESHAPE;
  if (P == 0) return 0;
  hipLaunchKernelGGL(lidarseg_sample_kernel, dim3(occf_cdiv((long)P * 64, 256)), dim3(256), 0,
                     (hipStream_t)stream, mask_pred, cls, pts, out, P, Q, K, X, Y, Z);
  OCCF_LAUNCH_CHECK();
}
