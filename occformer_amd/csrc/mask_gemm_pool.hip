// Fused  mask_pred = mask_embed . mask_feature  ->  adaptive 3-D max-pool  (preserve-pooling)
// for the decoder layers whose full-resolution mask logits are never consumed.
//
// Reference: projects/mmdet3d_plugin/occformer/mask2former/mask2former_nusc_occ.py:448-466
//   (einsum 'bqc,bcxyz->bqxyz', F.adaptive_max_pool3d, sigmoid < 0.5).  In `simple_test` only the
//   LAST layer's mask_pred is used (:713-731); the other nine exist only to be pooled into the next
//   layer's attention mask.  The reference still writes and re-reads each of them (10 x 2 x 256 MB
//   at the 200-grid); here the [B,Q,X,Y,Z] tensor of those layers is never written.
//
// Geometry (uniform windows: ox | X, oy | Y, oz | Z, 128 % Z == 0, (128/Z) % (Y/oy) == 0, Y % (128/Z) == 0):
// a GEMM column tile = 128 consecutive voxels of the channels-last volume = RY = 128/Z complete z-rows of
// one x-plane, i.e. whole (y, z) pooling windows.  A workgroup owns (batch, x-window slice, y-tile): it
// walks the x-planes of its slice, multiplies [Q <= 128] x [128 voxels] per plane (split-bf16 MFMA core
// and LDS layout of gemm_bf16.hip; the voxel features arrive pre-split, they are the shared "weight" of
// all ten contractions), drops the tile into LDS and folds it into per-thread running window maxima.
// No atomics: one writer per (query, window[, slice]); a second small kernel takes the max over the
// slices, emits the blocked bytes and the row_open flags of occf_mask_pool_fwd.
// Global loads are unconditional and run PF k-tiles ahead in a register ring (see gemm_bf16.hip).
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define MG_BK 32

struct MaskPoolArgs {
  const float* me;            // [B, Q, E]
  const uint16_t* Fh;         // [B, V, E]
  const uint16_t* Fl;
  float* part;                // [S][B][Q][L] window maxima of every x-slice of the pooling windows
  int B, Q, E;
  int X, Y, Z, ox, oy, oz;
  int S;                      // x-planes of a window are split over S workgroups (load balance)
};

__device__ __forceinline__ uint32_t mg_bf16_rne(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
#else
  const uint32_t u = __float_as_uint(x);
#endif
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float mg_bf16_up(uint32_t h) {
#ifdef OCCF_EMU
  uint32_t u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(h << 16);
#endif
}
__device__ __forceinline__ void mg_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  occf_bf16_split2(a, b, hi, lo);
}
__device__ __forceinline__ int mg_slot(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float occf_nanmax_mg(float m, float v) { return (v > m || v != v) ? v : m; }

typedef uint32_t mg_u4 __attribute__((ext_vector_type(4)));

#define MG_PF 2

template <int TERMS>
__global__ void __launch_bounds__(256) mask_gemm_pool_kernel(MaskPoolArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  unsigned char* Ah = lds;                 // [128][64 B]
  unsigned char* Al = lds + 8192;
  unsigned char* Bh = lds + 16384;
  unsigned char* Bl = lds + 24576;
  float* tile = (float*)lds;               // [128 q][128 vox] after a plane's K loop (aliases the operands)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;
  const long V = (long)p.X * p.Y * p.Z;
  const int RY = 128 / p.Z;                       // y-rows per tile
  const int wx = p.X / p.ox, wy = p.Y / p.oy, wz = p.Z / p.oz;
  const int ytiles = p.Y / RY;
  const int ncy = RY / wy, ncz = p.oz;            // windows per tile along y, z
  const int xs = wx / p.S;                        // x-planes per workgroup
  // workgroup -> (b, cx, slice, ty)
  unsigned wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const int ty = wg % ytiles; wg /= ytiles;
  const int sl = wg % p.S; wg /= p.S;
  const int cx = wg % p.ox;
  const int b = wg / p.ox;
  const int x0 = cx * wx + sl * xs;

  // A (mask_embed rows; rows >= Q read row Q-1, their outputs are never folded) and B (voxel rows)
  int a_m[4], a_kq[4];
  long a_base[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    a_m[i] = idx >> 3;
    a_kq[i] = idx & 7;
    a_base[i] = ((long)b * p.Q + (a_m[i] < p.Q ? a_m[i] : p.Q - 1)) * p.E + a_kq[i] * 4;
  }
  int b_n[2], b_slot[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    b_n[i] = idx >> 2;
    b_slot[i] = idx & 3;
  }
  const int nk = p.E / MG_BK;
  const int F = xs * nk;                          // flat (plane, k-tile) stream
  float4 ra[MG_PF][4];
  mg_u4 rbh[MG_PF][2], rbl[MG_PF][2];
  auto load_tile = [&](int f, int d) __attribute__((always_inline)) {
    const int fc = f < F ? f : F - 1;
    const int xi = fc / nk, kt = fc - xi * nk;
    const int k0 = kt * MG_BK;
    const long n0 = ((long)(x0 + xi) * p.Y + (long)ty * RY) * p.Z;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[d][i] = *(const float4*)(p.me + a_base[i] + k0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long o = ((long)b * V + n0 + b_n[i]) * p.E + k0 + b_slot[i] * 8;
      rbh[d][i] = *(const mg_u4*)(p.Fh + o);
      if (TERMS == 3) rbl[d][i] = *(const mg_u4*)(p.Fl + o);
    }
  };
  auto store_tile = [&](int d) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      uint32_t h0, l0, h1, l1;
      mg_split2(ra[d][i].x, ra[d][i].y, h0, l0);
      mg_split2(ra[d][i].z, ra[d][i].w, h1, l1);
      const int off = mg_slot(a_m[i], a_kq[i] >> 1) + (a_kq[i] & 1) * 8;
      *(uint32_t*)(Ah + off) = h0;
      *(uint32_t*)(Ah + off + 4) = h1;
      if (TERMS == 3) {
        *(uint32_t*)(Al + off) = l0;
        *(uint32_t*)(Al + off + 4) = l1;
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int off = mg_slot(b_n[i], b_slot[i]);
      *(mg_u4*)(Bh + off) = rbh[d][i];
      if (TERMS == 3) *(mg_u4*)(Bl + off) = rbl[d][i];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // max over the x-planes commutes with the (y, z) window max: the planes are folded element-wise in
  // registers and the window reduction runs ONCE per workgroup
  f32x16 pmax[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) pmax[i][j][r] = -INFINITY;
  const int n_items = p.Q * ncy * ncz;            // (query, window) pairs of this tile column

#pragma unroll
  for (int d = 0; d < MG_PF; ++d) load_tile(d, d);
  int kt = 0;
  for (int f0 = 0; f0 < F; f0 += MG_PF) {
#pragma unroll
    for (int d = 0; d < MG_PF; ++d) {
      const int f = f0 + d;
      if (f < F) {
        __syncthreads();                           // previous fragment reads are done
        store_tile(d);
        __syncthreads();
        load_tile(f + MG_PF, d);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kslot = s * 2 + lk;
          bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int off = mg_slot(wm * 64 + i * 32 + li, kslot);
            ah[i] = *(const bf16x8*)(Ah + off);
            if (TERMS == 3) al[i] = *(const bf16x8*)(Al + off);
          }
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int off = mg_slot(wn * 64 + j * 32 + li, kslot);
            bh[j] = *(const bf16x8*)(Bh + off);
            if (TERMS == 3) bl[j] = *(const bf16x8*)(Bl + off);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (TERMS == 3) {
                acc[i][j] = occf_mfma_bf16_32x32x16(al[i], bh[j], acc[i][j]);
                acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bl[j], acc[i][j]);
              }
              acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bh[j], acc[i][j]);
            }
        }
        if (++kt == nk) {                          // one x-plane done
          kt = 0;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                pmax[i][j][r] = occf_nanmax_mg(pmax[i][j][r], acc[i][j][r]);
                acc[i][j][r] = 0.f;
              }
        }
      }
    }
  }
  // ---- plane maxima -> LDS [q][voxel] (columns rotated by the row so that lanes walking the queries hit
  //      distinct banks), then one thread per (query, window) folds its wy x wz voxels
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        tile[row * 128 + ((wn * 64 + j * 32 + li + row) & 127)] = pmax[i][j][r];
      }
  __syncthreads();
  const long L = (long)p.ox * p.oy * p.oz;
  for (int w = tid; w < n_items; w += 256) {
    const int q = w % p.Q, c = w / p.Q;
    const int cz = c % ncz, cyl = c / ncz;
    float m = -INFINITY;
    for (int dy = 0; dy < wy; ++dy)
      for (int dz = 0; dz < wz; ++dz)
        m = occf_nanmax_mg(m, tile[q * 128 + (((cyl * wy + dy) * p.Z + cz * wz + dz + q) & 127)]);
    p.part[(((long)sl * p.B + b) * p.Q + q) * L + ((long)cx * p.oy + ty * ncy + cyl) * p.oz + cz] = m;
  }
}

// ---------------------------------------------------------------------------------------
// Streaming variant (E = 16 * NS channels, Z a power of two <= 32, power-of-two windows with wy a multiple
// of the 32/Z rows of a voxel group): the tile kernel above re-stages the (constant) mask embeddings for
// every k-tile and synchronises twice per k-tile -- it reached 2.3 TB/s on a contraction whose only large
// operand (the pre-split voxel features, 492 MB at the 200-grid) has to be read exactly once.  Here the
// problem is transposed: D[voxel][query] = F[voxel][k] * M^T[k][query].
//   * M^T lives in LDS for the whole launch as ready-made B fragments (hi and lo, 4 query tiles x NS k-slots
//     x 64 lanes x 16 B = 96 KB for E = 192); workgroups are persistent (one per CU).
//   * every wave streams its own 32-voxel groups: the A fragments are the voxels' bf16 rows read straight
//     from global memory in fragment order (lane = voxel row, 16 B = 8 channels), one whole group
//     (2 * NS loads of 16 B per lane) ahead in registers -- no LDS staging, no barriers in the main loop.
//   * a wave owns complete pooling windows: the groups of a unit (the x-planes and y-row groups of one
//     (x-window slice, y-window)) are folded element-wise into a running maximum; the z (and in-group y)
//     reduction happens across the accumulator registers (+ one lane swap for z bit 2).
struct MaskStreamArgs {
  const float* me;            // [B, Q, E]
  const uint16_t* Fh;         // [B, V, E]
  const uint16_t* Fl;
  float* part;                // [S][B][Q][L]
  int B, Q, E;
  int X, Y, Z, ox, oy, oz;
  int S;                      // x-slices per window
  int units;                  // per batch: ox * oy * S
  int reverse;                // walk the units from the far end (see occf_mask_gemm_pool_fwd)
  float* pooled;              // S == 1: final outputs written here, no finish kernel
  uint8_t* blocked;
  int* row_open;
};

template <int TERMS, int NS>
__global__ void __launch_bounds__(256) mask_gemm_pool_stream_kernel(MaskStreamArgs p) {
  OCCF_DYN_SMEM(smem);
  mg_u4* Mh = reinterpret_cast<mg_u4*>(smem);                         // [4][NS][64] fragments
  mg_u4* Ml = reinterpret_cast<mg_u4*>(smem + 4 * NS * 64 * 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const long V = (long)p.X * p.Y * p.Z;
  // ---- stage the mask embeddings as B fragments: entry (nt, s, l) = M[nt*32 + (l&31)][16 s + 8 (l>>5) .. +7]
  for (int e = tid; e < 4 * NS * 64; e += 256) {
    const int l = e & 63, s = (e >> 6) % NS, nt = e / (64 * NS);
    const int q = nt * 32 + (l & 31);
    const float* src = p.me + ((long)b * p.Q + (q < p.Q ? q : p.Q - 1)) * p.E + 16 * s + 8 * (l >> 5);
    const float4 v0 = *(const float4*)src, v1 = *(const float4*)(src + 4);
    mg_u4 h, lo;
    uint32_t hh, ll;
    mg_split2(v0.x, v0.y, hh, ll); h[0] = hh; lo[0] = ll;
    mg_split2(v0.z, v0.w, hh, ll); h[1] = hh; lo[1] = ll;
    mg_split2(v1.x, v1.y, hh, ll); h[2] = hh; lo[2] = ll;
    mg_split2(v1.z, v1.w, hh, ll); h[3] = hh; lo[3] = ll;
    Mh[e] = h;
    if (TERMS == 3) Ml[e] = lo;
  }
  __syncthreads();

  const int wx = p.X / p.ox, wy = p.Y / p.oy, wz = p.Z / p.oz;
  const int GY = 32 / p.Z;                        // y-rows of a 32-voxel group
  const int xs = wx / p.S, ygroups = wy / GY;
  const int gpu = xs * ygroups;                   // groups per unit
  const int nwaves = gridDim.x * 4;
  const int wid = blockIdx.x * 4 + wave;
  const int my_units = wid < p.units ? (p.units - wid + nwaves - 1) / nwaves : 0;
  const long T = (long)my_units * gpu;            // this wave's flat (unit, group) stream
  // voxel-row byte-free element offset of the lane inside a group: row (l & 31), channel half (l >> 5)
  const long lane_off = (long)(lane & 31) * p.E + 8 * (lane >> 5);
  auto group_base = [&](long t) -> long {         // element offset of the group's first voxel row
    const long tc = t < T ? t : T - 1;
    const int uf = wid + (int)(tc / gpu) * nwaves, g = (int)(tc % gpu);
    const int u = p.reverse ? p.units - 1 - uf : uf;
    const int sl = u % p.S, cy = (u / p.S) % p.oy, cx = u / (p.S * p.oy);
    const int x = cx * wx + sl * xs + g / ygroups, y = cy * wy + (g % ygroups) * GY;
    return ((long)b * V + ((long)x * p.Y + y) * p.Z) * p.E;
  };
  // A fragments of ONE group; slot s is refilled with the next group's slot s as soon as its MFMAs are
  // issued, so a whole group of loads (2 * NS x 16 B per lane) is always in flight
  mg_u4 ah[NS], al[NS];
  f32x16 pm[4];
  mg_u4 bq[2][8];                                 // embedding fragments (4 query tiles x {hi, lo}) of two k-slots
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    bq[0][nt] = Mh[(nt * NS) * 64 + lane];
    if (TERMS == 3) bq[0][4 + nt] = Ml[(nt * NS) * 64 + lane];
  }
  // reduced voxel bits: v = (r & 3) + 4 (lane >> 5) + 8 (r >> 2); z = v % Z, in-group row = v / Z
  int lwz = 0;
  while ((1 << lwz) < wz) ++lwz;
  const int red = ((1 << lwz) - 1) | (31 & ~(p.Z - 1));      // z bits below the window size + all row bits
  const long L = (long)p.ox * p.oy * p.oz;

  auto run = [&](long t) __attribute__((always_inline)) {
    const int g = (int)(t % gpu);
    const long onext = group_base(t + 1) + lane_off;          // past the end: re-reads the last group
    if (g == 0) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) pm[nt][r] = -INFINITY;
    }
    f32x16 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      // The embedding fragments are loop-invariant LDS reads: without the memory fence the compiler hoists all
      // 8 * NS of them into registers (384 VGPRs) and spills.  The fragments of slot s+1 are read while slot s
      // is multiplied (bq double buffer; slot 0 of the next group closes the ring), and the three split
      // products are issued term-major so that consecutive MFMAs write different accumulators (left alone,
      // the compiler emits read-wait-3 dependent MFMAs per query tile: LDS latency and MFMA latency in series).
      asm volatile("" ::: "memory");
      const int sn = s + 1 < NS ? s + 1 : 0;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        bq[(s + 1) & 1][nt] = Mh[(nt * NS + sn) * 64 + lane];
        if (TERMS == 3) bq[(s + 1) & 1][4 + nt] = Ml[(nt * NS + sn) * 64 + lane];
      }
      OCCF_SCHED_FENCE();
      bf16x8 a_h, a_l, b_h[4], b_l[4];
      __builtin_memcpy(&a_h, &ah[s], 16);
      if (TERMS == 3) __builtin_memcpy(&a_l, &al[s], 16);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        __builtin_memcpy(&b_h[nt], &bq[s & 1][nt], 16);
        if (TERMS == 3) __builtin_memcpy(&b_l[nt], &bq[s & 1][4 + nt], 16);
      }
      if (TERMS == 3) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = occf_mfma_bf16_32x32x16(a_h, b_l[nt], acc[nt]);   // feature_hi * embed_lo
        OCCF_SCHED_FENCE();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = occf_mfma_bf16_32x32x16(a_l, b_h[nt], acc[nt]);   // feature_lo * embed_hi
        OCCF_SCHED_FENCE();
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = occf_mfma_bf16_32x32x16(a_h, b_h[nt], acc[nt]);
      OCCF_SCHED_FENCE();
      ah[s] = *(const mg_u4*)(p.Fh + onext + 16 * s);
      if (TERMS == 3) al[s] = *(const mg_u4*)(p.Fl + onext + 16 * s);
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) pm[nt][r] = occf_nanmax_mg(pm[nt][r], acc[nt][r]);
    if (g == gpu - 1) {
      // ---- window reduction across the accumulator registers, then one value per (query, z-window)
      const int uf = wid + (int)(t / gpu) * nwaves;
      const int u = p.reverse ? p.units - 1 - uf : uf;
      const int sl = u % p.S, cy = (u / p.S) % p.oy, cx = u / (p.S * p.oy);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        if (red & 1) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) pm[nt][r] = occf_nanmax_mg(pm[nt][r], pm[nt][r + 1]);
        }
        if (red & 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!(r & 2)) pm[nt][r] = occf_nanmax_mg(pm[nt][r], pm[nt][r + 2]);
        }
        if (red & 8) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (!(r & 4)) pm[nt][r] = occf_nanmax_mg(pm[nt][r], pm[nt][r + 4]);
        }
        if (red & 16) {
#pragma unroll
          for (int r = 0; r < 8; ++r) pm[nt][r] = occf_nanmax_mg(pm[nt][r], pm[nt][r + 8]);
        }
        if (red & 4) {
#pragma unroll
          for (int r = 0; r < 16; ++r) pm[nt][r] = occf_nanmax_mg(pm[nt][r], __shfl_xor(pm[nt][r], 32));
        }
        const int q = nt * 32 + (lane & 31);
        if (q < p.Q) {
          const long cell0 = ((long)cx * p.oy + cy) * p.oz;
          if (p.S == 1) {
            // no x-slices to merge: the epilogue of mask_pool_finish_kernel right here (same expressions)
            const long row = (long)b * p.Q + q;
            bool open = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int v = (r & 3) + 4 * (lane >> 5) + 8 * (r >> 2);
              if ((v & red) == 0) {
                const long c = row * L + cell0 + ((v & (p.Z - 1)) >> lwz);
                const float m = pm[nt][r];
                p.pooled[c] = m;
                const float sg = 1.0f / (1.0f + expf(-m));
                const bool blk = sg < 0.5f;
                p.blocked[c] = blk ? 1 : 0;
                open |= !blk;
              }
            }
            if (open && p.row_open[row] == 0) atomicOr((unsigned*)&p.row_open[row], 1u);
          } else {
            float* dst = p.part + (((long)sl * p.B + b) * p.Q + q) * L + cell0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int v = (r & 3) + 4 * (lane >> 5) + 8 * (r >> 2);
              if ((v & red) == 0) dst[(v & (p.Z - 1)) >> lwz] = pm[nt][r];
            }
          }
        }
      }
    }
  };

  if (T > 0) {
    const long o0 = group_base(0) + lane_off;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      ah[s] = *(const mg_u4*)(p.Fh + o0 + 16 * s);
      if (TERMS == 3) al[s] = *(const mg_u4*)(p.Fl + o0 + 16 * s);
    }
    for (long t = 0; t < T; ++t) run(t);
  }
}

// max over the x-slices, blocked bytes, "any key open" per (batch, query) row (all-masked-row fix);
// grid = (rows, 2048-cell chunks); row_open is zeroed by the host and OR-ed (order independent)
__global__ void __launch_bounds__(256) mask_pool_finish_kernel(const float* __restrict__ part, float* __restrict__ pooled,
                                                               uint8_t* __restrict__ blocked, int* __restrict__ row_open,
                                                               long L, long BQ, int S) {
  const long row = blockIdx.x;
  const long c0 = (long)blockIdx.y * 2048;
  bool open = false;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const long c = c0 + u * 256 + threadIdx.x;
    if (c < L) {
      float m = part[row * L + c];
      for (int s = 1; s < S; ++s) m = occf_nanmax_mg(m, part[((long)s * BQ + row) * L + c]);
      pooled[row * L + c] = m;
      const float sg = 1.0f / (1.0f + expf(-m));            // as the reference: fp32 sigmoid, then compare
      const bool blk = sg < 0.5f;
      blocked[row * L + c] = blk ? 1 : 0;
      open |= !blk;
    }
  }
  if (__ballot(open) != 0 && (threadIdx.x & 63) == 0) atomicOr((unsigned*)&row_open[row], 1u);
}

static int mg_slices(int B, int X, int Y, int Z, int ox) {
  // split a window's x-planes over S workgroups until the grid has >= ~1000 workgroups
  const int wx = X / ox, ytiles = Y / (128 / Z);
  int S = 1;
  while (S * 2 <= wx && wx % (S * 2) == 0 && (long)B * ox * S * ytiles < 1000) S *= 2;
  return S;
}
static bool mg_geometry_ok(int Q, int E, int X, int Y, int Z, int ox, int oy, int oz) {
  if (Q <= 0 || Q > 128 || E % MG_BK != 0) return false;
  if (ox <= 0 || oy <= 0 || oz <= 0 || X % ox || Y % oy || Z % oz) return false;
  if (Z > 128 || 128 % Z) return false;
  const int RY = 128 / Z, wy = Y / oy;
  if (Y % RY || RY % wy) return false;
  return true;
}

static bool mg_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
// streaming kernel: E = 192, Z a power of two <= 32, power-of-two z windows, y windows made of whole groups
static bool mg_stream_ok(int Q, int E, int X, int Y, int Z, int ox, int oy, int oz) {
  static const bool on = [] {
    const char* e = getenv("OCCF_MASK_POOL_STREAM");       // diagnostics: 0 keeps the tile kernel
    return e == nullptr || atoi(e) != 0;
  }();
  if (!on || Q <= 0 || Q > 128 || E != 192) return false;
  if (ox <= 0 || oy <= 0 || oz <= 0 || X % ox || Y % oy || Z % oz) return false;
  if (!mg_pow2(Z) || Z > 32 || !mg_pow2(Z / oz)) return false;
  return (Y / oy) % (32 / Z) == 0;
}
static int mg_stream_slices(int B, int X, int Y, int Z, int ox, int oy) {
  // split a window's x-planes until every wave of the chip has a few units (load balance of the tail)
  const int wx = X / ox;
  int S = 1;
  while (S * 2 <= wx && wx % (S * 2) == 0 && (long)ox * oy * S < 4096) S *= 2;
  return S;
}

// floats of scratch for the per-slice window maxima; 0 when the geometry is not taken by the fused kernel
extern "C" long occf_mask_gemm_pool_workspace(int B, int Q, int E, int X, int Y, int Z, int ox, int oy, int oz) {
  if (B <= 0) return 0;
  if (mg_stream_ok(Q, E, X, Y, Z, ox, oy, oz)) return (long)mg_stream_slices(B, X, Y, Z, ox, oy) * B * Q * ox * oy * oz;
  if (!mg_geometry_ok(Q, E, X, Y, Z, ox, oy, oz)) return 0;
  return (long)mg_slices(B, X, Y, Z, ox) * B * Q * ox * oy * oz;
}

extern "C" int occf_mask_gemm_pool_fwd(const float* mask_embed, const uint16_t* feat_hi, const uint16_t* feat_lo,
                                       float* pooled, uint8_t* blocked, int32_t* row_open, float* workspace, int B,
                                       int Q, int E, int X, int Y, int Z, int ox, int oy, int oz, int terms,
                                       int reverse, void* stream) {
  const bool stream_k = B > 0 && mg_stream_ok(Q, E, X, Y, Z, ox, oy, oz);
  if (B <= 0 || (!stream_k && !mg_geometry_ok(Q, E, X, Y, Z, ox, oy, oz))) return OCCF_ESHAPE;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  if ((terms == 3 && feat_lo == nullptr) || workspace == nullptr) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long L = (long)ox * oy * oz;
  int S;
  if (stream_k) {
    S = mg_stream_slices(B, X, Y, Z, ox, oy);
    MaskStreamArgs a = {mask_embed, feat_hi, feat_lo, workspace, B, Q, E, X, Y, Z, ox, oy, oz, S, ox * oy * S,
                        reverse != 0, pooled, blocked, (int*)row_open};
    if (S == 1) {              // the kernel writes the final outputs and ORs row_open itself
#ifndef OCCF_EMU
      hipError_t e0 = hipMemsetAsync(row_open, 0, sizeof(int32_t) * (size_t)B * Q, st);
      if (e0 != hipSuccess) return (int)e0;
#else
      memset(row_open, 0, sizeof(int32_t) * (size_t)B * Q);
#endif
    }
    const size_t lds = (size_t)4 * 12 * 64 * 16 * (terms == 3 ? 2 : 1);
#ifndef OCCF_EMU
    static bool attr_set[2] = {};
    if (!attr_set[terms == 3]) {
      const void* fn = terms == 3 ? (const void*)mask_gemm_pool_stream_kernel<3, 12>
                                  : (const void*)mask_gemm_pool_stream_kernel<1, 12>;
      hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return (int)e;
      attr_set[terms == 3] = true;
    }
#endif
    const int nblk = a.units >= 1024 ? 256 : occf_cdiv(a.units, 4);      // persistent: one workgroup per CU
    if (terms == 3)
      hipLaunchKernelGGL((mask_gemm_pool_stream_kernel<3, 12>), dim3(nblk, B), dim3(256), lds, st, a);
    else
      hipLaunchKernelGGL((mask_gemm_pool_stream_kernel<1, 12>), dim3(nblk, B), dim3(256), lds, st, a);
    if (S == 1) OCCF_LAUNCH_CHECK();
  } else {
    S = mg_slices(B, X, Y, Z, ox);
    MaskPoolArgs a = {mask_embed, feat_hi, feat_lo, workspace, B, Q, E, X, Y, Z, ox, oy, oz, S};
    const long blocks = (long)B * ox * S * (Y / (128 / Z));
    if (blocks >= 2147483647L) return OCCF_ESHAPE;
    if (terms == 3) hipLaunchKernelGGL(mask_gemm_pool_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(mask_gemm_pool_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, a);
  }
#ifndef OCCF_EMU
  hipError_t e = hipMemsetAsync(row_open, 0, sizeof(int32_t) * (size_t)B * Q, st);
  if (e != hipSuccess) return (int)e;
#else
  memset(row_open, 0, sizeof(int32_t) * (size_t)B * Q);
#endif
  hipLaunchKernelGGL(mask_pool_finish_kernel, dim3((unsigned)(B * Q), (unsigned)occf_cdiv(L, 2048)), dim3(256), 0, st,
                     workspace, pooled, blocked, (int*)row_open, L, (long)B * Q, S);
  OCCF_LAUNCH_CHECK();
}
