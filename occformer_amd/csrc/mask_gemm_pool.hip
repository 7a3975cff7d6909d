// Fused  mask_pred = mask_embed . mask_feature  ->  adaptive 3-D max-pool  (preserve-pooling)
// for the decoder layers whose full-resolution mask logits are never consumed.
//
// Reference: projects/mmdet3d_plugin/occformer/mask2former/mask2former_nusc_occ.py:448-466
//   (einsum 'bqc,bcxyz->bqxyz', F.adaptive_max_pool3d, sigmoid < 0.5).  In `simple_test` only the
//   LAST layer's mask_pred is used (:713-731); the other nine exist only to be pooled into the next
//   layer's attention mask.  The reference still writes and re-reads each of them (10 x 2 x 256 MB
//   at the 200-grid); here the GEMM epilogue pools its 128-voxel tile and the [B,Q,X,Y,Z] tensor of
//   those layers is never written.
//
// GEMM: rows = queries (<= 128, one M tile), columns = voxels (tile of 128 consecutive voxels in
// channels-last order = whole (y, z) rows), K = E.  Same split-bf16 MFMA core and LDS layout as
// gemm_bf16.hip; the voxel features arrive pre-split (they are the shared "weight" of all ten
// contractions).  Epilogue: tile -> LDS [q][voxel]; each thread folds, for one query, the
// voxels of one pooling-cell column and merges into the global pooled logits with an ordered-int
// atomicMax (max is order independent -> deterministic).  A second tiny kernel decodes the pooled
// logits into the blocked bytes / row_open flags of occf_mask_pool_fwd.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define MG_BK 32

struct MaskPoolArgs {
  const float* me;            // [B, Q, E]
  const uint16_t* Fh;         // [B, V, E]
  const uint16_t* Fl;
  int* pooled_enc;            // [B, Q, L] ordered-int encoded maxima
  int B, Q, E;
  int X, Y, Z, ox, oy, oz;
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
  const uint32_t ha = mg_bf16_rne(a), hb = mg_bf16_rne(b);
  const uint32_t la = mg_bf16_rne(a - mg_bf16_up(ha)), lb = mg_bf16_rne(b - mg_bf16_up(hb));
  hi = ha | (hb << 16);
  lo = la | (lb << 16);
}
__device__ __forceinline__ int mg_slot(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }
// monotone float -> int map (for atomicMax on floats) and its inverse
__device__ __forceinline__ int mg_enc(float f) {
#ifdef OCCF_EMU
  int i;
  memcpy(&i, &f, 4);
#else
  const int i = __float_as_int(f);
#endif
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float mg_dec(int i) {
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
#ifdef OCCF_EMU
  float f;
  memcpy(&f, &j, 4);
  return f;
#else
  return __int_as_float(j);
#endif
}
// adaptive pooling: cell c covers [floor(c*in/out), ceil((c+1)*in/out)); the cells covering input
// index i form the range [floor(i*out/in), ceil((i+1)*out/in) - 1] (one cell when out | in).
__device__ __forceinline__ void mg_cells(int i, int in, int out, int& c0, int& c1) {
  c0 = (int)(((long)i * out) / in);
  c1 = (int)((((long)i + 1) * out + in - 1) / in) - 1;
  if (c1 > out - 1) c1 = out - 1;
}

struct mg_u4 {
  uint32_t x, y, z, w;
};

template <int TERMS>
__global__ void __launch_bounds__(256) mask_gemm_pool_kernel(MaskPoolArgs p) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  unsigned char* Ah = lds;                 // [128][64 B]
  unsigned char* Al = lds + 8192;
  unsigned char* Bh = lds + 16384;
  unsigned char* Bl = lds + 24576;
  float* tile = (float*)lds;               // [128 q][128 vox] after the K loop (aliases the operands)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long V = (long)p.X * p.Y * p.Z;
  const int b = blockIdx.y;
  const long n0 = (long)occf_xcd_remap(blockIdx.x, gridDim.x) * 128;

  int a_m[4], a_kq[4];
  bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    a_m[i] = idx >> 3;
    a_kq[i] = idx & 7;
    a_ok[i] = a_m[i] < p.Q;
  }
  int b_n[2], b_slot[2];
  bool b_ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 256;
    b_n[i] = idx >> 2;
    b_slot[i] = idx & 3;
    b_ok[i] = n0 + b_n[i] < V;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lk = lane >> 5;
  const int nk = p.E / MG_BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * MG_BK;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (a_ok[i]) v = *(const float4*)(p.me + ((long)b * p.Q + a_m[i]) * p.E + k0 + a_kq[i] * 4);
      uint32_t h0, l0, h1, l1;
      mg_split2(v.x, v.y, h0, l0);
      mg_split2(v.z, v.w, h1, l1);
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
      mg_u4 vh = {0, 0, 0, 0}, vl = {0, 0, 0, 0};
      if (b_ok[i]) {
        const long o = ((long)b * V + n0 + b_n[i]) * p.E + k0 + b_slot[i] * 8;
        vh = *(const mg_u4*)(p.Fh + o);
        if (TERMS == 3) vl = *(const mg_u4*)(p.Fl + o);
      }
      const int off = mg_slot(b_n[i], b_slot[i]);
      *(mg_u4*)(Bh + off) = vh;
      if (TERMS == 3) *(mg_u4*)(Bl + off) = vl;
    }
    __syncthreads();
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
  }
  __syncthreads();
  // ---- tile -> LDS [q][voxel]; columns rotated by the row so that lanes walking the queries
  //      (stride 128 floats) hit distinct banks
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        tile[row * 128 + ((wn * 64 + j * 32 + li + row) & 127)] = acc[i][j][r];
      }
  __syncthreads();
  const long L = (long)p.ox * p.oy * p.oz;
  if (p.Z <= 128 && 128 % p.Z == 0) {
    // tile = 128/Z complete z-rows.  Work item = (query, z-cell): walk the rows, keep a running
    // max while the (x, y) cell range is unchanged, flush with one ordered-int atomicMax per cell.
    const int rows = 128 / p.Z;
    for (int w = tid; w < p.Q * p.oz; w += 256) {
      const int q = w % p.Q, cz = w / p.Q;
      const int z0 = (int)(((long)cz * p.Z) / p.oz), z1 = (int)((((long)cz + 1) * p.Z + p.oz - 1) / p.oz);
      int* dst = p.pooled_enc + ((long)b * p.Q + q) * L + cz;
      float m = -INFINITY;
      int pcx0 = -1, pcx1 = -1, pcy0 = -1, pcy1 = -1;
      for (int rr = 0; rr <= rows; ++rr) {
        int cx0 = -2, cx1 = -2, cy0 = -2, cy1 = -2;
        const long v0 = n0 + (long)rr * p.Z;
        const bool valid = rr < rows && v0 < V;
        if (valid) {
          mg_cells((int)(v0 / ((long)p.Z * p.Y)), p.X, p.ox, cx0, cx1);
          mg_cells((int)((v0 / p.Z) % p.Y), p.Y, p.oy, cy0, cy1);
        }
        if (pcx0 >= 0 && (cx0 != pcx0 || cx1 != pcx1 || cy0 != pcy0 || cy1 != pcy1)) {
          const int e = mg_enc(m);
          for (int cx = pcx0; cx <= pcx1; ++cx)
            for (int cy = pcy0; cy <= pcy1; ++cy) atomicMax(dst + ((long)cx * p.oy + cy) * p.oz, e);
          m = -INFINITY;
        }
        if (!valid) break;
        pcx0 = cx0; pcx1 = cx1; pcy0 = cy0; pcy1 = cy1;
        for (int z = z0; z < z1; ++z) m = fmaxf(m, tile[q * 128 + ((rr * p.Z + z + q) & 127)]);
      }
    }
  } else {
    for (int w = tid; w < p.Q * 128; w += 256) {       // generic geometry: one voxel at a time
      const int q = w % p.Q, c = w / p.Q;
      const long v = n0 + c;
      if (v >= V) continue;
      const int z = (int)(v % p.Z), y = (int)((v / p.Z) % p.Y), x = (int)(v / ((long)p.Z * p.Y));
      int cx0, cx1, cy0, cy1, cz0, cz1;
      mg_cells(x, p.X, p.ox, cx0, cx1);
      mg_cells(y, p.Y, p.oy, cy0, cy1);
      mg_cells(z, p.Z, p.oz, cz0, cz1);
      const int e = mg_enc(tile[q * 128 + ((c + q) & 127)]);
      for (int cx = cx0; cx <= cx1; ++cx)
        for (int cy = cy0; cy <= cy1; ++cy)
          for (int cz = cz0; cz <= cz1; ++cz)
            atomicMax(p.pooled_enc + ((long)b * p.Q + q) * L + ((long)cx * p.oy + cy) * p.oz + cz, e);
    }
  }
}

__global__ void __launch_bounds__(256) mask_pool_fill_kernel(int* __restrict__ enc, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) enc[i] = (int)0x80000000;
}

// decode pooled logits in place (int -> float) and emit the blocked bytes; one workgroup per
// (batch, query) row also reduces "any key open" for the all-masked-row fix (no atomics).
__global__ void __launch_bounds__(256) mask_pool_decode_kernel(int* __restrict__ enc, uint8_t* __restrict__ blocked,
                                                               int* __restrict__ row_open, long L) {
  __shared__ int any_open;
  const long row = blockIdx.x;
  if (threadIdx.x == 0) any_open = 0;
  __syncthreads();
  bool open = false;
  for (long c = threadIdx.x; c < L; c += blockDim.x) {
    const long i = row * L + c;
    const float m = mg_dec(enc[i]);
    ((float*)enc)[i] = m;
    const float sg = 1.0f / (1.0f + expf(-m));
    const bool blk = sg < 0.5f;
    blocked[i] = blk ? 1 : 0;
    open |= !blk;
  }
  if (open) any_open = 1;          // benign race: every writer stores the same value
  __syncthreads();
  if (threadIdx.x == 0) row_open[row] = any_open;
}

extern "C" int occf_mask_gemm_pool_fwd(const float* mask_embed, const uint16_t* feat_hi, const uint16_t* feat_lo,
                                       float* pooled, uint8_t* blocked, int32_t* row_open, int B, int Q, int E,
                                       int X, int Y, int Z, int ox, int oy, int oz, int terms, void* stream) {
  if (B <= 0 || Q <= 0 || Q > 128 || E % MG_BK != 0) return OCCF_ESHAPE;
  if (ox <= 0 || oy <= 0 || oz <= 0 || ox > X || oy > Y || oz > Z) return OCCF_ESHAPE;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  if (terms == 3 && feat_lo == nullptr) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const long V = (long)X * Y * Z, L = (long)ox * oy * oz;
  const long BQL = (long)B * Q * L;
  hipLaunchKernelGGL(mask_pool_fill_kernel, dim3(occf_cdiv(BQL, 256)), dim3(256), 0, st, (int*)pooled, BQL);
  MaskPoolArgs a = {mask_embed, feat_hi, feat_lo, (int*)pooled, B, Q, E, X, Y, Z, ox, oy, oz};
  const dim3 grid((unsigned)occf_cdiv(V, 128), B);
  if (terms == 3) hipLaunchKernelGGL(mask_gemm_pool_kernel<3>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(mask_gemm_pool_kernel<1>, grid, dim3(256), 0, st, a);
  hipLaunchKernelGGL(mask_pool_decode_kernel, dim3((unsigned)(B * Q)), dim3(256), 0, st, (int*)pooled, blocked,
                     (int*)row_open, L);
  OCCF_LAUNCH_CHECK();
}
