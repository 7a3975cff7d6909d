// Weight-gradient contraction ("TN" GEMM) for gfx950: dW[n, tap*Cin + c] = sum_m dY[m, n] * X_tap[m, c]
// -- the weight gradient of every nn.Linear / 1^3 / 3^3 / dilated / strided convolution of the path
// (ATen autograd in the reference: occupancyformer.py:132-199 -> loss.backward()).  Optionally also
// db[n] = sum_m dY[m, n] from the same pass over dY.
//
// Both operands are "k-major": the contraction index m (voxels / tokens, up to 680 000) is the SLOW index of
// the two row-major activations, the fast index is the channel.  v_mfma_f32_32x32x16_bf16 wants 8 consecutive
// k per lane, so the staging transposes in REGISTERS on the way into LDS: a thread loads float4 from 8 (or 4)
// consecutive rows, packs the 8 row values of each of its 4 columns to bf16x8 (hi and lo halves of the 3-term split)
// and writes them as 16-byte (8-byte) pieces; LDS holds [m / 8][channel] 16-byte groups, so a fragment is ONE
// ds_read_b128 with consecutive lanes on consecutive 16-byte slots -- both directions bank-conflict free without
// padding (four ds_read_b32 per fragment made the kernel LDS-issue bound).
//
// Workgroup = 4 waves (2 x 2), tile 128 (n) x BC (c, 128 or 64) of ONE tap, walks 64-row chunks of its M-slice
// with the next chunk's global loads in flight (issued unconditionally, clamped).  M is split over blockIdx.y
// into slabs reduced in fixed order (deterministic).  fp32 accumulate; terms = 3: hi*hi + hi*lo + lo*hi.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

struct WgGeom {
  int B, Xo, Yo, Zo, Xi, Yi, Zi, kX, kY, kZ, stride, dil, pad_x, pad_y, pad_z;
  long sb, sx, sy, sz;
};
struct WgArgs {
  const float* dY;
  const float* X;
  float* out;        // [S][N][taps * Cin]
  float* bias_out;   // [S][N] or NULL
  long M;
  int N, Cin, taps;
  long ldy, ldx;
  int conv;
  int swz;           // 7 = XOR-swizzled LDS slots, 0 = plain (diagnostic switch OCCF_WG_SWZ)
  // pre-split operands (bf16 hi / lo arrays, same shapes / strides in ELEMENTS as dY / X): the 27 taps of a
  // convolution stage the same rows 27 x (n, c tiles) times -- splitting once up front takes the fp32 -> (hi, lo)
  // conversion (4 VALU per element, 60 % of the kernel's issue slots by PMC) out of the loop and halves its loads
  const uint16_t* dYh;
  const uint16_t* dYl;
  const uint16_t* Xh;
  const uint16_t* Xl;
  long rows_per_split;
  // zfast (pre-split convolutions with Zo in {8, 16, 32, 64}): the 8 consecutive rows a staging thread owns are 8
  // consecutive z of ONE (b, x, y) column, so the voxel decode, the x / y bounds and the base address are per
  // thread and chunk, not per row, and an out-of-range row reads a 16-byte zero constant instead of being masked
  // after the load.  PMC on the 192 -> 192 convolution before: 14.9 VALU instructions per MFMA, VALU issue = 50 % of
  // all SIMD cycles (MFMA 27 %) -- the address arithmetic and the 4-dword selects of 16 loads per thread and chunk
  int zfast;
  int zshift;          // log2(Zo)
  // xcd_tiles > 0: 1-D grid; workgroup w runs on XCD w % 8 (round-robin dispatch), so ALL (tap, tile) workgroups of one
  // M-slab are placed on ONE XCD (slab = xcd + 8 * (k / xcd_tiles), tile = k % xcd_tiles with k = w / 8): the slab's
  // dY / X rows are then fetched from HBM once into that XCD's L2 and shared by its 27 x tiles workgroups.  With the
  // slabs spread round-robin over all XCDs every XCD streamed every slab: 21 GB of HBM fetch for 0.99 GB of operands on
  // the 192 -> 192 convolution (PMC FETCH_SIZE), i.e. the kernel ran at the HBM roof, not the MFMA roof.
  int xcd_tiles;
  int n_slabs;
  WgGeom g;
};

__device__ __attribute__((aligned(16))) const uint32_t wg_zero16[4] = {0u, 0u, 0u, 0u};

typedef uint32_t wg_u4 __attribute__((ext_vector_type(4)));

// 16-byte LDS slot of column `col` inside a row group: XOR swizzle that makes BOTH access patterns hit eight
// distinct 16-byte bank groups per eight lanes -- the staging writes (a thread owns 4 consecutive columns, so one
// store instruction covers columns 4*lane + e: stride 4) and the fragment reads (32 consecutive columns).  Unswizzled,
// every staging store was a 4-way bank conflict.
__device__ __forceinline__ int wg_swz(int col, int on) { return col ^ (((col >> 3) & 7) & on); }

// bf16 pair packing for the pre-split path: a dword holds two ADJACENT COLUMNS of one row; the LDS group wants two
// ADJACENT ROWS of one column.  lo16(a) | lo16(b) << 16 and hi16(a) | hi16(b) << 16 (v_perm_b32 on the GPU).
__device__ __forceinline__ uint32_t wg_pack_lo(uint32_t a, uint32_t b) {
#ifdef OCCF_EMU
  return (a & 0xFFFFu) | (b << 16);
#else
  return __builtin_amdgcn_perm(b, a, 0x05040100u);
#endif
}
__device__ __forceinline__ uint32_t wg_pack_hi(uint32_t a, uint32_t b) {
#ifdef OCCF_EMU
  return (a >> 16) | (b & 0xFFFF0000u);
#else
  return __builtin_amdgcn_perm(b, a, 0x07060302u);
#endif
}

template <int BC, int TERMS, bool PRE>
__global__ void __launch_bounds__(256) wgrad_kernel(WgArgs p) {
  constexpr int TC = BC / 64;                 // 32-wide tiles per wave along c
  constexpr int QB = BC / 4;                  // channel quads of the X tile
  constexpr int RPT = 64 * QB / 256;          // rows per thread of the X tile (8 or 4)
  // [8 row groups][columns] of 16-byte groups (8 bf16 = 8 consecutive rows of one column)
  __shared__ __attribute__((aligned(16))) wg_u4 Ah[8 * 128], Al[8 * 128], Bh[8 * BC], Bl[8 * BC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c_tiles = (p.Cin + BC - 1) / BC;
  int bx = blockIdx.x;
  int slab = blockIdx.y;
  if (p.xcd_tiles > 0) {
    const int xcd = (int)(blockIdx.x & 7u), k = (int)(blockIdx.x >> 3);
    slab = xcd + 8 * (k / p.xcd_tiles);
    bx = k % p.xcd_tiles;
    if (slab >= p.n_slabs) return;                 // (uniform over the workgroup; only when n_slabs % 8 != 0)
  }
  const int ct = bx % c_tiles;
  bx /= c_tiles;
  const int tap = bx % p.taps;
  const int nt = bx / p.taps;
  const int n0 = nt * 128, c0 = ct * BC;
  const long m_begin = (long)slab * p.rows_per_split;
  long m_end = m_begin + p.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nchunks = m_end > m_begin ? (int)((m_end - m_begin + 63) / 64) : 0;

  // ---- loader roles
  const int a_c4 = tid & 31, a_rg = tid >> 5;                 // dY: 8 rows x 4 columns per thread
  const int a_col = n0 + a_c4 * 4;
  const bool a_col_ok = a_col < p.N;
  const int b_c4 = tid % QB, b_rg = tid / QB;                 // X: RPT rows x 4 channels per thread
  const int b_ch = c0 + b_c4 * 4;
  const bool b_ch_ok = b_ch < p.Cin;
  int tdx = 0, tdy = 0, tdz = 0;
  if (p.conv) {
    tdz = tap % p.g.kZ;
    tdy = (tap / p.g.kZ) % p.g.kY;
    tdx = tap / (p.g.kZ * p.g.kY);
  }

  // ---- pre-split loader roles: threads 0..127 stage the hi halves, 128..255 the lo halves; a thread owns 8 rows x
  // 8 columns (one 16-byte load per row)
  const int p_half = tid >> 7, p_t = tid & 127;
  const int pa_c8 = p_t & 15, pa_rg = p_t >> 4;
  const int pa_col = n0 + pa_c8 * 8;
  const bool pa_ok = pa_col < p.N;
  constexpr int PBQ = BC / 8;                                  // 8-column groups of the X tile
  const int pb_c8 = p_t % PBQ, pb_rg = p_t / PBQ;
  const bool pb_act = pb_rg < 8;
  const int pb_ch = c0 + pb_c8 * 8;
  const bool pb_ok = pb_act && pb_ch < p.Cin;
  wg_u4 pa[8], pb[8];
  auto load_chunk_pre = [&](int ck) __attribute__((always_inline)) {
    const long mb = m_begin + (long)ck * 64;
    const uint16_t* ya = p_half ? p.dYl : p.dYh;
    const uint16_t* xa = p_half ? p.Xl : p.Xh;
    const wg_u4 z4 = {0u, 0u, 0u, 0u};
    if (p.zfast) {
      const wg_u4* zp = (const wg_u4*)wg_zero16;
      {
        const long m0 = mb + pa_rg * 8;
        const uint16_t* base = ya + (m0 < p.M ? m0 : 0) * p.ldy + (pa_ok ? pa_col : 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = pa_ok && m0 + j < m_end;
          pa[j] = *(ok ? (const wg_u4*)(base + j * p.ldy) : zp);
        }
      }
      {
        const long ml = mb + (pb_act ? pb_rg : 0) * 8;
        const unsigned m = (unsigned)(ml < p.M ? ml : 0);
        const int z0 = (int)(m & (unsigned)(p.g.Zo - 1));
        unsigned t = m >> p.zshift;
        const int yo = (int)(t % (unsigned)p.g.Yo);
        t /= (unsigned)p.g.Yo;
        const int xo = (int)(t % (unsigned)p.g.Xo);
        const long b = t / (unsigned)p.g.Xo;
        const int xi = xo * p.g.stride - p.g.pad_x + tdx * p.g.dil, yi = yo * p.g.stride - p.g.pad_y + tdy * p.g.dil;
        const int zi0 = z0 * p.g.stride - p.g.pad_z + tdz * p.g.dil;
        const bool okxy = pb_ok && ml < p.M && xi >= 0 && xi < p.g.Xi && yi >= 0 && yi < p.g.Yi;
        const uint16_t* base = xa + (okxy ? b * p.g.sb + xi * p.g.sx + yi * p.g.sy + pb_ch : 0);
        const int zstep = p.g.stride * (int)p.g.sz;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int zi = zi0 + j * p.g.stride;
          const bool ok = okxy && (unsigned)zi < (unsigned)p.g.Zi;
          pb[j] = *(ok ? (const wg_u4*)(base + (zi0 * (int)p.g.sz + j * zstep)) : zp);
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long m = mb + pa_rg * 8 + j;
      const bool ok = m < m_end && pa_ok;
      if (m >= p.M) m = p.M - 1;
      const wg_u4 v = *(const wg_u4*)(ya + m * p.ldy + (pa_ok ? pa_col : 0));
      pa[j] = ok ? v : z4;
    }
    if (p.conv) {
      long ml = mb + (pb_act ? pb_rg : 0) * 8;
      if (ml >= p.M) ml = p.M - 1;
      const unsigned m = (unsigned)ml;
      int zo = (int)(m % (unsigned)p.g.Zo);
      unsigned t = m / (unsigned)p.g.Zo;
      int yo = (int)(t % (unsigned)p.g.Yo);
      t /= (unsigned)p.g.Yo;
      int xo = (int)(t % (unsigned)p.g.Xo);
      long b = t / (unsigned)p.g.Xo;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int xi = xo * p.g.stride - p.g.pad_x + tdx * p.g.dil, yi = yo * p.g.stride - p.g.pad_y + tdy * p.g.dil,
                  zi = zo * p.g.stride - p.g.pad_z + tdz * p.g.dil;
        const bool ok = pb_ok && xi >= 0 && xi < p.g.Xi && yi >= 0 && yi < p.g.Yi && zi >= 0 && zi < p.g.Zi;
        const int xc = occf_clampi(xi, p.g.Xi - 1), yc = occf_clampi(yi, p.g.Yi - 1), zc = occf_clampi(zi, p.g.Zi - 1);
        const long bc = b < p.g.B ? b : p.g.B - 1;
        const wg_u4 v = *(const wg_u4*)(xa + bc * p.g.sb + xc * p.g.sx + yc * p.g.sy + zc * p.g.sz + (pb_ok ? pb_ch : 0));
        pb[j] = ok ? v : z4;
        if (++zo == p.g.Zo) {
          zo = 0;
          if (++yo == p.g.Yo) {
            yo = 0;
            if (++xo == p.g.Xo) {
              xo = 0;
              ++b;
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        long m = mb + (pb_act ? pb_rg : 0) * 8 + j;
        if (m >= p.M) m = p.M - 1;
        const wg_u4 v = *(const wg_u4*)(xa + m * p.ldx + (pb_ok ? pb_ch : 0));
        pb[j] = pb_ok ? v : z4;
      }
    }
  };
  auto store_chunk_pre = [&]() __attribute__((always_inline)) {
    wg_u4* Ad = p_half ? Al : Ah;
    wg_u4* Bd = p_half ? Bl : Bh;
    const uint32_t* ar = (const uint32_t*)pa;                  // [row j][dword d]: columns 2d, 2d + 1
    const uint32_t* br = (const uint32_t*)pb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wg_u4 v, w;
      const int d = e >> 1;
      if (e & 1) {
        v.x = wg_pack_hi(ar[0 * 4 + d], ar[1 * 4 + d]); v.y = wg_pack_hi(ar[2 * 4 + d], ar[3 * 4 + d]);
        v.z = wg_pack_hi(ar[4 * 4 + d], ar[5 * 4 + d]); v.w = wg_pack_hi(ar[6 * 4 + d], ar[7 * 4 + d]);
        w.x = wg_pack_hi(br[0 * 4 + d], br[1 * 4 + d]); w.y = wg_pack_hi(br[2 * 4 + d], br[3 * 4 + d]);
        w.z = wg_pack_hi(br[4 * 4 + d], br[5 * 4 + d]); w.w = wg_pack_hi(br[6 * 4 + d], br[7 * 4 + d]);
      } else {
        v.x = wg_pack_lo(ar[0 * 4 + d], ar[1 * 4 + d]); v.y = wg_pack_lo(ar[2 * 4 + d], ar[3 * 4 + d]);
        v.z = wg_pack_lo(ar[4 * 4 + d], ar[5 * 4 + d]); v.w = wg_pack_lo(ar[6 * 4 + d], ar[7 * 4 + d]);
        w.x = wg_pack_lo(br[0 * 4 + d], br[1 * 4 + d]); w.y = wg_pack_lo(br[2 * 4 + d], br[3 * 4 + d]);
        w.z = wg_pack_lo(br[4 * 4 + d], br[5 * 4 + d]); w.w = wg_pack_lo(br[6 * 4 + d], br[7 * 4 + d]);
      }
      if (TERMS == 3 || p_half == 0) {
        Ad[pa_rg * 128 + wg_swz(pa_c8 * 8 + e, p.swz)] = v;
        if (pb_act) Bd[pb_rg * BC + wg_swz(pb_c8 * 8 + e, p.swz)] = w;
      }
    }
  };

  float4 ra[8], rb[RPT];
  auto load_chunk = [&](int ck) __attribute__((always_inline)) {
    const long mb = m_begin + (long)ck * 64;
    {
      // one base address per thread and chunk; a row outside the slice reads the 16-byte zero constant (a select on
      // the ADDRESS, two instructions, instead of four on the loaded dwords)
      const float4* zp = (const float4*)wg_zero16;
      const long m0 = mb + a_rg * 8;
      const float* base = p.dY + (m0 < p.M ? m0 : 0) * p.ldy + (a_col_ok ? a_col : 0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = a_col_ok && m0 + j < m_end;
        ra[j] = *(ok ? (const float4*)(base + j * p.ldy) : zp);
      }
    }
    if (p.conv) {
      long ml = mb + b_rg * RPT;
      if (ml >= p.M) ml = p.M - 1;
      const unsigned m = (unsigned)ml;                      // M < 2^31: 32-bit divisions
      int zo = (int)(m % (unsigned)p.g.Zo);
      unsigned t = m / (unsigned)p.g.Zo;
      int yo = (int)(t % (unsigned)p.g.Yo);
      t /= (unsigned)p.g.Yo;
      int xo = (int)(t % (unsigned)p.g.Xo);
      long b = t / (unsigned)p.g.Xo;
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const int xi = xo * p.g.stride - p.g.pad_x + tdx * p.g.dil, yi = yo * p.g.stride - p.g.pad_y + tdy * p.g.dil,
                  zi = zo * p.g.stride - p.g.pad_z + tdz * p.g.dil;
        const bool ok = b_ch_ok && xi >= 0 && xi < p.g.Xi && yi >= 0 && yi < p.g.Yi && zi >= 0 && zi < p.g.Zi;
        const int xc = occf_clampi(xi, p.g.Xi - 1), yc = occf_clampi(yi, p.g.Yi - 1), zc = occf_clampi(zi, p.g.Zi - 1);
        const long bc = b < p.g.B ? b : p.g.B - 1;
        const float4 v = *(const float4*)(p.X + bc * p.g.sb + xc * p.g.sx + yc * p.g.sy + zc * p.g.sz +
                                          (b_ch_ok ? b_ch : 0));
        rb[j] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        // next output voxel (rows beyond M carry zero dY rows; their batch index is clamped at the address)
        if (++zo == p.g.Zo) {
          zo = 0;
          if (++yo == p.g.Yo) {
            yo = 0;
            if (++xo == p.g.Xo) {
              xo = 0;
              ++b;
            }
          }
        }
      }
    } else {
      const float4* zp = (const float4*)wg_zero16;
      const long m0 = mb + b_rg * RPT;
      const float* base = p.X + (m0 < p.M ? m0 : 0) * p.ldx + (b_ch_ok ? b_ch : 0);
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        const bool ok = b_ch_ok && m0 + j < p.M;        // (rows in [m_end, M) meet zero dY rows)
        rb[j] = *(ok ? (const float4*)(base + j * p.ldx) : zp);
      }
    }
  };
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool do_bias = !PRE && p.bias_out != nullptr && tap == 0 && ct == 0;
  auto store_chunk = [&]() __attribute__((always_inline)) {
    // dY: rows a_rg*8 .. +7 of columns a_c4*4 .. +3 -> four 16-byte groups (one per column) in row group a_rg
    {
      const float cx[4][8] = {{ra[0].x, ra[1].x, ra[2].x, ra[3].x, ra[4].x, ra[5].x, ra[6].x, ra[7].x},
                              {ra[0].y, ra[1].y, ra[2].y, ra[3].y, ra[4].y, ra[5].y, ra[6].y, ra[7].y},
                              {ra[0].z, ra[1].z, ra[2].z, ra[3].z, ra[4].z, ra[5].z, ra[6].z, ra[7].z},
                              {ra[0].w, ra[1].w, ra[2].w, ra[3].w, ra[4].w, ra[5].w, ra[6].w, ra[7].w}};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        wg_u4 h, l;
        uint32_t hh, ll;
        occf_bf16_split2(cx[e][0], cx[e][1], hh, ll); h.x = hh; l.x = ll;
        occf_bf16_split2(cx[e][2], cx[e][3], hh, ll); h.y = hh; l.y = ll;
        occf_bf16_split2(cx[e][4], cx[e][5], hh, ll); h.z = hh; l.z = ll;
        occf_bf16_split2(cx[e][6], cx[e][7], hh, ll); h.w = hh; l.w = ll;
        const int off = a_rg * 128 + wg_swz(a_c4 * 4 + e, p.swz);
        Ah[off] = h;
        if (TERMS == 3) Al[off] = l;
      }
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { bsum[0] += ra[j].x; bsum[1] += ra[j].y; bsum[2] += ra[j].z; bsum[3] += ra[j].w; }
    }
    // X: RPT = 8 -> whole 16-byte groups; RPT = 4 -> the thread owns half a group (rows 4*(b_rg&1) .. +3)
    {
      float cx[4][RPT];
#pragma unroll
      for (int j = 0; j < RPT; ++j) { cx[0][j] = rb[j].x; cx[1][j] = rb[j].y; cx[2][j] = rb[j].z; cx[3][j] = rb[j].w; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t hh[RPT / 2], ll[RPT / 2];
#pragma unroll
        for (int j = 0; j < RPT; j += 2) occf_bf16_split2(cx[e][j], cx[e][j + 1], hh[j / 2], ll[j / 2]);
        if (RPT == 8) {
          const int off = b_rg * BC + wg_swz(b_c4 * 4 + e, p.swz);
          wg_u4 h, l;
          h.x = hh[0]; h.y = hh[1]; h.z = hh[RPT / 2 - 2]; h.w = hh[RPT / 2 - 1];
          l.x = ll[0]; l.y = ll[1]; l.z = ll[RPT / 2 - 2]; l.w = ll[RPT / 2 - 1];
          Bh[off] = h;
          if (TERMS == 3) Bl[off] = l;
        } else {
          const int off = (b_rg >> 1) * BC + wg_swz(b_c4 * 4 + e, p.swz);
          uint32_t* dh = (uint32_t*)(Bh + off) + (b_rg & 1) * 2;
          uint32_t* dl = (uint32_t*)(Bl + off) + (b_rg & 1) * 2;
          dh[0] = hh[0];
          dh[1] = hh[1];
          if (TERMS == 3) { dl[0] = ll[0]; dl[1] = ll[1]; }
        }
      }
    }
  };

  f32x16 acc[2][TC];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int li = lane & 31, lk = lane >> 5;
  auto frag = [&](const wg_u4* base, int ld, int col, int ks) __attribute__((always_inline)) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, base[(ks * 2 + lk) * ld + wg_swz(col, p.swz)]);
  };
  auto compute = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 ah[2], al[2], bh[TC], bl[TC];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = frag(Ah, 128, wm * 64 + i * 32 + li, ks);
        if (TERMS == 3) al[i] = frag(Al, 128, wm * 64 + i * 32 + li, ks);
      }
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        bh[j] = frag(Bh, BC, wn * (BC / 2) + j * 32 + li, ks);
        if (TERMS == 3) bl[j] = frag(Bl, BC, wn * (BC / 2) + j * 32 + li, ks);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          if (TERMS == 3) {
            acc[i][j] = occf_mfma_bf16_32x32x16(al[i], bh[j], acc[i][j]);
            acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bl[j], acc[i][j]);
          }
          acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bh[j], acc[i][j]);
        }
    }
  };

  if (PRE) {
    if (nchunks > 0) load_chunk_pre(0);
    for (int ck = 0; ck < nchunks; ++ck) {
      if (ck > 0) __syncthreads();
      store_chunk_pre();
      __syncthreads();
      load_chunk_pre(ck + 1 < nchunks ? ck + 1 : ck);
      compute();
    }
  } else {
    if (nchunks > 0) load_chunk(0);
    for (int ck = 0; ck < nchunks; ++ck) {
      if (ck > 0) __syncthreads();              // previous chunk's fragment reads are done
      store_chunk();
      __syncthreads();
      load_chunk(ck + 1 < nchunks ? ck + 1 : ck);   // unconditional (the last one re-reads its own chunk)
      compute();
    }
  }

  // ---- epilogue: raw partial sums of this M-slice
  const int Kt = p.taps * p.Cin;
  float* o = p.out + (long)slab * p.N * Kt;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = c0 + wn * (BC / 2) + j * 32 + li;
      if (c >= p.Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (n < p.N) o[(long)n * Kt + (long)tap * p.Cin + c] = acc[i][j][r];
      }
    }
  if (do_bias) {
    __syncthreads();
    float* red = (float*)Ah;                   // [8 row groups][128 columns]
#pragma unroll
    for (int e = 0; e < 4; ++e) red[a_rg * 128 + a_c4 * 4 + e] = bsum[e];
    __syncthreads();
    if (tid < 128 && n0 + tid < p.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) s += red[r * 128 + tid];
      p.bias_out[(long)slab * p.N + n0 + tid] = s;
    }
  }
}

// out[i] = sum_s slab[s][i]  (fixed order: four interleaved slab groups, each summed in order with two alternating
// accumulators, combined 0..3 -- deterministic).  64 elements x 4 slab groups per workgroup: one thread per element
// walking all S slabs serially was latency-bound (23 us per call at S ~ 170, 257 calls = 5.9 ms per training step).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                           long n, int S) {
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + e;
  float v0 = 0.f, v1 = 0.f;
  if (i < n) {
    int s = g;
    for (; s + 4 < S; s += 8) {
      v0 += slab[(long)s * n + i];
      v1 += slab[(long)(s + 4) * n + i];
    }
    if (s < S) v0 += slab[(long)s * n + i];
  }
  part[g][e] = v0 + v1;
  __syncthreads();
  if (g == 0 && i < n) out[i] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

// small problems (M <= 1024 rows, or shapes the tile kernel does not take): exact fp32, 8 lanes per output walk
// the rows (stride 8) and combine with a 3-step butterfly -- the decoder's 100-row linears launch ~120 of these
// per training step
__global__ void __launch_bounds__(256) wgrad_small_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                          float* __restrict__ dW, float* __restrict__ db, long M, int N,
                                                          int K, long ldy, long ldx) {
  const long gid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int sub = threadIdx.x & 7;
  const long NK = (long)N * K;
  const bool w_ok = gid < NK, b_ok = db != nullptr && gid < N;
  const int n = w_ok ? (int)(gid / K) : 0, k = w_ok ? (int)(gid % K) : 0;
  float s = 0.f, sb = 0.f;
  for (long m = sub; m < M; m += 8) {
    s = fmaf(dY[m * ldy + n], X[m * ldx + k], s);
    if (b_ok) sb += dY[m * ldy + gid];
  }
  s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
  sb += __shfl_xor(sb, 1); sb += __shfl_xor(sb, 2); sb += __shfl_xor(sb, 4);
  if (sub == 0) {
    if (w_ok) dW[gid] = s;
    if (b_ok) db[gid] = sb;
  }
}

static int wg_pick_splits(long M, int N, int Cin, int taps, int BC) {
  const long tiles = (long)occf_cdiv(N, 128) * occf_cdiv(Cin, BC) * taps;
  const long chunks = (M + 63) / 64;
  static const int target = [] {
    const char* e = getenv("OCCF_WG_TARGET");          // diagnostics: workgroups aimed at per launch
    return e ? atoi(e) : 0;
  }();
  // measured (scripts/bwd_probe.py, r02 probes): the 27-tap convolutions want ~2048 workgroups (3.95 vs 4.92 ms at
  // 128 -> 128), a plain linear with 3 tiles pays for every extra slab in the reduction (0.63 ms at 512 vs 0.83)
  if (target <= 0 && taps > 1) {
    // convolutions: 2 workgroups per CU are resident (512 slots); choose the slab count (whole slabs per XCD: a
    // multiple of 8) whose workgroup count fills whole rounds of 512 best, between ~1 000 and ~4 600 workgroups.
    // Measured on the 192 -> 192 convolution (108 tiles): S = 16 (3.4 rounds) 7.85 ms, S = 40 (8.4 rounds) 7.15 ms.
    long best_s = 0;
    double best_e = 0.0;
    for (long S = 8; S <= 128 && S <= chunks / 4; S += 8) {
      const long total = tiles * S;
      if (total < 1024 && S + 8 <= 128 && S + 8 <= chunks / 4) continue;
      if (total > 4608 && best_s) break;
      const double e = (double)total / (double)(((total + 511) / 512) * 512);
      if (e > best_e + 0.01) {
        best_e = e;
        best_s = S;
      }
    }
    if (best_s) return (int)best_s;
  }
  const int tgt = target > 0 ? target : (taps > 1 ? 2048 : 512);
  long S = (tgt + tiles - 1) / tiles;
  if (S > chunks / 4) S = chunks / 4;
  if (S < 1) S = 1;
  if (S > 256) S = 256;
  if (S >= 8) S = (S + 4) / 8 * 8;                    // whole slabs per XCD (xcd_tiles placement)
  return (int)S;
}
static int wg_bc(int Cin) {
  static const int env = [] {
    const char* e = getenv("OCCF_WG_BC");               // diagnostics: force the channel tile (64 / 128)
    return e ? atoi(e) : 0;
  }();
  if (env == 64 || env == 128) return env;
  // 192 channels: two 128-wide tiles (the second half-masked) stage dY twice instead of three times -- the kernel is
  // bound by its staging, not by the matrix cores
  return (Cin % 128 == 0 || Cin > 128) ? 128 : 64;
}

static long wg_workspace(long M, int N, int Cin, int taps) {
  const int S = wg_pick_splits(M, N, Cin, taps, wg_bc(Cin));
  return S > 1 ? (long)S * N * ((long)taps * Cin + 1) : 0;
}

static int wg_launch(WgArgs a, float* dW, float* db, float* workspace, long workspace_floats, int terms, hipStream_t st) {
  const int BC = wg_bc(a.Cin);
  static const int swz_env = [] {
    const char* e = getenv("OCCF_WG_SWZ");
    return e ? atoi(e) : 1;
  }();
  a.swz = swz_env ? 7 : 0;
  int S = wg_pick_splits(a.M, a.N, a.Cin, a.taps, BC);
  const long Kt = (long)a.taps * a.Cin;
  while (S > 1 && (long)S * a.N * (Kt + 1) > workspace_floats) --S;
  if (S > 1 && !workspace) S = 1;
  long rows = (a.M + S - 1) / S;
  rows = (rows + 63) / 64 * 64;
  S = (int)((a.M + rows - 1) / rows);
  a.rows_per_split = rows;
  a.out = S > 1 ? workspace : dW;
  a.bias_out = db ? (S > 1 ? workspace + (long)S * a.N * Kt : db) : nullptr;
  const long tiles = (long)occf_cdiv(a.N, 128) * occf_cdiv(a.Cin, BC) * a.taps;
  dim3 grid((unsigned)tiles, S);
  static const int xcd_env = [] {
    const char* e = getenv("OCCF_WG_XCD");
    return e ? atoi(e) : 1;
  }();
  a.n_slabs = S;
  if (xcd_env && S >= 8 && tiles * ((S + 7) / 8) * 8 < 2147483647L) {
    a.xcd_tiles = (int)tiles;
    grid = dim3((unsigned)(tiles * ((S + 7) / 8) * 8), 1);
  }
  const bool pre = a.dYh != nullptr;
#define OCCF_WG_LAUNCH(BC_, T_)                                                                        \
  do {                                                                                                 \
    if (pre) hipLaunchKernelGGL((wgrad_kernel<BC_, T_, true>), grid, dim3(256), 0, st, a);             \
    else hipLaunchKernelGGL((wgrad_kernel<BC_, T_, false>), grid, dim3(256), 0, st, a);                \
  } while (0)
  if (BC == 128) {
    if (terms == 3) OCCF_WG_LAUNCH(128, 3);
    else OCCF_WG_LAUNCH(128, 1);
  } else {
    if (terms == 3) OCCF_WG_LAUNCH(64, 3);
    else OCCF_WG_LAUNCH(64, 1);
  }
#undef OCCF_WG_LAUNCH
  if (S > 1) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(occf_cdiv((long)a.N * Kt, 64)), dim3(256), 0, st, workspace, dW,
                       (long)a.N * Kt, S);
    if (db)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(occf_cdiv(a.N, 64)), dim3(256), 0, st, a.bias_out, db, (long)a.N, S);
  }
  return (int)hipGetLastError();
}

extern "C" long occf_linear_wgrad_workspace(long M, int N, int K) {
  if (M <= 1024 || N % 4 || K % 4) return 0;
  return wg_workspace(M, N, K, 1);
}

extern "C" int occf_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, float* workspace,
                                 long workspace_floats, long M, int N, int K, long ldy, long ldx, int terms,
                                 void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return OCCF_EINVAL;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 1024 || N % 4 || K % 4 || ldy % 4 || ldx % 4) {
    if (M > 65536) return OCCF_ESHAPE;
    const long total = ((long)N * K > N ? (long)N * K : N) * 8;
    hipLaunchKernelGGL(wgrad_small_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, st, dy, x, dw, dbias, M, N, K,
                       ldy, ldx);
    return (int)hipGetLastError();
  }
  WgArgs a = {};
  a.dY = dy; a.X = x; a.M = M; a.N = N; a.Cin = K; a.taps = 1; a.ldy = ldy; a.ldx = ldx; a.conv = 0;
  return wg_launch(a, dw, dbias, workspace, workspace_floats, terms, st);
}

// hi = bf16_rne(x), lo = bf16_rne(x - hi)  (the same split as occf_split_bf16; local copy: separate translation unit)
__global__ void __launch_bounds__(256) wg_split_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi,
                                                       uint16_t* __restrict__ lo, long n2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  uint32_t h, l;
  occf_bf16_split2(x[2 * i], x[2 * i + 1], h, l);
  ((uint32_t*)hi)[i] = h;
  ((uint32_t*)lo)[i] = l;
}

static bool wg_presplit_ok(int Cin, int Cout, int taps) {
  static const int env = [] {
    const char* e = getenv("OCCF_WG_PRESPLIT");
    return e ? atoi(e) : 1;
  }();
  return env && taps >= 9 && Cin % 8 == 0 && Cout % 8 == 0;
}

extern "C" long occf_conv3d_wgrad_workspace(int B, int Xi, int Yi, int Zi, int Cin, int Cout, int kX, int kY, int kZ,
                                            int stride, int dil, int pad_x, int pad_y, int pad_z) {
  const int Xo = (Xi + 2 * pad_x - dil * (kX - 1) - 1) / stride + 1;
  const int Yo = (Yi + 2 * pad_y - dil * (kY - 1) - 1) / stride + 1;
  const int Zo = (Zi + 2 * pad_z - dil * (kZ - 1) - 1) / stride + 1;
  long need = wg_workspace((long)B * Xo * Yo * Zo, Cout, Cin, kX * kY * kZ);
  if (wg_presplit_ok(Cin, Cout, kX * kY * kZ))        // bf16 (hi, lo) copies of dy and x: 4 bytes per element
    need += (long)B * Xo * Yo * Zo * Cout + (long)B * Xi * Yi * Zi * Cin;
  return need;
}

extern "C" int occf_conv3d_wgrad(const float* dy, const float* x, float* dw_tapmajor, float* dbias, float* workspace,
                                 long workspace_floats, int B, int Xi, int Yi, int Zi, int Cin, int Cout, int kX,
                                 int kY, int kZ, int stride, int dil, int pad_x, int pad_y, int pad_z, long in_sb,
                                 long in_sx, long in_sy, long in_sz, int terms, void* stream) {
  if (B <= 0 || Cin % 4 || Cout % 4 || stride <= 0 || dil <= 0) return OCCF_ESHAPE;
  if (in_sb % 4 || in_sx % 4 || in_sy % 4 || in_sz % 4) return OCCF_ESHAPE;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  WgArgs a = {};
  WgGeom& g = a.g;
  g.B = B; g.Xi = Xi; g.Yi = Yi; g.Zi = Zi; g.kX = kX; g.kY = kY; g.kZ = kZ; g.stride = stride; g.dil = dil;
  g.pad_x = pad_x; g.pad_y = pad_y; g.pad_z = pad_z;
  g.Xo = (Xi + 2 * pad_x - dil * (kX - 1) - 1) / stride + 1;
  g.Yo = (Yi + 2 * pad_y - dil * (kY - 1) - 1) / stride + 1;
  g.Zo = (Zi + 2 * pad_z - dil * (kZ - 1) - 1) / stride + 1;
  g.sb = in_sb; g.sx = in_sx; g.sy = in_sy; g.sz = in_sz;
  if (g.Xo <= 0 || g.Yo <= 0 || g.Zo <= 0) return OCCF_ESHAPE;
  a.dY = dy; a.X = x; a.M = (long)B * g.Xo * g.Yo * g.Zo; a.N = Cout; a.Cin = Cin; a.taps = kX * kY * kZ;
  a.ldy = Cout; a.ldx = 0; a.conv = 1;
  hipStream_t st = (hipStream_t)stream;
  // pre-split operands: dense x only (the strides then address the bf16 copy unchanged), no bias sum
  const long nx = (long)B * Xi * Yi * Zi * Cin, ny = a.M * Cout;
  const bool dense = in_sz == Cin && in_sy == (long)Zi * Cin && in_sx == (long)Yi * Zi * Cin &&
                     in_sb == (long)Xi * Yi * Zi * Cin;
  if (wg_presplit_ok(Cin, Cout, a.taps) && dense && !dbias && workspace && workspace_floats >= nx + ny) {
    uint16_t* yh = (uint16_t*)(workspace + workspace_floats - (nx + ny));
    uint16_t* yl = yh + ny;
    uint16_t* xh = yl + ny;
    uint16_t* xl = xh + nx;
    hipLaunchKernelGGL(wg_split_kernel, dim3(occf_cdiv(ny / 2, 256)), dim3(256), 0, st, dy, yh, yl, ny / 2);
    hipLaunchKernelGGL(wg_split_kernel, dim3(occf_cdiv(nx / 2, 256)), dim3(256), 0, st, x, xh, xl, nx / 2);
    a.dYh = yh; a.dYl = yl; a.Xh = xh; a.Xl = xl;
    workspace_floats -= nx + ny;
    static const int zfast_env = [] {
      const char* e = getenv("OCCF_WG_ZFAST");
      return e ? atoi(e) : 1;
    }();
    if (zfast_env && (g.Zo == 8 || g.Zo == 16 || g.Zo == 32 || g.Zo == 64) && a.M < 2147483647L) {
      a.zfast = 1;
      a.zshift = g.Zo == 8 ? 3 : g.Zo == 16 ? 4 : g.Zo == 32 ? 5 : 6;
    }
  }
  return wg_launch(a, dw_tapmajor, dbias, workspace, workspace_floats, terms, st);
}
