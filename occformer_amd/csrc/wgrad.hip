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
  int zfast;           // per-column staging with bounded buffer loads and incremental decode (kernel template ZF)
  int zshift;          // log2(Zo)
  uint32_t ybytes, xbytes;   // sizes of ONE pre-split array (hi or lo) of dY / X in bytes (zfast == 2: < 2 GiB each)
  // xcd_tiles > 0: 1-D grid; workgroup w runs on XCD w % 8 (round-robin dispatch), so ALL (tap, tile) workgroups of one
  // M-slab are placed on ONE XCD (slab = xcd + 8 * (k / xcd_tiles), tile = k % xcd_tiles with k = w / 8): the slab's
  // dY / X rows are then fetched from HBM once into that XCD's L2 and shared by its 27 x tiles workgroups.  With the
  // slabs spread round-robin over all XCDs every XCD streamed every slab: 21 GB of HBM fetch for 0.99 GB of operands on
  // the 192 -> 192 convolution (PMC FETCH_SIZE), i.e. the kernel ran at the HBM roof, not the MFMA roof.
  int xcd_tiles;
  int n_slabs;
  // TERMS == 2 (two fp16-piece products): scale[0] = bit pattern of max |dY| (wg_absmax_kernel); dY enters as ONE fp16
  // piece of dY * 2^k, X as fp16 (hi, lo); the partial sums leave multiplied by 2^-k
  const uint32_t* scale;
  int bc;              // channel tile width (128 = 128-wide tiles + remainder tile, 64 = uniform 64-wide tiles)
  int mix;             // remainder tiles of <= 64 columns run as 64-wide tiles (both dimensions)
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

// One workgroup tile: TN (n, 128 or 64) x BC (c, 128 or 64) of one tap over one M-slab.  N and Cin are covered by
// 128-wide tiles plus ONE remainder tile that is 64 wide when the remainder fits (192 = 128 + 64: no padded MFMAs;
// with uniform 128 x 128 tiles the 192 -> 192 convolution issued 207 M MFMAs for 117 M needed -- PMC).
template <int TN, int BC, int TERMS, bool PRE, bool ZF>
__device__ __forceinline__ void wgrad_body(const WgArgs& p, wg_u4* __restrict__ Ah, wg_u4* __restrict__ Al,
                                           wg_u4* __restrict__ Bh, wg_u4* __restrict__ Bl, const int n0, const int c0,
                                           const int tap, const int slab, const bool first_c) {
  constexpr int TI = TN / 64;                 // 32-wide tiles per wave along n
  constexpr int TC = BC / 64;                 // 32-wide tiles per wave along c
  constexpr int QA = TN / 4;                  // column quads of the dY tile
  constexpr int RPA = 64 * QA / 256;          // rows per thread of the dY tile (8 or 4)
  constexpr int QB = BC / 4;                  // channel quads of the X tile
  constexpr int RPT = 64 * QB / 256;          // rows per thread of the X tile (8 or 4)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long m_begin = (long)slab * p.rows_per_split;
  long m_end = m_begin + p.rows_per_split;
  if (m_end > p.M) m_end = p.M;
  const int nchunks = m_end > m_begin ? (int)((m_end - m_begin + 63) / 64) : 0;

  // ---- loader roles
  const int a_c4 = tid % QA, a_rg = tid / QA;                 // dY: RPA rows x 4 columns per thread
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
  constexpr int PAQ = TN / 8;                                  // 8-column groups of the dY tile
  const int pa_c8 = p_t % PAQ, pa_rg = p_t / PAQ;
  const bool pa_act = pa_rg < 8;
  const int pa_col = n0 + pa_c8 * 8;
  const bool pa_ok = pa_act && pa_col < p.N;
  constexpr int PBQ = BC / 8;                                  // 8-column groups of the X tile
  const int pb_c8 = p_t % PBQ, pb_rg = p_t / PBQ;
  const bool pb_act = pb_rg < 8;
  const int pb_ch = c0 + pb_c8 * 8;
  const bool pb_ok = pb_act && pb_ch < p.Cin;
  wg_u4 pa[8], pb[8];
  int zc_next = -1, zc_b = 0, zc_x = 0, zc_y = 0;                      // incremental column decode (zfast == 2)
  const int zb_z0 = ((pb_act ? pb_rg : 0) * 8) & (p.g.Zo - 1);         // (chunk bases are multiples of 64 >= Zo)
  auto load_chunk_pre = [&](int ck) __attribute__((always_inline)) {
    const long mb = m_begin + (long)ck * 64;
    const uint16_t* ya = p_half ? p.dYl : p.dYh;
    const uint16_t* xa = p_half ? p.Xl : p.Xh;
    const wg_u4 z4 = {0u, 0u, 0u, 0u};
    if constexpr (ZF) {
      // bounded buffer loads ("load or zero" = one 32-bit select on the byte offset; the row stride folds into the
      // instruction's immediate) and an INCREMENTAL column decode: PMC on the 192 -> 192 convolution counted 6.6 VALU
      // per MFMA in this loop, ~100 of the ~320 per chunk in the two runtime divisions of the decode and ~130 in
      // 64-bit address selects / adds of the 16 loads
      const occf_bbuf ybuf = occf_make_bbuf(ya, p.ybytes), xbuf = occf_make_bbuf(xa, p.xbytes);
      {
        const long m0 = mb + (pa_act ? pa_rg : 0) * 8;
        const uint32_t off0 = pa_ok && m0 < p.M ? (uint32_t)((m0 * p.ldy + pa_col) * 2) : OCCF_BUF_OOB;
        const uint32_t rs = (uint32_t)p.ldy * 2u;
        const int left = (int)(m_end - m0 < 8 ? m_end - m0 : 8);       // rows of this group inside the slab
#pragma unroll
        for (int j = 0; j < 8; ++j)
          pa[j] = __builtin_bit_cast(wg_u4, occf_bbuf_load_b128(ybuf, j < left ? off0 + (uint32_t)j * rs : OCCF_BUF_OOB));
      }
      {
        // (zc_b, zc_x, zc_y) = the (batch, x, y) column of this thread's row group in chunk `zc_next`
        if (zc_next != ck) {                                            // first chunk (or a re-read): decode by division
          const long ml = mb + (pb_act ? pb_rg : 0) * 8;
          unsigned t = (unsigned)((ml < p.M ? ml : 0) >> p.zshift);
          zc_y = (int)(t % (unsigned)p.g.Yo);
          t /= (unsigned)p.g.Yo;
          zc_x = (int)(t % (unsigned)p.g.Xo);
          zc_b = (int)(t / (unsigned)p.g.Xo);
        }
        const long ml = mb + (pb_act ? pb_rg : 0) * 8;
        const int xi = zc_x * p.g.stride - p.g.pad_x + tdx * p.g.dil, yi = zc_y * p.g.stride - p.g.pad_y + tdy * p.g.dil;
        const bool okxy = pb_ok && ml < p.M && xi >= 0 && xi < p.g.Xi && yi >= 0 && yi < p.g.Yi;
        const int zi0 = zb_z0 * p.g.stride - p.g.pad_z + tdz * p.g.dil;
        const uint32_t off0 = okxy ? (uint32_t)((zc_b * p.g.sb + xi * p.g.sx + yi * p.g.sy + zi0 * p.g.sz + pb_ch) * 2)
                                   : OCCF_BUF_OOB;
        const uint32_t zs = (uint32_t)(p.g.stride * (int)p.g.sz) * 2u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = okxy && (unsigned)(zi0 + j * p.g.stride) < (unsigned)p.g.Zi;
          pb[j] = __builtin_bit_cast(wg_u4, occf_bbuf_load_b128(xbuf, ok ? off0 + (uint32_t)j * zs : OCCF_BUF_OOB));
        }
        // advance the column to the next chunk: 64 rows = 64 >> zshift columns further along y
        zc_next = ck + 1;
        zc_y += 64 >> p.zshift;
        while (zc_y >= p.g.Yo) {
          zc_y -= p.g.Yo;
          if (++zc_x == p.g.Xo) { zc_x = 0; ++zc_b; }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      long m = mb + (pa_act ? pa_rg : 0) * 8 + j;
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
      if ((TERMS == 3 || p_half == 0) && pa_act) Ad[pa_rg * TN + wg_swz(pa_c8 * 8 + e, p.swz)] = v;
      if ((TERMS >= 2 || p_half == 0) && pb_act) Bd[pb_rg * BC + wg_swz(pb_c8 * 8 + e, p.swz)] = w;
    }
  };

  uint32_t inv_bits = 0x3F800000u;
  const float f16_scale = TERMS == 2 ? occf_u2f(occf_f16_scale_bits(p.scale[0], inv_bits)) : 1.0f;
  float4 ra[RPA], rb[RPT];
  auto load_chunk = [&](int ck) __attribute__((always_inline)) {
    const long mb = m_begin + (long)ck * 64;
    {
      // one base address per thread and chunk; a row outside the slice reads the 16-byte zero constant (a select on
      // the ADDRESS, two instructions, instead of four on the loaded dwords)
      const float4* zp = (const float4*)wg_zero16;
      const long m0 = mb + a_rg * RPA;
      const float* base = p.dY + (m0 < p.M ? m0 : 0) * p.ldy + (a_col_ok ? a_col : 0);
#pragma unroll
      for (int j = 0; j < RPA; ++j) {
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
  const bool do_bias = !PRE && p.bias_out != nullptr && tap == 0 && first_c;
  auto store_chunk = [&]() __attribute__((always_inline)) {
    // dY: rows a_rg*RPA .. of columns a_c4*4 .. +3: RPA = 8 -> four whole 16-byte groups (one per column) in row group
    // a_rg; RPA = 4 (TN = 64) -> the thread owns half a group (rows 4*(a_rg&1) .. +3), as the X side does for BC = 64
    {
      float cx[4][RPA];
#pragma unroll
      for (int j = 0; j < RPA; ++j) { cx[0][j] = ra[j].x; cx[1][j] = ra[j].y; cx[2][j] = ra[j].z; cx[3][j] = ra[j].w; }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        uint32_t hh[RPA / 2], ll[RPA / 2];
#pragma unroll
        for (int j = 0; j < RPA; j += 2) {
          if (TERMS == 2) {
            hh[j / 2] = occf_f16_pack2(cx[e][j] * f16_scale, cx[e][j + 1] * f16_scale);
            ll[j / 2] = 0u;
          } else {
            occf_bf16_split2(cx[e][j], cx[e][j + 1], hh[j / 2], ll[j / 2]);
          }
        }
        if (RPA == 8) {
          const int off = a_rg * TN + wg_swz(a_c4 * 4 + e, p.swz);
          wg_u4 h, l;
          h.x = hh[0]; h.y = hh[1]; h.z = hh[RPA / 2 - 2]; h.w = hh[RPA / 2 - 1];
          l.x = ll[0]; l.y = ll[1]; l.z = ll[RPA / 2 - 2]; l.w = ll[RPA / 2 - 1];
          Ah[off] = h;
          if (TERMS == 3) Al[off] = l;
        } else {
          const int off = (a_rg >> 1) * TN + wg_swz(a_c4 * 4 + e, p.swz);
          uint32_t* dh = (uint32_t*)(Ah + off) + (a_rg & 1) * 2;
          uint32_t* dl = (uint32_t*)(Al + off) + (a_rg & 1) * 2;
          dh[0] = hh[0];
          dh[1] = hh[1];
          if (TERMS == 3) { dl[0] = ll[0]; dl[1] = ll[1]; }
        }
      }
    }
    if (do_bias) {
#pragma unroll
      for (int j = 0; j < RPA; ++j) { bsum[0] += ra[j].x; bsum[1] += ra[j].y; bsum[2] += ra[j].z; bsum[3] += ra[j].w; }
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
        for (int j = 0; j < RPT; j += 2) {
          if (TERMS == 2) occf_f16_split2(cx[e][j], cx[e][j + 1], hh[j / 2], ll[j / 2]);
          else occf_bf16_split2(cx[e][j], cx[e][j + 1], hh[j / 2], ll[j / 2]);
        }
        if (RPT == 8) {
          const int off = b_rg * BC + wg_swz(b_c4 * 4 + e, p.swz);
          wg_u4 h, l;
          h.x = hh[0]; h.y = hh[1]; h.z = hh[RPT / 2 - 2]; h.w = hh[RPT / 2 - 1];
          l.x = ll[0]; l.y = ll[1]; l.z = ll[RPT / 2 - 2]; l.w = ll[RPT / 2 - 1];
          Bh[off] = h;
          if (TERMS >= 2) Bl[off] = l;
        } else {
          const int off = (b_rg >> 1) * BC + wg_swz(b_c4 * 4 + e, p.swz);
          uint32_t* dh = (uint32_t*)(Bh + off) + (b_rg & 1) * 2;
          uint32_t* dl = (uint32_t*)(Bl + off) + (b_rg & 1) * 2;
          dh[0] = hh[0];
          dh[1] = hh[1];
          if (TERMS >= 2) { dl[0] = ll[0]; dl[1] = ll[1]; }
        }
      }
    }
  };

  f32x16 acc[TI][TC];
#pragma unroll
  for (int i = 0; i < TI; ++i)
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
      bf16x8 ah[TI], al[TI], bh[TC], bl[TC];
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        ah[i] = frag(Ah, TN, wm * (TN / 2) + i * 32 + li, ks);
        if (TERMS == 3) al[i] = frag(Al, TN, wm * (TN / 2) + i * 32 + li, ks);
      }
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        bh[j] = frag(Bh, BC, wn * (BC / 2) + j * 32 + li, ks);
        if (TERMS >= 2) bl[j] = frag(Bl, BC, wn * (BC / 2) + j * 32 + li, ks);
      }
#pragma unroll
      for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TC; ++j) {
          if (TERMS == 2) {
            acc[i][j] = occf_mfma_f16_32x32x16(ah[i], bl[j], acc[i][j]);
            acc[i][j] = occf_mfma_f16_32x32x16(ah[i], bh[j], acc[i][j]);
            continue;
          }
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
  const float unscale = occf_u2f(inv_bits);                     // (TERMS == 2: 2^-k, exact; else 1)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = c0 + wn * (BC / 2) + j * 32 + li;
      if (c >= p.Cin) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * (TN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (n < p.N) o[(long)n * Kt + (long)tap * p.Cin + c] = TERMS == 2 ? acc[i][j][r] * unscale : acc[i][j][r];
      }
    }
  if (do_bias) {
    __syncthreads();
    constexpr int RG = 256 / QA;               // row groups of the dY loader (8 or 16)
    float* red = (float*)Ah;                   // [RG row groups][TN columns]  (RG * TN = 1024 floats)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[a_rg * TN + a_c4 * 4 + e] = bsum[e];
    __syncthreads();
    if (tid < TN && n0 + tid < p.N) {
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < RG; ++r) s += red[r * TN + tid];
      p.bias_out[(long)slab * p.N + n0 + tid] = s;
    }
  }
}

// tile table of one dimension: full 128-wide tiles, then one remainder tile (64 wide if the remainder is <= 64 and
// mixed tiles are enabled, else 128 wide and partly masked); uniform 64-wide tiles when `w` = 64
struct WgTiles { int count, w, rem_w; };
__host__ __device__ __forceinline__ WgTiles wg_tiles(int n, int w, int mix) {
  WgTiles t;
  t.w = w;
  if (w == 64) { t.count = (n + 63) / 64; t.rem_w = 64; return t; }
  const int full = n / 128, rem = n % 128;
  t.count = full + (rem ? 1 : 0);
  t.rem_w = (rem && rem <= 64 && mix) ? 64 : 128;
  return t;
}

template <int TERMS, bool PRE, bool ZF>
__global__ void __launch_bounds__(256, 2) wgrad_kernel(WgArgs p) {
  // [8 row groups][columns] of 16-byte groups (8 bf16 = 8 consecutive rows of one column); sized for the 128 x 128 tile
  __shared__ __attribute__((aligned(16))) wg_u4 Ah[8 * 128], Al[8 * 128], Bh[8 * 128], Bl[8 * 128];
  const WgTiles tn = wg_tiles(p.N, 128, p.mix), tc = wg_tiles(p.Cin, p.bc, p.mix);
  int bx = blockIdx.x;
  int slab = blockIdx.y;
  int nt, ct, tap;
  if (p.xcd_tiles > 0) {
    // All workgroups of one M-slab run on ONE XCD (workgroup w is dispatched to XCD w % 8) and share the slab's rows in
    // that L2 -- as long as they walk the slab in step.  Tiles of different widths do not (a 64 x 64 tile stages half
    // the bytes of a 128 x 128 tile per chunk and runs ahead): with the mixed tiles of the 192 -> 192 convolution in
    // plain tile order the launch fetched 15 GB for 0.99 GB of operands (PMC r03b).  So the order on an XCD is
    // CLASS-major: every (slab of this XCD, tap, tile) of the 128 x 128 class first, then 128 x 64, 64 x 128, 64 x 64 --
    // the ~64 workgroups resident on the XCD at any time are of ONE class and a couple of slabs.
    const int xcd = (int)(blockIdx.x & 7u);
    int k = (int)(blockIdx.x >> 3);
    const int sx = (p.n_slabs - xcd + 7) >> 3;                         // slabs of this XCD: xcd, xcd + 8, ...
    const int n_small = (p.N % 128 && tn.rem_w == 64) ? 1 : 0, n_big = tn.count - n_small;
    const int c_small = tc.w == 64 ? tc.count : ((p.Cin % 128 && tc.rem_w == 64) ? 1 : 0), c_big = tc.count - c_small;
    const int cn[4] = {n_big, n_big, n_small, n_small}, cc[4] = {c_big, c_small, c_big, c_small};
    int cl = 0, T = 0;
    for (; cl < 4; ++cl) {
      T = p.taps * cn[cl] * cc[cl];
      if (k < sx * T) break;
      k -= sx * T;
    }
    if (cl == 4) return;                           // (uniform over the workgroup: XCDs with one slab fewer)
    slab = xcd + 8 * (k / T);
    k %= T;
    ct = k % cc[cl] + ((cl & 1) ? c_big : 0);
    k /= cc[cl];
    tap = k % p.taps;
    nt = k / p.taps + ((cl & 2) ? n_big : 0);
  } else {
    ct = bx % tc.count;
    bx /= tc.count;
    tap = bx % p.taps;
    nt = bx / p.taps;
  }
  const int n0 = nt * 128, c0 = ct * tc.w;
  const int wn_ = nt == tn.count - 1 && p.N % 128 ? tn.rem_w : 128;
  const int wc_ = tc.w == 64 ? 64 : (ct == tc.count - 1 && p.Cin % 128 ? tc.rem_w : 128);
  if (wn_ == 128) {
    if (wc_ == 128) wgrad_body<128, 128, TERMS, PRE, ZF>(p, Ah, Al, Bh, Bl, n0, c0, tap, slab, ct == 0);
    else wgrad_body<128, 64, TERMS, PRE, ZF>(p, Ah, Al, Bh, Bl, n0, c0, tap, slab, ct == 0);
  } else {
    if (wc_ == 128) wgrad_body<64, 128, TERMS, PRE, ZF>(p, Ah, Al, Bh, Bl, n0, c0, tap, slab, ct == 0);
    else wgrad_body<64, 64, TERMS, PRE, ZF>(p, Ah, Al, Bh, Bl, n0, c0, tap, slab, ct == 0);
  }
}

// out[i] = sum_s slab[s][i]  (fixed order: four interleaved slab groups, each summed in order with two alternating
// accumulators, combined 0..3 -- deterministic).  64 elements x 4 slab groups per workgroup: one thread per element
// walking all S slabs serially was latency-bound (23 us per call at S ~ 170, 257 calls = 5.9 ms per training step).
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out,
                                                           long n, int S, const float* __restrict__ slab2,
                                                           float* __restrict__ out2, long n2) {
  // workgroups [0, ceil(n / 64)) reduce the weight partials, the rest (slab2 != NULL) the bias partials: one launch for
  // both (the bias reduction was a launch of its own: ~120 eight-microsecond launches per training step)
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long nb1 = (n + 63) / 64;
  long i = (long)blockIdx.x * 64 + e;
  if ((long)blockIdx.x >= nb1) {
    i = ((long)blockIdx.x - nb1) * 64 + e;
    slab = slab2;
    out = out2;
    n = n2;
  }
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

// the same for M <= 2048 as a 32 x 32 register-tiled contraction staged through LDS: the kernel above re-reads dY and X
// once per OUTPUT (8 lanes walking M with strided loads: 28 us per call for the decoder's 100-row linears, ~120 calls
// per training step); here a workgroup reads its 32 columns of each operand once.  MC = rows staged per pass: 32, or
// -- M <= 128, the decoder's 100-query linears -- 128: ALL rows in one pass, i.e. every load of the call in flight at
// once and ONE barrier instead of a load round trip + two barriers per 32 rows (17 us per call, 2 ms per training
// step in 120 launches, profiles/r05/r05z_train_kernel_stats.txt).
template <int MC>
__global__ void __launch_bounds__(256) wgrad_small_tile_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                               float* __restrict__ dW, float* __restrict__ db, long M,
                                                               int N, int K, long ldy, long ldx) {
  __shared__ float ys[MC][33], xs[MC][33];
  const int ktiles = (K + 31) / 32;
  const int n0 = (int)(blockIdx.x / ktiles) * 32, k0 = (int)(blockIdx.x % ktiles) * 32;
  const int tid = threadIdx.x, n = tid >> 3, kq = (tid & 7) * 4;
  const int lr = tid >> 5, lc = tid & 31;                       // loader: rows lr, lr + 8, .. of column lc
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, bsum = 0.f;
  const bool do_b = db != nullptr && k0 == 0 && (tid & 7) == 0;
  for (long m0 = 0; m0 < M; m0 += MC) {
    // (clamped addresses, masked values: the loads of a pass are one batch)
    float yv[MC / 8], xv[MC / 8];
#pragma unroll
    for (int j = 0; j < MC / 8; ++j) {
      const long m = m0 + lr + 8 * j;
      const long mc = m < M ? m : M - 1;
      yv[j] = dY[mc * ldy + (n0 + lc < N ? n0 + lc : N - 1)];
      xv[j] = X[mc * ldx + (k0 + lc < K ? k0 + lc : K - 1)];
    }
#pragma unroll
    for (int j = 0; j < MC / 8; ++j) {
      const long m = m0 + lr + 8 * j;
      ys[lr + 8 * j][lc] = (m < M && n0 + lc < N) ? yv[j] : 0.f;
      xs[lr + 8 * j][lc] = (m < M && k0 + lc < K) ? xv[j] : 0.f;
    }
    __syncthreads();
    const int rows = M - m0 < MC ? (int)(M - m0) : MC;          // (zero rows beyond M add nothing; skipped)
#pragma unroll 8
    for (int r = 0; r < rows; ++r) {
      const float a = ys[r][n];
      acc[0] = fmaf(a, xs[r][kq + 0], acc[0]);
      acc[1] = fmaf(a, xs[r][kq + 1], acc[1]);
      acc[2] = fmaf(a, xs[r][kq + 2], acc[2]);
      acc[3] = fmaf(a, xs[r][kq + 3], acc[3]);
      if (do_b) bsum += a;
    }
    if (m0 + MC < M) __syncthreads();
  }
  if (n0 + n < N) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + kq + j < K) dW[(long)(n0 + n) * K + k0 + kq + j] = acc[j];
    if (do_b) db[n0 + n] = bsum;
  }
}

#include "occf_absmax.h"

#include "wgrad_g8.h"

static int wg_mix() {
  static const int env = [] {
    const char* e = getenv("OCCF_WG_MIX");               // diagnostics: 0 = uniform (padded) tiles as in round 2
    return e ? atoi(e) : 1;
  }();
  return env;
}
static long wg_tile_count(int N, int Cin, int taps, int BC) {
  return (long)wg_tiles(N, 128, wg_mix()).count * wg_tiles(Cin, BC, wg_mix()).count * taps;
}
static int wg_pick_splits(long M, int N, int Cin, int taps, int BC) {
  const long tiles = wg_tile_count(N, Cin, taps, BC);
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
  const long tiles = wg_tile_count(a.N, a.Cin, a.taps, BC);
  a.bc = BC;
  a.mix = wg_mix();
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
  if (pre && a.zfast) {
    if (terms == 3) hipLaunchKernelGGL((wgrad_kernel<3, true, true>), grid, dim3(256), 0, st, a);
    else if (terms == 2) hipLaunchKernelGGL((wgrad_kernel<2, true, true>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<1, true, true>), grid, dim3(256), 0, st, a);
  } else if (pre) {
    if (terms == 3) hipLaunchKernelGGL((wgrad_kernel<3, true, false>), grid, dim3(256), 0, st, a);
    else if (terms == 2) hipLaunchKernelGGL((wgrad_kernel<2, true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<1, true, false>), grid, dim3(256), 0, st, a);
  } else {
    if (terms == 3) hipLaunchKernelGGL((wgrad_kernel<3, false, false>), grid, dim3(256), 0, st, a);
    else if (terms == 2) hipLaunchKernelGGL((wgrad_kernel<2, false, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((wgrad_kernel<1, false, false>), grid, dim3(256), 0, st, a);
  }
  if (S > 1) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(occf_cdiv((long)a.N * Kt, 64) + (db ? occf_cdiv(a.N, 64) : 0)),
                       dim3(256), 0, st, workspace, dW, (long)a.N * Kt, S, db ? a.bias_out : (const float*)nullptr, db,
                       (long)a.N);
  }
  return (int)hipGetLastError();
}

#define WG_SCALE_SLOT OCCF_ABSMAX_SLOT             // floats kept at the END of every workspace for the fp16 scale (terms == 2)
extern "C" long occf_linear_wgrad_workspace(long M, int N, int K) {
  if (M <= 1024 || N % 4 || K % 4) return 0;
  return wg_workspace(M, N, K, 1) + WG_SCALE_SLOT;
}

extern "C" int occf_linear_wgrad(const float* dy, const float* x, float* dw, float* dbias, float* workspace,
                                 long workspace_floats, long M, int N, int K, long ldy, long ldx, int terms,
                                 void* stream) {
  if (M <= 0 || N <= 0 || K <= 0) return OCCF_EINVAL;
  if (terms != 1 && terms != 2 && terms != 3) return OCCF_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (M <= 1024 || N % 4 || K % 4 || ldy % 4 || ldx % 4) {
    if (M > 65536) return OCCF_ESHAPE;
    if (M <= 2048) {
      if (M <= 128)
        hipLaunchKernelGGL(wgrad_small_tile_kernel<128>, dim3((unsigned)(((N + 31) / 32) * ((K + 31) / 32))), dim3(256), 0,
                           st, dy, x, dw, dbias, M, N, K, ldy, ldx);
      else
        hipLaunchKernelGGL(wgrad_small_tile_kernel<32>, dim3((unsigned)(((N + 31) / 32) * ((K + 31) / 32))), dim3(256), 0,
                           st, dy, x, dw, dbias, M, N, K, ldy, ldx);
      return (int)hipGetLastError();
    }
    const long total = ((long)N * K > N ? (long)N * K : N) * 8;
    hipLaunchKernelGGL(wgrad_small_kernel, dim3(occf_cdiv(total, 256)), dim3(256), 0, st, dy, x, dw, dbias, M, N, K,
                       ldy, ldx);
    return (int)hipGetLastError();
  }
  WgArgs a = {};
  a.dY = dy; a.X = x; a.M = M; a.N = N; a.Cin = K; a.taps = 1; a.ldy = ldy; a.ldx = ldx; a.conv = 0;
  if (terms == 2) {
    // two fp16-piece products: dy * 2^k in ONE piece (k from max |dy|, one more read of dy), x in (hi, lo).  (The
    // product's op layer asks for this in the 27-tap convolutions only: a plain linear is bound by its staging, not by
    // the matrix cores -- r06a: 11.6 ms per step with three bf16 products, no faster with two.)
    if (!workspace || workspace_floats < WG_SCALE_SLOT) terms = 3;
    else {
      workspace_floats -= WG_SCALE_SLOT;
      a.scale = (const uint32_t*)(workspace + workspace_floats);
      wg_absmax(dy, M, N, ldy, (uint32_t*)a.scale, st);
    }
  }
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

// the fp16 forms (terms == 2): y = f16(x * 2^k) with k from scale[0] (no lo array); (hi, lo) = fp16 halves of x
__global__ void __launch_bounds__(256) wg_split_f16_y_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi,
                                                             long n2, const uint32_t* __restrict__ scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  uint32_t inv;
  const float sc = occf_u2f(occf_f16_scale_bits(scale[0], inv));
  ((uint32_t*)hi)[i] = occf_f16_pack2(x[2 * i] * sc, x[2 * i + 1] * sc);
}
__global__ void __launch_bounds__(256) wg_split_f16_x_kernel(const float* __restrict__ x, uint16_t* __restrict__ hi,
                                                             uint16_t* __restrict__ lo, long n2) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  uint32_t h, l;
  occf_f16_split2(x[2 * i], x[2 * i + 1], h, l);
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
  const long M = (long)B * Xo * Yo * Zo, nx = (long)B * Xi * Yi * Zi * Cin, ny = M * Cout;
  if (wg_presplit_ok(Cin, Cout, kX * kY * kZ))        // bf16 (hi, lo) copies of dy and x: 4 bytes per element
    need += ny + nx;
  need += WG_SCALE_SLOT;
  if (Xo > 0 && Yo > 0 && Zo == Zi && wg8_eligible(Zi, Cin, Cout, kX, kY, kZ, stride, dil, pad_z, M, nx, ny)) {
    const long g8 = wg8_workspace(B * Xo, Yo, Zi / 8, Cout, Cin, kX * kY * kZ, nx, ny);  // x in three z-shifted copies
    if (g8 + WG_SCALE_SLOT > need) need = g8 + WG_SCALE_SLOT;
  }
  return need;
}

extern "C" int occf_conv3d_wgrad(const float* dy, const float* x, float* dw_tapmajor, float* dbias, float* workspace,
                                 long workspace_floats, int B, int Xi, int Yi, int Zi, int Cin, int Cout, int kX,
                                 int kY, int kZ, int stride, int dil, int pad_x, int pad_y, int pad_z, long in_sb,
                                 long in_sx, long in_sy, long in_sz, int terms, void* stream) {
  if (B <= 0 || Cin % 4 || Cout % 4 || stride <= 0 || dil <= 0) return OCCF_ESHAPE;
  if (in_sb % 4 || in_sx % 4 || in_sy % 4 || in_sz % 4) return OCCF_ESHAPE;
  if (terms != 1 && terms != 2 && terms != 3 && terms != 4) return OCCF_EINVAL;
  // terms == 4: ONE product -- dy and x each in one fp16 piece -- where the G8 kernel applies; elsewhere it means 2
  const bool x1 = terms == 4;
  if (x1) terms = 2;
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
  if (dense && !dbias && workspace && g.Zo == Zi &&
      wg8_eligible(Zi, Cin, Cout, kX, kY, kZ, stride, dil, pad_z, a.M, nx, ny) &&
      workspace_floats >= wg8_workspace(B * g.Xo, g.Yo, Zi / 8, Cout, Cin, a.taps, nx, ny)) {
    Wg8Args q = {};
    const Wg8Geom gm = wg8_geometry(B * g.Xo, g.Yo, Zi / 8, Cout, Cin, a.taps);
    const int S = gm.n_slabs;
    const long Kt = (long)a.taps * Cin;
    float* part = S > 1 ? workspace : dw_tapmajor;
    uint16_t* yh = (uint16_t*)(workspace + (long)S * Cout * Kt);
    uint16_t* yl = yh + ny;
    uint16_t* xh = yl + ny;
    uint16_t* xl = xh + 3 * nx;
    const int ZG = Zi / 8;
    const long cols = (long)B * Xi * Yi;
    // terms == 2: ONE fp16 piece of dy * 2^k (k from max |dy|) times the fp16 (hi, lo) halves of x -- two products per
    // product; the scale slot lives in the (then unused) dY-lo array
    uint32_t* scale = (uint32_t*)yl;                         // (ny * 2 bytes >= WG_SCALE_SLOT floats: M >= 1024, Cout >= 64)
    if (terms == 2) {
      wg_absmax(dy, a.M, Cout, (long)Cout, scale, st);
      hipLaunchKernelGGL(wg8_split_y_kernel<true>, dim3(occf_cdiv((a.M / 8) * Cout, 256)), dim3(256), 0, st, dy,
                         (long)Cout, a.M / 8, Cout, (wg_u4*)yh, (wg_u4*)yl, scale);
      if (x1)
        hipLaunchKernelGGL((wg8_split_x_kernel<true, true>), dim3(occf_cdiv(cols * ZG * Cin, 256)), dim3(256), 0, st, x,
                           cols, ZG, Cin, (wg_u4*)xh, (wg_u4*)xl, cols * ZG * Cin);
      else
        hipLaunchKernelGGL(wg8_split_x_kernel<true>, dim3(occf_cdiv(cols * ZG * Cin, 256)), dim3(256), 0, st, x, cols, ZG,
                           Cin, (wg_u4*)xh, (wg_u4*)xl, cols * ZG * Cin);
    } else {
      hipLaunchKernelGGL(wg8_split_y_kernel<false>, dim3(occf_cdiv((a.M / 8) * Cout, 256)), dim3(256), 0, st, dy,
                         (long)Cout, a.M / 8, Cout, (wg_u4*)yh, (wg_u4*)yl, scale);
      hipLaunchKernelGGL(wg8_split_x_kernel<false>, dim3(occf_cdiv(cols * ZG * Cin, 256)), dim3(256), 0, st, x, cols, ZG,
                         Cin, (wg_u4*)xh, (wg_u4*)xl, cols * ZG * Cin);
    }
    q.scale = scale;
    q.Yh = yh; q.Yl = yl; q.Xh = xh; q.Xl = xl; q.out = part; q.xcopy_elems = nx;
    q.n_strips = gm.n_strips; q.strip_w = gm.strip_w; q.seg_planes = gm.seg_planes; q.planes = B * g.Xo;
    q.slabs_per_xcd = (S + 7) / 8;
    q.N = Cout; q.Cin = Cin; q.taps = a.taps; q.ybytes = (uint32_t)(ny * 2); q.xbytes = (uint32_t)(nx * 2);
    q.B = B; q.Xo = g.Xo; q.Yo = g.Yo; q.Xi = Xi; q.Yi = Yi; q.ZG = ZG;
    q.zg_shift = ZG == 1 ? 0 : ZG == 2 ? 1 : ZG == 4 ? 2 : 3;
    q.kX = kX; q.kY = kY; q.kZ = kZ; q.pad_x = pad_x; q.pad_y = pad_y; q.n_slabs = S;
    Wg8Seg sn[2], sc[2];
    const int nn = wg8_segments(Cout, sn), nc = wg8_segments(Cin, sc);
    for (int i = 0; i < nn; ++i)
      for (int j = 0; j < nc; ++j) {
        if (!sn[i].count || !sc[j].count) continue;
        q.n_base = sn[i].base; q.tn_count = sn[i].count; q.c_base = sc[j].base; q.tc_count = sc[j].count;
        q.tiles = a.taps * sn[i].count * sc[j].count;
        if (terms == 2 && x1) wg8_launch_class_x1(q, sn[i].w / 64, sc[j].w / 64, st);
        else if (terms == 2) wg8_launch_class<true>(q, sn[i].w / 64, sc[j].w / 64, st);
        else wg8_launch_class<false>(q, sn[i].w / 64, sc[j].w / 64, st);
      }
    if (S > 1)
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(occf_cdiv((long)Cout * Kt, 64)), dim3(256), 0, st, part, dw_tapmajor,
                         (long)Cout * Kt, S, (const float*)nullptr, (float*)nullptr, 0L);
    return (int)hipGetLastError();
  }
  if (terms == 2) {
    // (only where the operands are pre-split anyway: >= 9 taps re-use each staged row often enough for the matrix
    // cores to matter)
    if (!workspace || workspace_floats < WG_SCALE_SLOT + nx + ny || !wg_presplit_ok(Cin, Cout, a.taps) || !dense || dbias)
      terms = 3;
    else {
      workspace_floats -= WG_SCALE_SLOT;
      a.scale = (const uint32_t*)(workspace + workspace_floats);
      wg_absmax(dy, a.M, Cout, (long)Cout, (uint32_t*)a.scale, st);
    }
  }
  if (wg_presplit_ok(Cin, Cout, a.taps) && dense && !dbias && workspace && workspace_floats >= nx + ny) {
    uint16_t* yh = (uint16_t*)(workspace + workspace_floats - (nx + ny));
    uint16_t* yl = yh + ny;
    uint16_t* xh = yl + ny;
    uint16_t* xl = xh + nx;
    if (terms == 2) {
      hipLaunchKernelGGL(wg_split_f16_y_kernel, dim3(occf_cdiv(ny / 2, 256)), dim3(256), 0, st, dy, yh, ny / 2, a.scale);
      hipLaunchKernelGGL(wg_split_f16_x_kernel, dim3(occf_cdiv(nx / 2, 256)), dim3(256), 0, st, x, xh, xl, nx / 2);
      yl = yh;                                             // (the lo-half staging threads load dY too; never stored)
    } else {
      hipLaunchKernelGGL(wg_split_kernel, dim3(occf_cdiv(ny / 2, 256)), dim3(256), 0, st, dy, yh, yl, ny / 2);
      hipLaunchKernelGGL(wg_split_kernel, dim3(occf_cdiv(nx / 2, 256)), dim3(256), 0, st, x, xh, xl, nx / 2);
    }
    a.dYh = yh; a.dYl = yl; a.Xh = xh; a.Xl = xl;
    workspace_floats -= nx + ny;
    static const int zfast_env = [] {
      const char* e = getenv("OCCF_WG_ZFAST");
      return e ? atoi(e) : 1;
    }();
    if (zfast_env && (g.Zo == 8 || g.Zo == 16 || g.Zo == 32 || g.Zo == 64) && a.M < 2147483647L) {
      static const int buf_env = [] {
        const char* e = getenv("OCCF_WG_BUFLOAD");
        return e ? atoi(e) : 1;
      }();
      a.zfast = (buf_env && ny * 2 < 2147483647L && nx * 2 < 2147483647L) ? 1 : 0;
      a.ybytes = (uint32_t)(ny * 2);
      a.xbytes = (uint32_t)(nx * 2);
      a.zshift = g.Zo == 8 ? 3 : g.Zo == 16 ? 4 : g.Zo == 32 ? 5 : 6;
    }
  }
  return wg_launch(a, dw_tapmajor, dbias, workspace, workspace_floats, terms, st);
}
