// 3x3x3 (stride 1, pad 1) convolution over a channels-last voxel volume with an LDS-resident
// halo tile -- the dominant FLOP item of the hot path (1885 + 1353 GFLOP at the
// 200-grid; dualpath_block.py:43-48, multiscale_deformattn_3d.py:101-110).
//
// The generic implicit-GEMM kernel (gemm_bf16.hip) re-reads and re-splits every input voxel
// once per tap (27x) and streams it through L2; here a workgroup stages the (2+2) x (TY+2) x
// (TZ+2) halo of a 2 x TY x TZ output tile ONCE per 32-channel chunk into LDS (already split
// into bf16 hi/lo) and all 27 taps read their shifted A fragments straight from that tile:
// 4.2x instead of 27x input traffic, one fp32->bf16x2 split per staged element.
//
// Workgroup = 512 threads = 8 waves as 4 (M) x 2 (N); output tile 256 voxels x (64*TN) channels;
// per chunk: 27 taps x (k = 32) on v_mfma_f32_32x32x16_bf16 (3-term split or plain bf16).
// The weight slab of one (tap, chunk) [BN x 32] (pre-split bf16) is double-buffered in LDS; the
// next tap's slab is fetched into registers while the current tap multiplies.
// LDS rows are 64 B with the 16-B k-slots XOR-swizzled by (row>>2)&3 (see gemm_bf16.hip).
// Optionally emits per-workgroup GroupNorm partial sums of the raw outputs.
#include <stdlib.h>
#include "occf_common.h"
#include "../../include/occformer_hip.h"

struct ConvHaloArgs {
  const float* x;
  const uint16_t* Wh;
  const uint16_t* Wl;
  const float* bias;
  const float* residual;
  float* out;
  int B, X, Y, Z, Cin, Cout;
  int TY, TZ;                 // TY * TZ == 128, TZ | Z
  long sb, sx, sy, sz;        // input element strides (channel stride 1)
  int act;
  float* gn_partial;          // optional [B][spatial tiles][Cout][2]: per-channel sum / sum of squares of the outputs
  // optional weights in FRAGMENT order (occf_conv3x3x3_halo_pack): [chunk][tap][k-step][32-column tile][lane][8] --
  // the B operand of one (tap, chunk, k-step, column tile) is 1 KB contiguous, one 16-byte load per lane
  const uint16_t* Fh;
  const uint16_t* Fl;
};

__device__ __forceinline__ uint32_t ch_bf16_rne(float x) {
#ifdef OCCF_EMU
  uint32_t u;
  memcpy(&u, &x, 4);
#else
  const uint32_t u = __float_as_uint(x);
#endif
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ float ch_bf16_up(uint32_t h) {
#ifdef OCCF_EMU
  uint32_t u = h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
#else
  return __uint_as_float(h << 16);
#endif
}
__device__ __forceinline__ void ch_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  occf_bf16_split2(a, b, hi, lo);
}
__device__ __forceinline__ int ch_slot(int row, int kslot) { return row * 64 + ((kslot ^ ((row >> 2) & 3)) << 4); }
__device__ __forceinline__ float ch_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// ds_read_b128 is serviced in four NON-contiguous 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): the
// FRAG variant lets operand row m of a 32-row MFMA tile hold tile voxel ch_pos(m), so that each group reads 16
// CONSECUTIVE halo rows (one z-line), which 80-byte rows place on sixteen distinct 4-bank slots for every tap offset.
// With the identity mapping a group spans two z-lines 18 rows apart and 2 of its 16 lanes always collide (PMC: bank
// conflicts = 50 % of the LDS cycles).  4-aligned runs of m map contiguously: run m/4 starts at {0,16,20,4,24,8,12,28}.
__device__ __forceinline__ int ch_pos(int m) {
  const int run = m >> 2;
  const int start = run == 0 ? 0 : run == 1 ? 16 : run == 2 ? 20 : run == 3 ? 4 : run == 4 ? 24 : run == 5 ? 8 : run == 6 ? 12 : 28;
  return start + (m & 3);
}

typedef uint32_t ch_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t ch_u2 __attribute__((ext_vector_type(2)));

// Global loads are unconditional (clamped address, masked value) and issued in batches so that the
// compiler never has to wait inside a branch: the halo of a chunk is fetched as one batch of <= 13
// float4 per thread -- for TN <= 2 already during the taps of the previous chunk -- and the weight slabs
// run two taps ahead in a register ring.
// FRAG: the weight fragments come straight from global memory (L1 / L2) in fragment order into registers, one
// k-step ahead of the MFMAs that use them -- no weight slabs in LDS and NO BARRIER PER TAP (the slab double buffer
// needed one; PMC r02: MFMA pipe 50 % busy, 48 % of the wave cycles waiting): the 8 waves only meet when the halo
// tile changes, every 27 taps.
// SCH (FRAG only): the k-step as an explicit software pipeline -- MFMAs issued term-major (six accumulators between two
// products into the same one), the NEXT k-step's four A reads and 2 * TN B loads placed one per MFMA at the head of
// the current k-step (sched_group_barrier), no conditional load in the loop.  The compiler's own order sank the
// loads to 1-5 MFMAs in front of their first use to save registers (ISA: s_waitcnt lgkmcnt one MFMA after four
// ds_read_b128, vmcnt four MFMAs after the loads).
// WM = waves along M: 4 (workgroup = 8 waves, 256 output voxels, halo tile 115 KB: ONE workgroup per CU) or 2 (round 5, opt-in:
// 4 waves, 128 voxels = 2 x-planes x (TY * TZ = 64), halo tile <= 69 KB: TWO workgroups per CU, so that one's chunk
// boundary -- barrier, a halo round trip to HBM / L2, the split pass, barrier: the whole CU idle with one workgroup --
// runs under the other's taps; FRAG only)
template <int TN, int TERMS, bool FRAG, int SCH = 0, int WM = 4>
__global__ void __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) conv3x3x3_halo_kernel(ConvHaloArgs p) {
  static_assert(WM == 4 || (WM == 2 && FRAG), "the half-size tile exists for the fragment variant only");
  constexpr int NT = WM * 128;                        // threads
  constexpr int PLS = WM == 4 ? 7 : 6;                // log2(TY * TZ): voxels of one x-plane of the tile
  constexpr int BN = 64 * TN;
  constexpr int NBP = (BN * 4 + 511) / 512;          // 16-B weight pieces per thread per array (slab variant)
  constexpr int NHI = WM == 4 ? 13 : 14;              // float4 halo pieces per thread (816 * 8 / 512; 432 * 8 / 256)
  constexpr bool HPF = TN <= 2;                       // prefetch the next chunk's halo across the taps
  OCCF_DYN_SMEM(smem);
  const int TY = p.TY, TZ = p.TZ;
  const int HY = TY + 2, HZ = TZ + 2;
  const int NH = 4 * HY * HZ;                         // halo rows (voxels)
  unsigned char* Hh = (unsigned char*)smem;           // [NH][64 B]
  // halo rows: 64 B with XOR-swizzled k-slots (slab variant), or -- FRAG -- 80 B rows, no swizzle: 16 consecutive rows
  // then start at banks 20 r mod 64 = sixteen disjoint 4-bank groups, and a fragment address is LINEAR in the row, so
  // the tap offset is one scalar-derived add per tap and everything else is an instruction immediate (the swizzle
  // arithmetic was ~30 of the ~85 VALU instructions per tap; VALU issue does not overlap the partner wave's MFMAs)
  constexpr int HROW = FRAG ? 80 : 64;
  unsigned char* Hl = Hh + (size_t)NH * HROW;
  unsigned char* Bh = Hl + (TERMS == 3 ? (size_t)NH * HROW : 0);     // [2][BN][64 B] (slab variant only)
  unsigned char* Bl = Bh + 2 * BN * 64;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;

  // workgroup -> (b, x-pair, y-tile, z-tile, n-tile); n fastest so the workgroups sharing a halo
  // are neighbours, then z, y, x: an XCD's contiguous range is a slab of x-planes
  const int n_tiles = (p.Cout + BN - 1) / BN;
  const int zt = p.Z / TZ, yt = (p.Y + TY - 1) / TY, xt = (p.X + 1) / 2;
  unsigned wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % n_tiles; wg /= n_tiles;
  const int tz0 = (wg % zt) * TZ; wg /= zt;
  const int ty0 = (wg % yt) * TY; wg /= yt;
  const int tx0 = (wg % xt) * 2;
  const int b = wg / xt;
  const int n0 = nt * BN;

  // halo base index of this lane's two A rows at tap (0,0,0)
  int hb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = wm * 64 + i * 32 + (FRAG ? ch_pos(li) : li);
    const int tx = r >> PLS, pp = r & ((1 << PLS) - 1);
    hb[i] = (tx * HY + pp / TZ) * HZ + pp % TZ;
  }
  // this thread's halo pieces: clamped element offset (channel 0 of the chunk) + validity bit
  const float* xb = p.x + (long)b * p.sb;
  const int nhp = NH * 8;
  int hofs[NHI];
  unsigned hok = 0;
#pragma unroll
  for (int i = 0; i < NHI; ++i) {
    const int idx = tid + i * NT;
    const int idc = idx < nhp ? idx : nhp - 1;
    const int h = idc >> 3, kq = idc & 7;
    const int hz = h % HZ, hy = (h / HZ) % HY, hx = h / (HZ * HY);
    const int x = tx0 + hx - 1, y = ty0 + hy - 1, z = tz0 + hz - 1;
    const bool ok = idx < nhp && x >= 0 && x < p.X && y >= 0 && y < p.Y && z >= 0 && z < p.Z;
    hofs[i] = (int)(occf_clampi(x, p.X - 1) * p.sx + occf_clampi(y, p.Y - 1) * p.sy +
                    occf_clampi(z, p.Z - 1) * p.sz) + kq * 4;
    hok |= (ok ? 1u : 0u) << i;
  }
  // weight piece bookkeeping (columns >= Cout read column Cout-1; never stored)
  int b_slot[NBP], b_row[NBP];
  long b_off[NBP];
  const long K = 27L * p.Cin;
#pragma unroll
  for (int i = 0; i < NBP; ++i) {
    const int idx = tid + i * 512;
    const int idc = idx < BN * 4 ? idx : BN * 4 - 1;
    b_row[i] = idc >> 2;
    b_slot[i] = idc & 3;
    const int n = n0 + b_row[i];
    b_off[i] = (long)(n < p.Cout ? n : p.Cout - 1) * K + b_slot[i] * 8;
  }

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  constexpr int HB = HPF ? NHI : 7;                   // halo pieces per batch (register budget of TN = 3)
  float4 hreg[HB];
  ch_u4 rbh[2][NBP], rbl[2][NBP];
  const int n_chunks = p.Cin / 32;
  const int G = n_chunks * 27;                        // flat (chunk, tap) stream
  auto load_halo = [&](int c0, int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < HB; ++i)
      if (i0 + i < NHI) hreg[i] = *(const float4*)(xb + hofs[i0 + i] + c0);
  };
  auto store_halo = [&](int i0) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < HB; ++ii) {
      const int i = i0 + ii;
      const int idx = tid + i * NT;
      if (i < NHI && idx < nhp) {
        const bool ok = (hok >> i) & 1u;
        const float4 v = hreg[ii];
        uint32_t h0, l0, h1, l1;
        ch_split2(ok ? v.x : 0.f, ok ? v.y : 0.f, h0, l0);
        ch_split2(ok ? v.z : 0.f, ok ? v.w : 0.f, h1, l1);
        const ch_u2 hi = {h0, h1}, lo = {l0, l1};
        const int h = idx >> 3, kq = idx & 7;
        const int off = FRAG ? h * HROW + kq * 8 : ch_slot(h, kq >> 1) + (kq & 1) * 8;
        *(ch_u2*)(Hh + off) = hi;
        if (TERMS == 3) *(ch_u2*)(Hl + off) = lo;
      }
    }
  };
  // weight slab of stream position g (clamped to the last one) -> ring slot d
  auto load_b = [&](int g, int d) __attribute__((always_inline)) {
    const int gc = g < G ? g : G - 1;
    const int cc = gc / 27, tap = gc - cc * 27;
    const long o = (long)tap * p.Cin + cc * 32;
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      rbh[d][i] = *(const ch_u4*)(p.Wh + b_off[i] + o);
      if (TERMS == 3) rbl[d][i] = *(const ch_u4*)(p.Wl + b_off[i] + o);
    }
  };
  auto store_b = [&](int buf, int d) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
      if (tid + i * 512 < BN * 4) {
        const int off = buf * BN * 64 + ch_slot(b_row[i], b_slot[i]);
        *(ch_u4*)(Bh + off) = rbh[d][i];
        if (TERMS == 3) *(ch_u4*)(Bl + off) = rbl[d][i];
      }
    }
  };

  if (FRAG) {
    const int ngrp = p.Cout >> 5;
#ifdef OCCF_EMU
    const int jg0 = (n0 + wn * (BN / 2)) >> 5;
#else
    const int jg0 = __builtin_amdgcn_readfirstlane((n0 + wn * (BN / 2)) >> 5);     // wave-uniform: scalar base address
#endif
    const unsigned lane8 = (unsigned)lane * 8u;
    // B fragments of stream position g (clamped), k-step s
    auto load_f = [&](int g, int s, bf16x8 (&fh)[TN], bf16x8 (&fl)[TN]) __attribute__((always_inline)) {
      const int gc = g < G ? g : G - 1;
      const long o = ((long)(gc * 2 + s) * ngrp + jg0) * 512;          // scalar
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // (scalar base + UNSIGNED 32-bit lane offset: the saddr + voffset addressing mode, no 64-bit VALU adds)
        fh[j] = *(const bf16x8*)(p.Fh + o + j * 512 + lane8);
        if (TERMS == 3) fl[j] = *(const bf16x8*)(p.Fl + o + j * 512 + lane8);
      }
    };
    // A fragments (this lane's two halo rows at tap offset toff, k-step s) are read from LDS one k-step AHEAD of the
    // MFMAs that use them as well: a ds_read issued right in front of its MFMAs costs its full latency every k-step
    // byte address of this lane's first row (i = 0), k-slot lk, tap (0,0,0); row i = 1 is 32 / TZ halo y-rows further
    const int a_base = hb[0] * HROW + lk * 16;
    const int a_i1 = (hb[1] - hb[0]) * HROW;            // wave-uniform
    const int hl_off = (int)((size_t)NH * HROW);
    auto load_a = [&](int toff, int s, bf16x8 (&ah)[2], bf16x8 (&al)[2]) __attribute__((always_inline)) {
      const unsigned char* ap = Hh + a_base + toff * HROW + s * 32;
      ah[0] = *(const bf16x8*)(ap);
      ah[1] = *(const bf16x8*)(ap + a_i1);
      if (TERMS == 3) {
        al[0] = *(const bf16x8*)(ap + hl_off);
        al[1] = *(const bf16x8*)(ap + hl_off + a_i1);
      }
    };
    auto mma_step = [&](const bf16x8 (&ah)[2], const bf16x8 (&al)[2], const bf16x8 (&fh)[TN],
                        const bf16x8 (&fl)[TN]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if (TERMS == 3) {
            acc[i][j] = occf_mfma_bf16_32x32x16(al[i], fh[j], acc[i][j]);
            acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fl[j], acc[i][j]);
          }
          acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fh[j], acc[i][j]);
        }
    };
    auto tap_off = [&](int tap) __attribute__((always_inline)) -> int {
      const int dz = tap % 3, dy = (tap / 3) % 3, dx = tap / 9;
      return (dx * HY + dy) * HZ + dz;
    };
    bf16x8 f0h[TN], f0l[TN], f1h[TN], f1l[TN];
    bf16x8 a0h[2], a0l[2], a1h[2], a1l[2];
    if (HPF) load_halo(0, 0);
    load_f(0, 0, f0h, f0l);
    int cc = 0, tap = 0;
    if (SCH) {
      auto mma_tm = [&](const bf16x8 (&ah)[2], const bf16x8 (&al)[2], const bf16x8 (&fh)[TN],
                        const bf16x8 (&fl)[TN]) __attribute__((always_inline)) {
        if (TERMS == 3) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(al[i], fh[j], acc[i][j]);
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fl[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fh[j], acc[i][j]);
      };
      constexpr int NA = TERMS == 3 ? 4 : 2;             // ds_read_b128 per k-step
      constexpr int NF = TERMS == 3 ? 2 * TN : TN;       // global 16-byte loads per k-step
      constexpr int NM = 2 * TN * TERMS;                 // MFMAs per k-step
      for (int g = 0; g < G; ++g) {
        if (tap == 0) {
          __syncthreads();
          if (HPF) {
            store_halo(0);
            load_halo((cc + 1 < n_chunks ? cc + 1 : cc) * 32, 0);
          } else {
#pragma unroll
            for (int i0 = 0; i0 < NHI; i0 += HB) {
              load_halo(cc * 32, i0);
              store_halo(i0);
            }
          }
          __syncthreads();
          load_a(0, 0, a0h, a0l);
        }
        const int toff = tap_off(tap);
        const int tnx = tap_off(tap < 26 ? tap + 1 : 0);   // (tap 26: a harmless read, replaced after the staging)
        OCCF_SCHED_FENCE();
        load_a(toff, 1, a1h, a1l);
        load_f(g, 1, f1h, f1l);
        mma_tm(a0h, a0l, f0h, f0l);
        load_a(tnx, 0, a0h, a0l);
        load_f(g + 1, 0, f0h, f0l);
        mma_tm(a1h, a1l, f1h, f1l);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int i = 0; i < NA; ++i) { OCCF_SCHED_GROUP(0x008, 1); OCCF_SCHED_GROUP(0x100, 1); }
#pragma unroll
          for (int i = 0; i < NF; ++i) { OCCF_SCHED_GROUP(0x008, 1); OCCF_SCHED_GROUP(0x020, 1); }
          OCCF_SCHED_GROUP(0x008, (NM > NA + NF ? NM - NA - NF : 0));   // (TN = 1, one term: 2 MFMAs per k-step, 3 slots asked)
        }
        OCCF_SCHED_FENCE();
        if (++tap == 27) { tap = 0; ++cc; }
      }
    } else
    for (int g = 0; g < G; ++g) {
      if (tap == 0) {
        __syncthreads();                                 // previous chunk's taps are done with the halo
        if (HPF) {
          store_halo(0);
          load_halo((cc + 1 < n_chunks ? cc + 1 : cc) * 32, 0);
        } else {
#pragma unroll
          for (int i0 = 0; i0 < NHI; i0 += HB) {
            load_halo(cc * 32, i0);
            store_halo(i0);
          }
        }
        __syncthreads();
        load_a(0, 0, a0h, a0l);                          // tap 0 of the new tile
      }
      const int toff = tap_off(tap);
      load_a(toff, 1, a1h, a1l);
      load_f(g, 1, f1h, f1l);
      mma_step(a0h, a0l, f0h, f0l);
      if (tap < 26) load_a(tap_off(tap + 1), 0, a0h, a0l);   // (tap 26: the next tile is staged first)
      load_f(g + 1, 0, f0h, f0l);
      mma_step(a1h, a1l, f1h, f1l);
      if (++tap == 27) { tap = 0; ++cc; }
    }
  } else {
  // ring slot (g & 1) holds slab g; LDS buffer (g & 1) holds slab g while it is multiplied
  if (HPF) load_halo(0, 0);
  load_b(0, 0);
  load_b(1, 1);
  store_b(0, 0);
  int cc = 0, tap = 0;
  for (int g0 = 0; g0 < G; g0 += 2) {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const int g = g0 + d;
      if (g < G) {
        if (g > 0) load_b(g + 1, (d + 1) & 1);          // slab g+1 (slot freed when slab g-1 went to LDS)
        if (tap == 0) {
          __syncthreads();                               // previous chunk's taps are done with the halo
          if (HPF) {
            store_halo(0);
            load_halo((cc + 1 < n_chunks ? cc + 1 : cc) * 32, 0);
          } else {
#pragma unroll
            for (int i0 = 0; i0 < NHI; i0 += HB) {
              load_halo(cc * 32, i0);
              store_halo(i0);
            }
          }
          __syncthreads();
        }
        const int dz = tap % 3, dy = (tap / 3) % 3, dx = tap / 9;
        const int toff = (dx * HY + dy) * HZ + dz;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int kslot = s * 2 + lk;
          bf16x8 ah[2], al[2], bh[TN], bl[TN];
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int off = ch_slot(hb[i] + toff, kslot);
            ah[i] = *(const bf16x8*)(Hh + off);
            if (TERMS == 3) al[i] = *(const bf16x8*)(Hl + off);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int off = d * BN * 64 + ch_slot(wn * (BN / 2) + j * 32 + li, kslot);
            bh[j] = *(const bf16x8*)(Bh + off);
            if (TERMS == 3) bl[j] = *(const bf16x8*)(Bl + off);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              if (TERMS == 3) {
                acc[i][j] = occf_mfma_bf16_32x32x16(al[i], bh[j], acc[i][j]);
                acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bl[j], acc[i][j]);
              }
              acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bh[j], acc[i][j]);
            }
        }
        if (g + 1 < G) store_b((d + 1) & 1, (d + 1) & 1);
        __syncthreads();
        if (++tap == 27) { tap = 0; ++cc; }
      }
    }
  }
  }

  // ---- epilogue: row r of the tile -> voxel (tx0 + r>>7, ty0 + (r&127)/TZ, tz0 + (r&127)%TZ)
  float gs[TN], gq[TN];                                   // GroupNorm partial sums of this lane's columns
#pragma unroll
  for (int j = 0; j < TN; ++j) gs[j] = gq[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + li;
      const bool n_ok = n < p.Cout;
      const int nc = n_ok ? n : p.Cout - 1;
      const float bv = p.bias ? p.bias[nc] : 0.f;
      long mrow[16];
      bool mok[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mrow_ = (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int row = wm * 64 + i * 32 + (FRAG ? ch_pos(mrow_) : mrow_);
        const int pp = row & ((1 << PLS) - 1);
        const int x = tx0 + (row >> PLS), y = ty0 + pp / TZ, z = tz0 + pp % TZ;
        mok[r] = n_ok && x < p.X && y < p.Y;
        mrow[r] = ((((long)b * p.X + occf_clampi(x, p.X - 1)) * p.Y + occf_clampi(y, p.Y - 1)) * p.Z + z) * p.Cout;
      }
      float rv[16];
      if (p.residual) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[r] = p.residual[mrow[r] + nc];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[i][j][r] + bv;
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = ch_gelu(v);
        if (p.residual) v += rv[r];
        if (mok[r]) {
          p.out[mrow[r] + n] = v;
          gs[j] += v;
          gq[j] = fmaf(v, v, gq[j]);
        }
      }
    }
  }
  if (p.gn_partial) {
    // deterministic workgroup reduction: lanes (lk) -> LDS [wm][channel] -> channel -> group
    float* red = (float*)smem;                             // [WM][BN][2]
    __syncthreads();                                        // every wave is out of the tap loop (halo LDS is free)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float a = gs[j] + __shfl_xor(gs[j], 32), q = gq[j] + __shfl_xor(gq[j], 32);
      if (lk == 0) {
        const int c = wn * (BN / 2) + j * 32 + li;
        red[(wm * BN + c) * 2 + 0] = a;
        red[(wm * BN + c) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < p.Cout) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { a += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
      const long sp = ((long)(tx0 >> 1) * yt + ty0 / TY) * zt + tz0 / TZ;
      float* o = p.gn_partial + ((((long)b * xt * yt * zt) + sp) * p.Cout + n0 + tid) * 2;
      o[0] = a;
      o[1] = q;
    }
  }
}

static size_t conv_halo_lds(int TY, int TZ, int TN, int terms, bool frag) {
  const size_t NH = 4 * (size_t)(TY + 2) * (TZ + 2);
  if (frag) return NH * 80 * (terms == 3 ? 2 : 1);                    // 80-byte halo rows, no weight slabs
  return NH * 64 * (terms == 3 ? 2 : 1) + (size_t)2 * 64 * TN * 64 * (terms == 3 ? 2 : 1);
}

typedef void (*conv_halo_fn_t)(ConvHaloArgs);
template <int TN>
static conv_halo_fn_t conv_halo_fn(bool t3, bool frag, int sch, bool small) {
  if (small && sch) return t3 ? conv3x3x3_halo_kernel<TN, 3, true, 1, 2> : conv3x3x3_halo_kernel<TN, 1, true, 1, 2>;
  if (small) return t3 ? conv3x3x3_halo_kernel<TN, 3, true, 0, 2> : conv3x3x3_halo_kernel<TN, 1, true, 0, 2>;
  if (frag && sch) return t3 ? conv3x3x3_halo_kernel<TN, 3, true, 1> : conv3x3x3_halo_kernel<TN, 1, true, 1>;
  if (frag) return t3 ? conv3x3x3_halo_kernel<TN, 3, true> : conv3x3x3_halo_kernel<TN, 1, true>;
  return t3 ? conv3x3x3_halo_kernel<TN, 3, false> : conv3x3x3_halo_kernel<TN, 1, false>;
}

// OCCF_HALO_SMALL: 1 = half-size tiles, two workgroups per CU (fragment variant); 0 (default) = 256-voxel tiles.
// Measured (r05j / r05zz): the same 3.19 ms on the 192-channel launch, +2 ... 4 % on the 128-channel ones, but 2.12 GB of
// HBM traffic per 192-channel launch instead of 1.72 GB (the halo is 3.4x the tile instead of 2.8x) -- an opt-in.  The
// choice depends on nothing but this switch and on the fragments being passed: occf_conv3x3x3_halo_gn_blocks answers
// for the fragment call, and a call WITHOUT fragments that asks for GroupNorm partials under small tiles is refused
// (OCCF_ESHAPE: the caller's buffer has the small-tile row count).
static int conv_halo_small() {
  const char* e = getenv("OCCF_HALO_SMALL");
  return e ? atoi(e) : 0;
}

// OCCF_HALO_SCHED: 1 (default) = the explicitly pipelined k-step of the FRAG variant, 0 = the compiler's order
static int conv_halo_sched() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("OCCF_HALO_SCHED");
    v = e ? atoi(e) : 1;
  }
  return v;
}

// w[Cout][27 * Cin] (tap-major rows, bf16) -> fragment order [chunk][tap][k-step][Cout / 32][lane = lk * 32 + li][8]:
// element e of lane (lk, li) = w[jg * 32 + li][tap * Cin + chunk * 32 + (s * 2 + lk) * 8 + e].  thread = one 16-byte group
__global__ void __launch_bounds__(256) conv_halo_pack_kernel(const uint16_t* __restrict__ w, uint16_t* __restrict__ f,
                                                             int Cin, int Cout) {
  const int ngrp = Cout >> 5, n_chunks = Cin >> 5;
  const long total = (long)n_chunks * 27 * 2 * ngrp * 64;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  long r = gid >> 6;
  const int jg = (int)(r % ngrp);
  r /= ngrp;
  const int s = (int)(r & 1);
  r >>= 1;
  const int tap = (int)(r % 27);
  const int cc = (int)(r / 27);
  const int li = lane & 31, lk = lane >> 5;
  const uint16_t* src = w + (long)(jg * 32 + li) * (27L * Cin) + (long)tap * Cin + cc * 32 + (s * 2 + lk) * 8;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  *(u4*)(f + gid * 8) = *(const u4*)src;
}

extern "C" long occf_conv3x3x3_halo_pack_elems(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return 0;
  return 27L * Cin * Cout;
}

extern "C" int occf_conv3x3x3_halo_pack(const uint16_t* w_hi, const uint16_t* w_lo, uint16_t* f_hi, uint16_t* f_lo,
                                        int Cin, int Cout, void* stream) {
  if (occf_conv3x3x3_halo_pack_elems(Cin, Cout) == 0 || !w_hi || !f_hi) return OCCF_ESHAPE;
  const long groups = 27L * Cin * Cout / 8;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv_halo_pack_kernel, dim3(occf_cdiv(groups, 256)), dim3(256), 0, st, w_hi, f_hi, Cin, Cout);
  if (w_lo && f_lo)
    hipLaunchKernelGGL(conv_halo_pack_kernel, dim3(occf_cdiv(groups, 256)), dim3(256), 0, st, w_lo, f_lo, Cin, Cout);
  OCCF_LAUNCH_CHECK();
}

// returns OCCF_ESHAPE when the shape is outside this kernel's envelope (caller falls back to the
// generic implicit-GEMM kernel)
extern "C" int occf_conv3x3x3_halo_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo,
                                       const float* bias, const float* residual, float* out, int B, int X,
                                       int Y, int Z, int Cin, int Cout, long in_sb, long in_sx, long in_sy,
                                       long in_sz, int act, int terms, float* gn_partial, const uint16_t* wfrag_hi,
                                       const uint16_t* wfrag_lo, void* stream) {
  if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || Cin % 32 != 0 || Cout <= 0) return OCCF_ESHAPE;
  if (terms != 1 && terms != 3) return OCCF_EINVAL;
  if (terms == 3 && w_lo == nullptr) return OCCF_EINVAL;
  if (in_sb % 4 || in_sx % 4 || in_sy % 4 || in_sz % 4) return OCCF_ESHAPE;
  if ((long)X * in_sx >= 2147483647L || (long)Y * in_sy >= 2147483647L) return OCCF_ESHAPE;  // int halo offsets
  int TZ = Z >= 16 ? 16 : Z;
  if (TZ != 16 && TZ != 8 && TZ != 4) return OCCF_ESHAPE;
  if (Z % TZ != 0) return OCCF_ESHAPE;
  // N tile: 64*TN channels; pick the widest that does not waste more than a quarter
  int TN;
  if (Cout % 128 == 0) TN = 2;
  else if (Cout % 192 == 0) TN = 3;
  else if (Cout % 64 == 0) TN = 1;
  else return OCCF_ESHAPE;
  const bool frag = wfrag_hi && (terms == 1 || wfrag_lo);
  const bool small = frag && conv_halo_small() != 0;
  if (!frag && gn_partial && conv_halo_small() != 0) return OCCF_ESHAPE;
  const int TY = (small ? 64 : 128) / TZ;
  const size_t lds = conv_halo_lds(TY, TZ, TN, terms, frag);
  if (lds > 160 * 1024) return OCCF_ESHAPE;
  ConvHaloArgs a = {};
  a.x = x; a.Wh = w_hi; a.Wl = w_lo; a.bias = bias; a.residual = residual; a.out = out;
  a.B = B; a.X = X; a.Y = Y; a.Z = Z; a.Cin = Cin; a.Cout = Cout; a.TY = TY; a.TZ = TZ;
  a.sb = in_sb; a.sx = in_sx; a.sy = in_sy; a.sz = in_sz; a.act = act;
  a.gn_partial = gn_partial;
  if (wfrag_hi && (terms == 1 || wfrag_lo)) { a.Fh = wfrag_hi; a.Fl = wfrag_lo; }
  const long blocks = (long)B * ((X + 1) / 2) * ((Y + TY - 1) / TY) * (Z / TZ) * ((Cout + 64 * TN - 1) / (64 * TN));
  if (blocks >= 2147483647L) return OCCF_ESHAPE;
  const bool t3 = terms == 3, fr = a.Fh != nullptr;
  const int sch = fr ? (conv_halo_sched() ? 1 : 0) : 0;
  const conv_halo_fn_t fn = TN == 1 ? conv_halo_fn<1>(t3, fr, sch, small) : TN == 2 ? conv_halo_fn<2>(t3, fr, sch, small)
                                                                                   : conv_halo_fn<3>(t3, fr, sch, small);
#ifndef OCCF_EMU
  static bool attr_set[4][2][2][2][2] = {};
  if (!attr_set[TN][t3][fr][sch][small]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[TN][t3][fr][sch][small] = true;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(small ? 256 : 512), lds, st, a);
  return (int)hipGetLastError();
}

// number of spatial workgroup tiles per batch element (= rows of the GroupNorm partial buffer) the halo
// kernel uses for this volume, or -1 when it does not take the shape
extern "C" long occf_conv3x3x3_halo_gn_blocks(int X, int Y, int Z) {
  const int TZ = Z >= 16 ? 16 : Z;
  if ((TZ != 16 && TZ != 8 && TZ != 4) || Z % TZ != 0) return -1;
  const int TY = (conv_halo_small() != 0 ? 64 : 128) / TZ;
  return (long)((X + 1) / 2) * ((Y + TY - 1) / TY) * (Z / TZ);
}
