// 3x3x3 (stride 1, pad 1) convolution as Winograd F(2, 3) ALONG X over the LDS-resident halo tile of conv_halo.hip
// (dualpath_block.py:43-48, multiscale_deformattn_3d.py:101-110: the dominant FLOP item of the path).
//
// The direct kernel spends 27 taps x Cin multiply-adds per output voxel and sits at the rate the matrix pipe sustains
// (65 % busy at the power-limited clock, three bf16 products per product): the only lever left is FEWER products.
// For a pair of outputs (x0, x0 + 1) and the four input planes d0..d3 = x0 - 1 .. x0 + 2, per (y, z, channel):
//     V0 = d0 - d2     V1 = d1 + d2     V2 = d2 - d1     V3 = d1 - d3                      (input transform)
//     U0 = g0          U1 = (g0 + g1 + g2) / 2    U2 = (g0 - g1 + g2) / 2    U3 = g2     (filter transform over dx)
//     M_t = sum over (dy, dz, cin) of V_t(y + dy - 1, z + dz - 1, cin) * U_t(dy, dz, cin, cout)
//     Y(x0) = M0 + M1 + M2          Y(x0 + 1) = M1 - M2 - M3
// i.e. four 9-tap contractions per output PAIR instead of two 27-tap ones: 18 instead of 27 multiply-adds per output
// voxel (2/3 of the MFMAs), the same LDS footprint (the 4 halo x-planes become the 4 transformed planes in place) and
// the same fragment machinery.  Products stay 3-term bf16 splits of the TRANSFORMED fp32 values; F(2, 3) amplifies
// the rounding of a product by ~1.3x in the root-mean-square (tests: <= 1e-4 of the output maximum as the direct kernel).
//
// Workgroup = 512 threads = 8 waves as 4 (t = Winograd position) x 2 (N halves); tile = one x-pair x TY x TZ
// (TY * TZ = 64 positions = 128 output voxels) x 64 TN channels.  Wave (t, wn) accumulates M_t for the 64 positions and
// its 32 TN columns (2 x TN accumulator tiles, as the direct kernel).  The weight fragments U_t come straight from
// global memory (L2) in fragment order, one k-step ahead; the transformed halo of chunk c + 1 is built into the OTHER
// LDS buffer half-way through chunk c's taps (global loads issued at the start of the chunk), so there is ONE barrier
// per 32-channel chunk and no staging latency on the critical path.  The epilogue exchanges the four M_t through LDS
// (two rounds of 32 positions), applies the output transform, bias / activation / residual, and emits the GroupNorm
// partial sums of the outputs.
//
// F16 variant (two products per product): the transformed activations as ONE fp16 piece of V_t * 2^k (k from the
// tensor's absolute maximum, one bit of headroom for the transform's sums), the filters as fp16 (hi, lo).  Used for the
// DATA GRADIENTS (the forward convolution of dy on the tap-flipped weight): scripts/precision_probe.py measures no
// change of any gradient figure with dy rounded to 11 bits there (profiles/r06: whole gradient 4.9e-5 vs 5.0e-5), while
// the same cut in the forward convolutions moves the per-parameter figures to the edge of their bound.
//
// W1 variant of it (ONE product per product): the filters too as ONE fp16 piece (the hi half of the same packed
// fragments; the lo array is not read): half the MFMAs and half the weight-fragment loads of the F16 variant.
// precision_probe.py `cd1` (profiles/r06/r06x_precision_cd1.txt): whole gradient 5.0e-5 -> 5.2e-5, 90 % of the
// parameters 1.3e-4 -> 2.0e-4, worst 5.9e-4 -> 6.9e-4 against bounds of 1e-3 / 1e-3 / 6e-3.
#include <stdlib.h>
#include "occf_common.h"
#include "occf_absmax.h"
#include "../../include/occformer_hip.h"

struct ConvWinoArgs {
  const float* x;
  const uint16_t* Fh;         // U fragments: [chunk][t][tap9][k-step][Cout / 32][lane][8]
  const uint16_t* Fl;
  const float* bias;
  const float* residual;
  float* out;
  int B, X, Y, Z, Cin, Cout;
  int TY, TZ, tz_shift;       // TY * TZ == 64, TZ | Z, TZ = 1 << tz_shift
  long sb, sx, sy, sz;        // input element strides (channel stride 1)
  int act;
  float* gn_partial;          // optional [B][spatial tiles][Cout][2]
  const uint32_t* scale;      // F16: scale slot (occf_absmax_f32): [0] = bit pattern of max |x|
};

typedef uint32_t cw_u2 __attribute__((ext_vector_type(2)));

// operand row m of a 32-row MFMA tile holds tile position cw_pos(m): each 16-lane service group of a ds_read_b128 then
// reads 16 consecutive halo rows (conv_halo.hip: ch_pos)
__device__ __forceinline__ int cw_pos(int m) {
  const int run = m >> 2;
  const int start = run == 0 ? 0 : run == 1 ? 16 : run == 2 ? 20 : run == 3 ? 4 : run == 4 ? 24 : run == 5 ? 8 : run == 6 ? 12 : 28;
  return start + (m & 3);
}
__device__ __forceinline__ float cw_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int TN, bool F16, bool W1 = false>
__global__ void __launch_bounds__(512, 1) conv3x3x3_wino_kernel(ConvWinoArgs p) {
  static_assert(F16 || !W1, "the one-piece filter belongs to the fp16 variant");
  constexpr int BN = 64 * TN;
  constexpr int HROW = 80;                             // bytes per halo row: 32 bf16 + pad (conflict-free b128 reads)
  constexpr int NUT = 2;                               // staging units per thread (<= 1024 units: HY * HZ * 8)
  OCCF_DYN_SMEM(smem);
  const int TY = p.TY, TZ = p.TZ;
  const int HY = TY + 2, HZ = TZ + 2;
  const int NH = 4 * HY * HZ;                          // transformed halo rows
  const int hl_off = NH * HROW;                        // lo array behind the hi array (F16: no lo array)
  const int buf_sz = (F16 ? 1 : 2) * NH * HROW;        // one (hi, lo) buffer; two of them
  uint32_t inv_bits = 0x3F800000u;
  const float sc = F16 ? occf_u2f(occf_f16_scale_bits(p.scale[0], inv_bits, 1u)) : 1.0f;
  unsigned char* H = (unsigned char*)smem;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wt = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lk = lane >> 5;

  const int n_tiles = p.Cout / BN;
  const int zt = p.Z / TZ, yt = (p.Y + TY - 1) / TY, xt = (p.X + 1) / 2;
  unsigned wg = occf_xcd_remap(blockIdx.x, gridDim.x);
  const int nt = wg % n_tiles; wg /= n_tiles;
  const int tz0 = (wg % zt) * TZ; wg /= zt;
  const int ty0 = (wg % yt) * TY; wg /= yt;
  const int tx0 = (wg % xt) * 2;
  const int b = wg / xt;
  const int n0 = nt * BN;

  // halo row of this lane's two A rows (positions i * 32 + cw_pos(li)) in plane t, tap (0, 0)
  int hb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = i * 32 + cw_pos(li);
    hb[i] = (wt * HY + (r >> p.tz_shift)) * HZ + (r & (TZ - 1));
  }
  // ---- staging: unit u = (hy, hz, channel quad kq) of the (y, z) halo; a unit reads its float4 from the FOUR x planes
  // tx0 - 1 .. tx0 + 2 (clamped address, masked value) and writes the four transformed (hi, lo) pieces
  const float* xb = p.x + (long)b * p.sb;
  const int NU = HY * HZ * 8;
  int uofs[NUT], urow[NUT];
  unsigned uok = 0;
#pragma unroll
  for (int j = 0; j < NUT; ++j) {
    const int u = tid + j * 512;
    const int uc = u < NU ? u : NU - 1;
    const int h2 = uc >> 3, kq = uc & 7;
    const int hz = h2 % HZ, hy = h2 / HZ;
    const int y = ty0 + hy - 1, z = tz0 + hz - 1;
    const bool ok = u < NU && y >= 0 && y < p.Y && z >= 0 && z < p.Z;
    uofs[j] = (int)(occf_clampi(y, p.Y - 1) * p.sy + occf_clampi(z, p.Z - 1) * p.sz) + kq * 4;
    urow[j] = (hy * HZ + hz) * HROW + kq * 8;          // byte offset inside plane 0 of a hi array
    uok |= (ok ? 1u : 0u) << j;
  }
  int xofs[4];
  unsigned xok = 0;
#pragma unroll
  for (int hx = 0; hx < 4; ++hx) {
    const int x = tx0 + hx - 1;
    xofs[hx] = (int)(occf_clampi(x, p.X - 1) * p.sx);
    xok |= ((x >= 0 && x < p.X) ? 1u : 0u) << hx;
  }
  const int plane_b = HY * HZ * HROW;                  // bytes between two t planes
  // (one unit in flight at a time: 4 float4 -- both units at once spilled the 192-channel variant)
  float4 hreg[4];
  auto load_halo = [&](int c0, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int hx = 0; hx < 4; ++hx) hreg[hx] = *(const float4*)(xb + xofs[hx] + uofs[j] + c0);
  };
  auto store_halo = [&](int bufsel, int j) __attribute__((always_inline)) {
    unsigned char* Hb = H + bufsel * buf_sz;
    {
      if (tid + j * 512 < NU) {
        const bool oky = (uok >> j) & 1u;
        float d[4][4];
#pragma unroll
        for (int hx = 0; hx < 4; ++hx) {
          const bool ok = oky && ((xok >> hx) & 1u);
          d[hx][0] = ok ? hreg[hx].x : 0.f;
          d[hx][1] = ok ? hreg[hx].y : 0.f;
          d[hx][2] = ok ? hreg[hx].z : 0.f;
          d[hx][3] = ok ? hreg[hx].w : 0.f;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = t == 0 ? d[0][e] - d[2][e] : t == 1 ? d[1][e] + d[2][e] : t == 2 ? d[2][e] - d[1][e] : d[1][e] - d[3][e];
          const int off = t * plane_b + urow[j];
          if (F16) {
            const cw_u2 hi = {occf_f16_pack2(v[0] * sc, v[1] * sc), occf_f16_pack2(v[2] * sc, v[3] * sc)};
            *(cw_u2*)(Hb + off) = hi;
          } else {
            uint32_t h0, l0, h1, l1;
            occf_bf16_split2(v[0], v[1], h0, l0);
            occf_bf16_split2(v[2], v[3], h1, l1);
            const cw_u2 hi = {h0, h1}, lo = {l0, l1};
            *(cw_u2*)(Hb + off) = hi;
            *(cw_u2*)(Hb + hl_off + off) = lo;
          }
        }
      }
    }
  };

  f32x16 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int n_chunks = p.Cin / 32;
  const int ngrp = p.Cout >> 5;
#ifdef OCCF_EMU
  const int jg0 = (n0 + wn * (BN / 2)) >> 5;
  const int wts = wt;
#else
  const int jg0 = __builtin_amdgcn_readfirstlane((n0 + wn * (BN / 2)) >> 5);     // wave-uniform: scalar base address
  const int wts = __builtin_amdgcn_readfirstlane(wt);
#endif
  const unsigned lane8 = (unsigned)lane * 8u;
  // B fragments of (chunk cc, tap, k-step s) of this wave's t (clamped to the last chunk)
  auto load_f = [&](int cc, int tap, int s, bf16x8 (&fh)[TN], bf16x8 (&fl)[TN]) __attribute__((always_inline)) {
    const int c = cc < n_chunks ? cc : n_chunks - 1;
    const long o = ((long)(((c * 4 + wts) * 9 + tap) * 2 + s) * ngrp + jg0) * 512;     // scalar
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      fh[j] = *(const bf16x8*)(p.Fh + o + j * 512 + lane8);
      if (!W1) fl[j] = *(const bf16x8*)(p.Fl + o + j * 512 + lane8);
    }
  };
  const int a_base = hb[0] * HROW + lk * 16;
  const int a_i1 = (hb[1] - hb[0]) * HROW;             // wave-uniform
  auto load_a = [&](int bufsel, int toff, int s, bf16x8 (&ah)[2], bf16x8 (&al)[2]) __attribute__((always_inline)) {
    const unsigned char* ap = H + bufsel * buf_sz + a_base + toff * HROW + s * 32;
    ah[0] = *(const bf16x8*)(ap);
    ah[1] = *(const bf16x8*)(ap + a_i1);
    if (!F16) {
      al[0] = *(const bf16x8*)(ap + hl_off);
      al[1] = *(const bf16x8*)(ap + hl_off + a_i1);
    }
  };
  // term-major: six accumulators between two products into the same one
  auto mma_tm = [&](const bf16x8 (&ah)[2], const bf16x8 (&al)[2], const bf16x8 (&fh)[TN],
                    const bf16x8 (&fl)[TN]) __attribute__((always_inline)) {
    if (F16) {
      if (!W1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_f16_32x32x16(ah[i], fl[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_f16_32x32x16(ah[i], fh[j], acc[i][j]);
      return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(al[i], fh[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fl[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], fh[j], acc[i][j]);
  };
  auto tap_off = [&](int tap) __attribute__((always_inline)) -> int { return (tap / 3) * HZ + tap % 3; };

  // B (weight) fragments: a ring of BD k-steps -- BD - 1 k-steps ahead of their MFMAs.  One k-step ahead (the direct
  // kernel's pipeline) is 12 / 18 MFMAs = 384 / 576 cycles of this wave plus as many of its SIMD partner: ~0.4 us at
  // 128 channels.  BD = 3 (two k-steps ahead, 8 TN more registers; the ISA then waits vmcnt(7) / (5) with 8 loads in
  // flight) was measured at 128 channels: 1.223 vs 1.210 ms (r06n vs r06m) -- the distance of the weight fragments is
  // not what idles the matrix pipe either.  The three-deep form stays compiled behind this constant.
  constexpr int BD = 2;
  bf16x8 fh[BD][TN], fl[BD][TN];
  bf16x8 a0h[2], a0l[2], a1h[2], a1l[2];
  constexpr int NA = F16 ? 2 : 4;                      // ds_read_b128 per k-step
  constexpr int NF = (W1 ? 1 : 2) * TN;                // global 16-byte loads per k-step
  constexpr int NM = (W1 ? 2 : F16 ? 4 : 6) * TN;      // MFMAs per k-step
  int b_cc = 0, b_tap = 0, b_s = 0;                    // stream position of the NEXT fragment set to load
  auto load_next_b = [&](bf16x8 (&h)[TN], bf16x8 (&l)[TN]) __attribute__((always_inline)) {
    load_f(b_cc, b_tap, b_s, h, l);
    if (++b_s == 2) {
      b_s = 0;
      if (++b_tap == 9) { b_tap = 0; ++b_cc; }
    }
  };
#pragma unroll
  for (int i = 0; i < BD - 1; ++i) load_next_b(fh[i], fl[i]);
#pragma unroll
  for (int j = 0; j < NUT; ++j) {
    load_halo(0, j);
    store_halo(0, j);
  }
  // one k-step: the NEXT k-step's A fragments (LDS) and the fragment set BD - 1 k-steps ahead (global) issued at the head
  // of this k-step's MFMAs, one per MFMA (explicit issue pipeline: conv_halo.hip)
#define CW_STEP(ACH, ACL, ANH, ANL, NEXT_TOFF, NEXT_S, BCUR, BLOAD)                                   \
  do {                                                                                                \
    OCCF_SCHED_FENCE();                                                                               \
    load_a(bufsel, NEXT_TOFF, NEXT_S, ANH, ANL);                                                      \
    load_next_b(fh[BLOAD], fl[BLOAD]);                                                                \
    mma_tm(ACH, ACL, fh[BCUR], fl[BCUR]);                                                             \
    _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) { OCCF_SCHED_GROUP(0x008, 1); OCCF_SCHED_GROUP(0x100, 1); } \
    _Pragma("unroll") for (int i_ = 0; i_ < NF; ++i_) { OCCF_SCHED_GROUP(0x008, (NM >= NA + NF ? 1 : 0)); OCCF_SCHED_GROUP(0x020, 1); } \
    OCCF_SCHED_GROUP(0x008, (NM > NA + NF ? NM - NA - NF : 0));                                       \
    OCCF_SCHED_FENCE();                                                                               \
  } while (0)
  for (int cc = 0; cc < n_chunks; ++cc) {
    const int bufsel = cc & 1;
    __syncthreads();                                   // chunk cc's planes are complete; chunk cc - 1's taps are done
    const bool more = cc + 1 < n_chunks;
    if (more) load_halo((cc + 1) * 32, 0);             // lands under the first taps
    load_a(bufsel, 0, 0, a0h, a0l);
    if constexpr (BD == 3) {
      // three taps (six k-steps) per iteration: the ring slots are static
#pragma unroll 1
      for (int t0 = 0; t0 < 9; t0 += 3) {
        // the next chunk's planes go into the OTHER buffer (last read in chunk cc - 1: before the barrier), one staging
        // unit at taps 3 and 6, each three taps after its loads were issued
        if (t0 == 3 && more) {
          store_halo(bufsel ^ 1, 0);
          load_halo((cc + 1) * 32, 1);
        }
        if (t0 == 6 && more) store_halo(bufsel ^ 1, 1);
        const int o0 = tap_off(t0), o1 = tap_off(t0 + 1), o2 = tap_off(t0 + 2);
        const int o3 = tap_off(t0 + 3 < 9 ? t0 + 3 : 0);  // (after tap 8: a harmless read, replaced after the barrier)
        CW_STEP(a0h, a0l, a1h, a1l, o0, 1, 0, 2);
        CW_STEP(a1h, a1l, a0h, a0l, o1, 0, 1, 0);
        CW_STEP(a0h, a0l, a1h, a1l, o1, 1, 2, 1);
        CW_STEP(a1h, a1l, a0h, a0l, o2, 0, 0, 2);
        CW_STEP(a0h, a0l, a1h, a1l, o2, 1, 1, 0);
        CW_STEP(a1h, a1l, a0h, a0l, o3, 0, 2, 1);
      }
    } else {
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap) {
        if (tap == 3 && more) {
          store_halo(bufsel ^ 1, 0);
          load_halo((cc + 1) * 32, 1);
        }
        if (tap == 6 && more) store_halo(bufsel ^ 1, 1);
        const int toff = tap_off(tap);
        const int tnx = tap_off(tap < 8 ? tap + 1 : 0);  // (tap 8: a harmless read, replaced after the barrier)
        CW_STEP(a0h, a0l, a1h, a1l, toff, 1, 0, 1);
        CW_STEP(a1h, a1l, a0h, a0l, tnx, 0, 1, 0);
      }
    }
  }
#undef CW_STEP

  // ---- epilogue: the four M_t of a position meet in LDS, two rounds of 32 positions; wave (t, wn) then owns the
  // outputs of x-parity t & 1, MFMA rows 16 (t >> 1) .. + 16, its 32 TN columns
  float* E = (float*)smem;                             // [4 t][32 rows][BN]
  const int xp = wt & 1, rh = wt >> 1;
  const float unscale = occf_u2f(inv_bits);            // (F16: 2^-k, exact; else 1)
  float gs[TN], gq[TN], bj[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    gs[j] = gq[j] = 0.f;
    // (this lane's TN bias values once: a load per (row, column tile) inside the loops below is waited for on the spot)
    bj[j] = p.bias ? p.bias[n0 + wn * (BN / 2) + j * 32 + li] : 0.f;
  }
  __syncthreads();                                     // every wave is out of the tap loop (halo LDS is free)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lk;
        E[(wt * 32 + m) * BN + wn * (BN / 2) + j * 32 + li] = acc[i][j][r];
      }
    __syncthreads();
    const int x = tx0 + xp;
    // the residual operand of this round: 8 x TN loads in ONE batch (clamped addresses, unconditional inside the branch)
    float rv[8][TN];
    if (p.residual) {
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int pos = i * 32 + cw_pos(16 * rh + 2 * q + lk);
        const int y = ty0 + (pos >> p.tz_shift), z = tz0 + (pos & (TZ - 1));
        const long vrow = ((((long)b * p.X + occf_clampi(x, p.X - 1)) * p.Y + occf_clampi(y, p.Y - 1)) * p.Z + z) * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) rv[q][j] = p.residual[vrow + n0 + wn * (BN / 2) + j * 32 + li];
      }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int m = 16 * rh + 2 * q + lk;
      const int pos = i * 32 + cw_pos(m);
      const int y = ty0 + (pos >> p.tz_shift), z = tz0 + (pos & (TZ - 1));
      const bool v_ok = x < p.X && y < p.Y;
      const long vrow = ((((long)b * p.X + occf_clampi(x, p.X - 1)) * p.Y + occf_clampi(y, p.Y - 1)) * p.Z + z) * p.Cout;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * (BN / 2) + j * 32 + li;
        const int n = n0 + col;
        const float e0 = E[(0 * 32 + m) * BN + col], e1 = E[(1 * 32 + m) * BN + col], e2 = E[(2 * 32 + m) * BN + col],
                    e3 = E[(3 * 32 + m) * BN + col];
        float v = xp == 0 ? (e0 + e1) + e2 : (e1 - e2) - e3;
        if (F16) v *= unscale;
        v += bj[j];
        if (p.act == 1) v = fmaxf(v, 0.f);
        else if (p.act == 2) v = cw_gelu(v);
        if (v_ok) {
          if (p.residual) v += rv[q][j];
          p.out[vrow + n] = v;
          gs[j] += v;
          gq[j] = fmaf(v, v, gq[j]);
        }
      }
    }
    __syncthreads();
  }
  if (p.gn_partial) {
    // deterministic workgroup reduction: lanes (lk) -> LDS [t][channel] -> channel
    float* red = (float*)smem;                         // [4][BN][2]
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float a = gs[j] + __shfl_xor(gs[j], 32), q = gq[j] + __shfl_xor(gq[j], 32);
      if (lk == 0) {
        const int c = wn * (BN / 2) + j * 32 + li;
        red[(wt * BN + c) * 2 + 0] = a;
        red[(wt * BN + c) * 2 + 1] = q;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float a = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { a += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
      const long sp = ((long)(tx0 >> 1) * yt + ty0 / TY) * zt + tz0 / TZ;
      float* o = p.gn_partial + ((((long)b * xt * yt * zt) + sp) * p.Cout + n0 + tid) * 2;
      o[0] = a;
      o[1] = q;
    }
  }
}

// w fp32 [Cout][27 * Cin] (tap-major rows, tap = (dx * 3 + dy) * 3 + dz) -> U (hi, lo) in fragment order
// [chunk][t][tap9 = dy * 3 + dz][k-step][Cout / 32][lane = lk * 32 + li][8]: element e of lane (lk, li) =
// U_t[jg * 32 + li][tap9][chunk * 32 + (s * 2 + lk) * 8 + e].  thread = one 16-byte group of each array
template <bool F16>
__global__ void __launch_bounds__(256) conv_wino_pack_kernel(const float* __restrict__ w, uint16_t* __restrict__ fh,
                                                             uint16_t* __restrict__ fl, int Cin, int Cout) {
  const int ngrp = Cout >> 5, n_chunks = Cin >> 5;
  const long total = (long)n_chunks * 4 * 9 * 2 * ngrp * 64;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  long r = gid >> 6;
  const int jg = (int)(r % ngrp);
  r /= ngrp;
  const int s = (int)(r & 1);
  r >>= 1;
  const int tap9 = (int)(r % 9);
  r /= 9;
  const int t = (int)(r & 3);
  const int cc = (int)(r >> 2);
  const int li = lane & 31, lk = lane >> 5;
  const float* src = w + (long)(jg * 32 + li) * (27L * Cin) + cc * 32 + (s * 2 + lk) * 8;
  const float* s0 = src + (long)(0 * 9 + tap9) * Cin;
  const float* s1 = src + (long)(1 * 9 + tap9) * Cin;
  const float* s2 = src + (long)(2 * 9 + tap9) * Cin;
  float u[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g0 = s0[e], g1 = s1[e], g2 = s2[e];
    u[e] = t == 0 ? g0 : t == 1 ? 0.5f * ((g0 + g2) + g1) : t == 2 ? 0.5f * ((g0 + g2) - g1) : g2;
  }
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (F16) occf_f16_split2(u[2 * q], u[2 * q + 1], hh[q], ll[q]);
    else occf_bf16_split2(u[2 * q], u[2 * q + 1], hh[q], ll[q]);
  }
  u4 h, l;
  h.x = hh[0]; h.y = hh[1]; h.z = hh[2]; h.w = hh[3];
  l.x = ll[0]; l.y = ll[1]; l.z = ll[2]; l.w = ll[3];
  *(u4*)(fh + gid * 8) = h;
  *(u4*)(fl + gid * 8) = l;
}

// OCCF_WINO: 1 (default) = stride-1 3^3 convolutions inside the envelope run as F(2, 3) along x; 0 = the direct kernel
static int conv_wino_enabled() {
  const char* e = getenv("OCCF_WINO");
  return e ? atoi(e) : 1;
}

static int conv_wino_tn(int Cout) {
  if (Cout % 128 == 0) return 2;
  if (Cout % 192 == 0) return 3;
  if (Cout % 64 == 0) return 1;
  return 0;
}

/* bf16 elements of one fragment array (hi or lo), or 0 when the shape is outside the kernel's envelope */
extern "C" long occf_conv3x3x3_wino_pack_elems(int Cin, int Cout) {
  if (Cin <= 0 || Cout <= 0 || Cin % 32 || conv_wino_tn(Cout) == 0 || !conv_wino_enabled()) return 0;
  return 36L * Cin * Cout;
}

extern "C" int occf_conv3x3x3_wino_pack(const float* w_tapmajor, uint16_t* f_hi, uint16_t* f_lo, int Cin, int Cout,
                                        int f16, void* stream) {
  if (occf_conv3x3x3_wino_pack_elems(Cin, Cout) == 0 || !w_tapmajor || !f_hi || !f_lo) return OCCF_ESHAPE;
  const long groups = 36L * Cin * Cout / 8;
  if (f16)
    hipLaunchKernelGGL(conv_wino_pack_kernel<true>, dim3(occf_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_tapmajor, f_hi, f_lo, Cin, Cout);
  else
    hipLaunchKernelGGL(conv_wino_pack_kernel<false>, dim3(occf_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_tapmajor, f_hi, f_lo, Cin, Cout);
  OCCF_LAUNCH_CHECK();
}

/* max |x| of x[rows][cols] (row stride ld, cols % 4 == 0, 16-byte aligned rows) into a scale slot of
 * occf_absmax_slot_words() uint32 words: slot[0] = the maximum's fp32 bit pattern.  Two launches, no atomics. */
extern "C" long occf_absmax_slot_words() { return OCCF_ABSMAX_SLOT; }
extern "C" int occf_absmax_f32(const float* x, long rows, int cols, long ld, uint32_t* slot, void* stream) {
  if (!x || !slot || rows <= 0 || cols <= 0 || cols % 4 || ld % 4) return OCCF_ESHAPE;
  wg_absmax(x, rows, cols, ld, slot, (hipStream_t)stream);
  OCCF_LAUNCH_CHECK();
}

extern "C" int occf_absmax_flat(const float* x, long n, uint32_t* slot, void* stream) {
  if (!x || !slot || n <= 0) return OCCF_ESHAPE;
  wg_absmax_flat(x, n, slot, (hipStream_t)stream);
  OCCF_LAUNCH_CHECK();
}

static int conv_wino_tz(int Z) {
  const int TZ = Z >= 16 ? 16 : Z;
  if ((TZ != 16 && TZ != 8 && TZ != 4) || Z % TZ != 0) return 0;
  return TZ;
}

/* spatial workgroup tiles per batch element (= rows of the GroupNorm partial buffer), or -1 */
extern "C" long occf_conv3x3x3_wino_gn_blocks(int X, int Y, int Z) {
  const int TZ = conv_wino_tz(Z);
  if (!TZ || !conv_wino_enabled()) return -1;
  const int TY = 64 / TZ;
  return (long)((X + 1) / 2) * ((Y + TY - 1) / TY) * (Z / TZ);
}

/* returns OCCF_ESHAPE when the shape is outside this kernel's envelope (the caller runs the direct kernel) */
extern "C" int occf_conv3x3x3_wino_fwd(const float* x, const uint16_t* wfrag_hi, const uint16_t* wfrag_lo,
                                       const float* bias, const float* residual, float* out, int B, int X, int Y,
                                       int Z, int Cin, int Cout, long in_sb, long in_sx, long in_sy, long in_sz,
                                       int act, float* gn_partial, const uint32_t* f16_scale, void* stream) {
  if (B <= 0 || X <= 0 || Y <= 0 || Z <= 0 || Cin <= 0 || Cin % 32 != 0 || Cout <= 0) return OCCF_ESHAPE;
  // (f16_scale given and wfrag_lo NULL: the one-product W1 variant)
  if (!conv_wino_enabled() || !wfrag_hi || (!wfrag_lo && !f16_scale)) return OCCF_ESHAPE;
  if (in_sb % 4 || in_sx % 4 || in_sy % 4 || in_sz % 4) return OCCF_ESHAPE;
  if ((long)X * in_sx >= 2147483647L || (long)Y * in_sy >= 2147483647L) return OCCF_ESHAPE;  // int halo offsets
  const int TZ = conv_wino_tz(Z);
  const int TN = conv_wino_tn(Cout);
  if (!TZ || !TN) return OCCF_ESHAPE;
  const int TY = 64 / TZ;
  const size_t NH = 4 * (size_t)(TY + 2) * (TZ + 2);
  size_t lds = 2 * NH * 80 * 2;                                          // two (hi, lo) buffers of 80-byte rows
  const size_t ex = (size_t)4 * 32 * 64 * TN * 4;                        // the epilogue's exchange buffer
  if (ex > lds) lds = ex;
  if (lds > 160 * 1024 || (TY + 2) * (TZ + 2) * 8 > 1024) return OCCF_ESHAPE;
  ConvWinoArgs a = {};
  a.x = x; a.Fh = wfrag_hi; a.Fl = wfrag_lo; a.bias = bias; a.residual = residual; a.out = out;
  a.B = B; a.X = X; a.Y = Y; a.Z = Z; a.Cin = Cin; a.Cout = Cout; a.TY = TY; a.TZ = TZ;
  a.tz_shift = TZ == 16 ? 4 : TZ == 8 ? 3 : 2;
  a.sb = in_sb; a.sx = in_sx; a.sy = in_sy; a.sz = in_sz; a.act = act;
  a.gn_partial = gn_partial;
  a.scale = f16_scale;
  const long blocks = (long)B * ((X + 1) / 2) * ((Y + TY - 1) / TY) * (Z / TZ) * (Cout / (64 * TN));
  if (blocks >= 2147483647L) return OCCF_ESHAPE;
  typedef void (*fn_t)(ConvWinoArgs);
  const int f16 = f16_scale == nullptr ? 0 : wfrag_lo ? 1 : 2;
  const fn_t fn = f16 == 2 ? (TN == 1 ? conv3x3x3_wino_kernel<1, true, true> : TN == 2 ? conv3x3x3_wino_kernel<2, true, true>
                                                                                       : conv3x3x3_wino_kernel<3, true, true>)
                  : f16 ? (TN == 1 ? conv3x3x3_wino_kernel<1, true> : TN == 2 ? conv3x3x3_wino_kernel<2, true>
                                                                              : conv3x3x3_wino_kernel<3, true>)
                        : (TN == 1 ? conv3x3x3_wino_kernel<1, false> : TN == 2 ? conv3x3x3_wino_kernel<2, false>
                                                                               : conv3x3x3_wino_kernel<3, false>);
#ifndef OCCF_EMU
  static bool attr_set[4][3] = {};
  if (!attr_set[TN][f16]) {
    hipError_t e = hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr_set[TN][f16] = true;
  }
#endif
  hipLaunchKernelGGL(fn, dim3((unsigned)blocks), dim3(512), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
