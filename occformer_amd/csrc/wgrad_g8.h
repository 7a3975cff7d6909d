// Weight gradient of the stride-1 3^3 convolutions on "G8" operands (included by wgrad.hip; same contraction, same
// output layout: dW[n, tap*Cin + c] = sum_m dY[m, n] * X_tap[m, c]).
//
// v_mfma_f32_32x32x16_bf16 wants 8 consecutive k (= voxels m) per lane of BOTH operands, while both activations are
// stored voxel-major with the channel fastest.  wgrad_kernel transposes 8 x 8 blocks in registers on the way into LDS,
// once per (tap, tile) -- 27 x tiles times per element: PMC on the 192 -> 192 convolution counted 9.1 VALU instructions
// per MFMA and the matrix pipe 29 % busy (profiles/r03b_wgrad_192_pmc.txt).  Here the transposition happens ONCE, in the
// bf16 pre-split pass that exists anyway: it writes (hi, lo) in G8 order
//     [m / 8][channel][8 consecutive m]          (16 bytes per (group, channel))
// which is exactly the LDS image a fragment read wants, so the main loop stages with LDS-DMA
// (buffer_load_dwordx4 ... lds: 1 KiB per wave instruction, no VGPRs, no VALU) and is left with fragment reads + MFMAs.
// A group is 8 consecutive z of one (b, x, y) column (Zo % 8 == 0).  A tap's x / y shift moves whole groups (an address
// offset; out-of-range columns read zeros through the buffer bounds); its z shift does not, so the pre-split pass
// writes THREE copies of x, shifted by dz = -1, 0, +1 along z with zeros shifted in (2 x 3 arrays of nx bf16).
//
// Workgroup = 4 waves (2 x 2), tile (64 TI) x (64 TC) of one tap over one M-slab, 16 KS rows per stage, two stages
// in LDS: the DMA of stage s + 1 is in flight while stage s feeds the matrix cores; one barrier per stage.  Wave w
// issues the DMA of ONE of the four arrays (dY hi, dY lo, x hi, x lo): every piece is then described by
// wave-uniform (scalar) values -- the column decode of the x pieces runs on the scalar unit.
#pragma once

struct Wg8Args {
  const uint16_t* Yh;
  const uint16_t* Yl;      // G8 [M / 8][N][8]
  const uint16_t* Xh;
  const uint16_t* Xl;      // G8 [kZ copies][B * Xi * Yi * ZG][Cin][8]
  float* out;              // [S][N][taps * Cin]
  long xcopy_elems;        // bf16 elements of ONE z-shifted copy
  int N, Cin, taps;
  uint32_t ybytes, xbytes; // bytes of one array (one copy), < 2 GiB
  int B, Xo, Yo, Xi, Yi, ZG, zg_shift, kX, kY, kZ, pad_x, pad_y;
  // M-slab = a RECTANGLE of output columns: the y-strip [strip * strip_w, + strip_w) of the planes
  // [seg * seg_planes, + seg_planes) (plane = b * Xo + x), walked plane by plane.  The taps with dx = +1, 0, -1 read the
  // same x plane of the strip strip_w columns apart, so two planes of one strip stay in the XCD's L2 between the three
  // uses; with slabs of whole planes (strip_w = Yo) the reuse distance is a whole plane of all six x arrays (24 MB at
  // the 200-grid) and every tap row streams from HBM: 11.9 GB fetched for 2.0 GB of operands (PMC r03i)
  int n_strips, strip_w, seg_planes, planes;
  int n_slabs, slabs_per_xcd, tiles;   // tiles = taps * tn_count * tc_count workgroups per slab in THIS launch
  int n_base, c_base, tn_count, tc_count;
  // two-product fp16 mode (kernel template F16): Yh holds ONE fp16 piece of dY * 2^k, Xh / Xl the fp16 (hi, lo) halves
  // of x; scale[1] = the bit pattern of 2^-k, applied to the partial sums in the epilogue.
  // one-product mode (F16 && X1): x too as ONE fp16 piece (Xh = x rounded to nearest even; no Xl): half the MFMAs and
  // two thirds of the staged bytes.  scripts/precision_probe.py `wg1c` (profiles/r06/r06x_precision_wg1c.txt): whole
  // gradient 4.5e-5 -> 5.0e-5, worst parameter 4.4e-4 -> 5.9e-4 against bounds of 1e-3 / 6e-3
  const uint32_t* scale;
};

#ifdef OCCF_EMU
template <int N>
static inline void wg8_wait_vm() {}
#else
template <int N>
__device__ __forceinline__ void wg8_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
#endif

template <int TI, int TC, int KS, int ST, bool F16 = false, bool X1 = false>
__global__ void __launch_bounds__(256, 2) wgrad_g8_kernel(Wg8Args p) {
  static_assert(F16 || !X1, "the one-piece x operand belongs to the fp16 mode");
  OCCF_DYN_SMEM(smem_raw);
  constexpr int TN = 64 * TI, BC = 64 * TC, G = 2 * KS;      // G = 8-row groups per stage
  constexpr int AS = G * TN, BS = G * BC;                      // 16-byte slots of one array of one stage
  constexpr int NA = F16 ? 1 : 2;                              // dY arrays (F16: one fp16 piece)
  constexpr int NB = X1 ? 1 : 2;                               // x arrays
  constexpr int STAGE = NA * AS + NB * BS;
  wg_u4* lds = (wg_u4*)smem_raw;                               // [ST stages][Ah | Al | Bh | Bl]  (F16: [A | Bh | Bl])

  // all (tap, tile) workgroups of an M-slab on ONE XCD (workgroup w runs on XCD w % 8): they walk the slab in step and
  // share its columns in that L2; an XCD owns a contiguous block of slabs, i.e. neighbouring strips (which share their
  // edge columns) run side by side
  const int xcd = (int)(blockIdx.x & 7u);
  int k = (int)(blockIdx.x >> 3);
  const int slab = xcd * p.slabs_per_xcd + k / p.tiles;
  if (k >= p.slabs_per_xcd * p.tiles || slab >= p.n_slabs) return;
  k %= p.tiles;
  const int ct = k % p.tc_count;
  k /= p.tc_count;
  const int tap = k % p.taps, nt = k / p.taps;
  const int n0 = p.n_base + nt * TN, c0 = p.c_base + ct * BC;
  const int tdz = tap % p.kZ, tdy = (tap / p.kZ) % p.kY, tdx = tap / (p.kZ * p.kY);

  const int tid = threadIdx.x, lane = tid & 63, wave = occf_wave_uniform(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int strip = slab % p.n_strips, seg = slab / p.n_strips;
  const int y0 = strip * p.strip_w, px0 = seg * p.seg_planes;
  const int wy = p.Yo - y0 < p.strip_w ? p.Yo - y0 : p.strip_w;
  const int npx = p.planes - px0 < p.seg_planes ? p.planes - px0 : p.seg_planes;
  const long ngroups = npx > 0 && wy > 0 ? ((long)npx * wy) << p.zg_shift : 0;
  const int nsteps = (int)((ngroups + G - 1) / G);

  // ---- DMA role of this wave: 0 = dY hi, 1 = dY lo, 2 = x hi, 3 = x lo  (F16: waves 0 and 1 share the one dY array,
  // wave 0 the first half of a stage's row groups, wave 1 the second)
  const bool is_b = wave >= 2, is_lo = (wave & 1) != 0;
  const occf_bbuf buf = is_b ? occf_make_bbuf(((is_lo && !X1) ? p.Xl : p.Xh) + (long)tdz * p.xcopy_elems, p.xbytes)
                             : occf_make_bbuf((is_lo && !F16) ? p.Yl : p.Yh, p.ybytes);
  const int arr_base = wave == 0 ? 0 : wave == 1 ? (F16 ? 0 : AS) : (wave == 2 || X1) ? NA * AS : NA * AS + BS;
  constexpr int GA = F16 ? G / 2 : G;                          // dY row groups one A wave issues per stage
  const int ga0 = F16 && wave == 1 ? G / 2 : 0;
  constexpr int GB = X1 ? G / 2 : G;                           // x row groups one B wave issues per stage (X1: the two
  const int gb0 = X1 && wave == 3 ? G / 2 : 0;                 // B waves share the one x array like the A waves)
  const uint32_t lane_off = (uint32_t)((is_b ? c0 : n0) + lane) * 16u;
  // position of the first group of the current stage inside the slab: z-group zg0, strip column cy, plane cpl (relative
  // to px0), and the (batch, x) of that plane
  int zg0 = 0, cy = 0, cpl = 0, cb = px0 / p.Xo, cx = px0 % p.Xo;
  auto issue = [&](int bufsel) __attribute__((always_inline)) {
    wg_u4* dst = lds + bufsel * STAGE + arr_base;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (F16 && !is_b && (g < ga0 || g >= ga0 + GA)) continue;
      if (X1 && is_b && (g < gb0 || g >= gb0 + GB)) continue;
      const int zz = zg0 + g;
      const int zg = zz & (p.ZG - 1);
      int y = cy + (zz >> p.zg_shift), pl = cpl, x = cx, b = cb;
      while (y >= wy) {
        y -= wy;
        ++pl;
        if (++x == p.Xo) { x = 0; ++b; }
      }
      const bool in_slab = pl < npx;                          // (the last stage may run past the slab: zero rows)
      uint32_t base;
      if (!is_b) {
        base = in_slab ? ((((uint32_t)(px0 + pl) * (uint32_t)p.Yo + (uint32_t)(y0 + y)) << p.zg_shift) + (uint32_t)zg) *
                             (uint32_t)p.N * 16u
                       : OCCF_BUF_OOB;
      } else {
        const int xs = x + tdx - p.pad_x, ys = y0 + y + tdy - p.pad_y;
        const bool ok = in_slab && (unsigned)xs < (unsigned)p.Xi && (unsigned)ys < (unsigned)p.Yi;
        // (32-bit: every array is < 2 GiB)
        base = ok ? ((((uint32_t)(b * p.Xi + xs) * (uint32_t)p.Yi + (uint32_t)ys) << p.zg_shift) + (uint32_t)zg) *
                        (uint32_t)p.Cin * 16u
                  : OCCF_BUF_OOB;
      }
      if (!is_b) {
#pragma unroll
        for (int q = 0; q < TI; ++q)
          occf_bbuf_load_lds_b128(buf, base + lane_off + (uint32_t)q * 1024u, dst + g * TN + q * 64);
      } else {
#pragma unroll
        for (int q = 0; q < TC; ++q)
          occf_bbuf_load_lds_b128(buf, base + lane_off + (uint32_t)q * 1024u, dst + g * BC + q * 64);
      }
    }
    zg0 += G;
    cy += zg0 >> p.zg_shift;
    zg0 &= p.ZG - 1;
    while (cy >= wy) {
      cy -= wy;
      ++cpl;
      if (++cx == p.Xo) { cx = 0; ++cb; }
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
  auto compute = [&](int bufsel) __attribute__((always_inline)) {
    const wg_u4* Ah = lds + bufsel * STAGE;
    const wg_u4* Al = Ah + AS;                                // (F16: unused)
    const wg_u4* Bh = Ah + NA * AS;
    const wg_u4* Bl = Bh + BS;                                // (X1: unused)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 ah[TI], al[TI], bh[TC], bl[TC];
      const int row = ks * 2 + lk;
#pragma unroll
      for (int i = 0; i < TI; ++i) {
        ah[i] = __builtin_bit_cast(bf16x8, Ah[row * TN + wm * (TN / 2) + i * 32 + li]);
        if (!F16) al[i] = __builtin_bit_cast(bf16x8, Al[row * TN + wm * (TN / 2) + i * 32 + li]);
      }
#pragma unroll
      for (int j = 0; j < TC; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, Bh[row * BC + wn * (BC / 2) + j * 32 + li]);
        if (!X1) bl[j] = __builtin_bit_cast(bf16x8, Bl[row * BC + wn * (BC / 2) + j * 32 + li]);
      }
      // term-major: consecutive MFMAs write different accumulators (no dependent-accumulator stalls)
      if (X1) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_f16_32x32x16(ah[i], bh[j], acc[i][j]);
      } else if (F16) {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_f16_32x32x16(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_f16_32x32x16(ah[i], bh[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(al[i], bh[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bl[j], acc[i][j]);
#pragma unroll
        for (int i = 0; i < TI; ++i)
#pragma unroll
          for (int j = 0; j < TC; ++j) acc[i][j] = occf_mfma_bf16_32x32x16(ah[i], bh[j], acc[i][j]);
      }
    }
  };

  // ST stages in LDS: while stage s feeds the matrix cores the DMA of stages s + 1 .. s + ST - 2 is in flight (the
  // L2 -> LDS latency under load is longer than one stage of MFMAs); ONE barrier per stage: the buffer refilled after
  // barrier s is the one compute(s - 1) read
#pragma unroll
  for (int t = 0; t < ST - 1; ++t)
    if (t < nsteps) issue(t);
  int cur = 0, nxt = ST - 1;
  for (int s = 0; s < nsteps; ++s) {
    // stage s has landed: this wave's pieces by its vmcnt (younger stages may stay in flight), everybody's by the barrier
    const int younger = nsteps - 1 - s;
    if (ST >= 3 && younger >= ST - 2) {
      if (is_b) wg8_wait_vm<GB * TC * (ST - 2)>(); else wg8_wait_vm<GA * TI * (ST - 2)>();
    } else if (ST >= 4 && younger == 1) {
      if (is_b) wg8_wait_vm<GB * TC>(); else wg8_wait_vm<GA * TI>();
    } else {
      wg8_wait_vm<0>();
    }
    __syncthreads();
    if (s + ST - 1 < nsteps) issue(nxt);
    compute(cur);
    cur = cur + 1 == ST ? 0 : cur + 1;
    nxt = nxt + 1 == ST ? 0 : nxt + 1;
  }

  // ---- epilogue: raw partial sums of this M-slab (tiles cover N and Cin exactly: no masks)
  const int Kt = p.taps * p.Cin;
  float* o = p.out + (long)slab * p.N * Kt;
  const float unscale = F16 ? occf_u2f(p.scale[1]) : 1.0f;     // (a power of two: exact)
#pragma unroll
  for (int i = 0; i < TI; ++i)
#pragma unroll
    for (int j = 0; j < TC; ++j) {
      const int c = c0 + wn * (BC / 2) + j * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * (TN / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        o[(long)n * Kt + (long)tap * p.Cin + c] = F16 ? acc[i][j][r] * unscale : acc[i][j][r];
      }
    }
}

// dY fp32 [M][ldy] -> G8 (hi, lo): a thread owns one 8-row group of ONE channel -- a wave reads 256 contiguous bytes
// per row and writes 1 KiB contiguous per array (a thread owning four channels wrote 16 of every 64 bytes per store
// instruction: 0.61 ms for the three x copies of the 192-channel grid)
template <bool F16>
__global__ void __launch_bounds__(256) wg8_split_y_kernel(const float* __restrict__ dy, long ldy, long groups, int N,
                                                          wg_u4* __restrict__ yh, wg_u4* __restrict__ yl,
                                                          uint32_t* __restrict__ scale) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= groups * N) return;
  const unsigned i32 = (unsigned)i;                         // (< 2^31 elements per array: 32-bit divisions)
  const int c = (int)(i32 % (unsigned)N);
  const long g = (long)(i32 / (unsigned)N);
  const float* src = dy + g * 8 * ldy + c;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = src[j * ldy];
  if (F16) {
    // scale[0] = bit pattern of max |dy| (wg_absmax_kernel, same stream) -> ONE fp16 piece of dy * 2^k; thread 0
    // leaves 2^-k in scale[1] for the contraction's epilogue
    uint32_t inv;
    const float sc = occf_u2f(occf_f16_scale_bits(scale[0], inv));
    if (i == 0) scale[1] = inv;
    wg_u4 h;
    h.x = occf_f16_pack2(v[0] * sc, v[1] * sc); h.y = occf_f16_pack2(v[2] * sc, v[3] * sc);
    h.z = occf_f16_pack2(v[4] * sc, v[5] * sc); h.w = occf_f16_pack2(v[6] * sc, v[7] * sc);
    yh[i] = h;
    return;
  }
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) occf_bf16_split2(v[2 * t], v[2 * t + 1], hh[t], ll[t]);
  wg_u4 h, l;
  h.x = hh[0]; h.y = hh[1]; h.z = hh[2]; h.w = hh[3];
  l.x = ll[0]; l.y = ll[1]; l.z = ll[2]; l.w = ll[3];
  yh[i] = h;
  yl[i] = l;
}

// x fp32 [cols][Z][C] (dense) -> three z-shifted G8 (hi, lo) copies: copy w holds x[z + w - 1] (zeros outside [0, Z))
template <bool F16, bool X1 = false>
__global__ void __launch_bounds__(256) wg8_split_x_kernel(const float* __restrict__ x, long cols, int ZG, int C,
                                                          wg_u4* __restrict__ xh, wg_u4* __restrict__ xl,
                                                          long copy_slots) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cols * ZG * C) return;
  const unsigned i32 = (unsigned)i;
  const int c = (int)(i32 % (unsigned)C);
  const unsigned gi32 = i32 / (unsigned)C;
  const long gi = (long)gi32;
  const int zg = (int)(gi32 % (unsigned)ZG);
  const long col = (long)(gi32 / (unsigned)ZG);
  const int Z = ZG * 8;
  const float* src = x + col * Z * C + c;
  float rows[10];
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const int z = zg * 8 - 1 + r;
    rows[r] = (z >= 0 && z < Z) ? src[(long)z * C] : 0.f;
  }
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    uint32_t hh[4], ll[4];
    if (X1) {                                     // x as ONE fp16 piece (round to nearest even): no lo copy
      wg_u4 h;
      h.x = occf_f16_pack2(rows[w], rows[w + 1]); h.y = occf_f16_pack2(rows[w + 2], rows[w + 3]);
      h.z = occf_f16_pack2(rows[w + 4], rows[w + 5]); h.w = occf_f16_pack2(rows[w + 6], rows[w + 7]);
      xh[w * copy_slots + i] = h;
      continue;
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (F16) occf_f16_split2(rows[w + 2 * t], rows[w + 2 * t + 1], hh[t], ll[t]);
      else occf_bf16_split2(rows[w + 2 * t], rows[w + 2 * t + 1], hh[t], ll[t]);
    }
    wg_u4 h, l;
    h.x = hh[0]; h.y = hh[1]; h.z = hh[2]; h.w = hh[3];
    l.x = ll[0]; l.y = ll[1]; l.z = ll[2]; l.w = ll[3];
    xh[w * copy_slots + i] = h;
    xl[w * copy_slots + i] = l;
  }
}

// ---- host side
struct Wg8Seg { int w, count, base; };
// a channel dimension (multiple of 64) as tiles of 192 / 128 / 64: at most two widths
static int wg8_segments(int n, Wg8Seg seg[2]) {
  if (n % 192 == 0) { seg[0] = Wg8Seg{192, n / 192, 0}; return 1; }
  if (n % 128 == 0) { seg[0] = Wg8Seg{128, n / 128, 0}; return 1; }
  if (n == 64) { seg[0] = Wg8Seg{64, 1, 0}; return 1; }
  seg[0] = Wg8Seg{128, (n - 192) / 128, 0};               // n % 128 == 64, n >= 192: ... + one 192-wide tile
  seg[1] = Wg8Seg{192, 1, n - 192};
  return 2;
}
static long wg8_tiles_per_slab(int N, int Cin, int taps) {
  Wg8Seg sn[2], sc[2];
  const int nn = wg8_segments(N, sn), nc = wg8_segments(Cin, sc);
  long t = 0;
  for (int a = 0; a < nn; ++a)
    for (int b = 0; b < nc; ++b) t += (long)sn[a].count * sc[b].count;
  return t * taps;
}
static bool wg8_eligible(int Zi, int Cin, int Cout, int kX, int kY, int kZ, int stride, int dil, int pad_z, long M,
                         long nx, long ny) {
  static const int env = [] {
    const char* e = getenv("OCCF_WG_G8");               // 0: the register-transposing kernel for every shape
    return e ? atoi(e) : 1;
  }();
  (void)kX; (void)kY;
  return env && stride == 1 && dil == 1 && kZ == 3 && pad_z == 1 && (Zi == 8 || Zi == 16 || Zi == 32 || Zi == 64) &&
         Cin % 64 == 0 && Cout % 64 == 0 && M >= 1024 && M < 2147483647L && nx * 2 < 2147483647L &&
         ny * 2 < 2147483647L;
}
struct Wg8Geom { int strip_w, n_strips, seg_planes, n_segs, n_slabs; };
// slab rectangles (see Wg8Args).  Measured on the 192 -> 192 convolution at the 200-grid (r03j): strips of 25 columns x 7
// plane segments (56 slabs, 2.95 rounds of the 512 resident workgroup slots) 3.94 ms per call, 8 / 10 / 13 columns
// (60 - 80 slabs) 4.10 -- 4.13, whole planes in 56 slabs 3.96, 16 slabs (0.84 rounds) 4.27, 378 workgroups 4.86: the
// kernel is bound by the matrix pipe at the clock the power limit allows (1.6 GHz at 68 % MFMA-busy), not by HBM
// (FETCH_SIZE 11.9 GB with whole planes, 8.0 GB with strips -- the same time).  So: strips = the grid's y extent
// divided by 1 / 2 / 4 / 8, plane segments so that the workgroups fill whole rounds, few slabs (each writes
// N x taps*Cin partial sums).
static Wg8Geom wg8_geometry(int planes, int Yo, int ZG, int N, int Cin, int taps) {
  static const int w_env = [] { const char* e = getenv("OCCF_WG8_W"); return e ? atoi(e) : 0; }();      // diagnostics
  static const int segs_env = [] { const char* e = getenv("OCCF_WG8_SEGS"); return e ? atoi(e) : 0; }();
  const long tiles = wg8_tiles_per_slab(N, Cin, taps);
  const double M = (double)planes * Yo * ZG * 8.0;
  Wg8Geom best = {Yo, 1, planes, 1, 1};
  double best_score = -1.0;
  for (int d = 8; d >= 1; d >>= 1) {
    const int w = w_env ? w_env : (Yo + d - 1) / d;
    if (w < 4 && d > 1) continue;
    const int strips = (Yo + w - 1) / w;
    const double fill = (double)Yo / ((double)strips * w);
    for (int segs = segs_env ? segs_env : 1; segs <= (segs_env ? segs_env : 64) && segs <= planes; ++segs) {
      const int sp = (planes + segs - 1) / segs;
      const int nseg = (planes + sp - 1) / sp;
      if ((long)sp * w * ZG * 8 < 1024 && segs > 1) break;               // >= 1024 rows per slab
      const long n = (long)strips * nseg, total = n * tiles;
      const double eff = (double)total / (double)(((total + 511) / 512) * 512) * fill;
      const double score = eff / (1.0 + 400.0 * (double)n / M) * (total >= 1024 ? 1.0 : (double)total / 1024.0);
      if (score > best_score + 1e-9) {
        best_score = score;
        best = Wg8Geom{w, strips, sp, nseg, (int)n};
      }
      if (total > 8192) break;
    }
    if (w_env) break;
  }
  return best;
}
static long wg8_workspace(int planes, int Yo, int ZG, int N, int Cin, int taps, long nx, long ny) {
  const Wg8Geom gm = wg8_geometry(planes, Yo, ZG, N, Cin, taps);
  return (long)gm.n_slabs * N * (long)taps * Cin + ny + 3 * nx;
}

template <int TI, int TC, int KS, int ST, bool F16, bool X1 = false>
static void wg8_launch_one(const Wg8Args& a, hipStream_t st) {
  const unsigned grid = (unsigned)(8 * a.slabs_per_xcd * a.tiles);
  const size_t smem = (size_t)ST * (2 * KS) * ((F16 ? 1 : 2) * 64 * TI + (X1 ? 1 : 2) * 64 * TC) * 16;
#ifndef OCCF_EMU
  static bool attr_set = false;
  if (smem > 65536 && !attr_set) {
    (void)hipFuncSetAttribute((const void*)wgrad_g8_kernel<TI, TC, KS, ST, F16, X1>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
#endif
  hipLaunchKernelGGL((wgrad_g8_kernel<TI, TC, KS, ST, F16, X1>), dim3(grid), dim3(256), smem, st, a);
}
// (rows per stage = 16 KS, stages ST): two workgroups per CU need <= 80 KB each
template <bool F16>
static void wg8_launch_class(const Wg8Args& a, int ti, int tc, hipStream_t st) {
  static const int ks_env = [] { const char* e = getenv("OCCF_WG8_KS"); return e ? atoi(e) : 0; }();
  static const int st_env = [] { const char* e = getenv("OCCF_WG8_ST"); return e ? atoi(e) : 0; }();
  const int big = ti + tc >= 5;
  const int ks = ks_env ? ks_env : (big ? 1 : 2);
  const int stg = st_env ? st_env : 2;
#define WG8_CASE(TI_, TC_)                                                                      \
  if (ti == TI_ && tc == TC_) {                                                                 \
    if (ks == 1 && stg == 2) return wg8_launch_one<TI_, TC_, 1, 2, F16>(a, st);                       \
    if (ks == 1 && stg == 3) return wg8_launch_one<TI_, TC_, 1, 3, F16>(a, st);                       \
    if (ks == 1 && stg >= 4) return wg8_launch_one<TI_, TC_, 1, 4, F16>(a, st);                       \
    if (stg == 2) return wg8_launch_one<TI_, TC_, 2, 2, F16>(a, st);                                  \
    return wg8_launch_one<TI_, TC_, 2, 3, F16>(a, st);                                                \
  }
  WG8_CASE(1, 1) WG8_CASE(1, 2) WG8_CASE(1, 3) WG8_CASE(2, 1) WG8_CASE(2, 2) WG8_CASE(2, 3) WG8_CASE(3, 1)
  WG8_CASE(3, 2) WG8_CASE(3, 3)
#undef WG8_CASE
}
// one-product mode: a stage of 16 rows is 9 MFMAs per wave at the largest tile -- 32 rows per stage (24.6 KB, two stages,
// two workgroups per CU) unless OCCF_WG8_KS=1
static void wg8_launch_class_x1(const Wg8Args& a, int ti, int tc, hipStream_t st) {
  static const int ks_env = [] { const char* e = getenv("OCCF_WG8_KS"); return e ? atoi(e) : 0; }();
  static const int st_env = [] { const char* e = getenv("OCCF_WG8_ST"); return e ? atoi(e) : 0; }();
#define WG8_CASE(TI_, TC_)                                                                      \
  if (ti == TI_ && tc == TC_) {                                                                 \
    if (ks_env == 1) return wg8_launch_one<TI_, TC_, 1, 2, true, true>(a, st);                        \
    if (st_env == 3) return wg8_launch_one<TI_, TC_, 2, 3, true, true>(a, st);                        \
    return wg8_launch_one<TI_, TC_, 2, 2, true, true>(a, st);                                         \
  }
  WG8_CASE(1, 1) WG8_CASE(1, 2) WG8_CASE(1, 3) WG8_CASE(2, 1) WG8_CASE(2, 2) WG8_CASE(2, 3) WG8_CASE(3, 1)
  WG8_CASE(3, 2) WG8_CASE(3, 3)
#undef WG8_CASE
}
