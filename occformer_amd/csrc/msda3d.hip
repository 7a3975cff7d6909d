// 3-D multi-scale deformable attention sampling core of the pixel decoder.
//
// Reference: projects/mmdet3d_plugin/occformer/necks/multi_scale_deform_attn_3d.py
//   multi_scale_deformable_attn_pytorch :17-80 (per level F.grid_sample, trilinear, zeros
//   padding, align_corners=False, weighted sum over levels*points) and the location /
//   softmax arithmetic of MultiScaleDeformableAttention3D.forward :246-273.
//
// The reference launches one grid_sample per level per layer and materialises a
// [B*heads, Dh, Nq, L*P] stack; here one pass reads the raw sampling offsets and attention
// logits, does the 12-way softmax in registers and gathers the 8 corners of every sample
// straight from the projected value tensor [B, Nv, heads*Dh] (a (key, head) row is Dh
// contiguous floats, read as 16-byte pieces by Dh/4 adjacent lanes).  Reference points are
// the queries' own normalised cell centres (point_generator.py:111-137), recomputed from
// the query index instead of being read from memory.
//
// Gather-bound: the value tensor (<= 70 MB at the 200-grid) is L2/Infinity-Cache resident.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#define MSDA_MAX_LEVELS 4

struct MsdaLevels {
  int n;
  int X[MSDA_MAX_LEVELS], Y[MSDA_MAX_LEVELS], Z[MSDA_MAX_LEVELS], start[MSDA_MAX_LEVELS];
};

template <int VEC, int LP_MAX, bool HM>
__global__ void __launch_bounds__(256) msda3d_fwd_kernel(
    const float* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    float* __restrict__ out, MsdaLevels lv, int B, int Nq, int H, int Dh, int P, long off_ld, long lg_ld) {
  const int lanes = Dh / VEC;
  // XCD remap: a contiguous eighth of the (head-major) index space per XCD -- with 8 heads, one head's
  // value planes per L2, walked in query (= spatial) order, instead of every XCD streaming all heads
  const long gid = (long)occf_xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const long total = (long)B * Nq * H * lanes;
  if (gid >= total) return;
  const int cv = (int)(gid % lanes) * VEC;
  long r = gid / lanes;
  int h, q, b;
  if (HM) {   // head-major: neighbouring lanes = neighbouring queries of ONE head -> their samples
              // (similar offsets, adjacent cells) fall into the same / adjacent cache lines
    q = (int)(r % Nq);
    r /= Nq;
    h = (int)(r % H);
    b = (int)(r / H);
  } else {
    h = (int)(r % H);
    r /= H;
    q = (int)(r % Nq);
    b = (int)(r / Nq);
  }
  const int L = lv.n;
  const int LP = L * P;

  // normalised reference point of this query: its own cell centre, in its own level
  int ql = 0;
  while (ql + 1 < L && q >= lv.start[ql + 1]) ++ql;
  const int local = q - lv.start[ql];
  const int qz = local % lv.Z[ql];
  const int qy = (local / lv.Z[ql]) % lv.Y[ql];
  const int qx = local / (lv.Z[ql] * lv.Y[ql]);
  const float rz = ((float)qz + 0.5f) / (float)lv.Z[ql];
  const float ry = ((float)qy + 0.5f) / (float)lv.Y[ql];
  const float rx = ((float)qx + 0.5f) / (float)lv.X[ql];

  // softmax over the L*P logits of this (query, head)
  const float* lg = logits + (long)(b * Nq + q) * lg_ld + h * LP;
  float w[LP_MAX];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    const float lgi = lg[i < LP ? i : LP - 1];
    w[i] = i < LP ? lgi : -3.0e38f;
    mx = fmaxf(mx, w[i]);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    w[i] = i < LP ? expf(w[i] - mx) : 0.f;
    sum += w[i];
  }
  const float inv = 1.0f / sum;

  const float* of = offs + (long)(b * Nq + q) * off_ld + h * LP * 3;
  const int E = H * Dh;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;

  // 24-bit multiplies (full rate; every factor here is < 2^24) and 32-bit element offsets from one 64-bit base
  // per level: the generic 64-bit index arithmetic per corner was ~45 % of this VALU-bound kernel
  const long Nv = lv.start[L - 1] + (long)lv.X[L - 1] * lv.Y[L - 1] * lv.Z[L - 1];
  const int kstride = HM ? Dh : E;                        // floats between consecutive keys
  int l = 0, pl = 0;                                      // level / point-in-level of sample i (no division)
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    if (i < LP) {
    const int Xl = lv.X[l], Yl = lv.Y[l], Zl = lv.Z[l];
    // loc = ref + off / (Z, Y, X);  grid = 2 loc - 1;  pixel = ((grid + 1) * size - 1) / 2
    const float lz = rz + of[i * 3 + 0] / (float)Zl;
    const float ly = ry + of[i * 3 + 1] / (float)Yl;
    const float lx = rx + of[i * 3 + 2] / (float)Xl;
    const float pz = ((2.f * lz - 1.f + 1.f) * (float)Zl - 1.f) * 0.5f;
    const float py = ((2.f * ly - 1.f + 1.f) * (float)Yl - 1.f) * 0.5f;
    const float px = ((2.f * lx - 1.f + 1.f) * (float)Xl - 1.f) * 0.5f;
    const float fz = floorf(pz), fy = floorf(py), fx = floorf(px);
    const float tz = pz - fz, ty = py - fy, tx = px - fx;
    const int iz = (int)fz, iy = (int)fy, ix = (int)fx;
    const float wgt = w[i] * inv;
    const float* vbase = HM ? value + (((long)b * H + h) * Nv + lv.start[l]) * Dh + cv
                            : value + ((long)b * Nv + lv.start[l]) * E + h * Dh + cv;
    // per axis: the two taps, clamped into the volume, and whether each lies inside it
    const bool vx[2] = {(unsigned)ix < (unsigned)Xl, (unsigned)(ix + 1) < (unsigned)Xl};
    const bool vy[2] = {(unsigned)iy < (unsigned)Yl, (unsigned)(iy + 1) < (unsigned)Yl};
    const bool vz[2] = {(unsigned)iz < (unsigned)Zl, (unsigned)(iz + 1) < (unsigned)Zl};
    const uint32_t cx[2] = {(uint32_t)occf_clampi(ix, Xl - 1), (uint32_t)occf_clampi(ix + 1, Xl - 1)};
    const uint32_t cy[2] = {(uint32_t)occf_clampi(iy, Yl - 1), (uint32_t)occf_clampi(iy + 1, Yl - 1)};
    const uint32_t zk[2] = {occf_umul24((uint32_t)occf_clampi(iz, Zl - 1), (uint32_t)kstride),
                            occf_umul24((uint32_t)occf_clampi(iz + 1, Zl - 1), (uint32_t)kstride)};
    const uint32_t rowx[2] = {occf_umul24(cx[0], (uint32_t)Yl), occf_umul24(cx[1], (uint32_t)Yl)};
    const uint32_t zs = occf_umul24((uint32_t)Zl, (uint32_t)kstride);
    // the 8 corner gathers of a sample are issued together and unconditionally (clamped address): a gather
    // under `if (inside)` is waited for on the spot
    float cw[8];
    uint32_t vo[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const int bx = c >> 2, by = (c >> 1) & 1, bz = c & 1;
      cw[c] = (bx ? tx : 1.f - tx) * (by ? ty : 1.f - ty) * (bz ? tz : 1.f - tz) * wgt;
      vo[c] = occf_umul24(rowx[bx] + cy[by], zs) + zk[bz];
    }
    constexpr int NV4 = VEC % 4 == 0 ? VEC / 4 : 1;
    constexpr int NS = VEC % 4 == 0 ? 1 : VEC;
    float4 t4[8][NV4];
    float t1[8][NS];
    if (VEC % 4 == 0) {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int u = 0; u < NV4; ++u) t4[c][u] = *(const float4*)(vbase + vo[c] + 4 * u);
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int v = 0; v < VEC; ++v) t1[c][v] = vbase[vo[c] + v];
    }
    auto add_corner = [&](int c) __attribute__((always_inline)) {
      if (VEC % 4 == 0) {
#pragma unroll
        for (int u = 0; u < NV4; ++u) {
          acc[(4 * u + 0) % VEC] = fmaf(cw[c], t4[c][u].x, acc[(4 * u + 0) % VEC]);
          acc[(4 * u + 1) % VEC] = fmaf(cw[c], t4[c][u].y, acc[(4 * u + 1) % VEC]);
          acc[(4 * u + 2) % VEC] = fmaf(cw[c], t4[c][u].z, acc[(4 * u + 2) % VEC]);
          acc[(4 * u + 3) % VEC] = fmaf(cw[c], t4[c][u].w, acc[(4 * u + 3) % VEC]);
        }
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[v] = fmaf(cw[c], t1[c][v], acc[v]);
      }
    };
    if (vx[0] && vx[1] && vy[0] && vy[1] && vz[0] && vz[1]) {
      // all 8 taps inside (the common case): plain weighted sum, as grid_sample computes it
#pragma unroll
      for (int c = 0; c < 8; ++c) add_corner(c);
    } else {
      // zeros padding: a tap outside the volume contributes nothing (its clamped stand-in was only loaded)
#pragma unroll
      for (int c = 0; c < 8; ++c)
        if (vx[c >> 2] && vy[(c >> 1) & 1] && vz[c & 1]) add_corner(c);
    }
    if (++pl == P) {
      pl = 0;
      ++l;
    }
    }
  }
  float* o = out + (long)(b * Nq + q) * E + h * Dh + cv;
  if (VEC % 4 == 0) {
#pragma unroll
    for (int u = 0; u < (VEC / 4 > 0 ? VEC / 4 : 1); ++u)
      *(float4*)(o + 4 * u) = make_float4(acc[(4 * u) % VEC], acc[(4 * u + 1) % VEC], acc[(4 * u + 2) % VEC],
                                          acc[(4 * u + 3) % VEC]);
  } else {
#pragma unroll
    for (int v = 0; v < VEC; ++v) o[v] = acc[v];
  }
}

extern "C" int occf_msda3d_fwd(const float* value, const float* sampling_offsets,
                               const float* attn_logits, float* out, const int32_t* level_shapes,
                               int num_levels, int B, int Nq, int heads, int head_dim,
                               int num_points, int value_head_major, long offsets_ld, long logits_ld, void* stream) {
  if (num_levels <= 0 || num_levels > MSDA_MAX_LEVELS || B <= 0 || heads <= 0 || head_dim <= 0 ||
      num_points <= 0 || num_levels * num_points > 16)
    return OCCF_ESHAPE;
  MsdaLevels lv;
  lv.n = num_levels;
  int start = 0;
  for (int l = 0; l < num_levels; ++l) {
    lv.X[l] = level_shapes[l * 3 + 0];
    lv.Y[l] = level_shapes[l * 3 + 1];
    lv.Z[l] = level_shapes[l * 3 + 2];
    lv.start[l] = start;
    start += lv.X[l] * lv.Y[l] * lv.Z[l];
  }
  for (int l = num_levels; l < MSDA_MAX_LEVELS; ++l) lv.X[l] = lv.Y[l] = lv.Z[l] = lv.start[l] = 0;
  if (start != Nq) return OCCF_ESHAPE;   // queries ARE the keys (encoder self-attention)
  hipStream_t st = (hipStream_t)stream;
  const long off_ld = offsets_ld > 0 ? offsets_ld : (long)heads * num_levels * num_points * 3;
  const long lg_ld = logits_ld > 0 ? logits_ld : (long)heads * num_levels * num_points;
  // channels per lane: the location / softmax / corner-weight arithmetic is repeated by every lane of a
  // (query, head), so wider lanes trade gather parallelism for less redundant VALU work (head_dim 24: 12
  // channels per lane = 2 lanes per (query, head) instead of 6: 2.86 -> 2.22 ms per step)
  static const int vec_env = [] {
    const char* e = getenv("OCCF_MSDA_VEC");
    return e ? atoi(e) : 0;
  }();
  int vec = head_dim % 12 == 0 ? 12 : head_dim % 8 == 0 ? 8 : 4;
  if (vec_env > 0 && vec_env % 4 == 0 && head_dim % vec_env == 0 && (vec_env == 4 || vec_env == 8 || vec_env == 12 || vec_env == 24)) vec = vec_env;
  if (head_dim % 4 == 0) {
    const long total = (long)B * Nq * heads * (head_dim / vec);
#define OCCF_MSDA_LAUNCH(V_, HM_)                                                                                  \
    hipLaunchKernelGGL((msda3d_fwd_kernel<V_, 16, HM_>), dim3(occf_cdiv(total, 256)), dim3(256), 0, st, value,     \
                       sampling_offsets, attn_logits, out, lv, B, Nq, heads, head_dim, num_points, off_ld, lg_ld)
    if (value_head_major) {
      if (vec == 24) OCCF_MSDA_LAUNCH(24, true);
      else if (vec == 12) OCCF_MSDA_LAUNCH(12, true);
      else if (vec == 8) OCCF_MSDA_LAUNCH(8, true);
      else OCCF_MSDA_LAUNCH(4, true);
    } else {
      if (vec == 24) OCCF_MSDA_LAUNCH(24, false);
      else if (vec == 12) OCCF_MSDA_LAUNCH(12, false);
      else if (vec == 8) OCCF_MSDA_LAUNCH(8, false);
      else OCCF_MSDA_LAUNCH(4, false);
    }
#undef OCCF_MSDA_LAUNCH
  } else {
    const long total = (long)B * Nq * heads * head_dim;
    if (value_head_major)
      hipLaunchKernelGGL((msda3d_fwd_kernel<1, 16, true>), dim3(occf_cdiv(total, 256)), dim3(256), 0, st, value,
                         sampling_offsets, attn_logits, out, lv, B, Nq, heads, head_dim, num_points, off_ld, lg_ld);
    else
      hipLaunchKernelGGL((msda3d_fwd_kernel<1, 16, false>), dim3(occf_cdiv(total, 256)), dim3(256), 0, st, value,
                         sampling_offsets, attn_logits, out, lv, B, Nq, heads, head_dim, num_points, off_ld, lg_ld);
  }
  OCCF_LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------------------------
// Backward of the sampling core (ATen autograd through softmax + F.grid_sample in the reference,
// multi_scale_deform_attn_3d.py:17-80,246-273).  With S_i[c] the trilinear sample of point i and a = softmax:
//   out[c] = sum_i a_i S_i[c]
//   g_i = <dout, S_i>;   dlogit_i = a_i (g_i - sum_j a_j g_j)
//   dvalue[corner, c] += a_i w_corner dout[c]                  (scatter, float atomics -- as grid_sample's backward)
//   doffset_i = a_i <dout, dS_i/dpos>   (pixel position = loc * size - 0.5 and loc = ref + off / size: d pos/d off = 1)
// LPG lanes share a (query, head), each owns VEC = Dh / LPG consecutive channels: the atomics of a corner cover
// one contiguous head row (coalesced at the L2), the channel sums are LPG-wide butterfly reductions.
// dvalue is TOKEN-major [B, Nv, H*Dh] (what the value projection's backward consumes) and zero-filled by the caller.
template <int VEC, int LP_MAX, bool HM>
__global__ void __launch_bounds__(256) msda3d_bwd_kernel(
    const float* __restrict__ value, const float* __restrict__ offs, const float* __restrict__ logits,
    const float* __restrict__ dout, float* __restrict__ dvalue, float* __restrict__ doffs, float* __restrict__ dlogits,
    MsdaLevels lv, int B, int Nq, int H, int Dh, int P, int LPG, long off_ld, long lg_ld, long doff_ld, long dlg_ld,
    int do_value, float4* __restrict__ rec_w, uint32_t* __restrict__ rec_c) {
  // rec_w / rec_c (optional): one RECORD per (b, q, h, sample) for the value-gradient tiles below -- softmax weight and
  // the three trilinear fractions (float4), base cell as 3 x int16 (two words) -- so that the tiles, which visit every
  // (query, head) once per sampled level AND channel pass (12 times at the 200-grid), do not repeat this pass's softmax
  // and position arithmetic (~550 VALU instructions per visit against ~1 300 for the 32 corners' atomics)
  // XCD remap as in the forward: a contiguous eighth of the (batch, head, query) space per XCD, i.e. with 8 heads the
  // scatter of one head stays on one L2
  const long gid = (long)occf_xcd_remap(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  const long total = (long)B * Nq * H * LPG;
  const bool live = gid < total;
  const long g2 = live ? gid : total - 1;                  // idle lanes shadow the last group (no stores)
  const int sub = (int)(g2 % LPG);
  const int cv = sub * VEC;
  long r = g2 / LPG;
  const int q = (int)(r % Nq);
  r /= Nq;
  const int h = (int)(r % H);
  const int b = (int)(r / H);
  const int L = lv.n;
  const int LP = L * P;
  int ql = 0;
  while (ql + 1 < L && q >= lv.start[ql + 1]) ++ql;
  const int local = q - lv.start[ql];
  const int qz = local % lv.Z[ql];
  const int qy = (local / lv.Z[ql]) % lv.Y[ql];
  const int qx = local / (lv.Z[ql] * lv.Y[ql]);
  const float rz = ((float)qz + 0.5f) / (float)lv.Z[ql];
  const float ry = ((float)qy + 0.5f) / (float)lv.Y[ql];
  const float rx = ((float)qx + 0.5f) / (float)lv.X[ql];
  const float* lg = logits + (long)(b * Nq + q) * lg_ld + h * LP;
  float w[LP_MAX], gi[LP_MAX];
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    const float lgi = lg[i < LP ? i : LP - 1];
    w[i] = i < LP ? lgi : -3.0e38f;
    mx = fmaxf(mx, w[i]);
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    w[i] = i < LP ? expf(w[i] - mx) : 0.f;
    sum += w[i];
  }
  const float inv = 1.0f / sum;
  const float* of = offs + (long)(b * Nq + q) * off_ld + h * LP * 3;
  float* dof = doffs + (long)(b * Nq + q) * doff_ld + h * LP * 3;
  const int E = H * Dh;
  const long Nv = lv.start[L - 1] + (long)lv.X[L - 1] * lv.Y[L - 1] * lv.Z[L - 1];
  float go[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) go[v] = dout[(long)(b * Nq + q) * E + h * Dh + cv + v];
  int l = 0, pl = 0;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) {
    gi[i] = 0.f;
    if (i < LP) {
      const int Xl = lv.X[l], Yl = lv.Y[l], Zl = lv.Z[l];
      const float lz = rz + of[i * 3 + 0] / (float)Zl;
      const float ly = ry + of[i * 3 + 1] / (float)Yl;
      const float lx = rx + of[i * 3 + 2] / (float)Xl;
      const float pz = ((2.f * lz - 1.f + 1.f) * (float)Zl - 1.f) * 0.5f;
      const float py = ((2.f * ly - 1.f + 1.f) * (float)Yl - 1.f) * 0.5f;
      const float px = ((2.f * lx - 1.f + 1.f) * (float)Xl - 1.f) * 0.5f;
      const float fz = floorf(pz), fy = floorf(py), fx = floorf(px);
      const float tz = pz - fz, ty = py - fy, tx = px - fx;
      const int iz = (int)fz, iy = (int)fy, ix = (int)fx;
      const float a = w[i] * inv;
      if (rec_w && live && sub == 0) {
        const long si = ((long)(b * Nq + q) * H + h) * LP + i;
        rec_w[si] = make_float4(a, tz, ty, tx);
        const int cz = iz < -2 ? -2 : (iz > 32000 ? 32000 : iz), cy = iy < -2 ? -2 : (iy > 32000 ? 32000 : iy),
                  cx = ix < -2 ? -2 : (ix > 32000 ? 32000 : ix);           // (beyond the grid either way: no valid corner)
        rec_c[2 * si] = ((uint32_t)cz & 0xFFFFu) | ((uint32_t)cy << 16);
        rec_c[2 * si + 1] = (uint32_t)cx & 0xFFFFu;
      }
      const long kstride = HM ? Dh : E;
      const float* vbase = HM ? value + (((long)b * H + h) * Nv + lv.start[l]) * Dh + cv
                              : value + ((long)b * Nv + lv.start[l]) * E + h * Dh + cv;
      float* dvb = dvalue + ((long)b * Nv + lv.start[l]) * E + h * Dh + cv;
      float s_acc = 0.f, gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int bx = c >> 2, by = (c >> 1) & 1, bz = c & 1;
        const int xx = ix + bx, yy = iy + by, zz = iz + bz;
        if ((unsigned)xx >= (unsigned)Xl || (unsigned)yy >= (unsigned)Yl || (unsigned)zz >= (unsigned)Zl) continue;
        const float wx = bx ? tx : 1.f - tx, wy = by ? ty : 1.f - ty, wz = bz ? tz : 1.f - tz;
        const long key = ((long)xx * Yl + yy) * Zl + zz;
        float dot = 0.f;                                  // <dout, v[corner]> over this lane's channels
        if (VEC % 4 == 0) {                               // (cv, Dh and E multiples of 4: 16-byte aligned rows)
#pragma unroll
          for (int v = 0; v < VEC; v += 4) {
            const float4 t4 = *(const float4*)(vbase + key * kstride + v);
            dot = fmaf(go[v], t4.x, dot);
            dot = fmaf(go[(v + 1) % VEC], t4.y, dot);
            dot = fmaf(go[(v + 2) % VEC], t4.z, dot);
            dot = fmaf(go[(v + 3) % VEC], t4.w, dot);
          }
        } else {
#pragma unroll
          for (int v = 0; v < VEC; ++v) dot = fmaf(go[v], vbase[key * kstride + v], dot);
        }
        s_acc = fmaf(wx * wy * wz, dot, s_acc);
        gx = fmaf((bx ? 1.f : -1.f) * wy * wz, dot, gx);
        gy = fmaf((by ? 1.f : -1.f) * wx * wz, dot, gy);
        gz = fmaf((bz ? 1.f : -1.f) * wx * wy, dot, gz);
        if (live && do_value) {
          const float cw = a * wx * wy * wz;
#pragma unroll
          for (int v = 0; v < VEC; ++v) atomicAdd(dvb + key * E + v, cw * go[v]);
        }
      }
      for (int o = 1; o < LPG; o <<= 1) {
        s_acc += __shfl_xor(s_acc, o);
        gx += __shfl_xor(gx, o);
        gy += __shfl_xor(gy, o);
        gz += __shfl_xor(gz, o);
      }
      gi[i] = s_acc;
      if (live && sub == 0) {
        dof[i * 3 + 0] = a * gz;
        dof[i * 3 + 1] = a * gy;
        dof[i * 3 + 2] = a * gx;
      }
      if (++pl == P) {
        pl = 0;
        ++l;
      }
    }
  }
  float mean = 0.f;
#pragma unroll
  for (int i = 0; i < LP_MAX; ++i) mean = fmaf(w[i] * inv, gi[i], mean);
  if (live && sub == 0) {
    float* dl = dlogits + (long)(b * Nq + q) * dlg_ld + h * LP;
#pragma unroll
    for (int i = 0; i < LP_MAX; ++i)
      if (i < LP) dl[i] = w[i] * inv * (gi[i] - mean);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// d(value) with LDS privatisation.  The plain scatter above issues B*Nq*H*L*P*8*Dh float atomics (1.7e9 per call
// at the 200-grid) onto 91 250 x 192 addresses -- ~770 contributions per address, serialised at the L2 (12.6 ms per
// call, 76 ms per training step).  Here a workgroup owns a TILE of one sampled level (T x T columns, all Z) of one
// head, accumulates in LDS (region = tile + M margin cells, CH channels of the head per pass) the contributions of
// every query whose own cell centre falls into the tile (queries of ALL levels: the levels are nested grids), and
// flushes the non-zero part of the region once.  A sample that leaves the region (offsets larger than the margin)
// falls back to the global atomic, so the result does not depend on the offsets being small.
// LDS float atomics (ds_add_f32) serialise the 64 lanes on gfx950: 768 cycles per wave instruction against 30 for
// ds_add_u32 / ds_add_u64 (scripts/lds_atomic_probe.hip).  The tiles therefore accumulate in 64-bit FIXED POINT:
// contribution * 2^30 / max|dout| rounded to an int32 (|contribution| <= max|dout|) and sign-extended -- 2^33 of them
// fit, the resolution is 2^-30 of the largest gradient entry (an fp32 sum of the same terms carries 2^-24 of its own
// magnitude), and the sum is order-independent (deterministic).
__global__ void __launch_bounds__(256) msda_absmax_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  // a NaN counts as +inf (fmaxf would drop it): with max|dout| = inf the fixed-point scale is 0 and its inverse inf, so
  // every value the tiles hand over is 0 * inf = NaN -- a diverged step stays visible in d(value) instead of going
  // through an undefined float -> int conversion
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = x[i];
    m = v != v ? occf_u2f(0x7F800000u) : fmaxf(m, fabsf(v));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  // one atomic per workgroup (non-negative floats order like ints): 2 048 same-address atomics cost 40 us
  if (threadIdx.x == 0) atomicMax((int*)out, (int)occf_f2u(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

__global__ void msda_zero_word_kernel(unsigned* p) { p[0] = 0u; }

struct MsdaTileCfg {
  int ls, T, M, tiles_x, tiles_y, groups;   // groups > 1: the level is ONE tile, its queries are split over `groups`
  int CH, passes, lpg;                      // channels per pass, passes per head, lanes per query (CH = cpl * lpg)
  int cpl;                                  // channels per lane: 3 or 6
  int stride;                               // 1: strided walk over the queries (conflict spreading), 0: linear
  int pad;                                  // extra 64-bit slots per cell in LDS (see msda_tile_pad)
};

// LDS cell stride = CH + pad 64-bit slots.  Unpadded (CH = 12: 24 dwords) two cells 8 apart share their banks
// (24 * 8 = 3 * 64), and on a level with Zs = 8 that is every pair of NEIGHBOURING (x, y) columns: queries adjacent in y
// hit the same banks in the same wave instruction -- PMC r06j: SQ_LDS_BANK_CONFLICT = 56 % of the kernel's LDS cycles.
// One slot of padding (26 dwords per cell) sends cells c .. c + 31 to 32 different bank pairs.  MEASURED (r06m, same
// tile geometry): 2.374 ms per call unpadded, 2.473 ms padded -- the conflicts are not what bounds the kernel (LDS 37 %
// busy, VALU 45 %, waves waiting 63 % of their cycles: the dependent chain per sample is).  Opt-in: OCCF_MSDA_PAD=1.
static int msda_tile_pad() {
  static const int v = [] {
    const char* e = getenv("OCCF_MSDA_PAD");
    return e ? atoi(e) : 0;
  }();
  return v > 0 ? 1 : 0;
}

__device__ __forceinline__ int msda_cdiv_pos(long num, long den) { return num <= 0 ? 0 : (int)((num + den - 1) / den); }

// signed fixed point of x (|x| <= 2^30): one rounding + one int32 conversion, sign-extended into the 64-bit
// accumulator (the generic float -> int64 conversion is ~25 VALU instructions, three per corner)
__device__ __forceinline__ unsigned long long msda_fx(float x) {
  return (unsigned long long)(long long)(int)rintf(x);
}

// workgroup = up to 1024 threads (16 waves: the tile takes most of the CU's LDS, so the waves that hide the VALU
// work behind the LDS atomics have to come from ONE workgroup; 256 threads: 8.0 ms per call, 512: 5.5, 1024: 4.5);
// work item = query, `lpg` lanes each (3 channels per lane).  Items are walked with a large odd stride so that the
// lanes of one wave instruction belong to queries far apart (no measurable effect at these sizes; kept).
// CPL = channels per lane.  Every lane of a query repeats the query's softmax, the four sample positions and the 32
// corner weights (~1 000 VALU instructions against 32 CPL LDS atomics): 6 channels per lane instead of 3 halve the
// lanes that repeat them.
// PL ("pass loop"): the channel passes of a head run INSIDE the workgroup (zero -> scatter -> flush per pass) instead of
// as separate workgroups, and the sample records of a thread's FIRST query -- softmax weight, fractions and corner of its
// P <= 4 points: everything that does not depend on the channel -- are kept in registers across the passes.  The
// ablation (scripts/msda_bwd_ablation_probe.py, r06af) put the per-query loop at 42 % of the backward and its LDS
// atomics at 2 %: ~1 700 VALU instructions per lane and pass for 192 atomics, repeated in each of the 4 passes of a
// level-0 head.  A thread's later queries (workgroups with more queries than slots) are recomputed per pass as before.
#define MSDA_TILE_THREADS 1024
template <int CPL, bool PL = false>
__global__ void __launch_bounds__(MSDA_TILE_THREADS) msda3d_bwd_value_tile_kernel(
    const float* __restrict__ offs, const float* __restrict__ logits, const float* __restrict__ dout,
    float* __restrict__ dvalue, float* __restrict__ scratch, const unsigned* __restrict__ absmax_bits, MsdaLevels lv,
    MsdaTileCfg tc, int B, int Nq, int H, int Dh, int P, long off_ld, long lg_ld, const float4* __restrict__ rec_w,
    const uint32_t* __restrict__ rec_c) {
  OCCF_DYN_SMEM(smem_raw);
  unsigned long long* tile = (unsigned long long*)smem_raw;
  const int NT = blockDim.x;
  const float gmax = occf_u2f(absmax_bits[0]);
  const float fx_scale = gmax > 0.f ? 1073741824.0f / gmax : 1.0f;          // 2^30 / max|dout|
  const float fx_inv = 1.0f / fx_scale;
  const int L = lv.n, LP = L * P;
  const int ls = tc.ls;
  const int Xs = lv.X[ls], Ys = lv.Y[ls], Zs = lv.Z[ls];
  int bx = blockIdx.x;
  const int pass0 = PL ? 0 : bx % tc.passes;
  if (!PL) bx /= tc.passes;
  const int npass = PL ? tc.passes : 1;
  const int grp = bx % tc.groups;
  bx /= tc.groups;
  const int ty_ = bx % tc.tiles_y, tx_ = bx / tc.tiles_y;
  const int h = blockIdx.y, b = blockIdx.z;
  const int tx0 = tx_ * tc.T, ty0 = ty_ * tc.T;
  const int rx0 = tx0 - tc.M, ry0 = ty0 - tc.M;
  const int RX = tc.T + 2 * tc.M, RY = RX;
  const int CH = tc.CH, CHP = tc.CH + tc.pad;
  const long ncell = (long)RX * RY * Zs;

  // query boxes per query level: cell(q) = floor((2q + 1) * Xs / (2 * Xq)) in [tx0, tx0 + T)
  int qx_lo[MSDA_MAX_LEVELS], qy_lo[MSDA_MAX_LEVELS], nqx[MSDA_MAX_LEVELS], nqy[MSDA_MAX_LEVELS], cum[MSDA_MAX_LEVELS + 1];
  cum[0] = 0;
  for (int lq = 0; lq < MSDA_MAX_LEVELS; ++lq) {
    if (lq < L) {
      const int Xq = lv.X[lq], Yq = lv.Y[lq];
      int a0 = msda_cdiv_pos(2L * tx0 * Xq - Xs, 2L * Xs), a1 = msda_cdiv_pos(2L * (tx0 + tc.T) * Xq - Xs, 2L * Xs);
      int c0 = msda_cdiv_pos(2L * ty0 * Yq - Ys, 2L * Ys), c1 = msda_cdiv_pos(2L * (ty0 + tc.T) * Yq - Ys, 2L * Ys);
      a1 = a1 > Xq ? Xq : a1;
      c1 = c1 > Yq ? Yq : c1;
      if (tx0 + tc.T >= Xs) a1 = Xq;
      if (ty0 + tc.T >= Ys) c1 = Yq;
      qx_lo[lq] = a0; qy_lo[lq] = c0;
      nqx[lq] = a1 > a0 ? a1 - a0 : 0;
      nqy[lq] = c1 > c0 ? c1 - c0 : 0;
      cum[lq + 1] = cum[lq] + nqx[lq] * nqy[lq] * lv.Z[lq];
    } else {
      qx_lo[lq] = qy_lo[lq] = nqx[lq] = nqy[lq] = 0;
      cum[lq + 1] = cum[lq];
    }
  }
  const int n_items = cum[L];
  const int lpg = tc.lpg, slots = NT / lpg;
  const int slot = threadIdx.x / lpg, sub = threadIdx.x % lpg;
  const int E = H * Dh;
  const long Nv = lv.start[L - 1] + (long)lv.X[L - 1] * lv.Y[L - 1] * lv.Z[L - 1];
  // this group's share of the queries; units = (query, point)
  const int per = (n_items + tc.groups - 1) / tc.groups;
  const int it_begin = grp * per, it_end = it_begin + per < n_items ? it_begin + per : n_items;
  const int n_q = it_end > it_begin ? it_end - it_begin : 0;
  unsigned stride = 1u;                                     // coprime with n_q; j * stride stays below 2^32
  if (tc.stride > 0 && n_q > 1 && n_q < 1000000) stride = (unsigned)(((n_q % 4099) ? 4099 : 4111) % n_q);
  // (PL) records of this thread's first query
  int r_q = 0, r_c0[4] = {0, 0, 0, 0}, r_c1[4] = {0, 0, 0, 0};
  float r_a[4] = {0.f, 0.f, 0.f, 0.f}, r_tz[4] = {0.f, 0.f, 0.f, 0.f}, r_ty[4] = {0.f, 0.f, 0.f, 0.f},
        r_tx[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ps = 0; ps < npass; ++ps) {
  const int pass = pass0 + ps;
  const int ch0 = pass * CH;
  if (ps > 0) __syncthreads();                              // the previous pass's flush has read the tile
  for (long i = threadIdx.x; i < ncell * CHP; i += NT) tile[i] = 0ull;
  __syncthreads();
  float* dvb = dvalue + ((long)b * Nv + lv.start[ls]) * E + h * Dh + ch0 + sub * CPL;
  for (int u = slot; u < n_q; u += slots) {
    const bool cached = PL && ps > 0 && u == slot;
    const bool keep = PL && ps == 0 && u == slot;
    int q = r_q;
    float rz = 0.f, ry = 0.f, rx = 0.f, mx = -3.0e38f, inv = 0.f;
    const float* lg = logits;
    if (!cached) {
    const int it = it_begin + (int)(((unsigned)u * stride) % (unsigned)n_q);
    int lq = 0;
    while (lq + 1 < L && it >= cum[lq + 1]) ++lq;
    int r = it - cum[lq];
    const int Zq = lv.Z[lq];
    const int qz = r % Zq;
    r /= Zq;
    const int qy = qy_lo[lq] + r % nqy[lq], qx = qx_lo[lq] + r / nqy[lq];
    q = lv.start[lq] + (qx * lv.Y[lq] + qy) * Zq + qz;
    rz = ((float)qz + 0.5f) / (float)Zq;
    ry = ((float)qy + 0.5f) / (float)lv.Y[lq];
    rx = ((float)qx + 0.5f) / (float)lv.X[lq];
    lg = logits + (long)(b * Nq + q) * lg_ld + h * LP;
    if (keep) r_q = q;
    if (!rec_w) {
      for (int i = 0; i < LP; ++i) mx = fmaxf(mx, lg[i]);
      float sum = 0.f;
      for (int i = 0; i < LP; ++i) sum += expf(lg[i] - mx);
      inv = 1.0f / sum;
    }
    }
    const float* of = offs + (long)(b * Nq + q) * off_ld + h * LP * 3;
    const float* gp = dout + (long)(b * Nq + q) * E + h * Dh + ch0 + sub * CPL;
    float gch[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) gch[c] = gp[c];
    auto point = [&](const int k) __attribute__((always_inline)) {
      const int i = ls * P + k;
      float tz, ty, tx, a;
      int iz, iy, ix;
      if (cached) {
        a = r_a[k & 3]; tz = r_tz[k & 3]; ty = r_ty[k & 3]; tx = r_tx[k & 3];
        iz = (int)(short)(r_c0[k & 3] & 0xFFFF); iy = (int)(short)((unsigned)r_c0[k & 3] >> 16); ix = r_c1[k & 3];
      } else if (rec_w) {                                      // the gather pass's record of this sample (see msda3d_bwd_kernel)
        const long si = ((long)(b * Nq + q) * H + h) * LP + i;
        const float4 rw = rec_w[si];
        const uint32_t c0 = rec_c[2 * si], c1 = rec_c[2 * si + 1];
        a = rw.x; tz = rw.y; ty = rw.z; tx = rw.w;
        iz = (int)(short)(c0 & 0xFFFFu); iy = (int)(short)(c0 >> 16); ix = (int)(short)(c1 & 0xFFFFu);
      } else {
        const float lz = rz + of[i * 3 + 0] / (float)Zs;
        const float ly = ry + of[i * 3 + 1] / (float)Ys;
        const float lx = rx + of[i * 3 + 2] / (float)Xs;
        const float pz = ((2.f * lz - 1.f + 1.f) * (float)Zs - 1.f) * 0.5f;
        const float py = ((2.f * ly - 1.f + 1.f) * (float)Ys - 1.f) * 0.5f;
        const float px = ((2.f * lx - 1.f + 1.f) * (float)Xs - 1.f) * 0.5f;
        const float fz = floorf(pz), fy = floorf(py), fx = floorf(px);
        tz = pz - fz; ty = py - fy; tx = px - fx;
        iz = (int)fz; iy = (int)fy; ix = (int)fx;
        a = expf(lg[i] - mx) * inv;
      }
      if (keep) {
        // (corner indices beyond +-32 767 cells cannot reach the grid: saturate them out of range)
        const int sz = iz < -32768 ? -32768 : iz > 32767 ? 32767 : iz, sy = iy < -32768 ? -32768 : iy > 32767 ? 32767 : iy;
        r_a[k & 3] = a; r_tz[k & 3] = tz; r_ty[k & 3] = ty; r_tx[k & 3] = tx;
        r_c0[k & 3] = (int)(((unsigned)sz & 0xFFFFu) | ((unsigned)sy << 16));
        r_c1[k & 3] = ix;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int cbx = c >> 2, cby = (c >> 1) & 1, cbz = c & 1;
        const int xx = ix + cbx, yy = iy + cby, zz = iz + cbz;
        if ((unsigned)xx >= (unsigned)Xs || (unsigned)yy >= (unsigned)Ys || (unsigned)zz >= (unsigned)Zs) continue;
        const float cw = a * (cbx ? tx : 1.f - tx) * (cby ? ty : 1.f - ty) * (cbz ? tz : 1.f - tz);
        const int lx_ = xx - rx0, ly_ = yy - ry0;
        if ((unsigned)lx_ < (unsigned)RX && (unsigned)ly_ < (unsigned)RY) {
          unsigned long long* t = tile + ((lx_ * RY + ly_) * Zs + zz) * CHP + sub * CPL;
          const float cs = cw * fx_scale;
#pragma unroll
          for (int c = 0; c < CPL; ++c) atomicAdd(t + c, msda_fx(cs * gch[c]));
        } else {
          float* d = dvb + (((long)xx * Ys + yy) * Zs + zz) * E;
#pragma unroll
          for (int c = 0; c < CPL; ++c) atomicAdd(d + c, cw * gch[c]);
        }
      }
    };
    if constexpr (PL) {                                  // (k is a constant in every copy: the records stay in registers)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < P) point(k);
    } else {
      for (int k = 0; k < P; ++k) point(k);
    }
  }
  __syncthreads();
  // hand the region over: plain coalesced stores into this workgroup's slab of the scratch buffer; the gather
  // kernel below sums, for every cell, the (at most 9) regions that cover it -- no atomics, fixed order
  float* slab = scratch + (((long)b * H + h) * ((long)gridDim.x * npass) + (long)blockIdx.x * npass + ps) * (ncell * CH);
  for (long i = threadIdx.x; i < ncell * CH; i += NT) {
    const long cell = i / CH;
    slab[i] = (float)((double)(long long)tile[cell * CHP + (i - cell * CH)] * (double)fx_inv);
  }
  }
}

// dvalue[cell, h*Dh + ch] += sum over the regions that contain the cell (tiles: <= 3 per axis; whole-level mode: the
// `groups` copies).  thread = (b, h, cell of level ls, channel)
__global__ void __launch_bounds__(256) msda3d_bwd_value_gather_kernel(const float* __restrict__ scratch,
                                                                      float* __restrict__ dvalue, MsdaLevels lv,
                                                                      MsdaTileCfg tc, int B, int H, int Dh) {
  const int ls = tc.ls;
  const int Xs = lv.X[ls], Ys = lv.Y[ls], Zs = lv.Z[ls];
  const long cells = (long)Xs * Ys * Zs;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)B * H * cells * Dh) return;
  const int ch = (int)(gid % Dh);
  long r = gid / Dh;
  const long cell = r % cells;
  r /= cells;
  const int h = (int)(r % H);
  const int b = (int)(r / H);
  const int z = (int)(cell % Zs);
  const int y = (int)((cell / Zs) % Ys), x = (int)(cell / ((long)Zs * Ys));
  const int pass = ch / tc.CH, chl = ch - pass * tc.CH;
  const int RX = tc.T + 2 * tc.M, RY = RX;
  const long region = (long)RX * RY * Zs * tc.CH;
  const long nblk = (long)tc.tiles_x * tc.tiles_y * tc.groups * tc.passes;
  const float* base = scratch + ((long)b * H + h) * nblk * region;
  int tx_lo = (x - tc.M - tc.T + 1), ty_lo = (y - tc.M - tc.T + 1);
  tx_lo = tx_lo <= 0 ? 0 : (tx_lo + tc.T - 1) / tc.T;
  ty_lo = ty_lo <= 0 ? 0 : (ty_lo + tc.T - 1) / tc.T;
  int tx_hi = (x + tc.M) / tc.T, ty_hi = (y + tc.M) / tc.T;
  tx_hi = tx_hi > tc.tiles_x - 1 ? tc.tiles_x - 1 : tx_hi;
  ty_hi = ty_hi > tc.tiles_y - 1 ? tc.tiles_y - 1 : ty_hi;
  float s = 0.f;
  for (int tx = tx_lo; tx <= tx_hi; ++tx)
    for (int ty = ty_lo; ty <= ty_hi; ++ty) {
      const long local = (((long)(x - (tx * tc.T - tc.M)) * RY + (y - (ty * tc.T - tc.M))) * Zs + z) * tc.CH + chl;
      for (int g = 0; g < tc.groups; ++g) {
        const long blk = (((long)tx * tc.tiles_y + ty) * tc.groups + g) * tc.passes + pass;
        s += base[blk * region + local];
      }
    }
  const long Nv = lv.start[lv.n - 1] + (long)lv.X[lv.n - 1] * lv.Y[lv.n - 1] * lv.Z[lv.n - 1];
  dvalue[((long)b * Nv + lv.start[ls] + cell) * ((long)H * Dh) + h * Dh + ch] += s;
}

// the same, four channels per thread and 32-bit index arithmetic: blockIdx.y = (b, h), thread = (cell, channel quad).
// The scalar kernel above decodes (b, h, cell, channel) from one 64-bit index -- six 64-bit and eight 32-bit run-time
// divisions per output ELEMENT: its vector pipe was 89 % busy (PMC r06t), 200 us per call for 0.2 GB of traffic.
// Needs Dh % 4 == 0 and tc.CH % 4 == 0 (a quad never straddles a pass).
__global__ void __launch_bounds__(256) msda3d_bwd_value_gather4_kernel(const float* __restrict__ scratch,
                                                                       float* __restrict__ dvalue, MsdaLevels lv,
                                                                       MsdaTileCfg tc, int B, int H, int Dh) {
  const int ls = tc.ls;
  const int Xs = lv.X[ls], Ys = lv.Y[ls], Zs = lv.Z[ls];
  const unsigned cells = (unsigned)(Xs * Ys * Zs), q4 = (unsigned)Dh >> 2;
  const unsigned g = blockIdx.x * 256u + threadIdx.x;
  if (g >= cells * q4) return;
  const unsigned cell = g / q4;
  const int ch = (int)(g - cell * q4) * 4;
  const int h = (int)(blockIdx.y % (unsigned)H), b = (int)(blockIdx.y / (unsigned)H);
  const unsigned cxy = cell / (unsigned)Zs;
  const int z = (int)(cell - cxy * (unsigned)Zs);
  const int x = (int)(cxy / (unsigned)Ys), y = (int)(cxy - (unsigned)x * (unsigned)Ys);
  const int pass = ch / tc.CH, chl = ch - pass * tc.CH;
  const int RX = tc.T + 2 * tc.M, RY = RX;
  const long region = (long)RX * RY * Zs * tc.CH;
  const long nblk = (long)tc.tiles_x * tc.tiles_y * tc.groups * tc.passes;
  const float* base = scratch + ((long)b * H + h) * nblk * region;
  int tx_lo = (x - tc.M - tc.T + 1), ty_lo = (y - tc.M - tc.T + 1);
  tx_lo = tx_lo <= 0 ? 0 : (tx_lo + tc.T - 1) / tc.T;
  ty_lo = ty_lo <= 0 ? 0 : (ty_lo + tc.T - 1) / tc.T;
  int tx_hi = (x + tc.M) / tc.T, ty_hi = (y + tc.M) / tc.T;
  tx_hi = tx_hi > tc.tiles_x - 1 ? tc.tiles_x - 1 : tx_hi;
  ty_hi = ty_hi > tc.tiles_y - 1 ? tc.tiles_y - 1 : ty_hi;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int tx = tx_lo; tx <= tx_hi; ++tx)
    for (int ty = ty_lo; ty <= ty_hi; ++ty) {
      const long local = (((long)(x - (tx * tc.T - tc.M)) * RY + (y - (ty * tc.T - tc.M))) * Zs + z) * tc.CH + chl;
      for (int gq = 0; gq < tc.groups; ++gq) {
        const long blk = (((long)tx * tc.tiles_y + ty) * tc.groups + gq) * tc.passes + pass;
        const float4 v = *(const float4*)(base + blk * region + local);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
    }
  const long Nv = lv.start[lv.n - 1] + (long)lv.X[lv.n - 1] * lv.Y[lv.n - 1] * lv.Z[lv.n - 1];
  float4* d = (float4*)(dvalue + ((long)b * Nv + lv.start[ls] + cell) * ((long)H * Dh) + h * Dh + ch);
  float4 o = *d;
  o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
  *d = o;
}

static bool msda_tile_cfg(const MsdaLevels& lv, int ls, int Dh, MsdaTileCfg& tc) {
  // bytes of LDS per workgroup (8 bytes per element).  Larger tiles are slower: 140 KB 2.51 ms, 156 KB 2.72 ms per call
  // against 2.50 (profiles/r04/r04k_msda_lds_sweep.txt) -- fewer, longer workgroups on the same LDS atomic rate
  const int X = lv.X[ls], Y = lv.Y[ls], Z = lv.Z[ls];
  tc.ls = ls;
  tc.M = 5;
  tc.pad = msda_tile_pad();
  // (the budget grows with the padding so that the tile GEOMETRY stays what the sweep chose: the first padded version
  // kept 124 KB, which turned the coarsest level from one whole-level tile with 32 query groups into four tiles and
  // cost 2.39 -> 3.41 ms per call, r06l)
  const long budget = 124L * 1024 * (12 + tc.pad) / 12;
  // channels per pass: 12 (4 lanes per query) if a tile of at least 4 x 4 columns fits, else 6
  for (int lpg = Dh >= 12 ? 4 : Dh / 3; lpg >= 2; lpg >>= 1) {
    if (Dh % (3 * lpg)) continue;
    tc.CH = 3 * lpg;
    tc.passes = Dh / tc.CH;
    static const int cpl_env = [] {
      const char* e = getenv("OCCF_MSDA_CPL");            // 3: the narrow split (one lane per 3 channels)
      return e ? atoi(e) : 6;
    }();
    // (12 per lane measured slower again: 2.38 vs 2.28 ms per call, profiles/r04/r04m_msda_cpl.txt)
    tc.cpl = (cpl_env >= 6 && tc.CH % 6 == 0) ? 6 : 3;
    tc.lpg = tc.CH / tc.cpl;
    const int CHP = tc.CH + tc.pad;                       // (the LDS footprint counts the padded cells)
    if ((long)X * Y * Z * CHP * 8 <= budget) {          // the whole level as one tile, its queries split over groups
      tc.T = X > Y ? X : Y;
      tc.M = 0;
      tc.tiles_x = tc.tiles_y = 1;
      tc.groups = 32;
      return true;
    }
    int T = 16;
    while (T > 2 && (long)(T + 2 * tc.M) * (T + 2 * tc.M) * Z * CHP * 8 > budget) --T;
    if ((long)(T + 2 * tc.M) * (T + 2 * tc.M) * Z * CHP * 8 > budget || (T < 4 && lpg > 2)) continue;
    tc.groups = 1;
    tc.T = T;
    tc.tiles_x = (X + T - 1) / T;
    tc.tiles_y = (Y + T - 1) / T;
    return true;
  }
  return false;
}

static long msda_scratch_floats(const MsdaLevels& lv, const MsdaTileCfg& tc, int B, int heads) {
  const long region = (long)(tc.T + 2 * tc.M) * (tc.T + 2 * tc.M) * lv.Z[tc.ls] * tc.CH;
  return (long)B * heads * tc.tiles_x * tc.tiles_y * tc.groups * tc.passes * region;
}

static void msda_levels(const int32_t* level_shapes, int num_levels, MsdaLevels& lv, int& total) {
  lv.n = num_levels;
  int start = 0;
  for (int l = 0; l < num_levels; ++l) {
    lv.X[l] = level_shapes[l * 3 + 0];
    lv.Y[l] = level_shapes[l * 3 + 1];
    lv.Z[l] = level_shapes[l * 3 + 2];
    lv.start[l] = start;
    start += lv.X[l] * lv.Y[l] * lv.Z[l];
  }
  for (int l = num_levels; l < MSDA_MAX_LEVELS; ++l) lv.X[l] = lv.Y[l] = lv.Z[l] = lv.start[l] = 0;
  total = start;
}

// OCCF_MSDA_RECORDS=1 (opt-in): the gather pass runs first and leaves per-sample records for the tiles.  Measured
// r05s at the 200-grid: 2.259 against 2.312 ms per call -- the ~550 VALU instructions a tile visit saves are not what
// bounds the tiles (nor were more channels per lane, r04m): the same-address LDS atomics of neighbouring queries are.
static bool msda_records_on() {
  const char* e = getenv("OCCF_MSDA_RECORDS");
  return e && atoi(e) != 0;
}
static long msda_record_floats(int B, long Nq, int heads) { return msda_records_on() ? (long)B * Nq * heads * 16 * 6 : 0; }

// floats of scratch for occf_msda3d_bwd (the largest level's region slabs; the levels run one after the other)
extern "C" long occf_msda3d_bwd_workspace(const int32_t* level_shapes, int num_levels, int B, int heads, int head_dim) {
  if (num_levels <= 0 || num_levels > MSDA_MAX_LEVELS || (head_dim != 12 && head_dim != 24)) return 0;
  MsdaLevels lv;
  int total;
  msda_levels(level_shapes, num_levels, lv, total);
  long need = 0;
  for (int l = 0; l < num_levels; ++l) {
    MsdaTileCfg tc;
    if (!msda_tile_cfg(lv, l, head_dim, tc)) return 0;
    const long n = msda_scratch_floats(lv, tc, B, heads);
    need = n > need ? n : need;
  }
  // + the per-sample records of the gather pass (float4 + two words per (b, q, h, sample); the sample count per head is
  // not an argument here: the 16 the op admits) + the max|dout| word
  return need + msda_record_floats(B, total, heads) + 4;
}

extern "C" int occf_msda3d_bwd(const float* value, const float* sampling_offsets, const float* attn_logits,
                               const float* dout, float* dvalue, float* doffsets, float* dlogits,
                               const int32_t* level_shapes, int num_levels, int B, int Nq, int heads, int head_dim,
                               int num_points, int value_head_major, long offsets_ld, long logits_ld, long doffsets_ld,
                               long dlogits_ld, float* workspace, long workspace_floats, void* stream) {
  if (num_levels <= 0 || num_levels > MSDA_MAX_LEVELS || B <= 0 || heads <= 0 || head_dim <= 0 ||
      num_points <= 0 || num_levels * num_points > 16)
    return OCCF_ESHAPE;
  MsdaLevels lv;
  lv.n = num_levels;
  int start = 0;
  for (int l = 0; l < num_levels; ++l) {
    lv.X[l] = level_shapes[l * 3 + 0];
    lv.Y[l] = level_shapes[l * 3 + 1];
    lv.Z[l] = level_shapes[l * 3 + 2];
    lv.start[l] = start;
    start += lv.X[l] * lv.Y[l] * lv.Z[l];
  }
  for (int l = num_levels; l < MSDA_MAX_LEVELS; ++l) lv.X[l] = lv.Y[l] = lv.Z[l] = lv.start[l] = 0;
  if (start != Nq) return OCCF_ESHAPE;
  hipStream_t st = (hipStream_t)stream;
  const long LP3 = (long)heads * num_levels * num_points * 3, LP1 = (long)heads * num_levels * num_points;
  const long off_ld = offsets_ld > 0 ? offsets_ld : LP3, lg_ld = logits_ld > 0 ? logits_ld : LP1;
  const long doff_ld = doffsets_ld > 0 ? doffsets_ld : LP3, dlg_ld = dlogits_ld > 0 ? dlogits_ld : LP1;
  // d(value): LDS-privatised tiles when the head width allows it (12 / 24 channels), else the plain scatter
  static const int tiled_env = [] {
    const char* e = getenv("OCCF_MSDA_TILED");
    return e ? atoi(e) : 1;
  }();
  float4* rec_w = nullptr;
  uint32_t* rec_c = nullptr;
  MsdaTileCfg cfgs[MSDA_MAX_LEVELS];
  bool tiles = false;
  if (tiled_env && workspace && (head_dim == 12 || head_dim == 24)) {
    bool ok = true;
    long smax = 0;
    for (int l = 0; l < num_levels; ++l) {
      ok = ok && msda_tile_cfg(lv, l, head_dim, cfgs[l]);
      if (ok) {
        const long n = msda_scratch_floats(lv, cfgs[l], B, heads);
        smax = n > smax ? n : smax;
      }
    }
    tiles = ok && smax + 4 <= workspace_floats;
    // (the gather pass runs FIRST either way; with OCCF_MSDA_RECORDS=1 it leaves its per-sample records for the tiles)
    const long rec = msda_record_floats(B, Nq, heads);
    if (tiles && rec > 0 && smax + rec + 4 <= workspace_floats && smax % 4 == 0) {
      rec_w = (float4*)(workspace + smax);
      rec_c = (uint32_t*)(workspace + smax + (long)B * Nq * heads * 16 * 4);
    }
  }
  int lpg = head_dim % 8 == 0 ? 8 : head_dim % 4 == 0 ? 4 : head_dim % 2 == 0 ? 2 : 1;
  int vec = head_dim / lpg;
  while (vec > 6 && lpg < 8) { lpg *= 2; vec = head_dim / lpg; }
  if (head_dim % lpg != 0 || vec > 6) { lpg = 1; vec = head_dim; }
  // Without the scatter (value gradient taken by the LDS tiles) this pass is a pure gather + dot product like the
  // forward, and the forward's split serves it best: 12 channels per lane, i.e. the softmax / position / corner-weight
  // arithmetic of a (query, head) is repeated by 2 lanes instead of 8 (measured r03c; OCCF_MSDA_BWD_VEC12=0 restores
  // the narrow split, which the atomics of the scatter want: one contiguous head row per corner)
  static const int vec12_env = [] {
    const char* e = getenv("OCCF_MSDA_BWD_VEC12");
    return e ? atoi(e) : 1;
  }();
  if (tiles && vec12_env && head_dim % 12 == 0 && (heads * head_dim) % 4 == 0) { vec = 12; lpg = head_dim / 12; }
  const long total = (long)B * Nq * heads * lpg;
  const int gather_do_value = tiles ? 0 : 1;
#define OCCF_MSDB_LAUNCH(V_, HM_)                                                                                   \
  hipLaunchKernelGGL((msda3d_bwd_kernel<V_, 16, HM_>), dim3(occf_cdiv(total, 256)), dim3(256), 0, st, value,        \
                     sampling_offsets, attn_logits, dout, dvalue, doffsets, dlogits, lv, B, Nq, heads, head_dim,    \
                     num_points, lpg, off_ld, lg_ld, doff_ld, dlg_ld, gather_do_value, rec_w, rec_c)
#define OCCF_MSDB_VEC(HM_)                     \
  switch (vec) {                               \
    case 1: OCCF_MSDB_LAUNCH(1, HM_); break;   \
    case 2: OCCF_MSDB_LAUNCH(2, HM_); break;   \
    case 3: OCCF_MSDB_LAUNCH(3, HM_); break;   \
    case 4: OCCF_MSDB_LAUNCH(4, HM_); break;   \
    case 5: OCCF_MSDB_LAUNCH(5, HM_); break;   \
    case 6: OCCF_MSDB_LAUNCH(6, HM_); break;   \
    case 12: OCCF_MSDB_LAUNCH(12, HM_); break; \
    default: return OCCF_ESHAPE;               \
  }
  if (value_head_major) { OCCF_MSDB_VEC(true) } else { OCCF_MSDB_VEC(false) }
#undef OCCF_MSDB_VEC
#undef OCCF_MSDB_LAUNCH
  if (tiles) {
    {
      unsigned* absmax = (unsigned*)(workspace + workspace_floats - 4);
      hipLaunchKernelGGL(msda_zero_word_kernel, dim3(1), dim3(1), 0, st, absmax);
      hipLaunchKernelGGL(msda_absmax_kernel, dim3(512), dim3(256), 0, st, dout, (long)B * Nq * heads * head_dim, absmax);
      static const int tile_threads = [] {
        const char* e = getenv("OCCF_MSDA_TILE_THREADS");
        const int v = e ? atoi(e) : MSDA_TILE_THREADS;
        return v >= 64 && v <= MSDA_TILE_THREADS && v % 64 == 0 ? v : MSDA_TILE_THREADS;
      }();
      static const int strided = [] {
        const char* e = getenv("OCCF_MSDA_STRIDED");
        return e ? atoi(e) : 1;
      }();
      for (int l = 0; l < num_levels; ++l) {
        MsdaTileCfg tc = cfgs[l];
        tc.stride = strided;
        const size_t lds = (size_t)(tc.T + 2 * tc.M) * (tc.T + 2 * tc.M) * lv.Z[l] * (tc.CH + tc.pad) * 8;
        // the channel passes inside the workgroup (kernel template PL) where a head takes several and the points' records
        // fit the four register slots; OCCF_MSDA_PASSLOOP=0: one workgroup per pass
        static const int pl_env = [] {
          const char* e = getenv("OCCF_MSDA_PASSLOOP");
          return e ? atoi(e) : 1;
        }();
        const bool pl = pl_env && tc.passes > 1 && num_points <= 4;
#ifndef OCCF_EMU
        static size_t lds_max[4] = {0, 0, 0, 0};
        const int ci = (tc.cpl == 6 ? 1 : 0) + (pl ? 2 : 0);
        if (lds > lds_max[ci]) {
          const void* fn = ci == 3   ? (const void*)msda3d_bwd_value_tile_kernel<6, true>
                           : ci == 2 ? (const void*)msda3d_bwd_value_tile_kernel<3, true>
                           : ci == 1 ? (const void*)msda3d_bwd_value_tile_kernel<6>
                                     : (const void*)msda3d_bwd_value_tile_kernel<3>;
          hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
          lds_max[ci] = lds;
        }
#endif
        const dim3 grid((unsigned)(tc.tiles_x * tc.tiles_y * tc.groups * (pl ? 1 : tc.passes)), heads, B);
        // threads: as many as the tile's queries fill evenly (level-0 tiles hold ~600 queries x lpg lanes)
        int threads = tile_threads;
        if (tc.groups == 1) {
          long items = 0;
          for (int lq = 0; lq < num_levels; ++lq) {
            const long nx = occf_cdiv((long)tc.T * lv.X[lq], lv.X[l]) + 1, ny = occf_cdiv((long)tc.T * lv.Y[lq], lv.Y[l]) + 1;
            items += (nx < lv.X[lq] ? nx : lv.X[lq]) * (ny < lv.Y[lq] ? ny : lv.Y[lq]) * lv.Z[lq];
          }
          const long lanes = items * tc.lpg;
          const long rounds = occf_cdiv(lanes, tile_threads);
          const long t = occf_cdiv(occf_cdiv(lanes, rounds), 64) * 64;
          threads = (int)(t < 64 ? 64 : (t > tile_threads ? tile_threads : t));
        }
        if (pl && tc.cpl == 6)
          hipLaunchKernelGGL((msda3d_bwd_value_tile_kernel<6, true>), grid, dim3(threads), lds, st, sampling_offsets,
                             attn_logits, dout, dvalue, workspace, absmax, lv, tc, B, Nq, heads, head_dim, num_points, off_ld,
                             lg_ld, (const float4*)rec_w, (const uint32_t*)rec_c);
        else if (pl)
          hipLaunchKernelGGL((msda3d_bwd_value_tile_kernel<3, true>), grid, dim3(threads), lds, st, sampling_offsets,
                             attn_logits, dout, dvalue, workspace, absmax, lv, tc, B, Nq, heads, head_dim, num_points, off_ld,
                             lg_ld, (const float4*)rec_w, (const uint32_t*)rec_c);
        else if (tc.cpl == 6)
          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<6>, grid, dim3(threads), lds, st, sampling_offsets, attn_logits,
                             dout, dvalue, workspace, absmax, lv, tc, B, Nq, heads, head_dim, num_points, off_ld, lg_ld,
                             (const float4*)rec_w, (const uint32_t*)rec_c);
        else
          hipLaunchKernelGGL(msda3d_bwd_value_tile_kernel<3>, grid, dim3(threads), lds, st, sampling_offsets, attn_logits,
                             dout, dvalue, workspace, absmax, lv, tc, B, Nq, heads, head_dim, num_points, off_ld, lg_ld,
                             (const float4*)rec_w, (const uint32_t*)rec_c);
        const long cells = (long)B * heads * lv.X[l] * lv.Y[l] * lv.Z[l] * head_dim;
        const long per_bh = (long)lv.X[l] * lv.Y[l] * lv.Z[l] * (head_dim / 4);
        if (head_dim % 4 == 0 && tc.CH % 4 == 0 && per_bh < 2147483647L && (long)B * heads < 65536 &&
            ((size_t)workspace & 15) == 0 && ((size_t)dvalue & 15) == 0)
          hipLaunchKernelGGL(msda3d_bwd_value_gather4_kernel, dim3((unsigned)occf_cdiv(per_bh, 256), (unsigned)(B * heads)),
                             dim3(256), 0, st, workspace, dvalue, lv, tc, B, heads, head_dim);
        else
          hipLaunchKernelGGL(msda3d_bwd_value_gather_kernel, dim3(occf_cdiv(cells, 256)), dim3(256), 0, st, workspace,
                             dvalue, lv, tc, B, heads, head_dim);
      }
    }
  }
  OCCF_LAUNCH_CHECK();
}
