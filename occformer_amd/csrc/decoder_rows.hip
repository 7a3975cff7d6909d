// The occupancy decoder's per-query work between its attention kernels as TWO kernels per layer (inference):
// DetrTransformerDecoderLayer with ('cross_attn', 'norm', 'self_attn', 'norm', 'ffn', 'norm') +
// Mask2FormerNuscOccHead.forward_head's per-query half (mask2former_nusc_occ.py:426-456, 640-667).
//
// Every operand of that work is [100 queries, 192] (the FFN's hidden layer [100, 1536]): as one launch per linear /
// LayerNorm / add it was ~19 launches of 5-9 us per layer and prediction set, 1.4 ms of a 29 ms forward at 0.1-4 TF
// (profiles/r05/r05z_fwd_kernel_stats.txt: 1 456 linear_small launches per 13 forwards).  The rows are independent
// everywhere except inside the two attentions, so a workgroup owns 16 query rows and walks the whole chain with the
// activations in LDS:
//   K1 (after the cross-attention):  q1 = LN0(o Wo^T + bo + q);  Qs | Ks = (q1 + qpos) [Wq; Wk]^T + b;  Vs = q1 Wv^T + bv
//   K2 (after the self-attention):   q2 = LN1(o2 Wo^T + bo + q1);  q3 = LN2(relu(q2 W1^T + b1) W2^T + b2 + q2);
//                                    d = LNpost(q3);  cls = d Wc^T + bc;  me = MLP3(d);  Qx = (q3 + qpos) Wq'^T + bq'
//                                    (Wq' = the NEXT layer's cross-attention query projection)
// GEMMs: v_mfma_f32_16x16x32_bf16 on the transposed problem D[feature][query] -- the weights are the A operand, read
// from global memory in fragment order (bf16 (hi, lo), packed once per weight version: occf_decoder_rows_pack), the 16
// query rows the B operand, read from LDS as bf16 (hi, lo) images of the current activation; three split products.
// A lane's four accumulator registers are four consecutive features of one query: the epilogue (bias, ReLU, residual)
// writes them as one 16-byte LDS store.  8 waves take the 16-feature tiles round-robin.
#include "occf_common.h"
#include "../../include/occformer_hip.h"

#ifdef OCCF_EMU
#define dr_mfma_16x16x32(a, b, c) emu_mfma_f32_16x16x32_bf16(a, b, c)
#else
#define dr_mfma_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

#define DR_ROWS 16
#define DR_THREADS 512
#define DR_RING 6          // k-steps of weight fragments in flight per wave (48 VGPRs)
// The 7 workgroups of a launch (100 queries) all read the SAME weights, each from first byte to last: placed on ONE XCD
// (workgroup b runs on XCD b % 8: the grid is 8x oversubscribed and only the b % 8 == 0 workgroups work) they share the
// weights in that L2 -- one fetch from HBM / the memory-side cache instead of one per XCD, and L2-hit latency for the
// followers (a wave keeps 12 KB in flight: at ~2 us per miss that is 48 GB/s per CU, 110 us per K2 launch, r06e)
#define DR_XCD_SPREAD 8

struct DrLinear {
  const uint16_t* fh;      // fragments [Npad / 16][K / 32][64 lanes][8]
  const uint16_t* fl;
  const float* bias;       // [N] or NULL
  int N, K;                // N = real outputs; tiles cover ceil(N / 16) * 16 (the pack pads with zero rows)
  int boff;                // float offset of this linear's bias copy in the LDS bias area (padded to 16 N)
};
struct DrNorm {
  const float* gamma;
  const float* beta;
  float eps;
};
struct DrArgs {
  int rows, E, H, Q;       // total rows (B * Q), embed dims, FFN hidden dims, queries per sample (qpos row = row % Q)
  const float* in_o;       // attention output [rows, E]
  const float* in_q;       // residual operand [rows, E]
  const float* qpos;       // [Q, E]
  int mode;                // K2: 1 = whole chain, 0 = head only (in_q is q3)
  int nbias;               // floats of the LDS bias area
  DrLinear out_proj, qk, v;                       // K1
  DrLinear ffn1, ffn2, cls, me0, me1, me2, qnext; // K2 (qnext.fh == NULL: no next layer)
  DrNorm ln_a, ln_b, ln_post;                     // K1: ln_a = norm0;  K2: ln_a = norm1, ln_b = norm2
  float* out_q;            // K1: q1;  K2: q3
  float* out_a;            // K1: Qs;  K2: cls [rows, cls.N]
  float* out_b;            // K1: Ks;  K2: mask embedding [rows, E]
  float* out_c;            // K1: Vs;  K2: Qx [rows, E]
};

typedef uint32_t dr_u2 __attribute__((ext_vector_type(2)));

struct DrSmem {
  float* a;                // [16][E]
  float* b;                // [16][E]
  float* h;                // [16][HC]  (K1: HC = 2E for [Qs | Ks];  K2: HC = 32 ceil(n_cls / 32) for the class logits)
  float* bias;             // every bias of the launch (zeros where a linear has none): the GEMM epilogues read LDS only
  unsigned char* oh;       // operand image of an [16][E] activation, hi: [16][E + 8] bf16
  unsigned char* ol;
  unsigned char* xh;       // operand image of the FFN's hidden activation [16][H + 8] (K2 only): written by the first FFN
  unsigned char* xl;       // GEMM's epilogue -- the fp32 hidden layer (98 KB at H = 1536) never exists
  char* dump;              // 1 KB per wave that nothing reads: where the L2 touches' LDS-DMA lands (dr_touch)
};

// rows [row0, row0 + 16) of a global [rows][E] matrix -> LDS (zeros beyond the last row); with ``mod`` > 0 the source row
// is (row % mod) (the per-sample positional rows)
__device__ __forceinline__ void dr_load(float* dst, const float* src, int row0, int rows, int E, int mod) {
  const int q4 = E >> 2;
  for (int i = threadIdx.x; i < DR_ROWS * q4; i += DR_THREADS) {
    const int r = i / q4, c = i - r * q4;
    const int row = row0 + r;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows) v = *(const float4*)(src + (long)(mod > 0 ? row % mod : row) * E + c * 4);
    *(float4*)(dst + r * E + c * 4) = v;
  }
}
__device__ __forceinline__ void dr_store(float* dst, const float* src, int src_ld, int col0, int row0, int rows, int N) {
  for (int i = threadIdx.x; i < DR_ROWS * N; i += DR_THREADS) {
    const int r = i / N, c = i - r * N;
    if (row0 + r < rows) dst[(long)(row0 + r) * N + c] = src[r * src_ld + col0 + c];
  }
}
// bf16 (hi, lo) images of x (+ add) [16][K] as the B operand of the next GEMM; row stride K + 8 elements: the sixteen
// 16-byte reads of a fragment then start on sixteen distinct 4-bank groups
__device__ __forceinline__ void dr_operand(const DrSmem& s, const float* x, const float* add, int K) {
  const int KP = K + 8, p2 = K >> 1;                  // (K = E: the [16][E + 8] image)
  for (int i = threadIdx.x; i < DR_ROWS * p2; i += DR_THREADS) {
    const int r = i / p2, c = (i - r * p2) * 2;
    float v0 = x[r * K + c], v1 = x[r * K + c + 1];
    if (add) { v0 += add[r * K + c]; v1 += add[r * K + c + 1]; }
    uint32_t hi, lo;
    occf_bf16_split2(v0, v1, hi, lo);
    *(uint32_t*)(s.oh + ((long)r * KP + c) * 2) = hi;
    *(uint32_t*)(s.ol + ((long)r * KP + c) * 2) = lo;
  }
}
// out[q][n] = act(sum_k W[n][k] x[q][k] + bias[n]) (+ res[q][n]) for the 16 rows of the operand image (ih, il); out row
// stride ldo.  out == NULL: the result goes to the hidden-layer operand image (xh, xl) as bf16 (hi, lo) instead.
// A wave's work is the FLAT sequence of (tile, k-step) pairs of its tiles (tile = wave, wave + 8, ...): a ring of DR_RING
// weight fragments stays in flight ACROSS tile boundaries, so a wave stalls on memory once per GEMM, not once per tile
// (with a per-tile ring every tile start paid a full round trip: 110 us per K2 launch, r06e / r06f).
template <int R>
__device__ __forceinline__ void dr_gemm_r(const DrSmem& s, const unsigned char* ih, const unsigned char* il,
                                          const DrLinear& L, int act, float* out, int ldo, const float* res, int ldr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NW = DR_THREADS / 64;
  const int j = lane & 15, g = lane >> 4;
  const int KP = L.K + 8, ksteps = L.K >> 5, ntiles = (L.N + 15) >> 4;      // (ksteps % R == 0: the caller picks R)
  const unsigned char* bh = ih + ((long)j * KP + g * 8) * 2;
  const unsigned char* bl = il + ((long)j * KP + g * 8) * 2;
  const int my_tiles = wave < ntiles ? (ntiles - wave + NW - 1) / NW : 0;
  const int total = my_tiles * ksteps;                      // flat (tile, k-step) items of this wave
  if (total == 0) return;
  // fragment address of flat item f (clamped to the last one: every refill is an unconditional load)
  auto frag = [&](const uint16_t* base, int f) __attribute__((always_inline)) -> bf16x8 {
    const int fc = f < total ? f : total - 1;
    const int t = fc / ksteps, ks = fc - t * ksteps;
    return *(const bf16x8*)(base + ((long)((wave + t * NW) * ksteps + ks) * 64 + lane) * 8);
  };
  bf16x8 rh[R], rl[R];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    rh[i] = frag(L.fh, i);
    rl[i] = frag(L.fl, i);
    OCCF_SCHED_FENCE();      // (slot order = issue order: the wait at the loop header is the one both entries need)
  }
  // The ring cycle (R k-steps) is STRAIGHT-LINE code: no branch, every refill unconditional (clamped).  With a branch in
  // the cycle (a tile boundary test per k-step) the compiler's wait-count pass gave up on the loop-carried ring and
  // waited vmcnt(0) at the top of every cycle -- the newest refill's full latency once per R k-steps: 103 us per K2
  // launch whatever the ring depth or the cache level the weights came from (r06e ... r06g).  A tile is a whole number of
  // cycles; its epilogue (between cycles) touches LDS only: the biases were staged there at kernel start.
  int f = 0;
  for (int t = 0; t < my_tiles; ++t) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};   // (alternating k-steps: two dependency chains)
    for (int ks0 = 0; ks0 < ksteps; ks0 += R) {
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const bf16x8 xh = *(const bf16x8*)(bh + (ks0 + i) * 64), xl = *(const bf16x8*)(bl + (ks0 + i) * 64);
        if (i & 1) {
          acc1 = dr_mfma_16x16x32(rl[i], xh, acc1);
          acc1 = dr_mfma_16x16x32(rh[i], xl, acc1);
          acc1 = dr_mfma_16x16x32(rh[i], xh, acc1);
        } else {
          acc0 = dr_mfma_16x16x32(rl[i], xh, acc0);
          acc0 = dr_mfma_16x16x32(rh[i], xl, acc0);
          acc0 = dr_mfma_16x16x32(rh[i], xh, acc0);
        }
        rh[i] = frag(L.fh, f + i + R);
        rl[i] = frag(L.fl, f + i + R);
        // (pin the refill behind its slot's MFMAs: left alone the scheduler sank all 2 R loads to the end of the cycle
        // and the loop waited vmcnt(0) at its top -- the ring was never in flight)
        OCCF_SCHED_FENCE();
      }
      f += R;
    }
    // ---- tile finished: this lane holds features n .. n + 3 of query row j
    const int n = (wave + t * NW) * 16 + g * 4;
    const float4 bv = *(const float4*)(s.bias + L.boff + n);
    float v[4] = {(acc0[0] + acc1[0]) + bv.x, (acc0[1] + acc1[1]) + bv.y, (acc0[2] + acc1[2]) + bv.z,
                  (acc0[3] + acc1[3]) + bv.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (act == 1) v[r] = fmaxf(v[r], 0.f);
      if (res) v[r] += res[j * ldr + n + r];
    }
    if (out) {
      *(float4*)(out + j * ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      uint32_t h0, l0, h1, l1;
      occf_bf16_split2(v[0], v[1], h0, l0);
      occf_bf16_split2(v[2], v[3], h1, l1);
      const dr_u2 hi = {h0, h1}, lo = {l0, l1};
      *(dr_u2*)(s.xh + ((long)j * (L.N + 8) + n) * 2) = hi;
      *(dr_u2*)(s.xl + ((long)j * (L.N + 8) + n) * 2) = lo;
    }
  }
}
__device__ __forceinline__ void dr_gemm(const DrSmem& s, const unsigned char* ih, const unsigned char* il,
                                        const DrLinear& L, int act, float* out, int ldo, const float* res, int ldr) {
  const int ksteps = L.K >> 5;
  if (ksteps % DR_RING == 0) dr_gemm_r<DR_RING>(s, ih, il, L, act, out, ldo, res, ldr);       // (192, 1536, ...)
  else if (ksteps % 3 == 0) dr_gemm_r<3>(s, ih, il, L, act, out, ldo, res, ldr);
  else if (ksteps % 2 == 0) dr_gemm_r<2>(s, ih, il, L, act, out, ldo, res, ldr);
  else dr_gemm_r<1>(s, ih, il, L, act, out, ldo, res, ldr);
}
// the bias of a linear (zeros when it has none, and in the padding up to 16 ceil(N / 16)) -> the LDS bias area
__device__ __forceinline__ void dr_stage_bias(const DrSmem& s, const DrLinear& L) {
  if (!L.fh) return;
  const int np = ((L.N + 15) >> 4) << 4;
  for (int i = threadIdx.x; i < np; i += DR_THREADS) s.bias[L.boff + i] = (L.bias && i < L.N) ? L.bias[i] : 0.f;
}
// every weight of the launch into this XCD's L2 before the chain starts: one 16-byte LDS-DMA touch per 128-byte line
// (buffer_load ... lds: no VGPR, no dependency, so a wave issues its touches back to back and only the first barrier waits
// for them), spread over the launch's workgroups.  The GEMMs then see L2-hit latency on every fragment; the first
// version touched with ordinary loads whose xor chain waited for each line in turn.  ``dump``: 1 KB of LDS per wave that
// nobody reads before it is overwritten.
__device__ __forceinline__ void dr_touch(const DrLinear& L, int part, int parts, char* dump) {
  if (!L.fh) return;
  const int lane = threadIdx.x & 63, wave = occf_wave_uniform(threadIdx.x >> 6);
  const uint32_t nbytes = (uint32_t)(((L.N + 15) >> 4) * (L.K >> 5)) * 1024u;        // bytes of one array
  const occf_bbuf bh = occf_make_bbuf(L.fh, nbytes), bl = occf_make_bbuf(L.fl, nbytes);
  const int nchunks = (int)((nbytes + 8191u) / 8192u);                                // 64 lines of 128 bytes
  for (int c = part * (DR_THREADS / 64) + wave; c < nchunks; c += parts * (DR_THREADS / 64)) {
    const uint32_t voff = (uint32_t)c * 8192u + (uint32_t)lane * 128u;
    occf_bbuf_load_lds_b128(bh, voff, dump + wave * 1024);
    occf_bbuf_load_lds_b128(bl, voff, dump + wave * 1024);
  }
}
// LayerNorm of the 16 rows in place (32 lanes per row, two-pass statistics in fp32 as ATen: mean, then centred squares)
__device__ __forceinline__ void dr_layernorm(float* x, int E, const DrNorm& n) {
  const int r = threadIdx.x >> 5, l = threadIdx.x & 31;
  float s = 0.f;
  for (int c = l; c < E; c += 32) s += x[r * E + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
  const float mean = s / (float)E;
  float q = 0.f;
  for (int c = l; c < E; c += 32) {
    const float d = x[r * E + c] - mean;
    q = fmaf(d, d, q);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o);
  const float rstd = 1.0f / sqrtf(q / (float)E + n.eps);
  for (int c = l; c < E; c += 32) x[r * E + c] = (x[r * E + c] - mean) * rstd * n.gamma[c] + n.beta[c];
}

// HC = columns of the h buffer, NB = floats of the bias area; H = 0: no hidden-layer image (K1)
__device__ __forceinline__ DrSmem dr_smem(char* smem, int E, int H, int HC, int NB) {
  DrSmem s;
  s.a = (float*)smem;
  s.b = s.a + DR_ROWS * E;
  s.h = s.b + DR_ROWS * E;
  s.bias = s.h + DR_ROWS * HC;
  s.oh = (unsigned char*)(s.bias + NB);
  s.ol = s.oh + (size_t)DR_ROWS * (E + 8) * 2;
  s.dump = (char*)(s.ol + (size_t)DR_ROWS * (E + 8) * 2);
  s.xh = (unsigned char*)s.dump + (DR_THREADS / 64) * 1024;
  s.xl = s.xh + (size_t)DR_ROWS * (H + 8) * 2;
  return s;
}
static size_t dr_smem_bytes(int E, int H, int HC, int NB) {
  return (size_t)DR_ROWS * (2 * E + HC) * 4 + (size_t)NB * 4 + (size_t)2 * DR_ROWS * (E + 8) * 2 + (DR_THREADS / 64) * 1024 +
         (H ? (size_t)2 * DR_ROWS * (H + 8) * 2 : 0);
}
static int dr_pad16(int n) { return (n + 15) / 16 * 16; }

__global__ void __launch_bounds__(DR_THREADS) decoder_rows_k1_kernel(DrArgs p) {
  OCCF_DYN_SMEM(smem);
  if (blockIdx.x % DR_XCD_SPREAD) return;
  const DrSmem s = dr_smem(smem, p.E, 0, 2 * p.E, p.nbias);
  const int row0 = (blockIdx.x / DR_XCD_SPREAD) * DR_ROWS, E = p.E;
  dr_stage_bias(s, p.out_proj);
  dr_stage_bias(s, p.qk);
  dr_stage_bias(s, p.v);
  {
    const int part = blockIdx.x / DR_XCD_SPREAD, parts = gridDim.x / DR_XCD_SPREAD;
    char* dump = s.dump;
    dr_touch(p.out_proj, part, parts, dump);
    dr_touch(p.qk, part, parts, dump);
    dr_touch(p.v, part, parts, dump);
  }
  dr_load(s.a, p.in_o, row0, p.rows, E, 0);
  dr_load(s.b, p.in_q, row0, p.rows, E, 0);
  __syncthreads();
  dr_operand(s, s.a, nullptr, E);
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.out_proj, 0, s.a, E, s.b, E);   // o Wo^T + bo + q   (s.a is no longer read: its image is)
  __syncthreads();
  dr_layernorm(s.a, E, p.ln_a);                       // q1
  dr_load(s.b, p.qpos, row0, p.rows, E, p.Q);
  __syncthreads();
  dr_store(p.out_q, s.a, E, 0, row0, p.rows, E);
  dr_operand(s, s.a, s.b, E);                         // q1 + qpos
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.qk, 0, s.h, 2 * E, nullptr, 0);   // [Qs | Ks]
  __syncthreads();
  dr_store(p.out_a, s.h, 2 * E, 0, row0, p.rows, E);
  dr_store(p.out_b, s.h, 2 * E, E, row0, p.rows, E);
  dr_operand(s, s.a, nullptr, E);                     // q1
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.v, 0, s.b, E, nullptr, 0);        // Vs
  __syncthreads();
  dr_store(p.out_c, s.b, E, 0, row0, p.rows, E);
}

__global__ void __launch_bounds__(DR_THREADS) decoder_rows_k2_kernel(DrArgs p) {
  OCCF_DYN_SMEM(smem);
  if (blockIdx.x % DR_XCD_SPREAD) return;
  const int HC = 32 * ((p.cls.N + 31) / 32);
  const DrSmem s = dr_smem(smem, p.E, p.H, HC, p.nbias);
  const int row0 = (blockIdx.x / DR_XCD_SPREAD) * DR_ROWS, E = p.E, H = p.H;
  dr_stage_bias(s, p.out_proj);
  dr_stage_bias(s, p.ffn1);
  dr_stage_bias(s, p.ffn2);
  dr_stage_bias(s, p.cls);
  dr_stage_bias(s, p.me0);
  dr_stage_bias(s, p.me1);
  dr_stage_bias(s, p.me2);
  dr_stage_bias(s, p.qnext);
  {
    const int part = blockIdx.x / DR_XCD_SPREAD, parts = gridDim.x / DR_XCD_SPREAD;
    char* dump = s.dump;
    if (p.mode) {
      dr_touch(p.out_proj, part, parts, dump);
      dr_touch(p.ffn1, part, parts, dump);
      dr_touch(p.ffn2, part, parts, dump);
    }
    dr_touch(p.qnext, part, parts, dump);
    dr_touch(p.cls, part, parts, dump);
    dr_touch(p.me0, part, parts, dump);
    dr_touch(p.me1, part, parts, dump);
    dr_touch(p.me2, part, parts, dump);
  }
  if (p.mode) {
    dr_load(s.a, p.in_o, row0, p.rows, E, 0);
    dr_load(s.b, p.in_q, row0, p.rows, E, 0);
    __syncthreads();
    dr_operand(s, s.a, nullptr, E);
    __syncthreads();
    dr_gemm(s, s.oh, s.ol, p.out_proj, 0, s.a, E, s.b, E);     // o2 Wo^T + bo + q1
    __syncthreads();
    dr_layernorm(s.a, E, p.ln_a);                     // q2
    __syncthreads();
    dr_operand(s, s.a, nullptr, E);
    __syncthreads();
    dr_gemm(s, s.oh, s.ol, p.ffn1, 1, nullptr, 0, nullptr, 0);   // relu(q2 W1^T + b1) -> the hidden layer's operand image
    __syncthreads();
    dr_gemm(s, s.xh, s.xl, p.ffn2, 0, s.b, E, s.a, E);           // . W2^T + b2 + q2
    __syncthreads();
    dr_layernorm(s.b, E, p.ln_b);                     // q3
  } else {
    dr_load(s.b, p.in_q, row0, p.rows, E, 0);         // q3 given (the prediction set in front of layer 0)
  }
  __syncthreads();
  if (p.out_q) dr_store(p.out_q, s.b, E, 0, row0, p.rows, E);
  if (p.qnext.fh) {
    dr_load(s.a, p.qpos, row0, p.rows, E, p.Q);
    __syncthreads();
    dr_operand(s, s.b, s.a, E);                       // q3 + qpos
    __syncthreads();
    dr_gemm(s, s.oh, s.ol, p.qnext, 0, s.a, E, nullptr, 0);     // the next layer's cross-attention queries
    __syncthreads();
    dr_store(p.out_c, s.a, E, 0, row0, p.rows, E);
  }
  __syncthreads();
  dr_layernorm(s.b, E, p.ln_post);                    // d = post_norm(q3)
  __syncthreads();
  dr_operand(s, s.b, nullptr, E);
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.cls, 0, s.h, HC, nullptr, 0);
  dr_gemm(s, s.oh, s.ol, p.me0, 1, s.a, E, nullptr, 0);
  __syncthreads();
  dr_store(p.out_a, s.h, HC, 0, row0, p.rows, p.cls.N);
  dr_operand(s, s.a, nullptr, E);
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.me1, 1, s.b, E, nullptr, 0);
  __syncthreads();
  dr_operand(s, s.b, nullptr, E);
  __syncthreads();
  dr_gemm(s, s.oh, s.ol, p.me2, 0, s.a, E, nullptr, 0);
  __syncthreads();
  dr_store(p.out_b, s.a, E, 0, row0, p.rows, E);
}

// w fp32 [N][K] (row stride ld) -> (hi, lo) fragments [ceil(N / 16)][K / 32][64 lanes][8]: element e of lane l =
// w[nt * 16 + (l & 15)][ks * 32 + 8 * (l >> 4) + e]; rows >= N are zeros.  thread = one 16-byte group of each array
__global__ void __launch_bounds__(256) decoder_rows_pack_kernel(const float* __restrict__ w, long ld, int N, int K,
                                                                uint16_t* __restrict__ fh, uint16_t* __restrict__ fl) {
  const int ksteps = K >> 5, ntiles = (N + 15) >> 4;
  const long total = (long)ntiles * ksteps * 64;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int lane = (int)(gid & 63);
  const long t = gid >> 6;
  const int ks = (int)(t % ksteps), nt = (int)(t / ksteps);
  const int n = nt * 16 + (lane & 15), k = ks * 32 + 8 * (lane >> 4);
  float u[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) u[e] = n < N ? w[(long)n * ld + k + e] : 0.f;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  uint32_t hh[4], ll[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) occf_bf16_split2(u[2 * q], u[2 * q + 1], hh[q], ll[q]);
  u4 h, l;
  h.x = hh[0]; h.y = hh[1]; h.z = hh[2]; h.w = hh[3];
  l.x = ll[0]; l.y = ll[1]; l.z = ll[2]; l.w = ll[3];
  *(u4*)(fh + gid * 8) = h;
  *(u4*)(fl + gid * 8) = l;
}

extern "C" long occf_decoder_rows_pack_elems(int N, int K) {
  if (N <= 0 || K <= 0 || K % 32) return 0;
  return (long)((N + 15) / 16) * 16 * K;
}
extern "C" int occf_decoder_rows_pack(const float* w, long ld, int N, int K, uint16_t* f_hi, uint16_t* f_lo, void* stream) {
  if (occf_decoder_rows_pack_elems(N, K) == 0 || !w || !f_hi || !f_lo || ld < K) return OCCF_ESHAPE;
  const long groups = occf_decoder_rows_pack_elems(N, K) / 8;
  hipLaunchKernelGGL(decoder_rows_pack_kernel, dim3(occf_cdiv(groups, 256)), dim3(256), 0, (hipStream_t)stream, w, ld, N,
                     K, f_hi, f_lo);
  OCCF_LAUNCH_CHECK();
}

static int dr_check(int rows, int E, int H, int Q) {
  if (rows <= 0 || Q <= 0 || rows % Q || E <= 0 || E % 32 || H <= 0 || H % 32) return OCCF_ESHAPE;
  return 0;
}
// (boff: running offset into the bias area, advanced by 16 ceil(N / 16) for every linear that exists)
static DrLinear dr_lin(const uint16_t* const* f, const float* bias, int N, int K, int& boff) {
  DrLinear L = {f ? f[0] : nullptr, f ? f[1] : nullptr, bias, N, K, boff};
  if (f) boff += dr_pad16(N);
  return L;
}

/* K1.  frags[i] = {hi, lo} of out_proj [E, E], [Wq; Wk] [2E, E], Wv [E, E]; biases alike; ln0 = {gamma, beta}. */
extern "C" int occf_decoder_rows_k1(const float* attn_out, const float* q_in, const float* qpos, int rows, int E, int Q,
                                    const uint16_t* const* f_out, const float* b_out, const float* ln_gamma,
                                    const float* ln_beta, float ln_eps, const uint16_t* const* f_qk, const float* b_qk,
                                    const uint16_t* const* f_v, const float* b_v, float* q1, float* qs, float* ks,
                                    float* vs, void* stream) {
  const int rc = dr_check(rows, E, 32, Q);
  if (rc) return rc;
  DrArgs a = {};
  a.rows = rows; a.E = E; a.H = 0; a.Q = Q; a.in_o = attn_out; a.in_q = q_in; a.qpos = qpos;
  int nb = 0;
  a.out_proj = dr_lin(f_out, b_out, E, E, nb); a.qk = dr_lin(f_qk, b_qk, 2 * E, E, nb); a.v = dr_lin(f_v, b_v, E, E, nb);
  a.nbias = nb;
  a.ln_a = DrNorm{ln_gamma, ln_beta, ln_eps};
  a.out_q = q1; a.out_a = qs; a.out_b = ks; a.out_c = vs;
  const size_t lds = dr_smem_bytes(E, 0, 2 * E, nb);
  if (lds > 160 * 1024) return OCCF_ESHAPE;
#ifndef OCCF_EMU
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)decoder_rows_k1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
#endif
  hipLaunchKernelGGL(decoder_rows_k1_kernel, dim3(DR_XCD_SPREAD * ((rows + DR_ROWS - 1) / DR_ROWS)), dim3(DR_THREADS), lds,
                     (hipStream_t)stream, a);
  OCCF_LAUNCH_CHECK();
}

/* K2.  mode 1: the whole chain from the self-attention output; mode 0: head only (q_in = the queries themselves,
 * attn_out unused).  f_qnext NULL: no next layer (qx unused).  n_cls = classes + 1. */
extern "C" int occf_decoder_rows_k2(int mode, const float* attn_out, const float* q_in, const float* qpos, int rows,
                                    int E, int H, int Q, const uint16_t* const* f_out, const float* b_out,
                                    const float* ln1_gamma, const float* ln1_beta, float ln1_eps,
                                    const uint16_t* const* f_ffn1, const float* b_ffn1, const uint16_t* const* f_ffn2,
                                    const float* b_ffn2, const float* ln2_gamma, const float* ln2_beta, float ln2_eps,
                                    const float* post_gamma, const float* post_beta, float post_eps,
                                    const uint16_t* const* f_cls, const float* b_cls, int n_cls,
                                    const uint16_t* const* f_me0, const float* b_me0, const uint16_t* const* f_me1,
                                    const float* b_me1, const uint16_t* const* f_me2, const float* b_me2,
                                    const uint16_t* const* f_qnext, const float* b_qnext, float* q3, float* cls,
                                    float* mask_embed, float* qx, void* stream) {
  const int rc = dr_check(rows, E, H, Q);
  if (rc) return rc;
  if (n_cls <= 0 || n_cls > 2 * E) return OCCF_ESHAPE;
  DrArgs a = {};
  a.rows = rows; a.E = E; a.H = H; a.Q = Q; a.in_o = attn_out; a.in_q = q_in; a.qpos = qpos; a.mode = mode;
  int nb = 0;
  a.out_proj = dr_lin(mode ? f_out : nullptr, b_out, E, E, nb);
  a.ffn1 = dr_lin(mode ? f_ffn1 : nullptr, b_ffn1, H, E, nb); a.ffn2 = dr_lin(mode ? f_ffn2 : nullptr, b_ffn2, E, H, nb);
  a.cls = dr_lin(f_cls, b_cls, n_cls, E, nb);
  a.me0 = dr_lin(f_me0, b_me0, E, E, nb); a.me1 = dr_lin(f_me1, b_me1, E, E, nb); a.me2 = dr_lin(f_me2, b_me2, E, E, nb);
  a.qnext = dr_lin(f_qnext, b_qnext, E, E, nb);
  a.nbias = nb;
  a.ln_a = DrNorm{ln1_gamma, ln1_beta, ln1_eps}; a.ln_b = DrNorm{ln2_gamma, ln2_beta, ln2_eps};
  a.ln_post = DrNorm{post_gamma, post_beta, post_eps};
  a.out_q = q3; a.out_a = cls; a.out_b = mask_embed; a.out_c = qx;
  const size_t lds = dr_smem_bytes(E, H, 32 * ((n_cls + 31) / 32), nb);
  if (lds > 160 * 1024) return OCCF_ESHAPE;
#ifndef OCCF_EMU
  static bool attr = false;
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)decoder_rows_k2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
    attr = true;
  }
#endif
  hipLaunchKernelGGL(decoder_rows_k2_kernel, dim3(DR_XCD_SPREAD * ((rows + DR_ROWS - 1) / DR_ROWS)), dim3(DR_THREADS), lds,
                     (hipStream_t)stream, a);
  OCCF_LAUNCH_CHECK();
}
