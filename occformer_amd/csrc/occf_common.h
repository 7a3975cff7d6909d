// Common device-side definitions for the gfx950 (CDNA4) kernels of the OccFormer
// forward hot path.  Wavefront = 64 lanes everywhere.
#pragma once
#include <stdint.h>

#ifdef OCCF_EMU
// test-only host build of the same kernel sources (tests/hipemu); never shipped
#include "hipemu.h"
typedef emu_f32x16 f32x16;
typedef emu_f32x4 f32x4;
typedef emu_bf16x8 bf16x8;
#define occf_mfma_f32_32x32x2(a, b, c) emu_mfma_f32_32x32x2f32(a, b, c)
#define occf_mfma_f32_16x16x4(a, b, c) emu_mfma_f32_16x16x4f32(a, b, c)
#define occf_mfma_bf16_32x32x16(a, b, c) emu_mfma_f32_32x32x16_bf16(a, b, c)
#define occf_mfma_f16_32x32x16(a, b, c) emu_mfma_f32_32x32x16_f16(a, b, c)
static inline float occf_fmul(float a, float b) {
  volatile float r = a * b;
  return r;
}
static inline float occf_fadd(float a, float b) {
  volatile float r = a + b;
  return r;
}
struct float4 {
  float x, y, z, w;
};
struct float2 {
  float x, y;
};
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
#else
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define OCCF_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define occf_mfma_f32_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define occf_mfma_f32_16x16x4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
#define occf_mfma_bf16_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
// the same tile on fp16 elements (operands travel as the 16-byte bit patterns bf16x8 carries)
typedef _Float16 occf_f16x8 __attribute__((ext_vector_type(8)));
#define occf_mfma_f16_32x32x16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(occf_f16x8, a), __builtin_bit_cast(occf_f16x8, b), c, 0, 0, 0)
// un-contracted fp32 ops: the reference's arithmetic is mul-then-add, never fma
__device__ __forceinline__ float occf_fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float occf_fadd(float a, float b) { return __fadd_rn(a, b); }
#endif

#define OCCF_WAVE 64

// ---- fp32 -> bf16 (round to nearest even) and the (hi, lo) split a = hi + lo used by the 3-term products.
// gfx950 converts two values per v_cvt_pk_bf16_f32; the host emulation uses the integer formulation (same
// results for finite inputs).
#ifdef OCCF_EMU
static inline uint32_t occf_f2u(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}
static inline float occf_u2f(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
static inline uint32_t occf_bf16_1(float x) {
  const uint32_t u = occf_f2u(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
static inline uint32_t occf_bf16_pack2(float a, float b) { return occf_bf16_1(a) | (occf_bf16_1(b) << 16); }
#else
typedef float occf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 occf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t occf_f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float occf_u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t occf_bf16_pack2(float a, float b) {
  const occf_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, occf_bf16x2));
}
__device__ __forceinline__ uint32_t occf_bf16_1(float x) { return occf_bf16_pack2(x, 0.f) & 0xFFFFu; }
#endif
// (a, b) -> packed hi pair and packed lo pair (lo = bf16(x - hi))
#ifdef OCCF_EMU
static inline
#else
__device__ __forceinline__
#endif
void occf_bf16_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = occf_bf16_pack2(a, b);
  lo = occf_bf16_pack2(a - occf_u2f(hi << 16), b - occf_u2f(hi & 0xFFFF0000u));
}

// ---- fp16 pieces (11 significant bits) for the two-product weight gradients: x = hi + lo with both halves fp16
// (22 bits; |x| is an activation, far inside the fp16 range), the other operand ONE fp16 piece after a per-tensor
// power-of-two scale (occf_f16_scale_bits).  Round to nearest even in both builds.
#ifdef OCCF_EMU
static inline uint32_t occf_f16_pack2(float a, float b) {
  return (uint32_t)emu_f32_to_f16(a) | ((uint32_t)emu_f32_to_f16(b) << 16);
}
static inline float occf_f16_lo_f32(uint32_t p) { return emu_f16_to_f32((uint16_t)(p & 0xFFFFu)); }
static inline float occf_f16_hi_f32(uint32_t p) { return emu_f16_to_f32((uint16_t)(p >> 16)); }
static inline
#else
typedef _Float16 occf_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t occf_f16_pack2(float a, float b) {
  const occf_f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, occf_f16x2));
}
__device__ __forceinline__ float occf_f16_lo_f32(uint32_t p) { return (float)__builtin_bit_cast(occf_f16x2, p)[0]; }
__device__ __forceinline__ float occf_f16_hi_f32(uint32_t p) { return (float)__builtin_bit_cast(occf_f16x2, p)[1]; }
__device__ __forceinline__
#endif
void occf_f16_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = occf_f16_pack2(a, b);
  lo = occf_f16_pack2(a - occf_f16_lo_f32(hi), b - occf_f16_hi_f32(hi));
}
// power-of-two scale that brings a tensor of absolute maximum `amax_bits` (fp32 bit pattern, >= 0) to [2^14, 2^15):
// returns the scale's bit pattern and, through `inv`, that of its reciprocal.  Elements below amax * 2^-29 flush.
#ifdef OCCF_EMU
static inline
#else
__device__ __forceinline__
#endif
uint32_t occf_f16_scale_bits(uint32_t amax_bits, uint32_t& inv, uint32_t headroom = 0u) {
  // ``headroom`` bits below that: a consumer that ADDS two scaled values before rounding to fp16 (the Winograd input
  // transform) asks for 1 -> maximum in [2^13, 2^14), sums below 2^15
  uint32_t e = (amax_bits >> 23) & 0xFFu;                 // biased exponent of the maximum
  e = e < 15u ? 15u : (e > 250u ? 250u : e);
  inv = (e - 14u + headroom) << 23;                       // 2^(e - 127 - 14 + headroom)
  return (268u - headroom - e) << 23;                     // 2^(14 - headroom - (e - 127))
}

// ---- reproducible scatter sums (ops.deterministic / OCCF_DETERMINISTIC=1): a float atomicAdd makes the result depend on
// the order the contributions arrive in; the same contributions as 64-bit FIXED POINT add up to the same bits in any
// order.  scale = 2^(29 - exponent of max |v|) from a scale slot (occf_absmax_f32): |contribution| < 2^30 converts with
// one rounding + one int32 conversion (the generic float -> int64 conversion is ~25 VALU instructions), sign-extended
// into the 64-bit accumulator; precision 2^-30 of the tensor's maximum.  occf_fx_to_f32 turns the sums back.
#ifdef OCCF_EMU
static inline
#else
__device__ __forceinline__
#endif
float occf_fx_scale(uint32_t amax_bits, float& inv) {
  uint32_t e = (amax_bits >> 23) & 0xFFu;
  e = e < 31u ? 31u : (e > 254u ? 254u : e);
  inv = occf_u2f((e - 29u) << 23);                       // 2^(e - 127 - 29)
  return occf_u2f((283u - e) << 23);                     // 2^(29 - (e - 127))
}
template <bool FX>
#ifdef OCCF_EMU
static inline
#else
__device__ __forceinline__
#endif
void occf_scatter_add(void* base, long idx, float v, float fx_scale) {
  if (FX) atomicAdd((unsigned long long*)base + idx, (unsigned long long)(long long)(int)rintf(v * fx_scale));
  else atomicAdd((float*)base + idx, v);
}

// scheduling fence: keeps the instruction groups on either side in program order (used where the
// compiler's register-minimising order would serialise LDS latency and dependent MFMAs)
#ifdef OCCF_EMU
#define OCCF_SCHED_FENCE() do { } while (0)
#define OCCF_SCHED_GROUP(mask, n) do { } while (0)
#else
#define OCCF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// one link of an explicit issue pipeline: the next `n` instructions of class `mask` (0x008 MFMA, 0x020 VMEM read,
// 0x100 DS read, 0x200 DS write, 0x002 VALU, 0x004 SALU) of this scheduling region, in the order the links appear
#define OCCF_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

// buffer-addressed loads: SGPR resource + 32-bit VGPR byte offset + SGPR byte offset, i.e. no 64-bit
// address arithmetic per load (a gather kernel that walks many same-shaped planes keeps the per-lane
// offsets fixed and moves only the scalar offset)
#ifdef OCCF_EMU
struct occf_buf {
  const char* base;
};
static inline occf_buf occf_make_buf(const void* p) { return occf_buf{(const char*)p}; }
static inline float occf_buf_load_f32(occf_buf b, uint32_t voff, uint32_t soff) {
  float f;
  memcpy(&f, b.base + voff + soff, 4);
  return f;
}
static inline float occf_rcp_fast(float x) { return 1.0f / x; }
// bounded buffer: a 16-byte load whose byte offset does not fit inside [0, nbytes) returns zeros (the hardware's
// out-of-range rule for raw buffers) -- "load or zero" with ONE 32-bit select on the offset
struct occf_bbuf {
  const char* base;
  uint32_t nbytes;
};
struct occf_u32x4 {
  uint32_t x, y, z, w;
};
static inline occf_bbuf occf_make_bbuf(const void* p, uint32_t nbytes) { return occf_bbuf{(const char*)p, nbytes}; }
static inline occf_u32x4 occf_bbuf_load_b128(occf_bbuf b, uint32_t voff) {
  occf_u32x4 v = {0u, 0u, 0u, 0u};
  if ((uint64_t)voff + 16u <= (uint64_t)b.nbytes) memcpy(&v, b.base + voff, 16);
  return v;
}
#else
typedef __amdgpu_buffer_rsrc_t occf_buf;
__device__ __forceinline__ occf_buf occf_make_buf(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0xFFFFFFFF, 0x00020000);
}
__device__ __forceinline__ float occf_buf_load_f32(occf_buf b, uint32_t voff, uint32_t soff) {
  return occf_u2f(__builtin_amdgcn_raw_buffer_load_b32(b, voff, soff, 0));
}
__device__ __forceinline__ float occf_rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }   // 1 ulp
typedef __amdgpu_buffer_rsrc_t occf_bbuf;
typedef uint32_t occf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ occf_bbuf occf_make_bbuf(const void* p, uint32_t nbytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, nbytes, 0x00020000);
}
__device__ __forceinline__ occf_u32x4 occf_bbuf_load_b128(occf_bbuf b, uint32_t voff) {
  return __builtin_amdgcn_raw_buffer_load_b128(b, voff, 0, 0);
}
#endif
#define OCCF_BUF_OOB 0x80000000u        // a byte offset no bounded buffer (< 2 GiB) contains

// LDS-DMA: a bounded 16-byte buffer load that lands in LDS without passing through VGPRs (buffer_load_dwordx4 ... lds).
// The destination is WAVE-UNIFORM base + lane * 16 (the hardware's rule), the source offset is per lane; an offset
// outside the buffer writes zeros.  Completion is tracked by vmcnt: the data may be read after s_waitcnt vmcnt(0) by
// the issuing wave + a workgroup barrier (__syncthreads emits both).  `wave_uniform` = readfirstlane.
#ifdef OCCF_EMU
static inline void occf_bbuf_load_lds_b128(occf_bbuf b, uint32_t voff, void* lds_wave_base) {
  const occf_u32x4 v = occf_bbuf_load_b128(b, voff);
  memcpy((char*)lds_wave_base + (threadIdx.x & 63u) * 16u, &v, 16);
}
static inline int occf_wave_uniform(int v) { return v; }
#else
__device__ __forceinline__ void occf_bbuf_load_lds_b128(occf_bbuf b, uint32_t voff, void* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(b, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, 0, 0, 0);
}
__device__ __forceinline__ int occf_wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#endif

// 24-bit unsigned multiply (v_mul_u32_u24: full rate, v_mul_lo_u32 is quarter rate); operands must be < 2^24
#ifdef OCCF_EMU
static inline uint32_t occf_umul24(uint32_t a, uint32_t b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
#else
__device__ __forceinline__ uint32_t occf_umul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
#endif

// error codes of the C ABI (0 = ok, >0 = hipError_t, <0 = argument error)
#define OCCF_EINVAL (-1)
#define OCCF_ESHAPE (-2)

static inline int occf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware workgroup remap (MI355X: 8 XCDs, workgroup b is observed to run on XCD b % 8, each
// XCD has a private 4 MB L2).  Gives every XCD a CONTIGUOUS range of tiles so that neighbouring
// tiles -- which share operand rows (GEMM) or halo voxels (implicit conv) -- hit the same L2.
// Bijective for any grid size; a wrong placement guess only costs speed.
__device__ __forceinline__ unsigned occf_xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// clamp an index into [0, hi]: the unconditional-load idiom (read a valid address, mask afterwards)
__device__ __forceinline__ int occf_clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// channel groups of the point-sampling kernels: enough threads (>= ~256 K) to hide the dependent gathers
static inline int occf_sample_cgroups(int N, int C, long P) {
  const long np = (long)N * P;
  long g = np > 0 ? (262144 + np - 1) / np : 1;
  return (int)(g < 1 ? 1 : (g > C ? C : g));
}

#define OCCF_LAUNCH_CHECK() return (int)hipGetLastError()
