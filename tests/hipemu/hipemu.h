// TEST-ONLY host emulation of the HIP execution model (workgroups of 64-lane
// wavefronts, LDS, barriers, cross-lane ops, MFMA) so that the kernel SOURCES under
// occformer_amd/csrc can be compiled for x86 and their index math / tiling / masking
// checked against the oracle in the GPU-less build container.
//
// It is NOT a fallback: the product loader (occformer_amd/_lib.py) only ever loads the
// gfx950 library and raises if it is missing; this header is reached only through
// tests/hipemu/build.py (-DOCCF_EMU).
//
// Model: every thread of a block is a ucontext fiber; blocks run sequentially.
// __syncthreads and wave collectives are rendezvous points between fibers.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 {
  unsigned x, y, z;
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }

namespace hipemu {
constexpr int WAVE = 64;
struct Wave {
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  alignas(16) unsigned char slot[64][64];  // up to 64 B per lane of exchange payload
};
// minimal x86-64 context switch (callee-saved registers + stack pointer); ucontext's
// swapcontext makes a sigprocmask syscall per switch, ~50x slower
extern "C" void hipemu_swap(void** save_sp, void* load_sp);

struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  emu_uint3 tid;
  int lane, wave;
};
struct Block {
  std::vector<Fiber> fibers;
  std::vector<Wave> waves;
  int alive = 0, arrived = 0;
  unsigned gen = 0;
  void* sched_sp = nullptr;
  int cur = -1;
};
extern Block* g_blk;
extern emu_uint3 g_bid;
extern dim3 g_bdim, g_gdim;
extern char* g_dyn_smem;
extern void (*g_entry)(void*);
extern void* g_entry_arg;

inline Fiber& cur() { return g_blk->fibers[g_blk->cur]; }
inline void yield() { hipemu_swap(&cur().sp, g_blk->sched_sp); }

inline void block_sync() {
  Block& b = *g_blk;
  b.arrived++;
  if (b.arrived >= b.alive) {
    b.arrived = 0;
    b.gen++;
  } else {
    unsigned g = b.gen;
    while (b.gen == g) yield();
  }
}
inline void wave_sync() {
  Wave& w = g_blk->waves[cur().wave];
  w.arrived++;
  if (w.arrived >= w.alive) {
    w.arrived = 0;
    w.gen++;
  } else {
    unsigned g = w.gen;
    while (w.gen == g) yield();
  }
}
template <class T>
inline T lane_exchange(T v, int src_lane) {
  static_assert(sizeof(T) <= 64, "payload too large");
  Wave& w = g_blk->waves[cur().wave];
  memcpy(w.slot[cur().lane], &v, sizeof(T));
  wave_sync();
  T r;
  int s = (src_lane >= 0 && src_lane < WAVE) ? src_lane : cur().lane;
  memcpy(&r, w.slot[s], sizeof(T));
  wave_sync();
  return r;
}
void fiber_main();
void run_block(void (*entry)(void*), void* arg, dim3 grid, dim3 block, emu_uint3 bid, size_t shmem);

template <class F>
void launch(F&& body, dim3 grid, dim3 block, size_t shmem) {
  auto tramp = [](void* p) { (*static_cast<std::remove_reference_t<F>*>(p))(); };
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx)
        run_block(tramp, (void*)&body, grid, block, emu_uint3{bx, by, bz}, shmem);
}
}  // namespace hipemu

#define threadIdx (hipemu::cur().tid)
#define blockIdx (hipemu::g_bid)
#define blockDim (hipemu::g_bdim)
#define gridDim (hipemu::g_gdim)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
// static LDS lives in one named section so that the emulator can poison it per workgroup (OCCF_EMU_LDS_POISON)
#define __shared__ static __attribute__((section("emu_lds")))
#define __launch_bounds__(...)
#define __restrict__
#define OCCF_DYN_SMEM(name) char* name = hipemu::g_dyn_smem

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(shmem))

static inline void __syncthreads() { hipemu::block_sync(); }
template <class T>
static inline T __shfl_xor(T v, int m, int width = 64) {
  int l = hipemu::cur().lane;
  return hipemu::lane_exchange(v, (l & ~(width - 1)) | ((l ^ m) & (width - 1)));
}
template <class T>
static inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = hipemu::cur().lane;
  int s = (l & (width - 1)) + (int)d;
  return hipemu::lane_exchange(v, s < width ? (l & ~(width - 1)) + s : l);
}
template <class T>
static inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = hipemu::cur().lane;
  int s = (l & (width - 1)) - (int)d;
  return hipemu::lane_exchange(v, s >= 0 ? (l & ~(width - 1)) + s : l);
}
template <class T>
static inline T __shfl(T v, int src, int width = 64) {
  int l = hipemu::cur().lane;
  return hipemu::lane_exchange(v, (l & ~(width - 1)) | (src & (width - 1)));
}
static inline unsigned long long __ballot(int pred) {
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) {
    int p = hipemu::lane_exchange(pred, i);
    if (p && i < hipemu::g_blk->waves[hipemu::cur().wave].alive) m |= 1ull << i;
  }
  return m;
}
template <class T>
static inline T atomicAdd(T* p, T v) {
  T o = *p;
  *p = o + v;
  return o;
}
static inline int atomicMax(int* p, int v) {
  int o = *p;
  if (v > o) *p = v;
  return o;
}
static inline unsigned atomicMin(unsigned* p, unsigned v) {
  unsigned o = *p;
  if (v < o) *p = v;
  return o;
}
static inline unsigned atomicOr(unsigned* p, unsigned v) {
  unsigned o = *p;
  *p = o | v;
  return o;
}
#define __expf(x) expf(x)
#define __ffsll(x) __builtin_ffsll(x)
#define __popcll(x) __builtin_popcountll(x)
static inline float __fdividef(float a, float b) { return a / b; }

// ---- MFMA (collective over the 64-lane wave); layouts per cdna_hip_programming.md §3
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
typedef float emu_f32x4 __attribute__((ext_vector_type(4)));
typedef short emu_bf16x8 __attribute__((ext_vector_type(8)));

static inline float emu_bf16_to_f32(short s) {
  uint32_t u = ((uint32_t)(uint16_t)s) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// D[i][j] += sum_k A[i][k] B[k][j];  lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// lane l reg r receives row i=(r&3)+8*(r>>2)+4*(l>>5), col j=l&31.
static inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
  using namespace hipemu;
  Wave& w = g_blk->waves[cur().wave];
  int l = cur().lane;
  float ab[2] = {a, b};
  memcpy(w.slot[l], ab, 8);
  wave_sync();
  emu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, w.slot[k * 32 + i], 4);
      memcpy(&bv, w.slot[k * 32 + j] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
// 16x16x4 f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D reg r: row=(l>>4)*4+r, col=l&15.
static inline emu_f32x4 emu_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c) {
  using namespace hipemu;
  Wave& w = g_blk->waves[cur().wave];
  int l = cur().lane;
  float ab[2] = {a, b};
  memcpy(w.slot[l], ab, 8);
  wave_sync();
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int i = (l >> 4) * 4 + r, j = l & 15;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, w.slot[k * 16 + i], 4);
      memcpy(&bv, w.slot[k * 16 + j] + 4, 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
// 16x16x32 bf16: lane l supplies A[i=l&15][k=8*(l>>4)+e], B[k=8*(l>>4)+e][j=l&15], e=0..7; D reg r: row=4*(l>>4)+r, col=l&15.
static inline emu_f32x4 emu_mfma_f32_16x16x32_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x4 c) {
  using namespace hipemu;
  Wave& w = g_blk->waves[cur().wave];
  int l = cur().lane;
  memcpy(w.slot[l], &a, 16);
  memcpy(w.slot[l] + 16, &b, 16);
  wave_sync();
  emu_f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * (l >> 4) + r, j = l & 15;
    float acc = c[r];
    for (int g = 0; g < 4; ++g) {
      emu_bf16x8 av, bv;
      memcpy(&av, w.slot[g * 16 + i], 16);
      memcpy(&bv, w.slot[g * 16 + j] + 16, 16);
      for (int e = 0; e < 8; ++e) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
    }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
// fp16 <-> fp32 in software (round to nearest even, subnormals kept): the host build does not rely on F16C
static inline uint16_t emu_f32_to_f16(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7FFFFFFFu;
  if (u >= 0x7F800000u) return (uint16_t)(sign | (u > 0x7F800000u ? 0x7E00u : 0x7C00u));
  if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                 // rounds to >= 65520: infinity
  if (u < 0x38800000u) {                                                     // below the smallest normal half
    if (u < 0x33000000u) return (uint16_t)sign;                              // < 2^-25: zero
    const int shift = 126 - (int)(u >> 23);                                  // 14 .. 24
    const uint32_t m = (u & 0x7FFFFFu) | 0x800000u;
    const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
  }
  const uint32_t e = (u >> 23) - 112u, m = u & 0x7FFFFFu;
  uint32_t h = (e << 10) | (m >> 13);
  const uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
  return (uint16_t)(sign | h);
}
static inline float emu_f16_to_f32(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3FFu;
  uint32_t u;
  if (e == 31u) u = sign | 0x7F800000u | (m << 13);
  else if (e) u = sign | ((e + 112u) << 23) | (m << 13);
  else {
    const float v = (float)m * 5.9604644775390625e-8f;                       // m * 2^-24
    memcpy(&u, &v, 4);
    u |= sign;
  }
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// 32x32x16 f16: the bf16 layout with fp16 elements
static inline emu_f32x16 emu_mfma_f32_32x32x16_f16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
  using namespace hipemu;
  Wave& w = g_blk->waves[cur().wave];
  int l = cur().lane;
  memcpy(w.slot[l], &a, 16);
  memcpy(w.slot[l] + 16, &b, 16);
  wave_sync();
  emu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    float acc = c[r];
    for (int h = 0; h < 2; ++h) {
      emu_bf16x8 av, bv;
      memcpy(&av, w.slot[h * 32 + i], 16);
      memcpy(&bv, w.slot[h * 32 + j] + 16, 16);
      for (int e = 0; e < 8; ++e) acc += emu_f16_to_f32((uint16_t)av[e]) * emu_f16_to_f32((uint16_t)bv[e]);
    }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
// 32x32x16 bf16: lane l supplies A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e=0..7.
static inline emu_f32x16 emu_mfma_f32_32x32x16_bf16(emu_bf16x8 a, emu_bf16x8 b, emu_f32x16 c) {
  using namespace hipemu;
  Wave& w = g_blk->waves[cur().wave];
  int l = cur().lane;
  memcpy(w.slot[l], &a, 16);
  memcpy(w.slot[l] + 16, &b, 16);
  wave_sync();
  emu_f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    float acc = c[r];
    for (int h = 0; h < 2; ++h) {
      emu_bf16x8 av, bv;
      memcpy(&av, w.slot[h * 32 + i], 16);
      memcpy(&bv, w.slot[h * 32 + j] + 16, 16);
      for (int e = 0; e < 8; ++e) acc += emu_bf16_to_f32(av[e]) * emu_bf16_to_f32(bv[e]);
    }
    d[r] = acc;
  }
  wave_sync();
  return d;
}
