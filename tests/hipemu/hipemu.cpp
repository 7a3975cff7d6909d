// TEST-ONLY: fiber scheduler behind tests/hipemu/hipemu.h (see the header).
#include <cstdlib>
#include "hipemu.h"
#include <cstring>
extern "C" char __start_emu_lds[] __attribute__((weak));
extern "C" char __stop_emu_lds[] __attribute__((weak));

asm(R"(
.text
.globl hipemu_swap
.type hipemu_swap, @function
hipemu_swap:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipemu_swap, .-hipemu_swap
)");

namespace hipemu {
Block* g_blk = nullptr;
emu_uint3 g_bid;
dim3 g_bdim, g_gdim;
char* g_dyn_smem = nullptr;
void (*g_entry)(void*) = nullptr;
void* g_entry_arg = nullptr;

static const size_t STACK = 96 * 1024;

void fiber_main() {
  g_entry(g_entry_arg);
  Block& b = *g_blk;
  Fiber& f = b.fibers[b.cur];
  f.done = true;
  // leaving threads no longer take part in rendezvous; release waiters if complete
  Wave& w = b.waves[f.wave];
  w.alive--;
  if (w.alive > 0 && w.arrived >= w.alive) {
    w.arrived = 0;
    w.gen++;
  }
  b.alive--;
  if (b.alive > 0 && b.arrived >= b.alive) {
    b.arrived = 0;
    b.gen++;
  }
  hipemu_swap(&f.sp, b.sched_sp);
  abort();   // a finished fiber is never resumed
}

void run_block(void (*entry)(void*), void* arg, dim3 grid, dim3 block, emu_uint3 bid, size_t shmem) {
  static std::vector<char*> stacks;
  const int nt = (int)(block.x * block.y * block.z);
  while ((int)stacks.size() < nt) stacks.push_back((char*)malloc(STACK));
  // OCCF_EMU_LDS_POISON=1: the LDS of every workgroup -- the dynamic block and every static __shared__ array (section
  // emu_lds, hipemu.h) -- starts as 0xFF bytes (NaN as f32 / bf16, -1 as an integer) instead of zeros / the previous
  // workgroup's values: on the GPU it holds whatever ran before, so a kernel that reads a word it has not written must
  // not pass the CPU suite because the emulation happened to hand it something benign
  static const char fill = (getenv("OCCF_EMU_LDS_POISON") && getenv("OCCF_EMU_LDS_POISON")[0] == '1') ? (char)0xFF : (char)0;
  std::vector<char> dyn(shmem + 64, fill);
  if (fill) memset(__start_emu_lds, 0xFF, (size_t)(__stop_emu_lds - __start_emu_lds));   // the static __shared__ arrays
  Block b;
  b.fibers.resize(nt);
  b.waves.resize((nt + WAVE - 1) / WAVE);
  b.alive = nt;
  g_blk = &b;
  g_bid = bid;
  g_bdim = block;
  g_gdim = grid;
  g_dyn_smem = (char*)(((uintptr_t)dyn.data() + 63) & ~(uintptr_t)63);
  g_entry = entry;
  g_entry_arg = arg;
  for (int t = 0; t < nt; ++t) {
    Fiber& f = b.fibers[t];
    f.tid.x = t % block.x;
    f.tid.y = (t / block.x) % block.y;
    f.tid.z = t / (block.x * block.y);
    f.lane = t % WAVE;
    f.wave = t / WAVE;
    b.waves[f.wave].alive++;
    // initial frame: 6 callee-saved slots, then the entry address `ret` jumps to, then a fake
    // return address so that the entry sees a call-aligned stack (rsp % 16 == 8)
    uintptr_t top = ((uintptr_t)stacks[t] + STACK) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;
    *--sp = (void*)fiber_main;
    for (int k = 0; k < 6; ++k) *--sp = nullptr;
    f.sp = (void*)sp;
  }
  int remaining = nt;
  long spins = 0;
  while (remaining > 0) {
    int progressed = 0;
    for (int t = 0; t < nt; ++t) {
      if (b.fibers[t].done) continue;
      b.cur = t;
      hipemu_swap(&b.sched_sp, b.fibers[t].sp);
      if (b.fibers[t].done) {
        remaining--;
        progressed++;
      }
    }
    if (++spins > 50000000L) {
      fprintf(stderr, "hipemu: deadlock suspected (block %u,%u,%u)\n", bid.x, bid.y, bid.z);
      abort();
    }
  }
  g_blk = nullptr;
}
}  // namespace hipemu
