// TEST INFRASTRUCTURE ONLY — fiber runtime behind tests/emu/hip_emu.h (see its header).
#include "hip_emu.h"

#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
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
.size emu_switch, .-emu_switch
)");

namespace emu {

static constexpr size_t kStack = 256 * 1024;

struct BlockState {
  std::vector<Fiber> fibers;
  std::vector<WaveState> waves;
  unsigned n = 0, alive = 0, arrived = 0, gen = 0;
  void* main_sp = nullptr;
  const std::function<void()>* body = nullptr;
};

struct Worker {
  BlockState blk;
  Fiber* current = nullptr;
  emu_uint3 bidx{0, 0, 0}, bdim{1, 1, 1}, gdim{1, 1, 1};
  std::vector<char*> stacks;
};
static thread_local Worker tl_worker;

Fiber* cur() { return tl_worker.current; }
emu_uint3& block_idx() { return tl_worker.bidx; }
emu_uint3& block_dim() { return tl_worker.bdim; }
emu_uint3& grid_dim() { return tl_worker.gdim; }
WaveState* wave() { return &tl_worker.blk.waves[tl_worker.current->linear >> 6]; }
unsigned lane() { return tl_worker.current->linear & 63; }

static void switch_to(Fiber* from, Fiber* to) {
  tl_worker.current = to;
  emu_switch(&from->sp, to->sp);
}

static void yield_next() {
  Worker& w = tl_worker;
  Fiber* me = w.current;
  BlockState& b = w.blk;
  unsigned i = me->linear;
  for (unsigned step = 1; step <= b.n; ++step) {
    Fiber* f = &b.fibers[(i + step) % b.n];
    if (!f->done && f != me) {
      switch_to(me, f);
      return;
    }
  }
  // nobody else runnable: keep running
}

void syncthreads() {
  BlockState& b = tl_worker.blk;
  unsigned g = b.gen;
  if (++b.arrived >= b.alive) {
    b.arrived = 0;
    b.gen++;
    return;
  }
  while (b.gen == g) yield_next();
}

void wave_barrier() {
  WaveState* w = wave();
  unsigned g = w->gen;
  if (++w->arrived >= w->alive) {
    w->arrived = 0;
    w->gen++;
    return;
  }
  while (w->gen == g) yield_next();
}

static void fiber_main() {
  Worker& w = tl_worker;
  Fiber* me = w.current;
  (*w.blk.body)();
  // retire
  BlockState& b = w.blk;
  me->done = true;
  b.alive--;
  WaveState* wv = &b.waves[me->linear >> 6];
  wv->alive--;
  if (wv->alive > 0 && wv->arrived >= wv->alive) { wv->arrived = 0; wv->gen++; }
  if (b.alive > 0 && b.arrived >= b.alive) { b.arrived = 0; b.gen++; }
  if (b.alive == 0) {
    tl_worker.current = nullptr;
    void* dummy;
    emu_switch(&dummy, b.main_sp);
  }
  for (;;) {
    yield_next();
    // if we come back here, everybody else is done too
    void* dummy;
    emu_switch(&dummy, b.main_sp);
  }
}

static void run_block(Worker& w, dim3 block, const std::function<void()>& body) {
  BlockState& b = w.blk;
  unsigned n = block.x * block.y * block.z;
  b.n = n;
  b.alive = n;
  b.arrived = 0;
  b.gen = 0;
  b.body = &body;
  b.fibers.assign(n, Fiber());
  unsigned nw = (n + 63) / 64;
  b.waves.assign(nw, WaveState());
  while (w.stacks.size() < n) w.stacks.push_back((char*)aligned_alloc(64, kStack));
  for (unsigned i = 0; i < n; ++i) {
    Fiber& f = b.fibers[i];
    f.linear = i;
    f.tid.x = i % block.x;
    f.tid.y = (i / block.x) % block.y;
    f.tid.z = i / (block.x * block.y);
    f.blk = &b;
    f.stack = w.stacks[i];
    b.waves[i >> 6].alive++;
    uintptr_t top = ((uintptr_t)f.stack + kStack) & ~(uintptr_t)15;
    void** sp = (void**)top;
    *--sp = nullptr;             // fake return address of fiber_main (keeps ABI alignment)
    *--sp = (void*)&fiber_main;  // popped by `ret` in emu_switch
    for (int r = 0; r < 6; ++r) *--sp = nullptr;
    f.sp = (void*)sp;
  }
  w.current = &b.fibers[0];
  emu_switch(&b.main_sp, b.fibers[0].sp);
  w.current = nullptr;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  if (nblocks == 0) return;
  unsigned hw = std::thread::hardware_concurrency();
  unsigned nthreads = (unsigned)std::min<size_t>(nblocks, hw ? hw : 4);
  const char* env = getenv("VQ_EMU_THREADS");
  if (env) nthreads = (unsigned)std::max(1, atoi(env));
  if (nthreads > nblocks) nthreads = (unsigned)nblocks;
  std::atomic<size_t> next{0};
  auto work = [&]() {
    Worker& w = tl_worker;
    w.bdim = {block.x, block.y, block.z};
    w.gdim = {grid.x, grid.y, grid.z};
    for (;;) {
      size_t bi = next.fetch_add(1);
      if (bi >= nblocks) break;
      w.bidx.x = (unsigned)(bi % grid.x);
      w.bidx.y = (unsigned)((bi / grid.x) % grid.y);
      w.bidx.z = (unsigned)(bi / ((size_t)grid.x * grid.y));
      run_block(w, block, body);
    }
    for (char* s : w.stacks) free(s);
    w.stacks.clear();
  };
  if (nthreads <= 1) {
    std::thread t(work);  // fresh thread => fresh thread_local __shared__ arrays
    t.join();
  } else {
    std::vector<std::thread> ts;
    for (unsigned i = 0; i < nthreads; ++i) ts.emplace_back(work);
    for (auto& t : ts) t.join();
  }
}

}  // namespace emu
