// tests/emu/emu_core.cpp -- TEST INFRASTRUCTURE: the fiber scheduler of the
// lock-step wavefront emulator (see hip/hip_runtime.h next to this file).
#include <ucontext.h>

#include <vector>

#include "hip/hip_runtime.h"

namespace emu {

Uint3 g_block_idx, g_block_dim, g_grid_dim;

namespace {

enum State { kRunnable, kAtWave, kAtBarrier, kAtSleep, kDone };

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  int tid = 0;
  State state = kDone;
  int op = 0;
  uint64_t val = 0;
};

constexpr size_t kStack = 1024 * 1024;
std::vector<Fiber> g_fibers;
std::vector<char*> g_stacks;
ucontext_t g_sched;
Fiber* g_cur = nullptr;
KernelThunk g_fn;
void* g_closure;
uint64_t g_snap[32][64];
uint64_t g_active[32];

void
fiber_main()
{
  g_fn(g_closure);
  g_cur->state = kDone;
  swapcontext(&g_cur->ctx, &g_sched);
}

void
yield_to_scheduler()
{
  swapcontext(&g_cur->ctx, &g_sched);
}

[[noreturn]] void
die(const char* what, int wave)
{
  fprintf(stderr, "emu: %s (block %u, wave %d)\n", what, g_block_idx.x, wave);
  for (auto& f : g_fibers)
    if (wave < 0 || f.tid / 64 == wave)
      fprintf(stderr, "  tid %d state %d op %d\n", f.tid, (int)f.state, f.op);
  abort();
}

void
run_block(unsigned nthreads)
{
  if (g_fibers.size() < nthreads)
    g_fibers.resize(nthreads);
  while (g_stacks.size() < nthreads)
    g_stacks.push_back((char*)malloc(kStack));
  for (unsigned t = 0; t < nthreads; t++) {
    Fiber& f = g_fibers[t];
    f.tid = (int)t;
    f.state = kRunnable;
    f.stack = g_stacks[t];
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = kStack;
    f.ctx.uc_link = &g_sched;
    makecontext(&f.ctx, fiber_main, 0);
  }
  const int nwaves = (int)((nthreads + 63) / 64);
  if (nwaves > 32)
    die("more than 32 wavefronts per workgroup", -1);
  unsigned long long idle_sweeps = 0;
  for (;;) {
    bool progress = false, all_done = true;
    for (int w = 0; w < nwaves; w++) {
      const unsigned t0 = w * 64, t1 = t0 + 64 < nthreads ? t0 + 64 : nthreads;
      for (unsigned t = t0; t < t1; t++) {
        Fiber& f = g_fibers[t];
        if (f.state == kAtSleep)
          f.state = kRunnable;
        if (f.state == kRunnable) {
          g_cur = &f;
          swapcontext(&g_sched, &f.ctx);
          g_cur = nullptr;
          if (f.state != kAtSleep)
            progress = true;
        }
      }
      // a wave collective completes once every live lane has arrived
      int live = 0, at_wave = 0, op = 0;
      bool same = true;
      for (unsigned t = t0; t < t1; t++) {
        const Fiber& f = g_fibers[t];
        if (f.state == kDone)
          continue;
        live++;
        if (f.state == kAtWave) {
          if (at_wave && f.op != op)
            same = false;
          op = f.op;
          at_wave++;
        }
      }
      if (live && at_wave == live) {
        if (!same)
          die("the lanes of a wavefront are at different collectives (divergent code around a wave operation)", w);
        g_active[w] = 0;
        for (unsigned t = t0; t < t1; t++) {
          Fiber& f = g_fibers[t];
          g_snap[w][t - t0] = 0;
          if (f.state == kAtWave) {
            g_snap[w][t - t0] = f.val;
            g_active[w] |= 1ull << (t - t0);
            f.state = kRunnable;
          }
        }
        progress = true;
      } else if (at_wave && at_wave < live) {
        // some lanes wait at a collective while others sit at a barrier: only
        // an error if nobody else can move (checked by the deadlock test below)
      }
    }
    int live = 0, at_bar = 0;
    for (unsigned t = 0; t < nthreads; t++) {
      const Fiber& f = g_fibers[t];
      if (f.state != kDone) {
        all_done = false;
        live++;
        at_bar += f.state == kAtBarrier;
      }
    }
    if (all_done)
      break;
    if (live && at_bar == live) {
      for (unsigned t = 0; t < nthreads; t++)
        if (g_fibers[t].state == kAtBarrier)
          g_fibers[t].state = kRunnable;
      progress = true;
    }
    if (!progress) {
      bool sleeper = false;
      for (unsigned t = 0; t < nthreads; t++)
        sleeper |= g_fibers[t].state == kAtSleep;
      if (!sleeper || ++idle_sweeps > 2000000ull)
        die(sleeper ? "livelock: threads only spin (waiting for a later workgroup?)"
                    : "deadlock: no thread can make progress", -1);
    } else {
      idle_sweeps = 0;
    }
  }
}

}  // namespace

Uint3
cur_thread_idx()
{
  const unsigned t = (unsigned)g_cur->tid;
  Uint3 r;
  r.x = t % g_block_dim.x;
  r.y = (t / g_block_dim.x) % g_block_dim.y;
  r.z = t / (g_block_dim.x * g_block_dim.y);
  return r;
}

int
cur_lane()
{
  return g_cur->tid & 63;
}

void
wave_exchange(int kind, uint64_t v, const uint64_t** snap, uint64_t* active)
{
  Fiber* me = g_cur;
  me->state = kAtWave;
  me->op = kind;
  me->val = v;
  yield_to_scheduler();
  g_cur = me;
  *snap = g_snap[me->tid / 64];
  *active = g_active[me->tid / 64];
}

void
block_barrier()
{
  Fiber* me = g_cur;
  me->state = kAtBarrier;
  me->op = kOpBarrier;
  yield_to_scheduler();
  g_cur = me;
}

void
sleep_yield()
{
  Fiber* me = g_cur;
  me->state = kAtSleep;
  me->op = kOpSleep;
  yield_to_scheduler();
  g_cur = me;
}

void
launch(dim3 grid, dim3 block, KernelThunk fn, void* closure)
{
  g_fn = fn;
  g_closure = closure;
  g_grid_dim = {grid.x, grid.y, grid.z};
  g_block_dim = {block.x, block.y, block.z};
  const unsigned nthreads = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
        g_block_idx = {bx, by, bz};
        run_block(nthreads);
      }
}

}  // namespace emu
