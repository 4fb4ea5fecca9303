// tests/emu/emu_core.cpp -- TEST INFRASTRUCTURE: the fiber scheduler of the
// lock-step wavefront emulator (see hip/hip_runtime.h next to this file).
#include <ucontext.h>

#include <algorithm>
#include <vector>

#include "hip/hip_runtime.h"

namespace emu {

Uint3 g_block_idx, g_block_dim, g_grid_dim;

namespace {

enum State { kRunnable, kAtWave, kAtBarrier, kAtSleep, kDone };

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  int tid = 0;   // thread in its workgroup
  int blk = 0;   // workgroup in the group that runs together (0 when workgroups run one by one)
  int gw = 0;    // wavefront in the group
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
std::vector<uint64_t> g_snap;    // [wavefronts of the group][64]
std::vector<uint64_t> g_active;  // [wavefronts of the group]
// Workgroups of a launch that run TOGETHER (set_concurrent_blocks): kernels whose workgroups wait
// for one another (ticket classes, mail-box granules) need their peers to be alive.  Only for
// kernels without __shared__ data -- that is one static object here, not one per workgroup.
unsigned g_concurrent = 1;
unsigned g_first_block = 0;  // linear index of the group's first workgroup

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
  fprintf(stderr, "emu: %s (first block of the group %u, wave %d)\n", what, g_first_block, wave);
  for (auto& f : g_fibers)
    if (wave < 0 || f.gw == wave)
      fprintf(stderr, "  block +%d tid %d state %d op %d\n", f.blk, f.tid, (int)f.state, f.op);
  abort();
}

void
run_blocks(unsigned nblocks, unsigned nthreads)
{
  const unsigned total = nblocks * nthreads;
  if (g_fibers.size() < total)
    g_fibers.resize(total);
  while (g_stacks.size() < total)
    g_stacks.push_back((char*)malloc(kStack));
  const int wpb = (int)((nthreads + 63) / 64);  // wavefronts per workgroup
  if (wpb > 32)
    die("more than 32 wavefronts per workgroup", -1);
  const int nwaves = wpb * (int)nblocks;
  g_snap.assign((size_t)nwaves * 64, 0);
  g_active.assign((size_t)nwaves, 0);
  for (unsigned b = 0; b < nblocks; b++)
    for (unsigned t = 0; t < nthreads; t++) {
      Fiber& f = g_fibers[b * nthreads + t];
      f.tid = (int)t;
      f.blk = (int)b;
      f.gw = (int)b * wpb + (int)(t / 64);
      f.state = kRunnable;
      f.stack = g_stacks[b * nthreads + t];
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack;
      f.ctx.uc_stack.ss_size = kStack;
      f.ctx.uc_link = &g_sched;
      makecontext(&f.ctx, fiber_main, 0);
    }
  unsigned long long idle_sweeps = 0;
  for (;;) {
    bool progress = false, all_done = true;
    for (int w = 0; w < nwaves; w++) {
      const unsigned blk = (unsigned)(w / wpb), wl = (unsigned)(w % wpb);
      const unsigned t0 = blk * nthreads + wl * 64;
      const unsigned t1 = blk * nthreads + (wl * 64 + 64 < nthreads ? wl * 64 + 64 : nthreads);
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
          g_snap[(size_t)w * 64 + (t - t0)] = 0;
          if (f.state == kAtWave) {
            g_snap[(size_t)w * 64 + (t - t0)] = f.val;
            g_active[w] |= 1ull << (t - t0);
            f.state = kRunnable;
          }
        }
        progress = true;
      }
    }
    // workgroup barriers: per workgroup
    for (unsigned b = 0; b < nblocks; b++) {
      int live = 0, at_bar = 0;
      for (unsigned t = b * nthreads; t < (b + 1) * nthreads; t++) {
        const Fiber& f = g_fibers[t];
        if (f.state != kDone) {
          all_done = false;
          live++;
          at_bar += f.state == kAtBarrier;
        }
      }
      if (live && at_bar == live) {
        for (unsigned t = b * nthreads; t < (b + 1) * nthreads; t++)
          if (g_fibers[t].state == kAtBarrier)
            g_fibers[t].state = kRunnable;
        progress = true;
      }
    }
    if (all_done)
      break;
    if (!progress) {
      bool sleeper = false;
      for (unsigned t = 0; t < total; t++)
        sleeper |= g_fibers[t].state == kAtSleep;
      if (!sleeper || ++idle_sweeps > 2000000ull)
        die(sleeper ? "livelock: threads only spin (waiting for a workgroup outside the group that runs together?)"
                    : "deadlock: no thread can make progress", -1);
    } else {
      idle_sweeps = 0;
    }
  }
}

}  // namespace

void
set_concurrent_blocks(unsigned n)
{
  g_concurrent = n ? n : 1;
}

// blockIdx of the calling thread
Uint3
cur_block_idx()
{
  const unsigned lin = g_first_block + (g_cur ? (unsigned)g_cur->blk : 0);
  Uint3 r;
  r.x = lin % g_grid_dim.x;
  r.y = (lin / g_grid_dim.x) % g_grid_dim.y;
  r.z = lin / (g_grid_dim.x * g_grid_dim.y);
  return r;
}

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
  *snap = &g_snap[(size_t)me->gw * 64];
  *active = g_active[me->gw];
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
  const unsigned nblocks = grid.x * grid.y * grid.z;
  // groups of g_concurrent workgroups at a time (1: one by one, the default); a group of more
  // than one needs whole wavefronts
  const unsigned group = nthreads % 64 == 0 ? g_concurrent : 1;
  for (unsigned first = 0; first < nblocks; first += group) {
    g_first_block = first;
    g_block_idx = {first % grid.x, (first / grid.x) % grid.y, first / (grid.x * grid.y)};
    run_blocks(std::min(group, nblocks - first), nthreads);
  }
  g_first_block = 0;
}

}  // namespace emu
