// gpcc_attr_mi355.hip -- C ABI (include/gpcc_attr_mi355.h) over the gfx950
// RAHT kernels: context / workspace management and the launch sequences.
//
// Launch sequence of one batched transform (S slices, N points):
//
//   tree_count -> tree_scan -> tree_emit      all octree levels at once
//   [ascend_leaf, ascend_level x nlev]        only integer Haar / region QP
//   schedule                                  per-slice level plan
//   for li = top .. 0:
//     lossy encoder : level<kAnalyze> -> rdoq classify/carry/apply
//                     -> level<kSynth>
//     Haar encoder  : level<kFused>
//     decoder       : level<kSynth>
//   finish                                    duplicates, write-back
//
// Every kernel is launched with a grid that depends only on host-known
// sizes; node counts stay on the device, so a whole transform is enqueued
// without a single host synchronisation.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <dlfcn.h>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <cmath>
#include <vector>

#include "gpcc_attr_mi355.h"
#include "raht_common.hpp"
#include "raht_edges.hpp"
#include "raht_levels.hpp"
#include "raht_rdoq.hpp"
#include "raht_subnode.hpp"
#include "raht_pipe.hpp"
#include "raht_sweep.hpp"
#include "raht_tile.hpp"
#include "raht_tree.hpp"
#include "cx_driver.hpp"
#include "raht_inter_driver.hpp"
#include "lift_kernels.hpp"
#include "lod_kernels.hpp"
#include "lod_scalable.hpp"
#include "pred_kernels.hpp"
#include "morton_sort.hpp"
#include "residual_bins.hpp"
#include "recolour_kernels.hpp"

using namespace gpcc;

namespace {

thread_local std::string g_last_error;

int
fail(int code, const std::string& msg)
{
  g_last_error = msg;
  return code;
}

#define HIP_TRY(expr)                                                        \
  do {                                                                       \
    hipError_t e_ = (expr);                                                  \
    if (e_ != hipSuccess)                                                    \
      return fail(                                                           \
        e_ == hipErrorOutOfMemory ? GPCC_ERR_OUT_OF_MEMORY : GPCC_ERR_HIP,   \
        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
  } while (0)

// Device indices are 32-bit and attribute arrays are strided by up to 3.
constexpr int32_t kMaxPoints = GPCC_MAX_POINTS;
constexpr int kGridMax = 2048;  // 256 CUs x 8 workgroups of 256 threads
constexpr int kLevelGridMax = 1 << 16;
#ifndef GPCC_SUB_GRID
#define GPCC_SUB_GRID 1024
#endif
constexpr int kSubGrid = GPCC_SUB_GRID;  // sub-node kernel: 4 workgroups per CU, resident
#ifndef GPCC_SUB_BLOCKS_PER_WG
#define GPCC_SUB_BLOCKS_PER_WG 64   // parents of the level per workgroup when sizing its grid
#endif

int
grid_for(int64_t items, int per_block)
{
  int64_t g = (items + per_block - 1) / per_block;
  g = std::min<int64_t>(std::max<int64_t>(g, 1), kGridMax);
  return (int)((g + 7) / 8 * 8);  // xcd_chunk() wants a multiple of 8
}

// ---- guard bands (GPCC_GUARD=1 in the environment; debugging aid, off by default) ---------------------------
// Every device allocation of the library gets a 4 KB canary band in front of it and behind it, and every
// sub-allocation of a context's arena a 256-byte band behind it; the bands are filled with 0xA5 when they are made
// and compared when the arena is carved again, when a pool block is released, in gpcc_ctx_synchronize and in
// gpcc_ctx_destroy.  A kernel of the library that writes outside what it was given -- into the next
// sub-allocation or into a neighbouring allocation of the process (torch's) -- then stops the process with a
// message that names the allocation, instead of surfacing later as somebody else's fault.
constexpr size_t kGuardOuter = 4096;
constexpr size_t kGuardInner = 256;
constexpr unsigned char kGuardByte = 0xA5;

bool
guard_mode()
{
  static const bool on = [] {
    const char* e = getenv("GPCC_GUARD");
    return e && e[0] == '1';
  }();
  return on;
}

struct GuardedBlock {
  char* raw;       // what hipMalloc returned
  size_t bytes;    // the user's bytes (bands excluded)
  const char* tag;
};
std::mutex g_guard_mu;
std::map<void*, GuardedBlock> g_guard_blocks;  // by user pointer
std::atomic<unsigned long long> g_guard_checks{0};  // bands compared so far (gpcc_debug_guard_checks)

[[noreturn]] void
guard_violation(const char* what, const char* tag, size_t index, size_t offset, const unsigned char* got, size_t len)
{
  size_t first = 0, count = 0;
  for (size_t i = 0; i < len; i++)
    if (got[i] != kGuardByte) {
      if (!count)
        first = i;
      count++;
    }
  fprintf(
    stderr,
    "gpcc: GUARD BAND OVERWRITTEN (%s, allocation '%s' #%zu, band at byte offset %zu): %zu of %zu bytes changed, "
    "first at +%zu = 0x%02x\n",
    what, tag, index, offset, count, len, first, got[first]);
  fflush(stderr);
  abort();
}

void
guard_compare(const void* dev, size_t len, const char* what, const char* tag, size_t index, size_t offset)
{
  static thread_local std::vector<unsigned char> buf;
  buf.resize(len);
  if (hipMemcpy(buf.data(), dev, len, hipMemcpyDeviceToHost) != hipSuccess) {
    fprintf(stderr, "gpcc: guard band of '%s' could not be read (%s)\n", tag, what);
    abort();
  }
  g_guard_checks++;
  for (size_t i = 0; i < len; i++)
    if (buf[i] != kGuardByte)
      guard_violation(what, tag, index, offset, buf.data(), len);
}

// hipMalloc / hipFree of the library (every persistent device allocation goes through these two)
hipError_t
guarded_malloc(void** out, size_t bytes, const char* tag)
{
  if (!guard_mode())
    return hipMalloc(out, bytes);
  char* raw = nullptr;
  const size_t padded = (bytes + 255) & ~size_t(255);
  hipError_t e = hipMalloc((void**)&raw, padded + 2 * kGuardOuter);
  if (e != hipSuccess)
    return e;
  e = hipMemset(raw, kGuardByte, kGuardOuter);
  if (e == hipSuccess)
    e = hipMemset(raw + kGuardOuter + padded, kGuardByte, kGuardOuter);
  if (e != hipSuccess) {
    hipFree(raw);
    return e;
  }
  *out = raw + kGuardOuter;
  std::lock_guard<std::mutex> lock(g_guard_mu);
  g_guard_blocks[*out] = {raw, padded, tag};
  return hipSuccess;
}

void
guarded_check(void* p, const char* what)
{
  if (!guard_mode() || !p)
    return;
  GuardedBlock b;
  {
    std::lock_guard<std::mutex> lock(g_guard_mu);
    auto it = g_guard_blocks.find(p);
    if (it == g_guard_blocks.end())
      return;
    b = it->second;
  }
  guard_compare(b.raw, kGuardOuter, what, b.tag, 0, 0);
  guard_compare(b.raw + kGuardOuter + b.bytes, kGuardOuter, what, b.tag, 1, kGuardOuter + b.bytes);
}

hipError_t
guarded_free(void* p)
{
  if (!guard_mode() || !p)
    return hipFree(p);
  guarded_check(p, "free");
  char* raw = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_guard_mu);
    auto it = g_guard_blocks.find(p);
    if (it != g_guard_blocks.end()) {
      raw = it->second.raw;
      g_guard_blocks.erase(it);
    }
  }
  return hipFree(raw ? (void*)raw : p);
}

struct Arena {
  char* base = nullptr;
  size_t cap = 0;
  size_t used = 0;
  // guard mode: where the 256-byte bands of the current carving lie (shared by the copies of an arena that
  // carve on behind a region, e.g. the lifting scratch behind the LoD workspace), and the stream they are filled on
  std::vector<size_t>* bands = nullptr;
  hipStream_t gstream = nullptr;

  // the band behind a sub-allocation that ends at `used` (called with `used` already advanced)
  void band()
  {
    if (!guard_mode())
      return;
    // (a carving that has run past the arena is reported by its caller's used > cap test: no band is
    // written outside the allocation, and a failed memset leaves a band that the next check reports)
    if (base && bands && used + kGuardInner <= cap) {
      if (hipMemsetAsync(base + used, kGuardByte, kGuardInner, gstream) == hipSuccess)
        bands->push_back(used);
    }
    used += kGuardInner;
  }
  // compare and forget the bands of the carving that ends here (the stream is idle afterwards)
  void check_bands(const char* what, bool keep = false)
  {
    if (!guard_mode() || !base || !bands)
      return;
    if (!bands->empty()) {
      hipStreamSynchronize(gstream);
      for (size_t i = 0; i < bands->size(); i++)
        guard_compare(base + (*bands)[i], kGuardInner, what, "arena", i, (*bands)[i]);
      if (!keep)
        bands->clear();
    }
  }
  void reset()
  {
    check_bands("arena carved again");
    used = 0;
  }
  template<typename T>
  T* take(size_t count)
  {
    size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    T* p = base ? reinterpret_cast<T*>(base + used) : nullptr;
    used += bytes;
    band();
    return p;
  }
};

}  // namespace

struct gpcc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  Arena arena;
  std::vector<size_t> arena_bands;  // guard mode (GPCC_GUARD=1): see Arena
  // allocation events since the context was made (gpcc_debug_alloc_events): arena (re)allocations, pool misses,
  // pinned staging (re)allocations -- a step that takes tens of milliseconds longer than its neighbours either shows up here or
  // was not the library's doing (tools/archive/r05_stall_probe.py)
  long long alloc_events[4] = {0, 0, 0, 0};
  int morton_bits = 0;  // hint for the device tier, 0 = unknown
  SharedLut* d_lut = nullptr;  // small-weight tables, built once
  int32_t* h_error = nullptr;  // pinned: copy of the sticky device-side error word
  int32_t* d_error = nullptr;  // device: set by a kernel whose bounded wait expired, cleared
                               // only when the error has been reported (check_device_error)
  TreeStats* h_stats = nullptr;   // pinned: what schedule_kernel tells the host about the tree
  CxLevelTab* h_cxtab = nullptr;  // pinned: blocks / real children per level (compact level pass)
  double* d_log2 = nullptr;       // log2 of 0 .. 2^20 from the host's libm (predicting encoder's rate model)
  void* sweep_mem = nullptr;      // records of the coarse-level sweep (raht_sweep.hpp), grown on demand
  size_t sweep_cap = 0;
  int pred_passes = 0;            // passes the last predicting encode with direct predictors took
  int64_t pred_pass_stats[4] = {0, 0, 0, 0};  // slices, passes, most passes, declined at the limit
  hipEvent_t ev_stats = nullptr;  // recorded behind schedule_kernel
  // recolour: the second tree's build runs on a stream of its own (recolour_kdtree.hpp KdLevelLoop)
  hipStream_t kd_stream = nullptr;
  hipEvent_t kd_event = nullptr;
  hipEvent_t inter_event = nullptr;  // inter-frame RAHT: the second candidate's stream joins the first
  int32_t* h_kd = nullptr;        // pinned: 2 x 4 counters
  // what the entries did since the context was created (gpcc_ctx_stats)
  gpcc_ctx_stats_t stats{};
  // device buffers of the host tiers, kept between calls (pool_malloc)
  struct PoolBlock {
    void* ptr;
    size_t cap;
    bool used;
  };
  std::vector<PoolBlock> pool;
  // host staging for the host tier
  void* h_pinned = nullptr;
  size_t h_pinned_cap = 0;
  // compact level pass: its own staging, two slots taken in turn (a call returns only after
  // the event behind its uploads, so the slot of the call before last is free: no wait for
  // the stream when a call starts)
  void* h_cx_stage = nullptr;
  size_t h_cx_stage_cap = 0;
  // small parameter blocks / tables / result words of synchronous host-tier calls (inter-frame RAHT): a pinned
  // block with a bump cursor, so that not even the small copies start from or land in the caller's pages or the stack
  char* h_small = nullptr;
  size_t h_small_used = 0;
  int cx_stage_flip = 0;
  // downloads into the caller's pageable memory go through this pinned buffer (d2h_user)
  void* h_bounce = nullptr;
  hipEvent_t ev_bounce[2] = {nullptr, nullptr};  // recorded behind the last upload out of each half
  bool bounce_busy[2] = {false, false};
  int bounce_turn = 0;
  // arithmetic back end of the dependency kernels (raht_arith.hpp): doubles where they are exact
  // (GPCC_F64=0 in the environment or gpcc_ctx_set_fast_arith(ctx, 0): int64 everywhere);
  // force_exact: the host tier's second attempt after GPCC_ERR_RANGE
  bool fast_arith = [] {
    const char* e = getenv("GPCC_F64");
    return !(e && e[0] == '0');
  }();
  bool force_exact = false;
  // profiling
  bool profiling = false;
  struct Span {
    const char* name;
    hipEvent_t a, b;
  };
  std::vector<Span> spans;
  std::vector<hipEvent_t> event_pool;
  std::vector<std::pair<const char*, std::pair<double, int>>> times;
  // device tier of the LoD build / lifting coder: slices of a batch run
  // concurrently, each on a lane (own stream + workspace); lane 0 is this context
  std::vector<gpcc_ctx*> lanes;
  hipEvent_t ev_lanes = nullptr;
  int lod_grid = 768;  // workgroups of the sub-sampling kernel (fewer while lanes share the device)
};

namespace {

// guard mode: every band the context owns (the stream is idle: the caller has synchronised)
void
guard_check_context(gpcc_ctx* ctx, const char* what)
{
  ctx->arena.check_bands(what, /*keep*/ true);
  guarded_check(ctx->arena.base, what);
  for (auto& b : ctx->pool)
    guarded_check(b.ptr, what);
  guarded_check(ctx->d_lut, what);
  guarded_check(ctx->d_error, what);
  guarded_check(ctx->d_log2, what);
  guarded_check(ctx->sweep_mem, what);
  for (gpcc_ctx* lane : ctx->lanes) {
    hipStreamSynchronize(lane->stream);
    guard_check_context(lane, what);
  }
}

// "<kernel>@<level>" names for per-level timings (GPCC_PROFILE_LEVELS=1); the
// strings live for the life of the process
const char*
level_name(const char* base, int li)
{
  static std::map<std::string, std::string> pool;
  static const bool on = [] {
    const char* e = getenv("GPCC_PROFILE_LEVELS");
    return e && e[0] == '1';
  }();
  if (!on)
    return base;
  char buf[96];
  snprintf(buf, sizeof buf, "%s@%02d", base, li);
  static std::mutex mu;  // lanes of the device tier call this from their own threads
  std::lock_guard<std::mutex> lock(mu);
  auto it = pool.emplace(buf, buf).first;
  return it->second.c_str();
}

struct Timer {
  gpcc_ctx* ctx;
  hipEvent_t a = nullptr, b = nullptr;
  const char* name;
  Timer(gpcc_ctx* c, const char* n) : ctx(c), name(n)
  {
    if (!ctx->profiling)
      return;
    auto get = [&]() {
      hipEvent_t e;
      if (!ctx->event_pool.empty()) {
        e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
      } else {
        hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    hipEventRecord(a, ctx->stream);
  }
  ~Timer()
  {
    if (!ctx->profiling)
      return;
    hipEventRecord(b, ctx->stream);
    ctx->spans.push_back({name, a, b});
  }
};

// ---- workspace plan ---------------------------------------------------

struct Plan {
  int n = 0, s = 0, c = 0, nlev = 0;
  bool encoder = false, haar = false, has_qp = false, lossy = false;
  std::vector<int64_t> cap;  // node capacity per level

  TreeView tv{};
  int32_t* pt_off = nullptr;
  uint32_t* tile_cnt = nullptr;
  int32_t* tile_attr = nullptr;
  int32_t* attr_prefix = nullptr;
  int32_t** haar_lf_tab = nullptr;
  int32_t** asc_qp_tab = nullptr;
  int32_t* dup_hf = nullptr;
  int64_t* rec[2] = {nullptr, nullptr};
  int64_t* rec_us[2] = {nullptr, nullptr};
  int32_t* nneigh[2] = {nullptr, nullptr};
  int32_t* dqp[2] = {nullptr, nullptr};
  SliceSched* sched = nullptr;
  gpcc_raht_params* params = nullptr;
  uint32_t* desc = nullptr;
  int64_t* ptrans = nullptr;
  int32_t* rtile_base = nullptr;
  unsigned long long* rtile_state = nullptr;
  int32_t* slice_l = nullptr;
  int32_t* worklist = nullptr;
  int32_t* work_count = nullptr;  // [kMaxLevels] then ticket[kMaxLevels*8], error[1] (one memset)
  unsigned long long* scan_state = nullptr;
  uint8_t* pocc = nullptr;
  uint32_t* mbox = nullptr;
  unsigned long long* rdoq_state = nullptr;
  bool sub = false;
  bool f64 = false;  // the sub-node kernels in ArithF64 (decided by dev_transform)
  // decoder with sub-node prediction: one launch over all levels (raht_pipe.hpp)
  bool pipe = false;
  int num_rtiles = 0;
  std::vector<int32_t*> haar_lf, asc_qp;
  LinkView lv{};  // neighbour links of the sub-node kernels (raht_links.hpp)
  bool links = false;
};

// Carve the workspace.  With arena.base == nullptr this only measures.
void
carve(Arena& ar, Plan& pl)
{
  ar.reset();
  const int n = pl.n, s = pl.s, c = pl.c, nlev = pl.nlev;
  pl.pt_off = ar.take<int32_t>(s + 1);
  pl.cap.assign(nlev, 0);
  for (int li = 0; li < nlev; li++) {
    // a level cannot hold more nodes than points, nor more than a full
    // octree below the one-node-per-slice top level
    int64_t cap = n;
    const int up = nlev - 1 - li;
    if (up < 11) {
      int64_t full = (int64_t)s << (3 * up);
      cap = std::min<int64_t>(cap, full);
    }
    pl.cap[li] = cap;
    pl.tv.cap[li] = (int32_t)cap;
    pl.tv.key[li] = ar.take<int64_t>(cap + 1);
    pl.tv.fp[li] = ar.take<int32_t>(cap + 2);
    pl.tv.fc[li] = ar.take<int32_t>(cap + 2);
    pl.tv.soff[li] = ar.take<int32_t>(s + 1);
  }
  pl.tv.nlev = nlev;
  pl.tv.num_slices = s;
  pl.tv.n_total = n;
  pl.tv.num_tiles = (n + kTilePoints - 1) / kTilePoints;
  pl.tv.pt_off = pl.pt_off;
  pl.tile_cnt = ar.take<uint32_t>((size_t)pl.tv.num_tiles * nlev);
  pl.tile_attr = ar.take<int32_t>((size_t)pl.tv.num_tiles * c);
  pl.sched = ar.take<SliceSched>(s);
  pl.worklist = ar.take<int32_t>((size_t)n + 1);
  pl.work_count = ar.take<int32_t>(kMaxLevels * 9);
  pl.scan_state = ar.take<unsigned long long>(1024);
  pl.pocc = pl.sub ? ar.take<uint8_t>((size_t)n + 1) : nullptr;
  pl.mbox = pl.sub ? ar.take<uint32_t>((size_t)n * c * 4) : nullptr;
  pl.rdoq_state = (pl.sub && pl.lossy) ? ar.take<unsigned long long>((size_t)n + 1) : nullptr;
  pl.params = ar.take<gpcc_raht_params>(1);
  for (int i = 0; i < 2; i++) {
    pl.rec[i] = ar.take<int64_t>((size_t)n * c);
    pl.rec_us[i] = ar.take<int64_t>((size_t)n * c);
    pl.nneigh[i] = ar.take<int32_t>(n);
    pl.dqp[i] = pl.has_qp ? ar.take<int32_t>((size_t)n * 2) : nullptr;
  }
  pl.attr_prefix = nullptr;
  pl.dup_hf = nullptr;
  pl.haar_lf.assign(nlev, nullptr);
  pl.asc_qp.assign(nlev, nullptr);
  pl.haar_lf_tab = nullptr;
  pl.asc_qp_tab = nullptr;
  if (pl.encoder && !pl.haar)
    pl.attr_prefix = ar.take<int32_t>(((size_t)n + 1) * c);
  if (pl.encoder && pl.haar) {
    pl.dup_hf = ar.take<int32_t>((size_t)n * c);
    pl.haar_lf_tab = ar.take<int32_t*>(nlev);
    for (int li = 0; li < nlev; li++)
      pl.haar_lf[li] = ar.take<int32_t>((size_t)(pl.cap[li] + 1) * c);
  }
  if (pl.has_qp) {
    pl.asc_qp_tab = ar.take<int32_t*>(nlev);
    for (int li = 0; li < nlev; li++)
      pl.asc_qp[li] = ar.take<int32_t>((size_t)(pl.cap[li] + 1) * 2);
  }
  pl.desc = nullptr;
  pl.ptrans = nullptr;
  if (pl.lossy) {
    pl.desc = ar.take<uint32_t>(n);
    pl.ptrans = pl.sub ? nullptr : ar.take<int64_t>((size_t)n * c);
    pl.rtile_base = ar.take<int32_t>(s + 1);
    pl.rtile_state = ar.take<unsigned long long>(pl.num_rtiles + 1);
    pl.slice_l = ar.take<int32_t>(2 * (size_t)s);  // sub-node path: [level parity][S]
  }
  pl.links = pl.sub && !pl.pipe && links_enabled();
  if (pl.links)
    link_carve([&](size_t bytes) { return ar.take<char>(bytes); }, pl.lv, pl.tv, n, s, nlev);
}

// Device buffers of the host tiers (uploads, downloads, scratch): a caching
// allocator instead of hipMalloc / hipFree per call -- in a steady state
// (slice after slice of similar size) no call allocates.  Reuse is safe without
// a synchronisation: all work of a context is ordered on its one stream.
hipError_t
pool_malloc(gpcc_ctx* ctx, void** out, size_t bytes)
{
  bytes = std::max<size_t>(bytes, 256);
  gpcc_ctx::PoolBlock* best = nullptr;
  for (auto& b : ctx->pool)
    if (!b.used && b.cap >= bytes && b.cap <= 2 * bytes + (1 << 20)
        && (!best || b.cap < best->cap))
      best = &b;
  if (best) {
    best->used = true;
    *out = best->ptr;
    return hipSuccess;
  }
  // a miss: the sizes have changed, drop what is idle before growing
  ctx->alloc_events[1]++;
  hipStreamSynchronize(ctx->stream);
  for (size_t i = 0; i < ctx->pool.size();) {
    if (!ctx->pool[i].used) {
      guarded_free(ctx->pool[i].ptr);
      ctx->pool.erase(ctx->pool.begin() + i);
    } else {
      i++;
    }
  }
  const size_t cap = bytes + bytes / 8;
  void* p = nullptr;
  hipError_t e = guarded_malloc(&p, cap, "pool block");
  if (e != hipSuccess)
    return e;
  ctx->pool.push_back({p, cap, true});
  *out = p;
  return hipSuccess;
}

void
pool_free(gpcc_ctx* ctx, void* p)
{
  if (!p)
    return;
  for (auto& b : ctx->pool)
    if (b.ptr == p) {
      if (guard_mode()) {
        hipStreamSynchronize(ctx->stream);
        guarded_check(p, "pool block released");
      }
      b.used = false;
    }
}

// A result for the caller's (pageable) memory.  An asynchronous copy straight into pageable
// memory makes the runtime pin the caller's pages on the fly, and it keeps such pins: a range it
// has pinned READ-ONLY once, as the source of an upload, is not pinned again when the same
// addresses later receive a download -- "Memory access fault ... Write access to a read-only
// page" in the middle of a long-running host process (the GPU test tier in one pytest process:
// numpy arrays of earlier tests freed, their heap addresses reused).  So large results are
// copied into a pinned buffer of the context and from there by the CPU; small ones take the
// runtime's own staging path.
constexpr size_t kBounceBytes = (size_t)8 << 20;
constexpr size_t kBounceMin = (size_t)64 << 10;

// ... and an input from the caller's memory: through the same buffer, so that the runtime never
// pins a page of the caller (it is those pins, read-only, that a later download trips over --
// this library's or anybody else's in the process).  The buffer is free again when the call
// returns.
// The two halves of the buffer are tracked by events: a half is waited for only when it is taken
// again (by the next chunk, the next upload or a download), so consecutive uploads of a call overlap
// with each other's DMA and with the kernels enqueued between them -- until round 3 every upload ended
// with a full stream synchronisation, which also waited for every kernel before it.
hipError_t
bounce_ready(gpcc_ctx* ctx)
{
  if (!ctx->h_bounce) {
    hipError_t e = hipHostMalloc(&ctx->h_bounce, 2 * kBounceBytes);
    if (e != hipSuccess)
      return e;
    for (int h = 0; h < 2; h++) {
      e = hipEventCreateWithFlags(&ctx->ev_bounce[h], hipEventDisableTiming);
      if (e != hipSuccess)
        return e;
      ctx->bounce_busy[h] = false;
    }
  }
  return hipSuccess;
}

// the half's last upload has left it
hipError_t
bounce_take(gpcc_ctx* ctx, int half)
{
  if (!ctx->bounce_busy[half])
    return hipSuccess;
  ctx->bounce_busy[half] = false;
  return hipEventSynchronize(ctx->ev_bounce[half]);
}

hipError_t
h2d_user(gpcc_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
  if (bytes < kBounceMin)
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
  hipError_t e = bounce_ready(ctx);
  if (e != hipSuccess)
    return e;
  size_t off = 0;
  while (off < bytes) {
    const int half = ctx->bounce_turn;
    ctx->bounce_turn ^= 1;
    e = bounce_take(ctx, half);
    if (e != hipSuccess)
      return e;
    const size_t chunk = std::min(kBounceBytes, bytes - off);
    char* buf = (char*)ctx->h_bounce + (size_t)half * kBounceBytes;
    memcpy(buf, (const char*)src + off, chunk);
    e = hipMemcpyAsync((char*)dst + off, buf, chunk, hipMemcpyHostToDevice, st);
    if (e != hipSuccess)
      return e;
    e = hipEventRecord(ctx->ev_bounce[half], st);
    if (e != hipSuccess)
      return e;
    ctx->bounce_busy[half] = true;
    off += chunk;
  }
  return hipSuccess;
}

hipError_t
d2h_user(gpcc_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
  if (bytes < kBounceMin)
    return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st);
  hipError_t e = bounce_ready(ctx);
  if (e != hipSuccess)
    return e;
  // two halves in turn: the CPU empties one while the next chunk arrives in the other
  size_t off = 0, prev_off = 0, prev_bytes = 0;
  int half = 0;
  while (off < bytes || prev_bytes) {
    size_t chunk = 0;
    char* buf = (char*)ctx->h_bounce + (size_t)half * kBounceBytes;
    if (off < bytes) {
      chunk = std::min(kBounceBytes, bytes - off);
      e = bounce_take(ctx, half);  // (an upload may still be reading this half)
      if (e != hipSuccess)
        return e;
      e = hipMemcpyAsync(buf, (const char*)src + off, chunk, hipMemcpyDeviceToHost, st);
      if (e != hipSuccess)
        return e;
    }
    if (prev_bytes)  // (its copy was waited for at the end of the previous turn)
      memcpy((char*)dst + prev_off, (char*)ctx->h_bounce + (size_t)(half ^ 1) * kBounceBytes, prev_bytes);
    if (chunk) {
      e = hipStreamSynchronize(st);
      if (e != hipSuccess)
        return e;
    }
    prev_off = off;
    prev_bytes = chunk;
    off += chunk;
    half ^= 1;
  }
  return hipSuccess;
}

// Small transfers of a synchronous host-tier call through the context's pinned block (the discipline of
// dev_transform's h_cx_stage: the runtime never touches the caller's pages or the stack).  small_reset() at the
// start of the call; a download's slot is read after the call's stream synchronisation.
constexpr size_t kSmallStage = 64 * 1024;

hipError_t
small_reset(gpcc_ctx* ctx)
{
  if (!ctx->h_small) {
    hipError_t e = hipHostMalloc((void**)&ctx->h_small, kSmallStage);
    if (e != hipSuccess)
      return e;
  }
  ctx->h_small_used = 0;
  return hipSuccess;
}

void*
small_slot(gpcc_ctx* ctx, size_t bytes)
{
  const size_t b = (bytes + 63) & ~size_t(63);
  if (!ctx->h_small || ctx->h_small_used + b > kSmallStage)
    return nullptr;
  void* p = ctx->h_small + ctx->h_small_used;
  ctx->h_small_used += b;
  return p;
}

hipError_t
small_h2d(gpcc_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st)
{
  void* slot = small_slot(ctx, bytes);
  if (!slot)
    return hipErrorOutOfMemory;
  memcpy(slot, src, bytes);
  return hipMemcpyAsync(dst, slot, bytes, hipMemcpyHostToDevice, st);
}

// the pinned slot the download lands in (valid once the stream has been synchronised)
hipError_t
small_d2h(gpcc_ctx* ctx, void** slot_out, const void* src, size_t bytes, hipStream_t st)
{
  void* slot = small_slot(ctx, bytes);
  if (!slot)
    return hipErrorOutOfMemory;
  *slot_out = slot;
  return hipMemcpyAsync(slot, src, bytes, hipMemcpyDeviceToHost, st);
}

int
ensure_arena(gpcc_ctx* ctx, size_t bytes)
{
  if (guard_mode()) {
    // the previous call's carving is over: its bands are compared before the workspace is laid out again
    ctx->arena.bands = &ctx->arena_bands;
    ctx->arena.gstream = ctx->stream;
    ctx->arena.check_bands("next call");
    if (ctx->arena.base) {
      HIP_TRY(hipStreamSynchronize(ctx->stream));
      guarded_check(ctx->arena.base, "next call");
    }
  }
  if (ctx->arena.cap >= bytes)
    return GPCC_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->arena.base)
    HIP_TRY(guarded_free(ctx->arena.base));
  ctx->arena.base = nullptr;
  ctx->arena.cap = 0;
  size_t want = bytes + bytes / 8;
  HIP_TRY(guarded_malloc((void**)&ctx->arena.base, want, "arena"));
  ctx->arena.cap = want;
  ctx->alloc_events[0]++;
  return GPCC_OK;
}

// the record block of the coarse-level sweep (raht_sweep.hpp): one allocation per context, grown on demand
// (its size follows the tree, not the point count: the pool's best-fit reuse would miss it)
int
ensure_sweep_mem(gpcc_ctx* ctx, size_t bytes)
{
  if (ctx->sweep_cap >= bytes)
    return GPCC_OK;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (ctx->sweep_mem)
    HIP_TRY(guarded_free(ctx->sweep_mem));
  ctx->sweep_mem = nullptr;
  ctx->sweep_cap = 0;
  const size_t want = bytes + bytes / 4;
  HIP_TRY(guarded_malloc(&ctx->sweep_mem, want, "sweep records"));
  ctx->sweep_cap = want;
  ctx->alloc_events[1]++;
  return GPCC_OK;
}

// log2 of every integer the rate estimates can ask for (0 .. 2^20), from THIS host's libm -- the
// reference's own log2: once per context (predicting encoder's rate model, inter-frame RAHT's per-layer decision)
int
ensure_log2(gpcc_ctx* ctx)
{
  if (!ctx->d_log2) {
    std::vector<double> tab(((size_t)1 << 20) + 1);
    for (size_t v = 0; v < tab.size(); v++)
      tab[v] = log2((double)v);
    HIP_TRY(guarded_malloc((void**)&ctx->d_log2, tab.size() * sizeof(double), "log2 table"));
    HIP_TRY(hipMemcpy(ctx->d_log2, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  return GPCC_OK;
}

int
check_params(const gpcc_raht_params* p, int c, bool encoder)
{
  if (!p)
    return fail(GPCC_ERR_INVALID_ARG, "params is null");
  if (c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "attribute count must be 1..3");
  if (p->num_qp_layers < 1 || p->num_qp_layers > GPCC_MAX_QP_LAYERS)
    return fail(GPCC_ERR_INVALID_ARG, "num_qp_layers out of range");
  if (p->num_ac_qp_layers < 0 || p->num_ac_qp_layers > GPCC_MAX_AC_QP_LAYERS)
    return fail(GPCC_ERR_INVALID_ARG, "num_ac_qp_layers out of range");
  return GPCC_OK;
}

template<int C>
int
launch_transform(
  gpcc_ctx* ctx, Plan& pl, const gpcc_raht_params* hp, bool encoder,
  const int64_t* offsets, const int64_t* d_morton, const int32_t* d_qp_off,
  int32_t* d_attrs, int32_t* d_coeffs)
{
  hipStream_t st = ctx->stream;
  const int n = pl.n, s = pl.s, nlev = pl.nlev;

  // ---- small host -> device tables (pinned staging, async) ------------
  std::vector<int32_t> h_off(s + 1), h_rbase(s + 1);
  for (int i = 0; i <= s; i++)
    h_off[i] = (int32_t)offsets[i];
  h_rbase[0] = 0;
  for (int i = 0; i < s; i++)
    h_rbase[i + 1] =
      h_rbase[i] + (h_off[i + 1] - h_off[i] + kRdoqTile - 1) / kRdoqTile;

  size_t stage_bytes = sizeof(gpcc_raht_params) + 2 * (s + 1) * sizeof(int32_t)
    + 2 * nlev * sizeof(void*) + 64;
  if (ctx->h_pinned_cap < stage_bytes) {
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->h_pinned)
      HIP_TRY(hipHostFree(ctx->h_pinned));
    HIP_TRY(hipHostMalloc(&ctx->h_pinned, stage_bytes * 2));
    ctx->alloc_events[3]++;
    ctx->h_pinned_cap = stage_bytes * 2;
  } else {
    // the previous call's async copies read this buffer
    HIP_TRY(hipStreamSynchronize(st));
  }
  char* hp_base = (char*)ctx->h_pinned;
  size_t o = 0;
  auto stage = [&](const void* src, size_t bytes, void* dst) -> hipError_t {
    memcpy(hp_base + o, src, bytes);
    hipError_t e =
      hipMemcpyAsync(dst, hp_base + o, bytes, hipMemcpyHostToDevice, st);
    o += (bytes + 15) & ~size_t(15);
    return e;
  };
  HIP_TRY(stage(hp, sizeof(*hp), pl.params));
  HIP_TRY(stage(h_off.data(), (s + 1) * sizeof(int32_t), pl.pt_off));
  if (pl.lossy)
    HIP_TRY(stage(h_rbase.data(), (s + 1) * sizeof(int32_t), pl.rtile_base));
  if (pl.haar_lf_tab)
    HIP_TRY(stage(pl.haar_lf.data(), nlev * sizeof(void*), pl.haar_lf_tab));
  if (pl.asc_qp_tab)
    HIP_TRY(stage(pl.asc_qp.data(), nlev * sizeof(void*), pl.asc_qp_tab));

  pl.tv.pos = d_morton;
  pl.tv.error = ctx->d_error;
  const TreeView tv = pl.tv;

  // ---- tree ------------------------------------------------------------
  const int32_t* sum_attrs = (encoder && !pl.haar) ? d_attrs : nullptr;
  {
    Timer t(ctx, "tree_count");
    tree_count_kernel<C><<<grid_for(tv.num_tiles, 4), 256, 0, st>>>(
      tv, sum_attrs, pl.tile_cnt, pl.tile_attr);
  }
  {
    Timer t(ctx, "tree_scan");
    tree_scan_kernel<C><<<1, 1024, 0, st>>>(
      tv, pl.tile_cnt, pl.tile_attr, pl.attr_prefix, sum_attrs != nullptr);
  }
  {
    Timer t(ctx, "tree_emit");
    tree_emit_kernel<C><<<grid_for(tv.num_tiles, 4), 256, 0, st>>>(
      tv, sum_attrs, pl.tile_cnt, pl.tile_attr, pl.attr_prefix);
  }
  // levels the coarse kernel may take: a slice's top levels with at most
  // kCoarseTiles tiles of parents each (sub-node prediction has its own path)
  const bool tiles = !pl.sub;
  // sub-node prediction: a slice's levels with at most kSweepMaxParents parents are walked by ONE
  // workgroup per slice inside one launch (raht_sweep.hpp); GPCC_SWEEP=0 keeps the per-level kernels
  static const bool sweep_on = [] {
    const char* e = getenv("GPCC_SWEEP");
    return !(e && e[0] == '0');
  }();
  // which levels: those where every slice has at most so many parents (GPCC_SWEEP_PARENTS; <= kSweepMaxParents)
  static const int sweep_parents = [] {
    const char* e = getenv("GPCC_SWEEP_PARENTS");
    const int v = e ? atoi(e) : kSweepDefaultParents;
    return v < 1 ? 1 : (v > kSweepMaxParents ? kSweepMaxParents : v);
  }();
  const bool sweep = pl.sub && !pl.haar && !pl.has_qp && !pl.pipe && !pl.links && sweep_on;
  {
    Timer t(ctx, "schedule");
    schedule_kernel<<<1, 256, 0, st>>>(
      tv, pl.sched, hp->num_qp_layers, tiles ? kCoarseParents : (sweep ? sweep_parents : 0), ctx->h_stats);
  }
  HIP_TRY(hipEventRecord(ctx->ev_stats, st));

  // ---- non-associative ascent (integer Haar / region QP) ----------------
  if ((encoder && pl.haar) || pl.has_qp) {
    AscendCtx ac{};
    ac.tv = tv;
    ac.attrs = (encoder && pl.haar) ? d_attrs : nullptr;
    ac.qp_off = d_qp_off;
    ac.haar_lf = (encoder && pl.haar) ? pl.haar_lf_tab : nullptr;
    ac.asc_qp = pl.has_qp ? pl.asc_qp_tab : nullptr;
    ac.dup_hf = pl.dup_hf;
    ac.dqp_root = pl.dqp[1];
    {
      Timer t(ctx, "ascend");
      ac.li = 0;
      ascend_leaf_kernel<C><<<grid_for(pl.cap[0], 256), 256, 0, st>>>(ac);
      for (int li = 1; li < nlev; li++) {
        ac.li = li;
        ascend_level_kernel<C><<<grid_for(pl.cap[li], 256), 256, 0, st>>>(ac);
      }
      if (pl.has_qp)
        qp_root_kernel<<<(s + 63) / 64, 64, 0, st>>>(ac, pl.sched);
    }
  }

  // ---- descent -----------------------------------------------------------
  LevelCtx lc{};
  lc.tv = tv;
  lc.params = pl.params;
  lc.sched = pl.sched;
  lc.attr_prefix = pl.attr_prefix;
  lc.haar_lf = (encoder && pl.haar) ? pl.haar_lf_tab : nullptr;
  lc.asc_qp = pl.has_qp ? pl.asc_qp_tab : nullptr;
  for (int i = 0; i < 2; i++) {
    lc.rec[i] = pl.rec[i];
    lc.rec_us[i] = pl.rec_us[i];
    lc.nneigh[i] = pl.nneigh[i];
    lc.dqp[i] = pl.dqp[i];
  }
  lc.coeffs = d_coeffs;
  lc.desc = pl.desc;
  lc.ptrans = pl.ptrans;
  lc.lut = ctx->d_lut;
  lc.worklist = pl.worklist;
  lc.work_count = pl.work_count;
  lc.scan_state = pl.scan_state;
  lc.pocc = pl.pocc;
  lc.mbox = pl.mbox;
  lc.ticket = pl.work_count + kMaxLevels;
  lc.error = ctx->d_error;
  HIP_TRY(hipMemsetAsync(pl.work_count, 0, kMaxLevels * 9 * sizeof(int32_t), st));
  HIP_TRY(hipMemsetAsync(pl.scan_state, 0, 1024 * sizeof(unsigned long long), st));
  if (pl.mbox)  // tags of a call are li + 1 >= 1
    HIP_TRY(hipMemsetAsync(pl.mbox, 0, (size_t)n * C * 4 * sizeof(uint32_t), st));
  if (pl.rdoq_state)
    HIP_TRY(hipMemsetAsync(pl.rdoq_state, 0, ((size_t)n + 1) * sizeof(unsigned long long), st));
  lc.rdoq_state = pl.rdoq_state;
  lc.slice_l = pl.slice_l;

  RdoqCtx rc{};
  if (pl.lossy) {
    rc.tv = tv;
    rc.sched = pl.sched;
    rc.tile_base = pl.rtile_base;
    rc.num_tiles = pl.num_rtiles;
    rc.desc = pl.desc;
    rc.coeffs = d_coeffs;
    rc.slice_l = pl.slice_l;
    rc.state = pl.rtile_state;
    rc.c = C;
    HIP_TRY(hipMemsetAsync(pl.rtile_state, 0, ((size_t)pl.num_rtiles + 1) * sizeof(unsigned long long), st));
    HIP_TRY(hipMemsetAsync(pl.slice_l, 0xff, 2 * (size_t)s * sizeof(int32_t), st));
  }

  // ---- the coarse levels of every slice: one launch ----------------------
  if (tiles) {
    lc.li = 0;
    const int cgrid = std::min(s, 2048);
    if (!encoder) {
      Timer t(ctx, "coarse_synth");
      raht_coarse_kernel<C, kCoarseDecode><<<cgrid, 1024, 0, st>>>(lc);
    } else if (pl.haar) {
      Timer t(ctx, "coarse_fused");
      raht_coarse_kernel<C, kCoarseHaar><<<cgrid, 1024, 0, st>>>(lc);
    } else {
      Timer t(ctx, "coarse_lossy");
      raht_coarse_kernel<C, kCoarseLossy><<<cgrid, 1024, 0, st>>>(lc);
    }
  }

  // The host learns the shape of the tree while the coarse kernel runs: the
  // remaining launches are sized by the real node counts and levels no slice
  // needs are not launched at all.
  HIP_TRY(hipEventSynchronize(ctx->ev_stats));
  const TreeStats ts = *ctx->h_stats;
  const int first_level = std::min(nlev - 1, tiles ? ts.fine_levels : ts.max_top);

  // ---- decoder with sub-node prediction: the DAG walked across levels --------
  bool piped = false;
  if (pl.pipe && first_level >= 1) {
    PipeCtx px{};
    int64_t total = 0;
    for (int li = 0; li <= first_level; li++) {
      px.uoff[li] = (int32_t)total;
      total += ts.nodes[li];
    }
    // the walk's node store, sized by the real node counts the host has just
    // read: from the context's caching pool (the arena is carved from host-known
    // bounds before the tree exists: 12 levels of N nodes for a lidar frame)
    const size_t need = (size_t)total * (4 + 1 + 4 + 32 * C) + 2048;
    void* pipe_mem = nullptr;
    if (total * C * 16 < ((int64_t)1 << 31) && pool_malloc(ctx, &pipe_mem, need) == hipSuccess) {
      Arena pa;
      pa.base = (char*)pipe_mem;
      pa.cap = need;
      px.g = pa.take<uint32_t>((size_t)total * C * 4);
      px.u = pa.take<uint32_t>((size_t)total * C * 4);
      px.src = pa.take<int32_t>(total);
      px.worklist = pa.take<int32_t>(total);
      px.pocc = pa.take<uint8_t>(total);
      px.total = (int32_t)total;
      px.top = first_level;
      px.tag = 1;
      px.ticket = lc.ticket;  // zeroed with work_count above
      HIP_TRY(hipMemsetAsync(px.g, 0, (size_t)total * C * 16, st));
      HIP_TRY(hipMemsetAsync(px.u, 0, (size_t)total * C * 16, st));
      {
        Timer t(ctx, "pipe_prepass");
        pipe_iota_kernel<<<grid_for(total, 256), 256, 0, st>>>(px.src, (int)total);
        for (int li = first_level - 1; li >= 0; li--) {
          lc.li = li;
          const int64_t parents = ts.nodes[li + 1];
          raht_pipe_prepass_kernel<C><<<(int)std::min<int64_t>((parents + 1023) / 1024, 1024), 256, 0, st>>>(lc, px);
        }
      }
      {
        Timer t(ctx, "pipe_synth");
        if (pl.f64)
          raht_pipe_synth_kernel<C, ArithF64><<<kSubGrid, 256, 0, st>>>(lc, px);
        else
          raht_pipe_synth_kernel<C><<<kSubGrid, 256, 0, st>>>(lc, px);
      }
      {
        Timer t(ctx, "pipe_leaf");
        pipe_leaf_kernel<C><<<grid_for(ts.nodes[0], 256), 256, 0, st>>>(lc, px);
      }
      piped = true;
      pool_free(ctx, pipe_mem);  // reuse is ordered on the context's stream
    }
  }

  // neighbour links of the sub-node kernels (raht_links.hpp): the occupancy pass and the top levels now,
  // every other level right before the level kernel that consumes it
  LinkSchedule links;
  auto link_prof = [&](const char* name, int li) { return Timer(ctx, li < 0 ? name : level_name(name, li)); };
  // (without the RAHT extension a parent with ONE child -- and one point -- searches too: the links cover the
  // nodes with more than one point only, those slices keep the bisection)
  const bool use_links = pl.links && !piped && !tiles && first_level >= 1 && hp->raht_extension != 0;
  if (use_links) {
    links.tv = tv;
    links.lv = pl.lv;
    links.begin(st, ts.nodes, first_level, link_prof);
  }
  // ---- sub-node prediction, the coarse levels of every slice: one launch (raht_sweep.hpp) ----
  int level_from = first_level;  // the per-level launches start below this level
  // Block records for the per-level kernels (raht_level_sub_kernel<.., REC>): the static half of every round's
  // prologue written by a full-occupancy pass per level.  OPT-IN (GPCC_REC=1), measured and not kept
  // (profiles/r06_rec_ab.txt): the level kernels lose 4-7 % of their time and the record pass costs 12-13 % -- a batch
  // of ten 1 M-point frames forward 28.2 -> 30.1 ms, one frame 7.38 -> 7.62 ms.  The ~25 dependent loads in front of a
  // round are not what a level waits for (round 3 found the same with partial records).
  static const int rec_mode = [] {
    const char* e = getenv("GPCC_REC");
    return e ? atoi(e) : 0;
  }();
  const bool ext_flag = hp->raht_extension != 0;
  const bool use_rec = pl.sub && !pl.haar && !pl.has_qp && !pl.links && !piped && rec_mode == 1;
  size_t rec_need = 0;
  if (use_rec)
    for (int li = std::min(first_level, sweep ? std::max(0, ts.fine_levels) : first_level) - 1; li >= 0; li--)
      rec_need = std::max(rec_need, sweep_rec_bytes(level_max_blocks(ts.nodes, li, ext_flag), C));
  if (sweep && !piped && ts.fine_levels < first_level) {
    const int lo = std::max(0, ts.fine_levels);
    const SweepCtx sw{first_level - 1, lo};
    SweepRec rec{};
    const int64_t sweep_parents = sweep_rec_layout(&rec, ts.nodes, sw.li_hi, sw.li_lo);
    // the records of the levels the sweep takes, sized by the real node counts the host has just read
    // (reuse of the context's block is ordered on the context's stream)
    if (ensure_sweep_mem(ctx, std::max(rec_need, sweep_rec_bytes(sweep_parents, C))) == GPCC_OK) {
      level_from = lo;
      sweep_rec_carve(&rec, ctx->sweep_mem, sweep_parents, C);
      Timer t(ctx, encoder ? "sub_sweep_lossy" : "sub_sweep_synth");
      sweep_launch<C>(st, lc, sw, rec, ts.nodes, s, encoder, pl.f64);
    }
  }
  for (int li = piped ? -1 : level_from - 1; li >= 0; li--) {
    lc.li = li;
    lc.mtag = (uint32_t)(li + 1);
    const int64_t parents = ts.nodes[li + 1];
    if (use_links) {
      links.produce(st, li + 1, link_prof);
      lc.link_rec = pl.lv.rec[(li + 1) & 1];
      lc.link_lrec = pl.lv.lrec[(li + 1) & 1];
    }
    if (tiles) {
      // a level where almost every parent has a single child takes the large
      // tiles (raht_tile.hpp): the node counts tell
      const bool sparse = ts.nodes[li] * 8 <= parents * 9 && parents >= 64 * kTileTSparse;
      const int tile_t = sparse ? kTileTSparse : kTileT;
      const int ntiles = (int)((parents + tile_t - 1) / tile_t);
      const int tgrid = std::min((ntiles + 7) / 8 * 8, kLevelGridMax);
#define GPCC_TILE_LAUNCH(MODE)                                                  \
  do {                                                                          \
    if (sparse)                                                                 \
      raht_tile_kernel<C, MODE, kTileTSparse><<<tgrid, kTileThreads, 0, st>>>(lc); \
    else                                                                        \
      raht_tile_kernel<C, MODE, kTileT><<<tgrid, kTileThreads, 0, st>>>(lc);    \
  } while (0)
      if (!encoder) {
        Timer t(ctx, level_name("tile_synth", li));
        GPCC_TILE_LAUNCH(kSynth);
      } else if (pl.haar) {
        Timer t(ctx, level_name("tile_fused", li));
        GPCC_TILE_LAUNCH(kFused);
      } else {
        {
          Timer t(ctx, level_name("tile_analyze", li));
          GPCC_TILE_LAUNCH(kAnalyze);
        }
        rc.li = li;
        {
          // one wavefront per 2048-coefficient tile of the batch, in order
          Timer t(ctx, "rdoq_resolve");
          rdoq_resolve_kernel<<<(pl.num_rtiles + 3) / 4, 256, 0, st>>>(rc);
        }
        {
          Timer t(ctx, level_name("tile_synth_rec", li));
          GPCC_TILE_LAUNCH(kSynthRec);
        }
      }
#undef GPCC_TILE_LAUNCH
      continue;
    }
    {
      Timer t(ctx, level_name("level_prepass", li));
      raht_level_prepass_kernel<C><<<(int)std::min<int64_t>((parents + 1023) / 1024, 1024), 256, 0, st>>>(lc);
    }
    // sub-node kernel: resident workgroups claim rounds of 8 blocks per wave by
    // ticket; a level with few parents gets few workgroups (measured: neutral
    // from 1 workgroup per 32 parents to the full grid, slower beyond 1 per 128)
    const int sgrid = (int)std::min<int64_t>(
      kSubGrid, std::max<int64_t>(8, (parents / (GPCC_SUB_BLOCKS_PER_WG) + 7) / 8 * 8));
    // Claims of several CONSECUTIVE rounds per wavefront with the zero-run state carried in registers
    // (raht_subnode.hpp, ctx.claim_rounds): built and measured in round 5, bit-exact, and 10-20 x SLOWER
    // (profiles/r05_claim_rounds_ab.txt: headline forward 7.5 -> 134 ms at 8 rounds per claim) -- a wavefront
    // walks its rounds one after the other and each starts with the ~100 us of dependent loads of its prologue,
    // which the one-round claims of many wavefronts overlap; the blocks of the NEXT claim wait for all of them.
    // Opt-in for experiments only: GPCC_SUB_CLAIM=R (default 1), GPCC_SUB_CLAIM_PARENTS (levels with at most
    // so many parents, default 100 000).
    lc.claim_rounds = sub_claim_rounds(encoder, pl.haar, parents);
    bool rec_level = false;
    if (use_rec && ensure_sweep_mem(ctx, rec_need) == GPCC_OK) {
      Timer t(ctx, level_name("level_record", li));
      lc.brec = level_record_launch<C>(st, lc, li, ctx->sweep_mem, level_max_blocks(ts.nodes, li, ext_flag), encoder, pl.f64);
      rec_level = true;
    }
    if (rec_level && !encoder) {
      Timer t(ctx, level_name("level_sub_synth", li));
      if (pl.f64)
        raht_level_sub_kernel<C, kSynth, ArithF64, false, true><<<sgrid, 256, 0, st>>>(lc);
      else
        raht_level_sub_kernel<C, kSynth, ArithI64, false, true><<<sgrid, 256, 0, st>>>(lc);
    } else if (rec_level) {
      Timer t(ctx, level_name("level_sub_lossy", li));
      if (pl.f64)
        raht_level_sub_kernel<C, kLossySub, ArithF64, false, true><<<sgrid, 256, 0, st>>>(lc);
      else
        raht_level_sub_kernel<C, kLossySub, ArithI64, false, true><<<sgrid, 256, 0, st>>>(lc);
    } else if (!encoder) {
      Timer t(ctx, level_name("level_sub_synth", li));
      if (pl.f64)
        raht_level_sub_kernel<C, kSynth, ArithF64><<<sgrid, 256, 0, st>>>(lc);
      else
        raht_level_sub_kernel<C, kSynth><<<sgrid, 256, 0, st>>>(lc);
    } else if (pl.haar) {
      Timer t(ctx, level_name("level_sub_fused", li));
      raht_level_sub_kernel<C, kFused><<<sgrid, 256, 0, st>>>(lc);
    } else {
      Timer t(ctx, level_name("level_sub_lossy", li));
      if (pl.f64)
        raht_level_sub_kernel<C, kLossySub, ArithF64><<<sgrid, 256, 0, st>>>(lc);
      else
        raht_level_sub_kernel<C, kLossySub><<<sgrid, 256, 0, st>>>(lc);
    }
  }

  // ---- duplicates + write-back ------------------------------------------
  FinishCtx fc{};
  fc.tv = tv;
  fc.params = pl.params;
  fc.sched = pl.sched;
  fc.attr_prefix = pl.attr_prefix;
  fc.haar_lf = (encoder && pl.haar) ? pl.haar_lf_tab : nullptr;
  fc.dup_hf = pl.dup_hf;
  fc.asc_qp = pl.has_qp ? pl.asc_qp_tab : nullptr;
  fc.qp_off = d_qp_off;
  for (int i = 0; i < 2; i++) {
    fc.rec[i] = pl.rec[i];
    fc.dqp[i] = pl.dqp[i];
  }
  fc.attrs = d_attrs;
  fc.coeffs = d_coeffs;
  fc.encoder = encoder;
  fc.lut = ctx->d_lut;
  {
    Timer t(ctx, "finish");
    finish_kernel<C><<<grid_for(pl.cap[0], 256), 256, 0, st>>>(fc);
  }
  if (ctx->h_error)
    HIP_TRY(hipMemcpyAsync(ctx->h_error, ctx->d_error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipGetLastError());
  return GPCC_OK;
}

// ---- the compact level pass (cx_driver.hpp): batches without sub-node prediction,
//      with the RAHT extension, without integer Haar and region QPs -------------------
template<int C>
int
launch_cx(
  gpcc_ctx* ctx, CxWork& w, const gpcc_raht_params* hp, const int64_t* offsets,
  const int64_t* d_morton, int32_t* d_attrs, int32_t* d_coeffs)
{
  hipStream_t st = ctx->stream;
  const int s = w.s;
  std::vector<int32_t> h_off(s + 1);
  for (int i = 0; i <= s; i++)
    h_off[i] = (int32_t)offsets[i];
  const size_t stage_bytes = (sizeof(gpcc_raht_params) + (s + 1) * sizeof(int32_t) + 64 + 255) & ~size_t(255);
  if (ctx->h_cx_stage_cap < stage_bytes) {
    HIP_TRY(hipStreamSynchronize(st));
    if (ctx->h_cx_stage)
      HIP_TRY(hipHostFree(ctx->h_cx_stage));
    ctx->h_cx_stage = nullptr;
    ctx->h_cx_stage_cap = 0;
    HIP_TRY(hipHostMalloc(&ctx->h_cx_stage, stage_bytes * 4));
    ctx->alloc_events[2]++;
    ctx->h_cx_stage_cap = stage_bytes * 2;
  }
  ctx->cx_stage_flip ^= 1;
  char* hp_base = (char*)ctx->h_cx_stage + (ctx->cx_stage_flip ? ctx->h_cx_stage_cap : 0);
  memcpy(hp_base, hp, sizeof(*hp));
  HIP_TRY(hipMemcpyAsync(w.params, hp_base, sizeof(*hp), hipMemcpyHostToDevice, st));
  const size_t o = (sizeof(*hp) + 15) & ~size_t(15);
  memcpy(hp_base + o, h_off.data(), (s + 1) * sizeof(int32_t));
  HIP_TRY(hipMemcpyAsync(w.pt_off, hp_base + o, (s + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  w.tv.pos = d_morton;
  w.tv.error = ctx->d_error;
  hipError_t e = cx_run<C>(
    st, w, ctx->d_lut, hp->num_qp_layers, d_attrs, d_coeffs, ctx->h_stats, ctx->h_cxtab,
    [&](const char* name, int li) { return Timer(ctx, li < 0 ? name : level_name(name, li)); },
    [&]() -> hipError_t { return hipEventRecord(ctx->ev_stats, st); },
    [&]() -> hipError_t { return hipEventSynchronize(ctx->ev_stats); });
  if (e != hipSuccess)
    return fail(GPCC_ERR_HIP, std::string("compact level pass: ") + hipGetErrorString(e));
  if (ctx->h_error)
    HIP_TRY(hipMemcpyAsync(ctx->h_error, ctx->d_error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  return GPCC_OK;
}

int
check_device_error(gpcc_ctx* ctx)
{
  if (ctx->h_error && *ctx->h_error) {
    // sticky on the device: every call since the failure copied the same word;
    // reported once, then cleared on both sides
    const int code = *ctx->h_error;
    *ctx->h_error = 0;
    hipMemsetAsync(ctx->d_error, 0, sizeof(int32_t), ctx->stream);
    if (code == 3)
      return fail(
        GPCC_ERR_RANGE,
        "values beyond the range of the fast arithmetic path (attributes wider than max_qp's bit "
        "depth says?); nothing was written -- gpcc_ctx_set_fast_arith(ctx, 0) and call again");
    if (code == 4)
      return fail(
        GPCC_ERR_UNSUPPORTED,
        "inter-frame RAHT: a coefficient magnitude beyond the rate estimate's log2 table; nothing was written");
    if (code == 2)
      return fail(
        GPCC_ERR_INVALID_ARG,
        "the Morton codes are wider than gpcc_ctx_set_morton_bits said (or a "
        "slice is not sorted): more than one node at the top level; the result "
        "is invalid");
    return fail(
      GPCC_ERR_HIP,
      "a dependency wait in the sub-node prediction kernel expired; the "
      "result is invalid");
  }
  return GPCC_OK;
}

int
dev_transform(
  gpcc_ctx* ctx, const gpcc_raht_params* params, bool encoder, int32_t s,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, void* d_coeffs, int32_t c, int morton_bits)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  int rcode = check_params(params, c, encoder);
  if (rcode)
    return rcode;
  if (s < 1 || !offsets || offsets[0] != 0)
    return fail(GPCC_ERR_INVALID_ARG, "bad slice offsets");
  for (int i = 0; i < s; i++)
    if (offsets[i + 1] <= offsets[i])
      return fail(GPCC_ERR_INVALID_ARG, "empty or unordered slice");
  if (offsets[s] > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per batch");
  if (!d_morton || !d_attrs || !d_coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null device buffer");
  HIP_TRY(hipSetDevice(ctx->device));

  const int bits = morton_bits > 0 ? std::min(morton_bits, 63) : 63;
  {
    static const bool cx_on = [] {
      const char* e = getenv("GPCC_CX");
      return !(e && e[0] == '0');
    }();
    if (cx_on && cx_supported(params, d_qp_off != nullptr, offsets[s])) {
      CxWork w;
      w.n = (int)offsets[s];
      w.s = s;
      w.c = c;
      w.nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
      w.encoder = encoder;
      {
        // ArithF64 (raht_arith.hpp) is available to the compact level pass under the same condition as
        // to the sub-node kernels below, but it is not the default here: measured on ten 1 M-point slices
        // (round 4) the encoder is SLOWER in doubles (cx_level_enc 2.66 against 2.33 ms: 96 registers and
        // 28 bytes of scratch at five wavefronts per SIMD, against 83 and none) and the decoder the same
        // (1.98 / 2.03 ms) -- the pass spends its issue slots on the neighbour search, not on the
        // products.  GPCC_CX_F64=1 selects it (tests/test_gpu_arith.py runs both).
        static const bool cx_f64 = [] {
          const char* e = getenv("GPCC_CX_F64");
          return e && e[0] == '1';
        }();
        int64_t n_max = 1;
        for (int i = 0; i < s; i++)
          n_max = std::max(n_max, offsets[i + 1] - offsets[i]);
        const int bdepth = 8 + std::max(0, (params->max_qp - 51 + 5) / 6);
        w.f64 = cx_f64 && ctx->fast_arith && !ctx->force_exact
          && 2 * bdepth + bitlen64((uint64_t)(n_max - 1)) <= 36;
      }
      w.links = links_enabled();
      size_t used = 0;
      // (guard mode: Arena::take puts a band behind every sub-allocation)
      cx_carve(
        [&](size_t bytes) {
          used += ((bytes + 255) & ~size_t(255)) + (guard_mode() ? kGuardInner : 0);
          return (char*)nullptr;
        },
        w);
      rcode = ensure_arena(ctx, used);
      if (rcode)
        return rcode;
      ctx->arena.reset();
      cx_carve([&](size_t bytes) { return ctx->arena.take<char>(bytes); }, w);
      if (ctx->arena.used > ctx->arena.cap)
        return fail(GPCC_ERR_OUT_OF_MEMORY, "compact level pass: workspace carved past the arena");
      switch (c) {
      case 1:
        return launch_cx<1>(ctx, w, params, offsets, (const int64_t*)d_morton, (int32_t*)d_attrs, (int32_t*)d_coeffs);
      case 2:
        return launch_cx<2>(ctx, w, params, offsets, (const int64_t*)d_morton, (int32_t*)d_attrs, (int32_t*)d_coeffs);
      default:
        return launch_cx<3>(ctx, w, params, offsets, (const int64_t*)d_morton, (int32_t*)d_attrs, (int32_t*)d_coeffs);
      }
    }
  }
  Plan pl;
  pl.n = (int)offsets[s];
  pl.s = s;
  pl.c = c;
  pl.nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  pl.encoder = encoder;
  pl.haar = params->integer_haar_enable_flag != 0;
  pl.has_qp = d_qp_off != nullptr;
  pl.lossy = encoder && !pl.haar;
  pl.sub = params->raht_prediction_enabled_flag
    && params->raht_subnode_prediction_enabled_flag;
  {
    static const bool pipe_on = [] {
      const char* e = getenv("GPCC_PIPE");
      return e && e[0] == '1';
    }();
    // The cross-level walk (opt-in since round 4: GPCC_PIPE=1) covers the decoder
    // without region QPs, one slice per call: it shortens the critical path of ONE
    // dependency DAG, but every round pays two more polling stages, and a batch is
    // bound by the rounds' wave time, not by a chain -- there the level-by-level
    // kernels interleave the slices' chains (2 / 3 / 5 / 10 frames: 6.0 / 7.2 /
    // 9.5 / 15.1 ms against 8.1 / 11.4 / 18.7 / 36.6).  Round 2 measured it ahead for
    // the single 1 M lidar frame (4.2 against 5.0 ms); with the level kernels in
    // doubles and their lean loop (round 4) it is behind: 4.80 against 4.28 ms
    // (kernel + prepass), headline 12.41 -> 11.77 ms.
    pl.pipe = pipe_on && !encoder && pl.sub && !pl.has_qp && s == 1;
  }
  pl.num_rtiles = 0;
  int64_t n_max = 1;
  for (int i = 0; i < s; i++) {
    pl.num_rtiles +=
      (int)((offsets[i + 1] - offsets[i] + kRdoqTile - 1) / kRdoqTile);
    n_max = std::max(n_max, offsets[i + 1] - offsets[i]);
  }
  {
    // ArithF64 where B-bit attributes (B from max_qp = 51 + 6 (B - 8), tmc3/quantization.cpp:151)
    // in slices of n points keep every product exact: values reach 2^(B + 15) sqrt(n), the kernels'
    // check sits at 2^34 (raht_arith.hpp) -- 2 B + ceil(log2 n) <= 36 leaves it a factor two
    const int bdepth = 8 + std::max(0, (params->max_qp - 51 + 5) / 6);
    pl.f64 = ctx->fast_arith && !ctx->force_exact && pl.sub && !pl.haar && params->raht_extension
      && 2 * bdepth + bitlen64((uint64_t)(n_max - 1)) <= 36;
  }

  Arena measure;
  carve(measure, pl);
  rcode = ensure_arena(ctx, measure.used);
  if (rcode)
    return rcode;
  carve(ctx->arena, pl);

  switch (c) {
  case 1:
    return launch_transform<1>(
      ctx, pl, params, encoder, offsets, (const int64_t*)d_morton,
      (const int32_t*)d_qp_off, (int32_t*)d_attrs, (int32_t*)d_coeffs);
  case 2:
    return launch_transform<2>(
      ctx, pl, params, encoder, offsets, (const int64_t*)d_morton,
      (const int32_t*)d_qp_off, (int32_t*)d_attrs, (int32_t*)d_coeffs);
  default:
    return launch_transform<3>(
      ctx, pl, params, encoder, offsets, (const int64_t*)d_morton,
      (const int32_t*)d_qp_off, (int32_t*)d_attrs, (int32_t*)d_coeffs);
  }
}

// host tier: stage one slice through HBM
int
host_transform(
  gpcc_ctx* ctx, const gpcc_raht_params* params, bool encoder,
  const int64_t* morton, const int32_t* qp_off, int32_t* attrs,
  int32_t* coeffs, int32_t n, int32_t c)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!morton || !attrs || !coeffs || n <= 0)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or n <= 0");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  int rcode = check_params(params, c, encoder);
  if (rcode)
    return rcode;
  for (int i = 1; i < n; i++)
    if (morton[i] < morton[i - 1])
      return fail(GPCC_ERR_UNSORTED, "Morton codes are not ascending");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int bits = bitlen64((uint64_t)(morton[0] ^ morton[n - 1]));

  int64_t* d_m = nullptr;
  int32_t *d_q = nullptr, *d_a = nullptr, *d_c = nullptr;
  auto cleanup = [&]() {
    pool_free(ctx, d_m);
    pool_free(ctx, d_q);
    pool_free(ctx, d_a);
    pool_free(ctx, d_c);
  };
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, (void**)&d_m, sizeof(int64_t) * n));
    HIP_TRY(pool_malloc(ctx, (void**)&d_a, sizeof(int32_t) * n * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_c, sizeof(int32_t) * n * c));
    HIP_TRY(h2d_user(ctx, d_m, morton, sizeof(int64_t) * n, st));
    if (qp_off) {
      HIP_TRY(pool_malloc(ctx, (void**)&d_q, sizeof(int32_t) * n * 2));
      HIP_TRY(h2d_user(ctx, d_q, qp_off, sizeof(int32_t) * n * 2, st));
    }
    if (encoder) {
      HIP_TRY(h2d_user(ctx, d_a, attrs, sizeof(int32_t) * n * c, st));
      // the reference's callers hand in a zero-initialised coefficient
      // vector; the one slot an all-duplicates slice leaves unwritten stays 0
      HIP_TRY(hipMemsetAsync(d_c, 0, sizeof(int32_t) * n * c, st));
    } else
      HIP_TRY(h2d_user(ctx, d_c, coeffs, sizeof(int32_t) * n * c, st));
    const int64_t offs[2] = {0, n};
    int r = dev_transform(
      ctx, params, encoder, 1, offs, d_m, d_q, d_a, d_c, c, std::max(bits, 1));
    if (r)
      return r;
    // the caller's buffers are written only once the transform is known to
    // have succeeded: on any failure `attrs` still holds the source, so the
    // caller can hand the slice to the reference CPU function
    HIP_TRY(hipStreamSynchronize(st));
    r = check_device_error(ctx);
    if (r)
      return r;
    HIP_TRY(d2h_user(ctx, attrs, d_a, sizeof(int32_t) * n * c, st));
    if (encoder)
      HIP_TRY(d2h_user(ctx, coeffs, d_c, sizeof(int32_t) * n * c, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  int r = run();
  if (r == GPCC_ERR_RANGE) {
    // the caller's buffers are untouched: once more in int64 arithmetic
    cleanup();
    d_m = nullptr;
    d_q = d_a = d_c = nullptr;
    ctx->force_exact = true;
    r = run();
    ctx->force_exact = false;
  }
  cleanup();
  return r;
}


// ---- RAHT with attribute inter prediction (raht_inter_driver.hpp): host tier -------------------
template<int C>
int
launch_inter(
  gpcc_ctx* ctx, InterWork& w, const InterTools& tl, const gpcc_raht_params* hp, const int64_t* d_ref_pos,
  const int32_t* d_ref_attrs, int32_t* d_attrs, int32_t* d_coeffs, const int32_t* d_qp_off)
{
  hipStream_t st = ctx->stream;
  // the encoder's second candidate under sub-node prediction runs on the context's second stream
  // (GPCC_INTER_STREAMS=0: one after the other); with the profiler on the spans are per stream order: sequential
  InterStreams streams;
  static const bool two = [] {
    const char* ev = getenv("GPCC_INTER_STREAMS");
    return !(ev && ev[0] == '0');
  }();
  if (two && w.sub && w.encoder && !ctx->profiling) {
    if (!ctx->kd_stream)
      HIP_TRY(hipStreamCreateWithFlags(&ctx->kd_stream, hipStreamNonBlocking));
    if (!ctx->kd_event)
      HIP_TRY(hipEventCreateWithFlags(&ctx->kd_event, hipEventDisableTiming));
    if (!ctx->inter_event)
      HIP_TRY(hipEventCreateWithFlags(&ctx->inter_event, hipEventDisableTiming));
    streams.second = ctx->kd_stream;
    streams.fork = ctx->kd_event;
    streams.join = ctx->inter_event;
  }
  hipError_t e = inter_run<C>(
    st, w, tl, hp, ctx->d_lut, ctx->d_log2, d_ref_pos, d_ref_attrs, d_attrs, d_coeffs, ctx->h_stats,
    [&](const char* name, int li) { return Timer(ctx, li < 0 ? name : level_name(name, li)); },
    [&]() -> hipError_t { return hipEventRecord(ctx->ev_stats, st); },
    [&]() -> hipError_t { return hipEventSynchronize(ctx->ev_stats); }, streams, d_qp_off);
  if (e != hipSuccess)
    return fail(GPCC_ERR_HIP, std::string("inter-frame RAHT: ") + hipGetErrorString(e));
  if (ctx->h_error)
    HIP_TRY(hipMemcpyAsync(ctx->h_error, ctx->d_error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  return GPCC_OK;
}

int
host_transform_inter(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_raht_inter_params* inter, bool encoder,
  const int64_t* morton, const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, const int64_t* morton_ref,
  const int32_t* attrs_ref, int32_t n_ref, int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps,
  int32_t* num_taps)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!morton || !attrs || !coeffs || n <= 0)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or n <= 0");
  if (!inter || !morton_ref || !attrs_ref || n_ref <= 0)
    return fail(GPCC_ERR_INVALID_ARG, "inter-frame RAHT: no reference frame or no tools");
  if (!layer_modes || !num_modes || !filter_taps || !num_taps)
    return fail(GPCC_ERR_INVALID_ARG, "inter-frame RAHT: null layer_modes / filter_taps");
  if (n > kMaxPoints || n_ref > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  int rcode = check_params(params, c, encoder);
  if (rcode)
    return rcode;
  if (!encoder && (*num_modes < 0 || *num_modes > 32 || *num_taps < 0 || *num_taps > 32))
    return fail(GPCC_ERR_INVALID_ARG, "at most 32 layer modes / filter taps");
  for (int i = 1; i < n; i++)
    if (morton[i] < morton[i - 1])
      return fail(GPCC_ERR_UNSORTED, "Morton codes are not ascending");
  for (int i = 1; i < n_ref; i++)
    if (morton_ref[i] < morton_ref[i - 1])
      return fail(GPCC_ERR_UNSORTED, "Morton codes of the reference frame are not ascending");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int rl = ensure_log2(ctx);
  if (rl)
    return rl;

  InterWork w;
  w.n = n;
  w.c = c;
  w.n_ref = n_ref;
  w.encoder = encoder;
  w.sub = inter_sub(params);
  InterTools tl;
  tl.depth_limit = inter->raht_inter_prediction_depth_minus1 + 1;
  tl.layer_rdo = inter->raht_enable_inter_intra_layer_rdo != 0;
  tl.filter_est = inter->enable_filter_estimation != 0;
  tl.skip_layers = inter->skip_init_layers_for_filtering;
  tl.bits_cur = bitlen64((uint64_t)(morton[0] ^ morton[n - 1]));
  tl.bits_ref = n_ref <= 1 ? -1 : bitlen64((uint64_t)(morton_ref[0] ^ morton_ref[n_ref - 1]));
  if (!encoder) {
    tl.modes = layer_modes;
    tl.num_modes = *num_modes;
    tl.taps = filter_taps;
    tl.num_taps = *num_taps;
  }
  if (!inter_supported(params, n, tl))
    return fail(
      GPCC_ERR_UNSUPPORTED,
      "inter-frame RAHT on the device: not a single point, nor the integer Haar kernel with trees that do not line up on "
      "octree levels");
  w.nlev = std::min((std::max(tl.bits_cur, 1) + 2) / 3 + 1, (int)kMaxLevels);
  w.haar = params->integer_haar_enable_flag != 0;
  w.nlev_ref = std::min((std::max(tl.bits_ref, 1) + 2) / 3 + 1, (int)kMaxLevels);
  w.has_qp = qp_off != nullptr;
  size_t need = 0;
  inter_carve(
    [&](size_t bytes) {
      need += (bytes + 255) & ~size_t(255);
      return (char*)nullptr;
    },
    w);
  char* work = nullptr;
  int64_t* d_m = nullptr;
  int64_t* d_mr = nullptr;
  int32_t *d_a = nullptr, *d_c = nullptr, *d_ar = nullptr, *d_q = nullptr;
  auto cleanup = [&]() {
    pool_free(ctx, d_q);
    pool_free(ctx, work);
    pool_free(ctx, d_m);
    pool_free(ctx, d_mr);
    pool_free(ctx, d_a);
    pool_free(ctx, d_c);
    pool_free(ctx, d_ar);
  };
  auto run = [&]() -> int {
    HIP_TRY(small_reset(ctx));
    HIP_TRY(pool_malloc(ctx, (void**)&work, need + 256));
    HIP_TRY(pool_malloc(ctx, (void**)&d_m, sizeof(int64_t) * n));
    HIP_TRY(pool_malloc(ctx, (void**)&d_mr, sizeof(int64_t) * n_ref));
    HIP_TRY(pool_malloc(ctx, (void**)&d_a, sizeof(int32_t) * n * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_c, sizeof(int32_t) * n * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_ar, sizeof(int32_t) * (size_t)n_ref * c));
    size_t off = 0;
    inter_carve(
      [&](size_t bytes) {
        char* p = work + off;
        off += (bytes + 255) & ~size_t(255);
        return p;
      },
      w);
    w.tv.pos = d_m;
    w.tv.error = ctx->d_error;
    HIP_TRY(h2d_user(ctx, d_m, morton, sizeof(int64_t) * n, st));
    HIP_TRY(h2d_user(ctx, d_mr, morton_ref, sizeof(int64_t) * n_ref, st));
    HIP_TRY(h2d_user(ctx, d_ar, attrs_ref, sizeof(int32_t) * (size_t)n_ref * c, st));
    if (qp_off) {
      HIP_TRY(pool_malloc(ctx, (void**)&d_q, sizeof(int32_t) * (size_t)n * 2));
      HIP_TRY(h2d_user(ctx, d_q, qp_off, sizeof(int32_t) * (size_t)n * 2, st));
      HIP_TRY(small_h2d(ctx, w.asc_qp_tab, w.asc_qp, sizeof(w.asc_qp), st));
    }
    if (encoder) {
      HIP_TRY(h2d_user(ctx, d_a, attrs, sizeof(int32_t) * n * c, st));
      HIP_TRY(hipMemsetAsync(d_c, 0, sizeof(int32_t) * n * c, st));
    } else {
      HIP_TRY(h2d_user(ctx, d_c, coeffs, sizeof(int32_t) * n * c, st));
    }
    const int32_t h_off[2] = {0, n};
    const int32_t h_rt[2] = {0, w.num_rtiles};
    HIP_TRY(small_h2d(ctx, w.params, params, sizeof(*params), st));
    HIP_TRY(small_h2d(ctx, w.pt_off, h_off, sizeof(h_off), st));
    if (w.rtile_base)
      HIP_TRY(small_h2d(ctx, w.rtile_base, h_rt, sizeof(h_rt), st));
    const int32_t h_off_ref[2] = {0, n_ref};
    if (w.haar) {
      if (w.haar_lf_tab)
        HIP_TRY(small_h2d(ctx, w.haar_lf_tab, w.haar_lf, sizeof(w.haar_lf), st));
      HIP_TRY(small_h2d(ctx, w.ref_lf_tab, w.ref_lf, sizeof(w.ref_lf), st));
      HIP_TRY(small_h2d(ctx, w.pt_off_ref, h_off_ref, sizeof(h_off_ref), st));
    }
    int r = GPCC_ERR_INVALID_ARG;
    switch (c) {
    case 1: r = launch_inter<1>(ctx, w, tl, params, d_mr, d_ar, d_a, d_c, d_q); break;
    case 2: r = launch_inter<2>(ctx, w, tl, params, d_mr, d_ar, d_a, d_c, d_q); break;
    case 3: r = launch_inter<3>(ctx, w, tl, params, d_mr, d_ar, d_a, d_c, d_q); break;
    }
    if (r)
      return r;
    HIP_TRY(hipStreamSynchronize(st));
    r = check_device_error(ctx);
    if (r)
      return r;
    HIP_TRY(d2h_user(ctx, attrs, d_a, sizeof(int32_t) * n * c, st));
    if (encoder) {
      HIP_TRY(d2h_user(ctx, coeffs, d_c, sizeof(int32_t) * n * c, st));
      void *p_rs, *p_nt, *p_modes, *p_taps;
      HIP_TRY(small_d2h(ctx, &p_rs, w.rs, sizeof(RateState), st));
      HIP_TRY(small_d2h(ctx, &p_nt, w.num_taps, sizeof(int32_t), st));
      HIP_TRY(small_d2h(ctx, &p_modes, w.modes, 32 * sizeof(int32_t), st));
      HIP_TRY(small_d2h(ctx, &p_taps, w.taps, 32 * sizeof(int32_t), st));
      HIP_TRY(hipStreamSynchronize(st));
      const RateState rs = *(const RateState*)p_rs;
      const int32_t nt = *(const int32_t*)p_nt;
      memcpy(layer_modes, p_modes, 32 * sizeof(int32_t));
      memcpy(filter_taps, p_taps, 32 * sizeof(int32_t));
      *num_modes = rs.num_modes;
      *num_taps = nt;
      for (int i = rs.num_modes; i < 32; i++)
        layer_modes[i] = 0;
      for (int i = nt; i < 32; i++)
        filter_taps[i] = 0;
    }
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  // the intra candidate under sub-node prediction in doubles where they are exact (the rule of dev_transform)
  {
    const int bdepth = 8 + std::max(0, (params->max_qp - 51 + 5) / 6);
    w.f64 = ctx->fast_arith && w.sub && encoder && params->raht_extension
      && 2 * bdepth + bitlen64((uint64_t)(n - 1)) <= 36;
  }
  int r = run();
  if (r == GPCC_ERR_RANGE && w.f64) {
    // the caller's buffers are untouched: once more in int64 arithmetic
    cleanup();
    work = nullptr;
    d_m = d_mr = nullptr;
    d_a = d_c = d_ar = d_q = nullptr;
    w.f64 = false;
    r = run();
  }
  cleanup();
  return r;
}

// ---- lifting transform ---------------------------------------------------

struct LiftDev {
  const int32_t *nc, *ni, *nw, *indexes, *qp_off;
  int32_t *attrs, *coeffs;
};

int
check_lift_params(const gpcc_lift_params* p, int n, int c)
{
  if (!p)
    return fail(GPCC_ERR_INVALID_ARG, "params is null");
  if (n <= 0 || c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "n <= 0 or attribute count not 1..3");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  if (p->num_lods < 1 || p->num_lods > GPCC_MAX_LODS)
    return fail(GPCC_ERR_INVALID_ARG, "num_lods out of range");
  if (p->num_points_in_lod[p->num_lods - 1] != n)
    return fail(GPCC_ERR_INVALID_ARG, "num_points_in_lod does not end at n");
  for (int l = 1; l < p->num_lods; l++)
    if (p->num_points_in_lod[l] < p->num_points_in_lod[l - 1])
      return fail(GPCC_ERR_INVALID_ARG, "num_points_in_lod not ascending");
  if (p->num_qp_layers < 1 || p->num_qp_layers > GPCC_MAX_QP_LAYERS)
    return fail(GPCC_ERR_INVALID_ARG, "num_qp_layers out of range");
  return GPCC_OK;
}

// Attribute inter prediction (n_frame > 0): the working arrays get n_frame entries
// BEHIND the n predictors.  a[n + r] holds the reference frame's attribute r in fixed
// point (h_frame, host memory, [n_frame][C]) and the caller has pointed every neighbour
// that lives in that frame at n + r: PCCLiftPredict then reads the frame's value
// (PCCTMC3Common.h:735-740) and what PCCLiftUpdate / PCCComputeQuantizationWeights skip for
// such a neighbour (:799-800, :845-846) lands in entries nobody reads -- the kernels are
// the intra ones, untouched.
template<int C>
int
launch_lift(
  gpcc_ctx* ctx, bool encoder, const gpcc_lift_params* p, int n,
  const LiftDev& d, int8_t* d_lcp_io, char* scratch, int n_frame = 0,
  const int64_t* h_frame = nullptr)
{
  hipStream_t st = ctx->stream;
  LiftCtx cx{};
  cx.n = n;
  cx.c = C;
  cx.num_lods = p->num_lods;
  for (int l = 0; l < p->num_lods; l++)
    cx.npl[l] = p->num_points_in_lod[l];
  // replay the reference's running counters over the distinct LoD
  // boundaries (AttributeEncoder.cpp:1430-1440)
  {
    int ql = 0, lod = 0, nr = 0;
    cx.range_start[nr] = 0;
    cx.range_qlayer[nr] = 0;
    cx.range_lcp[nr] = 0;
    nr++;
    int prev = -1;
    for (int l = 0; l < p->num_lods; l++) {
      const int b = p->num_points_in_lod[l];
      if (b == prev || b >= n || b == 0) {
        prev = b;
        if (b == 0) {
          // index 0 itself is a boundary: the counters step before the
          // first coefficient
          if (cx.range_start[0] == 0 && nr == 1) {
            if (0 == p->num_points_in_lod[ql])
              ql = std::min(p->num_qp_layers - 1, ql + 1);
            if (lod < p->num_lods && 0 == p->num_points_in_lod[lod])
              lod++;
            cx.range_qlayer[0] = ql;
            cx.range_lcp[0] = lod;
          }
        }
        continue;
      }
      prev = b;
      if (b == p->num_points_in_lod[ql])
        ql = std::min(p->num_qp_layers - 1, ql + 1);
      if (lod < p->num_lods && b == p->num_points_in_lod[lod])
        lod++;
      cx.range_start[nr] = b;
      cx.range_qlayer[nr] = ql;
      cx.range_lcp[nr] = std::min(lod, GPCC_MAX_LODS - 1);
      nr++;
    }
    cx.num_ranges = nr;
  }
  cx.lcp_enabled = p->last_component_prediction_enabled_flag && C == 3;
  cx.bitdepth = p->bitdepth;
  cx.num_qp_layers = p->num_qp_layers;
  memcpy(cx.layer_qp, p->layer_qp, sizeof(cx.layer_qp));
  cx.max_qp = p->max_qp;
  cx.fixed_point_qp_offset = p->fixed_point_qp_offset;
  cx.nc = d.nc;
  cx.ni = d.ni;
  cx.nw = d.nw;
  cx.indexes = d.indexes;
  cx.qp_off = d.qp_off;
  cx.attrs = d.attrs;
  cx.coeffs = d.coeffs;
  cx.lcp = d_lcp_io;
  Arena ar;
  ar.base = scratch;
  ar.cap = ~size_t(0);
  const size_t n_ext = (size_t)n + (size_t)n_frame;
  cx.a = ar.take<int64_t>(n_ext * C);
  cx.qw = ar.take<unsigned long long>(n_ext);
  cx.uw = ar.take<unsigned long long>(n_ext);
  cx.up = ar.take<unsigned long long>(n_ext * C);
  cx.lcp_sums = ar.take<long long>(2 * GPCC_MAX_LODS);
  cx.rsqrt = &ctx->d_lut->rsqrt;
  if (n_frame > 0)
    HIP_TRY(h2d_user(ctx, cx.a + (size_t)n * C, h_frame, sizeof(int64_t) * (size_t)n_frame * C, st));

  const int* npl = cx.npl;
  auto grid = [&](int items) { return grid_for(std::max(items, 1), 256); };
  {
    Timer t(ctx, "lift_init");
    lift_init_kernel<<<grid(n), 256, 0, st>>>(cx, encoder);
  }
  {
    Timer t(ctx, "lift_quant_weights");
    if (p->scalable_lifting_enabled_flag) {
      LodSizes t{};
      t.num_lods = p->num_lods;
      for (int l = 0; l < p->num_lods; l++)
        t.npl[l] = p->num_points_in_lod[l];
      quant_weights_scalable_kernel<<<grid(n), 256, 0, st>>>(n, t, cx.qw);
    } else
      for (int l = p->num_lods - 1; l >= 1; l--)
        if (npl[l] > npl[l - 1])
          lift_quant_weights_kernel<<<grid(npl[l] - npl[l - 1]), 256, 0, st>>>(cx, npl[l - 1], npl[l]);
  }
  if (encoder) {
    Timer t(ctx, "lift_forward");
    for (int l = p->num_lods - 1; l >= 1; l--) {
      if (npl[l] == npl[l - 1])
        continue;
      const int cnt = npl[l] - npl[l - 1];
      lift_predict_kernel<C><<<grid(cnt), 256, 0, st>>>(cx, npl[l - 1], npl[l], 1);
      lift_update_scatter_kernel<C><<<grid(cnt), 256, 0, st>>>(cx, npl[l - 1], npl[l]);
      lift_update_apply_kernel<C><<<grid(npl[l - 1]), 256, 0, st>>>(cx, npl[l - 1], 1);
    }
    if (cx.lcp_enabled) {
      lift_lcp_sums_kernel<<<grid(n), 256, 0, st>>>(cx);
      lift_lcp_resolve_kernel<<<1, 64, 0, st>>>(cx);
    }
  }
  {
    Timer t(ctx, "lift_quantise");
    lift_quantise_kernel<C><<<grid(n), 256, 0, st>>>(cx, encoder);
  }
  {
    Timer t(ctx, "lift_inverse");
    for (int l = 1; l < p->num_lods; l++) {
      if (npl[l] == npl[l - 1])
        continue;
      const int cnt = npl[l] - npl[l - 1];
      lift_update_scatter_kernel<C><<<grid(cnt), 256, 0, st>>>(cx, npl[l - 1], npl[l]);
      lift_update_apply_kernel<C><<<grid(npl[l - 1]), 256, 0, st>>>(cx, npl[l - 1], 0);
      lift_predict_kernel<C><<<grid(cnt), 256, 0, st>>>(cx, npl[l - 1], npl[l], 0);
    }
    lift_writeback_kernel<<<grid(n), 256, 0, st>>>(cx);
  }
  HIP_TRY(hipGetLastError());
  return GPCC_OK;
}

size_t
lift_scratch_bytes(int n, int c)
{
  Arena ar;
  ar.take<int64_t>((size_t)n * c);
  ar.take<unsigned long long>(n);
  ar.take<unsigned long long>(n);
  ar.take<unsigned long long>((size_t)n * c);
  ar.take<long long>(2 * GPCC_MAX_LODS);
  return ar.used;
}

int
host_lift(
  gpcc_ctx* ctx, bool encoder, const gpcc_lift_params* p, int n, int c,
  const int32_t* nc, const int32_t* ni, const int32_t* nw,
  const int32_t* indexes, const int32_t* qp_off, int32_t* attrs,
  int32_t* coeffs, int8_t* lcp,
  // attribute inter prediction (null: none): inter_ref [n][3] marks the neighbours that live in
  // the reference frame (ni is then a point index there), attrs_ref [n_frame][c] its attributes
  const int32_t* inter_ref = nullptr, const int32_t* attrs_ref = nullptr, int n_frame = 0)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  int rcode = check_lift_params(p, n, c);
  if (rcode)
    return rcode;
  if (!nc || !ni || !nw || !indexes || !attrs || !coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer");
  const bool lcp_on = c == 3 && p->last_component_prediction_enabled_flag;
  if (lcp_on && !lcp)
    return fail(GPCC_ERR_INVALID_ARG, "lcp_coeffs is null");
  if (inter_ref) {
    if (!attrs_ref || n_frame <= 0 || n_frame > kMaxPoints)
      return fail(GPCC_ERR_INVALID_ARG, "reference frame: null, empty or too large");
    if (c != 1 || p->scalable_lifting_enabled_flag)
      return fail(
        GPCC_ERR_UNSUPPORTED,
        "inter prediction exists in the reference's reflectance lifting driver only, and not with scalable lifting");
  } else
    n_frame = 0;
  for (int i = 0; i < n; i++) {
    if (nc[i] < 0 || nc[i] > 3 || indexes[i] < 0 || indexes[i] >= n)
      return fail(GPCC_ERR_INVALID_ARG, "bad neighbour count / index table");
    for (int j = 0; j < nc[i]; j++) {
      const int32_t v = ni[3 * (size_t)i + j];
      if (inter_ref && inter_ref[3 * (size_t)i + j]) {
        if (v < 0 || v >= n_frame)
          return fail(GPCC_ERR_INVALID_ARG, "a neighbour outside the reference frame");
      } else if (v < 0 || v >= i)
        return fail(GPCC_ERR_INVALID_ARG, "a neighbour does not precede its predictor");
    }
  }
  // neighbours in the reference frame are addressed behind the n predictors (launch_lift)
  std::vector<int32_t> ni_frame;
  std::vector<int64_t> a_frame;
  if (inter_ref) {
    ni_frame.assign(ni, ni + (size_t)n * 3);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < nc[i]; j++)
        if (inter_ref[3 * (size_t)i + j])
          ni_frame[3 * (size_t)i + j] += n;
    ni = ni_frame.data();
    a_frame.resize((size_t)n_frame * c);
    for (size_t t = 0; t < a_frame.size(); t++)
      a_frame[t] = (int64_t)attrs_ref[t] * 256;  // << kFixedPointAttributeShift
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  // one arena: inputs, outputs, scratch
  Arena m;
  auto carve = [&](Arena& ar, LiftDev& d, int32_t*& d_nc, int32_t*& d_ni,
                   int32_t*& d_nw, int32_t*& d_ix, int32_t*& d_qp,
                   int8_t*& d_lcp, char*& scratch) {
    ar.reset();
    d_nc = ar.take<int32_t>(n);
    d_ni = ar.take<int32_t>((size_t)n * 3);
    d_nw = ar.take<int32_t>((size_t)n * 3);
    d_ix = ar.take<int32_t>(n);
    d_qp = qp_off ? ar.take<int32_t>((size_t)n * 2) : nullptr;
    d.attrs = ar.take<int32_t>((size_t)n * c);
    d.coeffs = ar.take<int32_t>((size_t)n * c);
    d_lcp = ar.take<int8_t>(GPCC_MAX_LODS);
    scratch = ar.base ? ar.base + ar.used : nullptr;
    ar.used += lift_scratch_bytes(n + n_frame, c);
  };
  LiftDev d{};
  int32_t *d_nc, *d_ni, *d_nw, *d_ix, *d_qp;
  int8_t* d_lcp;
  char* scratch;
  carve(m, d, d_nc, d_ni, d_nw, d_ix, d_qp, d_lcp, scratch);
  rcode = ensure_arena(ctx, m.used);
  if (rcode)
    return rcode;
  carve(ctx->arena, d, d_nc, d_ni, d_nw, d_ix, d_qp, d_lcp, scratch);
  d.nc = d_nc;
  d.ni = d_ni;
  d.nw = d_nw;
  d.indexes = d_ix;
  d.qp_off = d_qp;
  HIP_TRY(h2d_user(ctx, d_nc, nc, sizeof(int32_t) * n, st));
  HIP_TRY(h2d_user(ctx, d_ni, ni, sizeof(int32_t) * n * 3, st));
  HIP_TRY(h2d_user(ctx, d_nw, nw, sizeof(int32_t) * n * 3, st));
  HIP_TRY(h2d_user(ctx, d_ix, indexes, sizeof(int32_t) * n, st));
  if (qp_off)
    HIP_TRY(h2d_user(ctx, d_qp, qp_off, sizeof(int32_t) * n * 2, st));
  if (encoder) {
    HIP_TRY(h2d_user(ctx, d.attrs, attrs, sizeof(int32_t) * n * c, st));
  } else {
    HIP_TRY(h2d_user(ctx, d.coeffs, coeffs, sizeof(int32_t) * n * c, st));
    if (lcp_on)
      HIP_TRY(hipMemcpyAsync(d_lcp, lcp, GPCC_MAX_LODS, hipMemcpyHostToDevice, st));
  }
  switch (c) {
  case 1: rcode = launch_lift<1>(ctx, encoder, p, n, d, d_lcp, scratch, n_frame, a_frame.data()); break;
  case 2: rcode = launch_lift<2>(ctx, encoder, p, n, d, d_lcp, scratch); break;
  default: rcode = launch_lift<3>(ctx, encoder, p, n, d, d_lcp, scratch); break;
  }
  if (rcode)
    return rcode;
  // (a_frame was staged by h2d_user: through the context's pinned buffer, or copied before the
  // call returned -- the vector may go)
  HIP_TRY(d2h_user(ctx, attrs, d.attrs, sizeof(int32_t) * n * c, st));
  if (encoder) {
    HIP_TRY(d2h_user(ctx, coeffs, d.coeffs, sizeof(int32_t) * n * c, st));
    if (lcp_on)
      HIP_TRY(hipMemcpyAsync(lcp, d_lcp, GPCC_MAX_LODS, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return GPCC_OK;
}


// ---- predicting transform ------------------------------------------------
struct PredDev {
  const int32_t *nc, *ni, *nw, *indexes, *qp_off;
  int32_t *attrs, *values;
};

int rc_scan(gpcc_ctx* ctx, int32_t* a, size_t n, long long* sums);  // (inclusive scan, defined with the recolour entry)

// the predicting encoder's mode decisions did not settle (see pred_kernels.hpp)
int
pred_encoder_unsettled()
{
  return fail(
    GPCC_ERR_UNSUPPORTED,
    "the encoder's choice among direct predictors did not settle within the pass limit: "
    "this slice stays on the reference CPU path");
}

constexpr int kPredMaxPasses = 64;

int
check_pred_params(const gpcc_pred_params* p, int n, int c, bool encoder)
{
  if (!p)
    return fail(GPCC_ERR_INVALID_ARG, "params is null");
  if (n <= 0 || (c != 1 && c != 3))
    return fail(GPCC_ERR_INVALID_ARG, "n <= 0 or attribute count not 1 / 3");
  if (n > (1 << 27))
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^27 points per call");
  if (p->num_lods < 1 || p->num_lods > GPCC_MAX_LODS)
    return fail(GPCC_ERR_INVALID_ARG, "num_lods out of range");
  if (p->num_points_in_lod[p->num_lods - 1] != n)
    return fail(GPCC_ERR_INVALID_ARG, "num_points_in_lod does not end at n");
  for (int l = 1; l < p->num_lods; l++)
    if (p->num_points_in_lod[l] < p->num_points_in_lod[l - 1])
      return fail(GPCC_ERR_INVALID_ARG, "num_points_in_lod not ascending");
  if (p->num_qp_layers < 1 || p->num_qp_layers > GPCC_MAX_QP_LAYERS)
    return fail(GPCC_ERR_INVALID_ARG, "num_qp_layers out of range");
  if (p->bitdepth < 1 || p->bitdepth > 16)
    return fail(GPCC_ERR_INVALID_ARG, "bitdepth out of range");
  if (p->max_num_direct_predictors < 0 || p->max_num_direct_predictors > 3)
    return fail(GPCC_ERR_INVALID_ARG, "max_num_direct_predictors out of range");
  if (p->max_num_detail_levels < p->num_lods || p->max_num_detail_levels > GPCC_MAX_LODS)
    return fail(GPCC_ERR_INVALID_ARG, "max_num_detail_levels out of range");
  return GPCC_OK;
}

size_t
pred_scratch_bytes(int n, int n_frame = 0)
{
  Arena ar;
  ar.take<int32_t>((size_t)n + n_frame);
  ar.take<int32_t>((size_t)n + n_frame);
  ar.take<unsigned long long>((size_t)n + n_frame);
  ar.take<unsigned long long>((size_t)n + n_frame);
  ar.take<uint32_t>((size_t)n * 4);
  ar.take<int32_t>(64);
  ar.take<unsigned long long>(GPCC_MAX_LODS * 18);
  // encoder with direct predictors: rate model per predictor, source copy, previous values,
  // ranks / events / event states of the probResGt1 recurrence, scan sums
  ar.take<int32_t>((size_t)n * 6);
  ar.take<int32_t>((size_t)n * 3);
  ar.take<int32_t>((size_t)n * 3);
  ar.take<int32_t>((size_t)n + 1);
  ar.take<uint8_t>((size_t)n + 1);
  ar.take<int32_t>((size_t)n + 1);
  ar.take<long long>((size_t)n / kKdScanBlock + 2);
  ar.take<int32_t>((size_t)n_frame + 1);
  return ar.used;
}

__global__ __launch_bounds__(256) void
fill_u64_kernel(unsigned long long* p, unsigned long long v, int n)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    p[i] = v;
}

template<int C>
int
launch_pred(
  gpcc_ctx* ctx, bool encoder, const gpcc_pred_params* p, int n, const PredDev& d,
  int8_t* d_icp, char* scratch,
  // attribute inter prediction (n_frame > 0, one component): h_frame [n_frame] the reference
  // frame's reflectances (host memory); the caller has pointed the neighbours that live in
  // that frame at n + r (PredCtx::frame_attr)
  int n_frame = 0, const int32_t* h_frame = nullptr)
{
  hipStream_t st = ctx->stream;
  PredCtx cx{};
  cx.n = n;
  cx.c = C;
  cx.num_lods = p->num_lods;
  for (int l = 0; l < p->num_lods; l++)
    cx.npl[l] = p->num_points_in_lod[l];
  // The reference's running counters step when the predictor index meets
  // numPointsInLod[counter] (one step per index, so a repeated boundary
  // stalls them): quantLayer and `lod` of the coding loops
  // (AttributeEncoder.cpp:1108-1121, AttributeDecoder.cpp:476-501) before the
  // index is processed, `lod` of computeInterComponentPredictionCoeffs
  // (:1033-1056) after index npl[lod] - 1.  Replayed over the distinct
  // boundaries.
  {
    int ql = 0, lod = 0, est = 0, nr = 1;
    std::vector<int> bs(p->num_points_in_lod, p->num_points_in_lod + p->num_lods);
    std::sort(bs.begin(), bs.end());
    bs.erase(std::unique(bs.begin(), bs.end()), bs.end());
    cx.range_start[0] = 0;
    for (int b : bs) {
      if (b >= n)
        break;
      if (ql < p->num_lods && b == p->num_points_in_lod[ql])
        ql = std::min(p->num_qp_layers - 1, ql + 1);
      if (lod < p->num_lods && b == p->num_points_in_lod[lod])
        lod++;
      if (b > 0 && est < p->num_lods && b == p->num_points_in_lod[est])
        est++;
      if (b > 0) {
        cx.range_start[nr] = b;
        nr++;
      }
      cx.range_qlayer[nr - 1] = ql;
      cx.range_lod[nr - 1] = std::min(lod, GPCC_MAX_LODS - 1);
      cx.range_est[nr - 1] = std::min(est, GPCC_MAX_LODS - 1);
    }
    cx.num_ranges = nr;
    cx.est_resolved = est < p->num_lods && p->num_points_in_lod[est] == n ? est + 1 : est;
  }
  cx.max_levels = p->max_num_detail_levels;
  cx.bitdepth = p->bitdepth;
  cx.num_qp_layers = p->num_qp_layers;
  memcpy(cx.layer_qp, p->layer_qp, sizeof(cx.layer_qp));
  cx.max_qp = p->max_qp;
  cx.max_direct = p->max_num_direct_predictors;
  cx.avg_disabled = p->direct_avg_predictor_disabled_flag != 0;
  cx.threshold = p->adaptive_prediction_threshold;
  cx.icp_enabled = C == 3 && p->inter_component_prediction_enabled_flag;
  for (int k = 0; k < 3; k++)
    cx.qnw[k] = p->quant_neigh_weight[k];
  cx.nc = d.nc;
  cx.ni = d.ni;
  cx.nw = d.nw;
  cx.indexes = d.indexes;
  cx.qp_off = d.qp_off;
  cx.attrs = d.attrs;
  cx.values = d.values;
  cx.icp = d_icp;
  Arena ar;
  ar.base = scratch;
  ar.cap = ~size_t(0);
  cx.indeg = ar.take<int32_t>((size_t)n + n_frame);
  cx.recv = ar.take<int32_t>((size_t)n + n_frame);
  cx.acc = ar.take<unsigned long long>((size_t)n + n_frame);
  cx.qw = ar.take<unsigned long long>((size_t)n + n_frame);
  cx.rec = ar.take<uint32_t>((size_t)n * 4);
  int32_t* small = ar.take<int32_t>(64);
  cx.ticket = small;
  cx.error = small + 8;
  cx.wide = small + 16;
  cx.packed_ok = cx.qnw[0] >= 0 && cx.qnw[1] >= 0 && cx.qnw[2] >= 0 && cx.qnw[0] + cx.qnw[1] + cx.qnw[2] < 256;
  cx.icp_sums = ar.take<unsigned long long>(GPCC_MAX_LODS * 18);
  // indeg .. rec, tickets, sums: one clear
  HIP_TRY(hipMemsetAsync(scratch, 0, ar.used, st));
  cx.tag = 1;
  const bool deciding = encoder && cx.max_direct > 0;
  int32_t* rm = ar.take<int32_t>((size_t)n * 6);
  int32_t* src_copy = ar.take<int32_t>((size_t)n * 3);
  int32_t* prev_values = ar.take<int32_t>((size_t)n * 3);
  int32_t* ev_rank = ar.take<int32_t>((size_t)n + 1);
  uint8_t* ev_up = ar.take<uint8_t>((size_t)n + 1);
  int32_t* ev_state = ar.take<int32_t>((size_t)n + 1);
  long long* scan_sums = ar.take<long long>((size_t)n / kKdScanBlock + 2);
  int32_t* d_frame = ar.take<int32_t>((size_t)n_frame + 1);
  if (n_frame > 0) {
    HIP_TRY(h2d_user(ctx, d_frame, h_frame, sizeof(int32_t) * (size_t)n_frame, st));
    cx.frame_attr = d_frame;
  }
  // the DAG pass: its inter-prediction build where neighbours may live in the reference frame
#define GPCC_PRED_DAG(ENC)                                                          \
  do {                                                                              \
    if (n_frame > 0)                                                                \
      pred_dag_kernel<C, ENC, true><<<std::max(pgrid, 1), 256, 0, st>>>(cx);        \
    else                                                                            \
      pred_dag_kernel<C, ENC><<<std::max(pgrid, 1), 256, 0, st>>>(cx);              \
  } while (0)
  if (deciding) {
    // log2 of every integer the rate estimate can ask for, from THIS host's libm (the
    // reference's own log2): once per context
    int rl = ensure_log2(ctx);
    if (rl)
      return rl;
    HIP_TRY(hipMemcpyAsync(src_copy, d.attrs, sizeof(int32_t) * (size_t)n * C, hipMemcpyDeviceToDevice, st));
    cx.src = src_copy;
    cx.rm = rm;
    cx.log2tab = ctx->d_log2;
  }
  auto grid = [&](int items) { return grid_for(std::max(items, 1), 256); };
  // persistent kernels: as many wavefronts as the device keeps resident
  const int pgrid = (int)std::min<int64_t>(2048, ((int64_t)n + 255) / 256);
  {
    Timer t(ctx, "pred_indegree");
    pred_indegree_kernel<<<grid(n), 256, 0, st>>>(cx);
  }
  if (p->scalable_lifting_enabled_flag) {
    // computeQuantizationWeightsScalable: by level of detail, quant_neigh_weight is not read
    Timer tm(ctx, "pred_quant_weights");
    LodSizes t{};
    t.num_lods = p->num_lods;
    for (int l = 0; l < p->num_lods; l++)
      t.npl[l] = p->num_points_in_lod[l];
    quant_weights_scalable_kernel<<<grid(n), 256, 0, st>>>(n, t, cx.qw);
  } else if (cx.qnw[0] || cx.qnw[1] || cx.qnw[2]) {
    Timer t(ctx, "pred_quant_weights");
    pred_quant_weights_kernel<<<std::max(pgrid, 1), 256, 0, st>>>(cx);
  } else {
    // no shares: every weight is 1 << kFixedPointWeightShift
    Timer t(ctx, "pred_quant_weights");
    fill_u64_kernel<<<grid(n), 256, 0, st>>>(cx.qw, 256ull, n);
  }
  if (encoder && cx.icp_enabled) {
    Timer t(ctx, "pred_icp");
    pred_icp_sums_kernel<<<std::min(grid(n), 1024), 256, 0, st>>>(cx);
    pred_icp_resolve_kernel<<<1, 64, 0, st>>>(cx);
  }
  if (!deciding) {
    Timer t(ctx, "pred_dag");
    if (encoder)
      GPCC_PRED_DAG(true);
    else
      GPCC_PRED_DAG(false);
  } else {
    // the DAG pass and the rate model's trajectory, iterated to their fixed point
    int32_t* flag = small + 24;
    {
      Timer t(ctx, "pred_rate");
      pred_rate_init_kernel<<<grid(n), 256, 0, st>>>(rm, n);
    }
    bool settled = false;
    int passes = 0;
    for (int pass = 0; pass < kPredMaxPasses && !settled; pass++) {
      passes++;
      cx.tag = (uint32_t)(pass + 1);
      HIP_TRY(hipMemsetAsync(cx.ticket, 0, 8 * sizeof(int32_t), st));
      HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int32_t), st));
      {
        Timer t(ctx, "pred_dag");
        GPCC_PRED_DAG(true);
      }
      int32_t changed = 1;
      {
        Timer t(ctx, "pred_rate");
        pred_values_diff_kernel<<<grid(n), 256, 0, st>>>(cx.values, prev_values, (size_t)n * C, flag);
        if (pass > 0) {
          HIP_TRY(hipMemcpyAsync(&changed, flag, sizeof(int32_t), hipMemcpyDeviceToHost, st));
          HIP_TRY(hipStreamSynchronize(st));
        }
        if (pass > 0 && !changed) {
          settled = true;
        } else {
          const int chunks = n / kRateChunk + 1;
          for (int k = 0; k < C; k++) {
            // probResGt0: an event at every predictor
            pred_rate_scan_kernel<<<(chunks + 63) / 64, 64, 0, st>>>(cx.values + k, C, nullptr, n, nullptr, rm + k, 6, 0);
            // probResGt1: the non-zero values, compacted
            pred_rate_flags_kernel<<<grid(n), 256, 0, st>>>(cx.values, n, C, k, ev_rank);
            int r = rc_scan(ctx, ev_rank, (size_t)n + 1, scan_sums);
            if (r)
              return r;
            pred_rate_events_kernel<<<grid(n), 256, 0, st>>>(cx.values, n, C, k, ev_rank, ev_up);
            pred_rate_scan_kernel<<<(chunks + 63) / 64, 64, 0, st>>>(nullptr, 0, ev_up, n, ev_rank + n, ev_state, 1, 1);
            pred_rate_gather_kernel<<<grid(n), 256, 0, st>>>(ev_rank, ev_state, n, k, rm);
          }
        }
      }
    }
    ctx->pred_passes = passes;
    ctx->pred_pass_stats[0]++;
    ctx->pred_pass_stats[1] += passes;
    ctx->pred_pass_stats[2] = std::max<int64_t>(ctx->pred_pass_stats[2], passes);
    ctx->pred_pass_stats[3] += settled ? 0 : 1;
    if (!settled) {
      // the caller's attrs hold a reconstruction that is not the reference's: restore the source
      HIP_TRY(hipMemcpyAsync(d.attrs, src_copy, sizeof(int32_t) * (size_t)n * C, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipStreamSynchronize(st));
      return pred_encoder_unsettled();
    }
  }
#undef GPCC_PRED_DAG
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ctx->h_error, cx.error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  return GPCC_OK;
}

int
pred_check_error(gpcc_ctx* ctx)
{
  // after a stream synchronisation: launch_pred copied the kernels' error word
  if (*ctx->h_error) {
    *ctx->h_error = 0;
    return fail(GPCC_ERR_HIP, "a dependency wait in the predicting transform expired");
  }
  return GPCC_OK;
}

int
host_pred(
  gpcc_ctx* ctx, bool encoder, const gpcc_pred_params* p, int n, int c, const int32_t* nc,
  const int32_t* ni, const int32_t* nw, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, int32_t* values, int8_t* icp,
  // attribute inter prediction (null: none), as host_lift
  const int32_t* inter_ref = nullptr, const int32_t* attrs_ref = nullptr, int n_frame = 0)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  int rcode = check_pred_params(p, n, c, encoder);
  if (rcode)
    return rcode;
  if (!nc || !ni || !nw || !indexes || !attrs || !values)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer");
  const bool icp_on = c == 3 && p->inter_component_prediction_enabled_flag;
  if (icp_on && !icp)
    return fail(GPCC_ERR_INVALID_ARG, "icp_coeffs is null");
  if (inter_ref) {
    if (!attrs_ref || n_frame <= 0 || n_frame > (1 << 27))
      return fail(GPCC_ERR_INVALID_ARG, "reference frame: null, empty or too large");
    if (c != 1 || p->scalable_lifting_enabled_flag)
      return fail(
        GPCC_ERR_UNSUPPORTED,
        "inter prediction exists in the reference's reflectance predicting driver only, and not over a scalable structure");
  } else
    n_frame = 0;
  for (int i = 0; i < n; i++) {
    if (nc[i] < 0 || nc[i] > 3 || indexes[i] < 0 || indexes[i] >= n)
      return fail(GPCC_ERR_INVALID_ARG, "bad neighbour count / index table");
    for (int j = 0; j < nc[i]; j++) {
      const int32_t v = ni[3 * (size_t)i + j];
      if (inter_ref && inter_ref[3 * (size_t)i + j]) {
        if (v < 0 || v >= n_frame)
          return fail(GPCC_ERR_INVALID_ARG, "a neighbour outside the reference frame");
      } else if (v < 0 || v >= i)
        return fail(GPCC_ERR_INVALID_ARG, "a neighbour does not precede its predictor");
    }
  }
  // neighbours in the reference frame are addressed behind the n predictors (launch_pred)
  std::vector<int32_t> ni_frame;
  if (inter_ref) {
    ni_frame.assign(ni, ni + (size_t)n * 3);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < nc[i]; j++)
        if (inter_ref[3 * (size_t)i + j])
          ni_frame[3 * (size_t)i + j] += n;
    ni = ni_frame.data();
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  PredDev d{};
  int32_t *d_nc = nullptr, *d_ni = nullptr, *d_nw = nullptr, *d_ix = nullptr, *d_qp = nullptr;
  int8_t* d_icp = nullptr;
  char* scratch = nullptr;
  auto carve = [&](Arena& ar) {
    ar.reset();
    d_nc = ar.take<int32_t>(N);
    d_ni = ar.take<int32_t>(N * 3);
    d_nw = ar.take<int32_t>(N * 3);
    d_ix = ar.take<int32_t>(N);
    d_qp = qp_off ? ar.take<int32_t>(N * 2) : nullptr;
    d.attrs = ar.take<int32_t>(N * c);
    d.values = ar.take<int32_t>(N * c);
    d_icp = ar.take<int8_t>(GPCC_MAX_LODS * 3);
    scratch = ar.base ? ar.base + ar.used : nullptr;
    ar.used += pred_scratch_bytes(n, n_frame);
  };
  Arena m;
  carve(m);
  rcode = ensure_arena(ctx, m.used);
  if (rcode)
    return rcode;
  carve(ctx->arena);
  d.nc = d_nc;
  d.ni = d_ni;
  d.nw = d_nw;
  d.indexes = d_ix;
  d.qp_off = d_qp;
  HIP_TRY(h2d_user(ctx, d_nc, nc, sizeof(int32_t) * N, st));
  HIP_TRY(h2d_user(ctx, d_ni, ni, sizeof(int32_t) * N * 3, st));
  HIP_TRY(h2d_user(ctx, d_nw, nw, sizeof(int32_t) * N * 3, st));
  HIP_TRY(h2d_user(ctx, d_ix, indexes, sizeof(int32_t) * N, st));
  if (qp_off)
    HIP_TRY(h2d_user(ctx, d_qp, qp_off, sizeof(int32_t) * N * 2, st));
  if (encoder) {
    HIP_TRY(h2d_user(ctx, d.attrs, attrs, sizeof(int32_t) * N * c, st));
  } else {
    HIP_TRY(h2d_user(ctx, d.values, values, sizeof(int32_t) * N * c, st));
    if (icp_on)
      HIP_TRY(hipMemcpyAsync(d_icp, icp, GPCC_MAX_LODS * 3, hipMemcpyHostToDevice, st));
  }
  rcode = c == 1 ? launch_pred<1>(ctx, encoder, p, n, d, d_icp, scratch, n_frame, attrs_ref)
                 : launch_pred<3>(ctx, encoder, p, n, d, d_icp, scratch);
  if (rcode)
    return rcode;
  HIP_TRY(d2h_user(ctx, attrs, d.attrs, sizeof(int32_t) * N * c, st));
  if (encoder) {
    HIP_TRY(d2h_user(ctx, values, d.values, sizeof(int32_t) * N * c, st));
    if (icp_on)
      HIP_TRY(hipMemcpyAsync(icp, d_icp, GPCC_MAX_LODS * 3, hipMemcpyDeviceToHost, st));
  }
  HIP_TRY(hipStreamSynchronize(st));
  return pred_check_error(ctx);
}

}  // namespace

// =========================================================================
extern "C" {

void
gpcc_raht_set_prediction_weights(gpcc_raht_params* p, const int32_t w[5])
{
  const int32_t child[12] = {w[4], w[4], w[3], w[4], w[3], w[3],
                             w[4], w[4], w[4], w[4], w[4], w[4]};
  const int32_t parent[19] = {w[0], w[1], w[1], w[1], w[2], w[2], w[2],
                              w[2], w[2], w[1], w[2], w[1], w[1], w[2],
                              w[2], w[2], w[2], w[2], w[2]};
  memcpy(p->pred_weight_child, child, sizeof(child));
  memcpy(p->pred_weight_parent, parent, sizeof(parent));
}

int
gpcc_abi_version(void)
{
  return GPCC_ABI_VERSION;
}

const char*
gpcc_last_error(void)
{
  return g_last_error.c_str();
}

extern "C" void
gpcc_clear_last_error(void)
{
  g_last_error.clear();
}

int
gpcc_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
    return 0;
  return n;
}

int
gpcc_ctx_create(int device, void* stream, gpcc_ctx** out)
{
  if (!out)
    return fail(GPCC_ERR_INVALID_ARG, "out is null");
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
    return fail(GPCC_ERR_NO_DEVICE, "no HIP device visible");
  if (device < 0 || device >= n)
    return fail(GPCC_ERR_INVALID_ARG, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(
      GPCC_ERR_NO_DEVICE,
      std::string("device is ") + prop.gcnArchName
        + ", this library is built for gfx950 only");
  gpcc_ctx* ctx = new gpcc_ctx();
  ctx->device = device;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete ctx;
      return fail(GPCC_ERR_HIP, hipGetErrorString(e));
    }
    ctx->own_stream = true;
  }
  if (guarded_malloc((void**)&ctx->d_lut, sizeof(SharedLut), "lut") != hipSuccess) {
    gpcc_ctx_destroy(ctx);
    return fail(GPCC_ERR_OUT_OF_MEMORY, "hipMalloc(lut)");
  }
  lut_init_kernel<<<1, 256, 0, ctx->stream>>>(ctx->d_lut);
  if (guarded_malloc((void**)&ctx->d_error, sizeof(int32_t), "error word") != hipSuccess
      || hipMemsetAsync(ctx->d_error, 0, sizeof(int32_t), ctx->stream) != hipSuccess
      || hipHostMalloc((void**)&ctx->h_error, sizeof(int32_t)) != hipSuccess
      || hipHostMalloc((void**)&ctx->h_stats, sizeof(TreeStats)) != hipSuccess
      || hipHostMalloc((void**)&ctx->h_cxtab, sizeof(CxLevelTab)) != hipSuccess
      || hipEventCreateWithFlags(&ctx->ev_stats, hipEventDisableTiming) != hipSuccess) {
    gpcc_ctx_destroy(ctx);
    return fail(GPCC_ERR_OUT_OF_MEMORY, "context bookkeeping allocations failed");
  }
  *ctx->h_error = 0;
  memset(ctx->h_stats, 0, sizeof(TreeStats));
  *out = ctx;
  return GPCC_OK;
}

void
gpcc_ctx_destroy(gpcc_ctx* ctx)
{
  if (!ctx)
    return;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  for (auto* l : ctx->lanes)
    gpcc_ctx_destroy(l);
  ctx->lanes.clear();
  if (ctx->ev_lanes)
    hipEventDestroy(ctx->ev_lanes);
  for (auto& sp : ctx->spans) {
    hipEventDestroy(sp.a);
    hipEventDestroy(sp.b);
  }
  for (auto e : ctx->event_pool)
    hipEventDestroy(e);
  ctx->arena.check_bands("context destroyed");
  if (ctx->arena.base)
    guarded_free(ctx->arena.base);
  for (auto& b : ctx->pool)
    guarded_free(b.ptr);
  ctx->pool.clear();
  if (ctx->d_lut)
    guarded_free(ctx->d_lut);
  if (ctx->h_error)
    hipHostFree(ctx->h_error);
  if (ctx->d_error)
    guarded_free(ctx->d_error);
  if (ctx->h_stats)
    hipHostFree(ctx->h_stats);
  if (ctx->h_cxtab)
    hipHostFree(ctx->h_cxtab);
  if (ctx->h_kd)
    hipHostFree(ctx->h_kd);
  if (ctx->kd_event)
    hipEventDestroy(ctx->kd_event);
  if (ctx->inter_event)
    hipEventDestroy(ctx->inter_event);
  if (ctx->kd_stream)
    hipStreamDestroy(ctx->kd_stream);
  if (ctx->d_log2)
    guarded_free(ctx->d_log2);
  if (ctx->sweep_mem)
    guarded_free(ctx->sweep_mem);
  if (ctx->ev_stats)
    hipEventDestroy(ctx->ev_stats);
  if (ctx->h_pinned)
    hipHostFree(ctx->h_pinned);
  if (ctx->h_cx_stage)
    hipHostFree(ctx->h_cx_stage);
  if (ctx->h_small)
    hipHostFree(ctx->h_small);
  if (ctx->h_bounce)
    hipHostFree(ctx->h_bounce);
  for (int h = 0; h < 2; h++)
    if (ctx->ev_bounce[h])
      hipEventDestroy(ctx->ev_bounce[h]);
  if (ctx->own_stream)
    hipStreamDestroy(ctx->stream);
  delete ctx;
}

int
gpcc_ctx_synchronize(gpcc_ctx* ctx)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (guard_mode())
    guard_check_context(ctx, "gpcc_ctx_synchronize");
  return check_device_error(ctx);
}

// {arena (re)allocations, pool misses, compact-pass staging reallocations, level-pass staging reallocations}
extern "C" int
gpcc_debug_alloc_events(const gpcc_ctx* ctx, long long out[4])
{
  if (!ctx || !out)
    return fail(GPCC_ERR_INVALID_ARG, "ctx / out is null");
  for (int i = 0; i < 4; i++)
    out[i] = ctx->alloc_events[i];
  return GPCC_OK;
}

// 1: the library was built with the round-5 experiments compiled in (GPCC_LINKS / GPCC_SUB_CLAIM work)
extern "C" int
gpcc_debug_has_experiments(void)
{
  return GPCC_EXPERIMENTS;
}

// bands compared so far in this process (0 unless GPCC_GUARD=1): lets a test tier prove the guards were armed
extern "C" unsigned long long
gpcc_debug_guard_checks(void)
{
  return g_guard_checks.load();
}

// rate_sum_kernel (raht_inter.hpp) alone, for tests/test_gpu_rate_sum.py: the two estimates' sums over
// terms[2][count] doubles, as the inter encoder's per-layer decision accumulates them
extern "C" int
gpcc_debug_rate_sum(gpcc_ctx* ctx, const double* terms, int32_t count, double out[2])
{
  if (!ctx || !out || count < 0 || (count > 0 && !terms))
    return fail(GPCC_ERR_INVALID_ARG, "bad arguments");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  void *d_terms = nullptr, *d_rs = nullptr, *d_err = nullptr;
  // (the blocks go back to the pool on every path)
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, &d_terms, std::max<size_t>(2 * (size_t)count * sizeof(double), 16)));
    HIP_TRY(pool_malloc(ctx, &d_rs, sizeof(RateState)));
    HIP_TRY(pool_malloc(ctx, &d_err, 16));
    HIP_TRY(hipMemsetAsync(d_rs, 0, sizeof(RateState), st));
    HIP_TRY(hipMemsetAsync(d_err, 0, 16, st));
    // (through the context's pinned buffers, as every entry: the runtime never pins the caller's pages)
    if (count > 0)
      HIP_TRY(h2d_user(ctx, d_terms, terms, 2 * (size_t)count * sizeof(double), st));
    HIP_TRY(small_reset(ctx));
    RateCtx cx{};
    cx.tv.error = (int32_t*)d_err;
    cx.n = count;
    cx.a = 0;
    cx.b = count;
    cx.c = 1;
    cx.term = (double*)d_terms;
    cx.rs = (RateState*)d_rs;
    hipLaunchKernelGGL(rate_sum_kernel, dim3(2), dim3(kAcSumThreads), 0, st, cx);
    HIP_TRY(hipGetLastError());
    void* p_rs = nullptr;
    HIP_TRY(small_d2h(ctx, &p_rs, d_rs, sizeof(RateState), st));
    HIP_TRY(hipStreamSynchronize(st));
    const RateState& h = *(const RateState*)p_rs;
    out[0] = h.bits[0];
    out[1] = h.bits[1];
    return GPCC_OK;
  };
  const int rc = run();
  for (void* p : {d_terms, d_rs, d_err})
    if (p)
      pool_free(ctx, p);
  return rc;
}

// guard mode's own test: writes 16 bytes past a pool block (mode 0) or past a sub-allocation of the arena
// (mode 1) and releases / re-carves it -- with GPCC_GUARD=1 the process must stop with the GUARD BAND message
// (tests/test_gpu_guard.py runs this in a child process); without guard mode nothing is checked.
extern "C" int
gpcc_debug_guard_selftest(gpcc_ctx* ctx, int mode)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  HIP_TRY(hipSetDevice(ctx->device));
  if (!guard_mode())  // (without the bands the write below would leave the allocation: nothing to test)
    return GPCC_OK;
  if (mode == 0) {
    void* p = nullptr;
    HIP_TRY(pool_malloc(ctx, &p, 1024));
    // (the pool rounds a request up by an eighth: the band starts behind the block's capacity)
    size_t cap = 0;
    for (auto& b : ctx->pool)
      if (b.ptr == p)
        cap = b.cap;
    cap = (cap + 255) & ~size_t(255);
    HIP_TRY(hipMemsetAsync((char*)p + cap, 0, 16, ctx->stream));
    pool_free(ctx, p);
  } else {
    int r = ensure_arena(ctx, 1 << 16);
    if (r)
      return r;
    ctx->arena.reset();
    char* a = ctx->arena.take<char>(1000);
    ctx->arena.take<char>(1000);
    HIP_TRY(hipMemsetAsync(a + 1024, 0, 16, ctx->stream));
    r = ensure_arena(ctx, 1 << 16);
    if (r)
      return r;
  }
  return GPCC_OK;
}

int
gpcc_ctx_stats(const gpcc_ctx* ctx, gpcc_ctx_stats_t* out)
{
  if (!ctx || !out)
    return fail(GPCC_ERR_INVALID_ARG, "ctx / out is null");
  *out = ctx->stats;
  return GPCC_OK;
}

extern "C" int
gpcc_ctx_pred_pass_stats(const gpcc_ctx* ctx, int64_t out[4])
{
  if (!ctx || !out)
    return fail(GPCC_ERR_INVALID_ARG, "ctx or out is null");
  // (the lanes of the device tier are contexts of their own: their slices are added up)
  for (int k = 0; k < 4; k++)
    out[k] = ctx->pred_pass_stats[k];
  for (const gpcc_ctx* lane : ctx->lanes) {
    out[0] += lane->pred_pass_stats[0];
    out[1] += lane->pred_pass_stats[1];
    out[2] = std::max(out[2], lane->pred_pass_stats[2]);
    out[3] += lane->pred_pass_stats[3];
  }
  return GPCC_OK;
}

size_t
gpcc_ctx_workspace_bytes(const gpcc_ctx* ctx)
{
  return ctx ? ctx->arena.cap : 0;
}

// Grow everything a RAHT call of up to max_points points in up to max_slices slices with max_c
// components allocates on demand -- the arena for the largest workspace any flag combination carves,
// the pooled record block of the coarse-level sweep, the pinned staging blocks -- touch the device
// memory once, and return with the device idle.  A transform that follows allocates nothing.
int
gpcc_ctx_reserve(gpcc_ctx* ctx, int64_t max_points, int32_t max_slices, int32_t max_c)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (max_points < 1 || max_points > kMaxPoints || max_slices < 1 || max_slices > max_points || max_c < 1 || max_c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "gpcc_ctx_reserve: sizes out of range");
  HIP_TRY(hipSetDevice(ctx->device));
  const int bits = ctx->morton_bits > 0 ? std::min(ctx->morton_bits, 63) : 63;
  const int nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  size_t need = 0;
  // the level kernels' plans: lossy encoder and integer-Haar encoder with sub-node prediction and region QPs
  // (the decoder's and the other flag combinations' workspaces are subsets)
  for (int haar = 0; haar < 2; haar++) {
    Plan pl;
    pl.n = (int)max_points;
    pl.s = max_slices;
    pl.c = max_c;
    pl.nlev = nlev;
    pl.encoder = true;
    pl.haar = haar != 0;
    pl.has_qp = true;
    pl.lossy = !pl.haar;
    pl.sub = true;
    pl.num_rtiles = (int)((max_points + kRdoqTile - 1) / kRdoqTile) + max_slices;
    Arena measure;
    carve(measure, pl);
    need = std::max(need, measure.used);
  }
  {
    // the compact level pass (sub-node prediction off)
    CxWork w;
    w.n = (int)max_points;
    w.s = max_slices;
    w.c = max_c;
    w.nlev = nlev;
    w.encoder = true;
    w.f64 = false;
    w.links = links_enabled();
    size_t used = 0;
    cx_carve(
      [&](size_t bytes) {
        used += ((bytes + 255) & ~size_t(255)) + (guard_mode() ? kGuardInner : 0);
        return (char*)nullptr;
      },
      w);
    need = std::max(need, used);
  }
  // (GPCC_RESERVE_TOUCH=0: allocate only, do not write the memory -- for measurements)
  static const bool touch = [] {
    const char* e = getenv("GPCC_RESERVE_TOUCH");
    return !(e && e[0] == '0');
  }();
  int rcode = ensure_arena(ctx, need);
  if (rcode)
    return rcode;
  if (touch)
    HIP_TRY(hipMemsetAsync(ctx->arena.base, 0, ctx->arena.cap, ctx->stream));
  // the sweep's records (raht_sweep.hpp): at most kSweepDefaultParents parents per slice and level
  rcode = ensure_sweep_mem(ctx, sweep_rec_bytes((int64_t)max_slices * nlev * kSweepDefaultParents, max_c));
  if (rcode)
    return rcode;
  if (touch)
    HIP_TRY(hipMemsetAsync(ctx->sweep_mem, 0, ctx->sweep_cap, ctx->stream));
  {
    // the host tier's device buffers for a slice of max_points points (host_transform): Morton codes,
    // attributes, coefficients, region QP offsets -- into the pool, where the calls find them
    const size_t n = (size_t)max_points;
    void* blk[4] = {nullptr, nullptr, nullptr, nullptr};
    const size_t sz[4] = {8 * n, 4 * n * (size_t)max_c, 4 * n * (size_t)max_c, 8 * n};
    for (int i = 0; i < 4; i++)
      if (pool_malloc(ctx, &blk[i], sz[i]) != hipSuccess)
        blk[i] = nullptr;
    for (int i = 0; i < 4; i++)
      if (blk[i]) {
        if (touch)
          hipMemsetAsync(blk[i], 0, sz[i], ctx->stream);
        pool_free(ctx, blk[i]);
      }
  }
  // pinned staging of the level kernels and of the compact level pass (launch_transform / launch_cx)
  const size_t stage_bytes = sizeof(gpcc_raht_params) + 2 * ((size_t)max_slices + 1) * sizeof(int32_t)
    + 2 * (size_t)nlev * sizeof(void*) + 64;
  if (ctx->h_pinned_cap < stage_bytes) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned)
      HIP_TRY(hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->h_pinned_cap = 0;
    HIP_TRY(hipHostMalloc(&ctx->h_pinned, stage_bytes * 2));
    ctx->alloc_events[3]++;
    ctx->h_pinned_cap = stage_bytes * 2;
  }
  const size_t cx_stage = (sizeof(gpcc_raht_params) + ((size_t)max_slices + 1) * sizeof(int32_t) + 64 + 255) & ~size_t(255);
  if (ctx->h_cx_stage_cap < cx_stage) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->h_cx_stage)
      HIP_TRY(hipHostFree(ctx->h_cx_stage));
    ctx->h_cx_stage = nullptr;
    ctx->h_cx_stage_cap = 0;
    HIP_TRY(hipHostMalloc(&ctx->h_cx_stage, cx_stage * 4));
    ctx->alloc_events[2]++;
    ctx->h_cx_stage_cap = cx_stage * 2;
  }
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return GPCC_OK;
}

int
gpcc_ctx_set_fast_arith(gpcc_ctx* ctx, int32_t on)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  ctx->fast_arith = on != 0;
  return GPCC_OK;
}

int
gpcc_ctx_set_morton_bits(gpcc_ctx* ctx, int32_t bits)
{
  if (!ctx || bits < 0 || bits > 63)
    return fail(GPCC_ERR_INVALID_ARG, "bad ctx / bits");
  ctx->morton_bits = bits;
  return GPCC_OK;
}

int
gpcc_ctx_set_profiling(gpcc_ctx* ctx, int enable)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  ctx->profiling = enable != 0;
  return GPCC_OK;
}

int
gpcc_ctx_kernel_times(gpcc_ctx* ctx, gpcc_kernel_time* out, int32_t max_entries)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (hipStreamSynchronize(ctx->stream) != hipSuccess)
    return fail(GPCC_ERR_HIP, "synchronize failed");
  std::vector<std::pair<const char*, std::pair<double, int>>> acc;
  for (auto& sp : ctx->spans) {
    float ms = 0;
    hipEventElapsedTime(&ms, sp.a, sp.b);
    bool found = false;
    for (auto& a : acc)
      if (!strcmp(a.first, sp.name)) {
        a.second.first += ms;
        a.second.second++;
        found = true;
      }
    if (!found)
      acc.push_back({sp.name, {ms, 1}});
    ctx->event_pool.push_back(sp.a);
    ctx->event_pool.push_back(sp.b);
  }
  ctx->spans.clear();
  int nout = 0;
  for (auto& a : acc) {
    if (nout >= max_entries)
      break;
    out[nout].name = a.first;
    out[nout].total_ms = a.second.first;
    out[nout].launches = a.second.second;
    nout++;
  }
  return nout;
}

static int
gpcc_raht_forward_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c)
{
  return host_transform(ctx, params, true, morton, qp_off, attrs, coeffs, n, c);
}

static int
gpcc_raht_inverse_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, const int32_t* coeffs, int32_t n,
  int32_t c)
{
  return host_transform(
    ctx, params, false, morton, qp_off, attrs, const_cast<int32_t*>(coeffs), n, c);
}

static int
gpcc_dev_raht_forward_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, void* d_coeffs, int32_t c)
{
  return dev_transform(
    ctx, params, true, num_slices, offsets, d_morton, d_qp_off, d_attrs,
    d_coeffs, c, ctx ? ctx->morton_bits : 0);
}

static int
gpcc_dev_raht_inverse_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, const void* d_coeffs, int32_t c)
{
  return dev_transform(
    ctx, params, false, num_slices, offsets, d_morton, d_qp_off, d_attrs,
    const_cast<void*>(d_coeffs), c, ctx ? ctx->morton_bits : 0);
}

static int
gpcc_dev_attr_morton_sort_impl(
  gpcc_ctx* ctx, int32_t num_slices, const int64_t* offsets, const void* d_xyz,
  void* d_morton, void* d_order)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (num_slices < 1 || !offsets || offsets[0] != 0 || !d_xyz || !d_morton || !d_order)
    return fail(GPCC_ERR_INVALID_ARG, "bad arguments");
  for (int i = 0; i < num_slices; i++)
    if (offsets[i + 1] <= offsets[i])
      return fail(GPCC_ERR_INVALID_ARG, "empty or unordered slice");
  if (offsets[num_slices] > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per batch");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int n = (int)offsets[num_slices];
  const int bits = ctx->morton_bits > 0 ? ctx->morton_bits : 63;
  const int passes = std::max(1, (bits + 7) / 8);

  // tiles never straddle a slice
  std::vector<int32_t> h_off(num_slices + 1), t_first, t_count, t_slice;
  for (int s = 0; s <= num_slices; s++)
    h_off[s] = (int32_t)offsets[s];
  for (int s = 0; s < num_slices; s++)
    for (int b = h_off[s]; b < h_off[s + 1]; b += kSortTile) {
      t_first.push_back(b);
      t_count.push_back(std::min(kSortTile, h_off[s + 1] - b));
      t_slice.push_back(s);
    }
  const int nt = (int)t_first.size();

  Arena measure;
  auto carve_sort = [&](Arena& ar, SortCtx& cx, int64_t*& kbuf, int32_t*& vbuf,
                        int32_t*& pt_off) {
    ar.reset();
    pt_off = ar.take<int32_t>(num_slices + 1);
    cx.tile_first = ar.take<int32_t>(nt);
    cx.tile_count = ar.take<int32_t>(nt);
    cx.tile_slice = ar.take<int32_t>(nt);
    cx.hist = ar.take<uint32_t>((size_t)256 * nt);
    kbuf = ar.take<int64_t>(n);
    vbuf = ar.take<int32_t>(n);
  };
  SortCtx cx{};
  int64_t* kbuf = nullptr;
  int32_t* vbuf = nullptr;
  int32_t* d_pt = nullptr;
  carve_sort(measure, cx, kbuf, vbuf, d_pt);
  int rcode = ensure_arena(ctx, measure.used);
  if (rcode)
    return rcode;
  carve_sort(ctx->arena, cx, kbuf, vbuf, d_pt);

  const size_t stage_bytes = (num_slices + 1 + 3 * (size_t)nt) * sizeof(int32_t) + 64;
  HIP_TRY(hipStreamSynchronize(st));
  if (ctx->h_pinned_cap < stage_bytes) {
    if (ctx->h_pinned)
      HIP_TRY(hipHostFree(ctx->h_pinned));
    HIP_TRY(hipHostMalloc(&ctx->h_pinned, stage_bytes * 2));
    ctx->h_pinned_cap = stage_bytes * 2;
  }
  char* hb = (char*)ctx->h_pinned;
  size_t o = 0;
  auto stage = [&](const void* src, size_t bytes, const void* dst) -> hipError_t {
    memcpy(hb + o, src, bytes);
    hipError_t e = hipMemcpyAsync(
      const_cast<void*>(dst), hb + o, bytes, hipMemcpyHostToDevice, st);
    o += (bytes + 15) & ~size_t(15);
    return e;
  };
  HIP_TRY(stage(h_off.data(), h_off.size() * 4, d_pt));
  HIP_TRY(stage(t_first.data(), nt * 4, cx.tile_first));
  HIP_TRY(stage(t_count.data(), nt * 4, cx.tile_count));
  HIP_TRY(stage(t_slice.data(), nt * 4, cx.tile_slice));

  cx.n = n;
  cx.num_tiles = nt;
  cx.num_slices = num_slices;
  cx.pt_off = d_pt;
  // ping-pong so that the last pass writes the caller's buffers
  int64_t* ka = passes % 2 ? kbuf : (int64_t*)d_morton;
  int32_t* va = passes % 2 ? vbuf : (int32_t*)d_order;
  int64_t* kb = passes % 2 ? (int64_t*)d_morton : kbuf;
  int32_t* vb = passes % 2 ? (int32_t*)d_order : vbuf;
  {
    Timer t(ctx, "morton_encode");
    morton_encode_kernel<<<grid_for(n, 256), 256, 0, st>>>(
      (const int32_t*)d_xyz, d_pt, num_slices, n, ka, va);
  }
  for (int p = 0; p < passes; p++) {
    cx.key_in = ka;
    cx.val_in = va;
    cx.key_out = kb;
    cx.val_out = vb;
    cx.shift = 8 * p;
    {
      Timer t(ctx, "sort_hist");
      sort_hist_kernel<<<std::min(nt, kGridMax), kSortThreads, 0, st>>>(cx);
    }
    {
      Timer t(ctx, "sort_scan");
      sort_scan_kernel<<<1, 1024, 0, st>>>(cx);
    }
    {
      Timer t(ctx, "sort_scatter");
      sort_scatter_kernel<<<std::min(nt, kGridMax), kSortThreads, 0, st>>>(cx);
    }
    std::swap(ka, kb);
    std::swap(va, vb);
  }
  HIP_TRY(hipGetLastError());
  return GPCC_OK;
}

static int
gpcc_attr_morton_sort_impl(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int64_t* morton, int32_t* order)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!xyz || !morton || !order || n <= 0)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or n <= 0");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  int32_t mx = 0;
  for (int64_t i = 0; i < (int64_t)n * 3; i++) {
    if (xyz[i] < 0 || xyz[i] >= (1 << 21))
      return fail(GPCC_ERR_INVALID_ARG, "coordinate outside [0, 2^21)");
    mx = std::max(mx, xyz[i]);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  int32_t* d_x = nullptr;
  int64_t* d_m = nullptr;
  int32_t* d_o = nullptr;
  auto cleanup = [&]() {
    pool_free(ctx, d_x);
    pool_free(ctx, d_m);
    pool_free(ctx, d_o);
  };
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, (void**)&d_x, sizeof(int32_t) * 3 * n));
    HIP_TRY(pool_malloc(ctx, (void**)&d_m, sizeof(int64_t) * n));
    HIP_TRY(pool_malloc(ctx, (void**)&d_o, sizeof(int32_t) * n));
    HIP_TRY(h2d_user(ctx, d_x, xyz, sizeof(int32_t) * 3 * n, st));
    const int saved = ctx->morton_bits;
    ctx->morton_bits = std::max(1, 3 * bitlen64((uint64_t)mx));
    const int64_t offs[2] = {0, n};
    int r = gpcc_dev_attr_morton_sort_impl(ctx, 1, offs, d_x, d_m, d_o);
    ctx->morton_bits = saved;
    if (r)
      return r;
    HIP_TRY(d2h_user(ctx, morton, d_m, sizeof(int64_t) * n, st));
    HIP_TRY(d2h_user(ctx, order, d_o, sizeof(int32_t) * n, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  int r = run();
  cleanup();
  return r;
}

static int
gpcc_lift_forward_impl(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs)
{
  return host_lift(
    ctx, true, params, n, c, neigh_count, neigh_index, neigh_weight, indexes,
    qp_off, attrs, coeffs, lcp_coeffs);
}

static int
gpcc_lift_inverse_impl(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, const int32_t* coeffs, const int8_t* lcp_coeffs)
{
  return host_lift(
    ctx, false, params, n, c, neigh_count, neigh_index, neigh_weight, indexes,
    qp_off, attrs, const_cast<int32_t*>(coeffs), const_cast<int8_t*>(lcp_coeffs));
}

static int
gpcc_lod_compute_weights_impl(
  gpcc_ctx* ctx, int32_t n, int32_t* neigh_count, const uint64_t* dist2,
  int32_t* neigh_weight)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (n <= 0 || !neigh_count || !dist2 || !neigh_weight)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or n <= 0");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  Arena m;
  m.take<int32_t>(n);
  m.take<uint64_t>((size_t)n * 3);
  m.take<int32_t>((size_t)n * 3);
  int rcode = ensure_arena(ctx, m.used);
  if (rcode)
    return rcode;
  Arena& ar = ctx->arena;
  ar.reset();
  int32_t* d_nc = ar.take<int32_t>(n);
  uint64_t* d_d = ar.take<uint64_t>((size_t)n * 3);
  int32_t* d_w = ar.take<int32_t>((size_t)n * 3);
  HIP_TRY(h2d_user(ctx, d_nc, neigh_count, sizeof(int32_t) * n, st));
  HIP_TRY(h2d_user(ctx, d_d, dist2, sizeof(uint64_t) * n * 3, st));
  lod_compute_weights_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, d_nc, d_d, d_w);
  HIP_TRY(hipGetLastError());
  HIP_TRY(d2h_user(ctx, neigh_count, d_nc, sizeof(int32_t) * n, st));
  HIP_TRY(d2h_user(ctx, neigh_weight, d_w, sizeof(int32_t) * n * 3, st));
  HIP_TRY(hipStreamSynchronize(st));
  return GPCC_OK;
}

}  // extern "C"

namespace {

// what the LoD build leaves on the device (valid until the arena is used again)
struct LodDeviceOut {
  int32_t* count = nullptr;        // [n]
  int32_t* neigh_index = nullptr;  // [n][3] predictor indices
  int32_t* weight = nullptr;       // [n][3]
  int32_t* indexes = nullptr;      // [n] predictor -> point
  int32_t* error = nullptr;        // device error word of the sub-sampling kernel
  int32_t* inter_ref = nullptr;    // [n][3] neighbour lives in the reference frame (inter builds only)
  const int32_t* xyz = nullptr;    // [n][3] the positions in POINT order, on the device
  std::vector<int32_t> npl;        // cumulative LoD sizes, coarse to fine
  size_t arena_end = 0;            // first free byte behind the build's workspace
};

// the reference frame of attribute inter prediction (host memory, point order)
struct LodInterFrame {
  const int32_t* xyz;
  int32_t n;
  int32_t search_range;    // abh.attrInterPredSearchRange: replaces both LoD search ranges
  int32_t frame_distance;  // AttributeInterPredParams::frameDistance
};

// -> the per-point offsets in `where` (device, [n][2]), or null when the block names no region
template<class Params>
int
qp_regions_to_points(
  const Params* p, const int32_t* d_xyz, int n, int32_t* where, hipStream_t st, const int32_t** out)
{
  *out = nullptr;
  if (p->num_qp_regions == 0)
    return GPCC_OK;
  if (p->num_qp_regions < 0 || p->num_qp_regions > GPCC_MAX_QP_REGIONS)
    return fail(GPCC_ERR_INVALID_ARG, "num_qp_regions out of range");
  QpRegionSet rs{};
  rs.n = p->num_qp_regions;
  for (int r = 0; r < rs.n; r++) {
    for (int k = 0; k < 3; k++) {
      rs.lo[r][k] = p->qp_region_min[r][k];
      rs.hi[r][k] = p->qp_region_max[r][k];
    }
    rs.off[r][0] = p->qp_region_offset[r][0];
    rs.off[r][1] = p->qp_region_offset[r][1];
  }
  qp_region_fill_kernel<<<grid_for(n, 256), 256, 0, st>>>(d_xyz, n, rs, where);
  HIP_TRY(hipGetLastError());
  *out = where;
  return GPCC_OK;
}

// AttributeLods::generate on the device; results stay there.  `extra_bytes`
// are reserved behind the workspace for the caller (same arena, no regrowth).
int
lod_build_core(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, const int32_t* xyz, int32_t n,
  size_t extra_bytes, LodDeviceOut* out, bool xyz_on_device = false,
  const LodInterFrame* frame = nullptr)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!lp || !xyz || n <= 0)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or n <= 0");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  if (frame) {
    if (!frame->xyz || frame->n <= 0 || frame->n > kMaxPoints || frame->search_range < 0 || xyz_on_device)
      return fail(GPCC_ERR_INVALID_ARG, "reference frame: null, empty, too large or a negative search range");
    if (lp->scalable_lifting_enabled_flag || lp->canonical_point_order_flag || lp->max_points_per_sort_log2_plus1)
      return fail(
        GPCC_ERR_UNSUPPORTED,
        "inter prediction together with scalable lifting / canonical point order stays on the reference CPU path");
    for (int64_t i = 0; i < (int64_t)frame->n * 3; i++)
      if (frame->xyz[i] < 0 || frame->xyz[i] >= (1 << 21))
        return fail(GPCC_ERR_INVALID_ARG, "reference frame coordinate outside [0, 2^21)");
  }
  const bool scalable = lp->scalable_lifting_enabled_flag != 0;
  if (!scalable && (lp->lod_decimation_type < 0 || lp->lod_decimation_type > 2))
    return fail(GPCC_ERR_INVALID_ARG, "lod_decimation_type out of range");
  if (scalable && (lp->max_neigh_range_minus1 < 0 || lp->max_neigh_range_minus1 > (1 << 20)))
    return fail(GPCC_ERR_INVALID_ARG, "max_neigh_range_minus1 out of range");
  // (scalable lifting: always 21 levels, hls.h:835-839 -- lod_scalable.hpp)
  const int max_levels = scalable ? 21 : lp->num_detail_levels_minus1 + 1;
  if (max_levels < 1 || max_levels > GPCC_MAX_LODS - 1)
    return fail(GPCC_ERR_INVALID_ARG, "num_detail_levels out of range");
  int32_t mx = 0;
  if (xyz_on_device) {
    // device tier: the positions are not inspected on the host; the Morton-bits
    // hint bounds the sort's passes (gpcc_ctx_set_morton_bits, 0 = all 63)
    mx = ctx->morton_bits > 0 ? (1 << std::min(21, (ctx->morton_bits + 2) / 3)) - 1 : (1 << 21) - 1;
  } else {
    for (int64_t i = 0; i < (int64_t)n * 3; i++) {
      if (xyz[i] < 0 || xyz[i] >= (1 << 21))
        return fail(GPCC_ERR_INVALID_ARG, "coordinate outside [0, 2^21)");
      mx = std::max(mx, xyz[i]);
    }
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  // Workspace from the context's arena (no allocation per call once it has
  // grown).  The Morton sort below carves its own scratch from the arena's
  // start; everything that has to outlive it is placed behind that region.
  const size_t NF = frame ? (size_t)frame->n : 0;
  const size_t sort_region = 32 * std::max((size_t)n, NF) + ((size_t)1 << 20);
  {
    // 26 arrays, the largest 24 B per point (see the DM list below); an inter build adds
    // the reference frame (positions twice, biased positions, codes, order, list, boxes)
    // and the flags of the result
    const size_t mine = (size_t)n * (scalable ? 244 : 232) + 64 * 1024 + (frame ? NF * 56 + (size_t)n * 12 + 64 * 1024 : 0);
    if (guard_mode())  // a 256-byte band behind every array here and behind the caller's few
      extra_bytes += 96 * 1024;
    int rc0 = ensure_arena(ctx, sort_region + mine + extra_bytes);
    if (rc0)
      return rc0;
  }
  size_t ar_used = sort_region;
  auto dmalloc = [&](size_t bytes) -> void* {
    const size_t b = (std::max<size_t>(bytes, 256) + 255) & ~size_t(255);
    if (ar_used + b > ctx->arena.cap)
      return nullptr;
    void* p = ctx->arena.base + ar_used;
    ar_used += b;
    if (guard_mode()) {  // (Arena::band() for this entry's own carving)
      if (ar_used + kGuardInner > ctx->arena.cap)
        return nullptr;
      ctx->arena.used = ar_used;
      ctx->arena.band();
      ar_used = ctx->arena.used;
    }
    return p;
  };
  auto run = [&]() -> int {
    const size_t N = (size_t)n;
    const int nb0 = (n + 31) >> 5, nb1 = (nb0 + 31) >> 5, nb2 = (nb1 + 31) >> 5;
#define DM(type, name, count)                                   \
  type* name = (type*)dmalloc(sizeof(type) * (count));          \
  if (!name)                                                    \
    return fail(GPCC_ERR_OUT_OF_MEMORY, "hipMalloc(" #name ")");
    DM(int32_t, d_xyz_own, 3 * N)
    const int32_t* d_xyz = xyz_on_device ? xyz : d_xyz_own;
    DM(int64_t, d_code, N)
    DM(int32_t, d_order, N)
    DM(int32_t, d_pos, 3 * N)
    DM(int32_t, d_bpos, 3 * N)
    DM(int32_t, d_list_a, N + 1)
    DM(int32_t, d_list_b, N + 1)
    DM(int32_t, d_refine, N + 1)
    DM(uint8_t, d_flags, N + 1)
    DM(uint8_t, d_heads, N + 1)
    DM(int32_t, d_positions, N + 1)
    DM(int32_t, d_cell_first, N + 2)
    DM(int32_t, d_cent_tmp, N + 2)
    DM(int64_t, d_cell_key, N + 1)
    DM(int64_t, d_ret_key, N + 1)
    DM(uint32_t, d_cell_state, 4 * (N + 1))
    DM(int32_t, d_small, 64)  // ticket[8], error, counts[2]
    DM(unsigned long long, d_scan, 1024)
    DM(long long, d_atlas_limit, 1)
    DM(int32_t, d_box, (size_t)2 * 2 * 3 * (nb0 + nb1 + nb2 + 3))
    DM(int32_t, d_pred_count, N)
    DM(int32_t, d_pred_point, 3 * N)
    DM(uint64_t, d_pred_dist2, 3 * N)
    DM(int32_t, d_pt2pred, N)
    DM(int32_t, d_indexes, N)
    DM(int32_t, d_neigh_index, 3 * N)
    DM(int32_t, d_weight, 3 * N)
    DM(int32_t, d_bpos_lod, scalable ? 3 * N : 1)
    // attribute inter prediction: the reference frame, sorted
    const int nfb0 = ((int)NF + 31) >> 5, nfb1 = (nfb0 + 31) >> 5, nfb2 = (nfb1 + 31) >> 5;
    DM(int32_t, d_fxyz, 3 * NF + 1)
    DM(int64_t, d_fcode, NF + 1)
    DM(int32_t, d_forder, NF + 1)
    DM(int32_t, d_fpos, 3 * NF + 1)
    DM(int32_t, d_fbpos, 3 * NF + 1)
    DM(int32_t, d_flist, NF + 1)
    DM(int32_t, d_fbox, (size_t)2 * 3 * (nfb0 + nfb1 + nfb2 + 3) + 1)
    DM(int32_t, d_inter_ref, frame ? 3 * N : 1)
#undef DM
    int32_t* d_ticket = d_small;
    int32_t* d_error = d_small + 8;
    int32_t* d_counts = d_small + 16;
    if (!xyz_on_device)
      HIP_TRY(h2d_user(ctx, d_xyz_own, xyz, sizeof(int32_t) * 3 * N, st));
    HIP_TRY(hipMemsetAsync(d_cell_state, 0, sizeof(uint32_t) * 4 * (N + 1), st));
    HIP_TRY(hipMemsetAsync(d_small, 0, sizeof(int32_t) * 64, st));
    HIP_TRY(hipMemsetAsync(d_scan, 0, sizeof(unsigned long long) * 1024, st));
    // Morton order (code, then index)
    {
      const int saved = ctx->morton_bits;
      ctx->morton_bits = std::max(1, 3 * bitlen64((uint64_t)mx));
      const int64_t offs[2] = {0, n};
      int r = gpcc_dev_attr_morton_sort_impl(ctx, 1, offs, d_xyz, d_code, d_order);
      ctx->morton_bits = saved;
      if (r)
        return r;
    }
    // canonical_point_order_flag / max_points_per_sort_log2_plus1 (PCCTMC3Common.h:2322-2331): the
    // reference takes the points as they come (or sorts them in chunks).  What the octree geometry
    // coder hands over IS in (Morton code, index) order -- then neither changes anything and the
    // build is the one below; any other order stays on the reference CPU path.
    if (lp->canonical_point_order_flag || lp->max_points_per_sort_log2_plus1) {
      int32_t moved = 0;
      lod_count_moved_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, d_order, d_small + 24);
      HIP_TRY(hipMemcpyAsync(&moved, d_small + 24, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (moved)
        return fail(
          GPCC_ERR_UNSUPPORTED,
          "canonical point order / chunked sort with points that are not in Morton order stay on the reference CPU path");
    }
    {
      Timer tm(ctx, "lod_gather");
      lod_gather_pos_kernel<<<grid_for(n, 256), 256, 0, st>>>(
        n, d_xyz, d_order, lp->lod_neigh_bias[0], lp->lod_neigh_bias[1],
        lp->lod_neigh_bias[2], d_pos, d_bpos, d_list_a);
    }
    // the reference frame of inter prediction: Morton order, biased positions, the list
    // 0, 1, 2, ... and one box hierarchy over it (buildPredictorsFast :2352-2376,
    // computeNearestNeighbors :1270-1292) -- the same at every level of detail
    int32_t* fbox[3][2] = {};
    if (frame) {
      HIP_TRY(h2d_user(ctx, d_fxyz, frame->xyz, sizeof(int32_t) * 3 * NF, st));
      int32_t fmx = 0;
      for (size_t i = 0; i < 3 * NF; i++)
        fmx = std::max(fmx, frame->xyz[i]);
      const int saved = ctx->morton_bits;
      ctx->morton_bits = std::max(1, 3 * bitlen64((uint64_t)fmx));
      const int64_t offs[2] = {0, (int64_t)NF};
      int r = gpcc_dev_attr_morton_sort_impl(ctx, 1, offs, d_fxyz, d_fcode, d_forder);
      ctx->morton_bits = saved;
      if (r)
        return r;
      lod_gather_pos_kernel<<<grid_for((int64_t)NF, 256), 256, 0, st>>>(
        (int)NF, d_fxyz, d_forder, lp->lod_neigh_bias[0], lp->lod_neigh_bias[1],
        lp->lod_neigh_bias[2], d_fpos, d_fbpos, d_flist);
      {
        int32_t* p = d_fbox;
        const int cnt[3] = {nfb0 + 1, nfb1 + 1, nfb2 + 1};
        for (int lev = 0; lev < 3; lev++)
          for (int m = 0; m < 2; m++) {
            fbox[lev][m] = p;
            p += 3 * cnt[lev];
          }
      }
      lod_box0_kernel<<<grid_for(std::max(nfb0, 1), 256), 256, 0, st>>>(
        (int)NF, d_flist, d_fbpos, fbox[0][0], fbox[0][1]);
      lod_box_up_kernel<<<grid_for(std::max(nfb1, 1), 256), 256, 0, st>>>(
        nfb0, fbox[0][0], fbox[0][1], fbox[1][0], fbox[1][1]);
      lod_box_up_kernel<<<1, 256, 0, st>>>(nfb1, fbox[1][0], fbox[1][1], fbox[2][0], fbox[2][1]);
      HIP_TRY(hipGetLastError());
    }

    // box storage: [list 0 = retained, 1 = refine][level][min/max]
    int32_t* box[2][3][2];
    {
      int32_t* p = d_box;
      const int cnt[3] = {nb0 + 1, nb1 + 1, nb2 + 1};
      for (int l = 0; l < 2; l++)
        for (int lev = 0; lev < 3; lev++)
          for (int m = 0; m < 2; m++) {
            box[l][lev][m] = p;
            p += 3 * cnt[lev];
          }
    }
    auto build_boxes = [&](int which, const int32_t* list, int cnt) {
      const int c0 = (cnt + 31) >> 5, c1 = (c0 + 31) >> 5;
      {
        Timer tm(ctx, "lod_boxes");
        lod_box0_kernel<<<grid_for(std::max(c0, 1), 256), 256, 0, st>>>(
          cnt, list, d_bpos, box[which][0][0], box[which][0][1]);
      }
      {
        Timer tm(ctx, "lod_boxes");
        lod_box_up_kernel<<<grid_for(std::max(c1, 1), 256), 256, 0, st>>>(
          c0, box[which][0][0], box[which][0][1], box[which][1][0], box[which][1][1]);
      }
      {
        Timer tm(ctx, "lod_boxes");
        lod_box_up_kernel<<<1, 256, 0, st>>>(
          c1, box[which][1][0], box[which][1][1], box[which][2][0], box[which][2][1]);
      }
    };

    int scan_epoch = 0;
    auto partition = [&](int cnt, const uint8_t* flags, const int32_t* list,
                         int32_t* out_true, int32_t* out_false, int* n_true) -> int {
      scan_epoch++;
      const int grid = (int)std::min<int64_t>(1024, ((int64_t)cnt + 1023) / 1024);
      {
        Timer tm(ctx, "lod_partition");
        lod_partition_kernel<<<std::max(grid, 1), 256, 0, st>>>(
          cnt, flags, list, out_true, out_false, d_counts, d_scan, scan_epoch);
      }
      int32_t h = 0;
      HIP_TRY(hipMemcpyAsync(&h, d_counts, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      *n_true = h;
      return GPCC_OK;
    };

    std::vector<int32_t> npl;
    npl.push_back(n);
    int32_t* d_input = d_list_a;
    int32_t* d_ret = d_list_b;
    int n_in = n, n_idx = 0;
    if (scalable) {
      // its own level loop (lod_scalable.hpp), same kernels
      LodWork w{};
      w.n = n;
      w.code = d_code;
      w.order = d_order;
      w.pos = d_pos;
      w.bpos = d_bpos;
      w.bpos_lod = d_bpos_lod;
      w.list_a = d_list_a;
      w.list_b = d_list_b;
      w.refine = d_refine;
      w.flags = d_flags;
      w.heads = d_heads;
      w.nxt0 = d_cell_first;
      w.nj0 = d_positions;
      w.nj1 = d_cent_tmp;
      w.ret_key = d_ret_key;
      w.counts = d_counts;
      w.scan = d_scan;
      w.atlas_limit = d_atlas_limit;
      for (int l = 0; l < 2; l++)
        for (int lev = 0; lev < 3; lev++)
          for (int m = 0; m < 2; m++)
            w.box[l][lev][m] = box[l][lev][m];
      w.pred_count = d_pred_count;
      w.pred_point = d_pred_point;
      w.pred_dist2 = d_pred_dist2;
      w.pt2pred = d_pt2pred;
      w.indexes = d_indexes;
      Timer tm(ctx, "lod_scalable_levels");
      HIP_TRY(lod_scalable_levels(lp, w, st, &npl, &scan_epoch));
      n_in = 0;  // the loop below has nothing left to do
    }
    for (int lod = 0; n_in > 0 && lod < max_levels; lod++) {
      const int start = n_idx;
      int n_ret = 0, n_ref = 0;
      const int shift_bits0 = lp->dist2 + lp->attr_dist2_delta + lod;
      if (lod == max_levels - 1 || (lp->lod_decimation_type != 1 && n_in == 1)) {
        HIP_TRY(hipMemcpyAsync(
          d_refine + start, d_input, sizeof(int32_t) * n_in, hipMemcpyDeviceToDevice, st));
        n_ref = n_in;
      } else {
        if (lp->lod_decimation_type == 1) {
          const int period = lp->lod_sampling_period[lod];
          if (period < 1)
            return fail(GPCC_ERR_INVALID_ARG, "lod_sampling_period < 1");
          lod_flag_periodic_kernel<<<grid_for(n_in, 256), 256, 0, st>>>(n_in, period, d_flags);
        } else if (lp->lod_decimation_type == 2) {
          const int period = lp->lod_sampling_period[lod];
          if (period < 1)
            return fail(GPCC_ERR_INVALID_ARG, "lod_sampling_period < 1");
          LodCtx lc{};
          lc.code = d_code;
          lc.pos = d_pos;
          lc.input = d_input;
          lc.n_in = n_in;
          lc.shift3 = 3 * (shift_bits0 + 1);
          lc.flags = d_flags;
          // group starts = orbit of 0 under next(): pointer doubling
          int32_t* nxt0 = d_cell_first;               // [n_in + 1]
          int32_t* nj[2] = {d_positions, d_cent_tmp};  // squared pointers
          Timer tm(ctx, "lod_centroid");
          lod_centroid_next_kernel<<<grid_for(n_in + 1, 256), 256, 0, st>>>(lc, period, nxt0);
          HIP_TRY(hipMemsetAsync(d_heads, 0, (size_t)n_in + 1, st));
          HIP_TRY(hipMemsetAsync(d_heads, 1, 1, st));
          const int32_t* cur = nxt0;
          for (int r = 0, reach = 1; reach < n_in; r++, reach *= 2) {
            lod_centroid_jump_kernel<<<grid_for(n_in + 1, 256), 256, 0, st>>>(
              n_in, cur, nj[r & 1], d_heads);
            cur = nj[r & 1];
          }
          lod_centroid_pick_kernel<<<grid_for(n_in, 256), 256, 0, st>>>(
            lc, shift_bits0, nxt0, d_heads, 1);
        } else {
          LodCtx lc{};
          lc.n = n;
          lc.code = d_code;
          lc.order = d_order;
          lc.pos = d_pos;
          lc.bpos = d_bpos;
          lc.input = d_input;
          lc.n_in = n_in;
          lc.shift3 = 3 * (shift_bits0 + 1);
          lc.boundary = std::min(63, lc.shift3 + 21);
          lc.radius2 = (int64_t)3 << (shift_bits0 << 1);
          lc.cell_first = d_cell_first;
          lc.cell_key = d_cell_key;
          lc.cell_state = d_cell_state;
          lc.ticket = d_ticket;
          lc.error = d_error;
          lc.epoch = lod + 1;
          lc.flags = d_flags;
          {
            Timer tm(ctx, "lod_cell_heads");
            lod_flag_cell_heads_kernel<<<grid_for(n_in, 256), 256, 0, st>>>(lc, d_heads, d_positions);
          }
          int ncell = 0;
          int r = partition(n_in, d_heads, d_positions, d_cell_first, nullptr, &ncell);
          if (r)
            return r;
          HIP_TRY(hipMemcpyAsync(
            d_cell_first + ncell, &n_in, sizeof(int32_t), hipMemcpyHostToDevice, st));
          HIP_TRY(hipMemsetAsync(d_ticket, 0, sizeof(int32_t) * 8, st));
          lc.ncell = ncell;
          lod_cell_keys_kernel<<<grid_for(ncell, 256), 256, 0, st>>>(lc);
          // 3 workgroups per CU stay resident (168 registers, no LDS).  One slice
          // alone takes them all (1 M dense points, the level with the most cells:
          // 6.4 / 4.5 / 3.85 / 3.8 ms with 256 / 512 / 768 / 1 024 workgroups);
          // concurrent lanes (run_slices) take 256 each (5 x 1 M points: 9.5-10 ms
          // per 1 M points from 192 to 768)
          const int grid = (int)std::min<int64_t>(ctx->lod_grid, ((int64_t)ncell + 255) / 256);
          {
            Timer tm(ctx, level_name("lod_subsample", lod));
            lod_subsample_distance_kernel<<<std::max(8, (grid + 7) / 8 * 8), 256, 0, st>>>(lc);
          }
        }
        int r = partition(n_in, d_flags, d_input, d_ret, d_refine + start, &n_ret);
        if (r)
          return r;
        n_ref = n_in - n_ret;
      }
      n_idx += n_ref;

      // nearest neighbours of this LoD's refinement points
      if (n_ref > 0) {
        NnCtx nc{};
        nc.n = n;
        nc.code = d_code;
        nc.order = d_order;
        nc.bpos = d_bpos;
        nc.retained = d_ret;
        nc.ret_key = d_ret_key;
        nc.n_ret = n_ret;
        nc.refine = d_refine + start;
        nc.n_ref = n_ref;
        nc.start = start;
        nc.shift3 = 3 * (1 + shift_bits0);
        nc.boundary = std::min(63, nc.shift3 + 21);
        nc.distribution = lp->prediction_with_distribution_enabled;
        nc.range_inter = frame ? frame->search_range : lp->inter_lod_search_range;
        nc.range_intra = frame ? frame->search_range : lp->intra_lod_search_range;
        nc.intra = lod >= lp->intra_lod_prediction_skip_layers;
        if (frame) {
          nc.frame_code = d_fcode;
          nc.frame_order = d_forder;
          nc.frame_bpos = d_fbpos;
          nc.frame_identity = d_flist;
          nc.n_frame = (int)NF;
          nc.frame_range = frame->search_range;
          nc.frame_boundary = std::min(63, nc.shift3 + 9);
          for (int lev = 0; lev < 3; lev++)
            for (int m = 0; m < 2; m++)
              nc.box_frame[lev][m] = fbox[lev][m];
        }
        nc.max_neigh = lp->num_pred_nearest_neighbours_minus1 + 1;
        for (int lev = 0; lev < 3; lev++)
          for (int m = 0; m < 2; m++) {
            nc.box_ret[lev][m] = box[0][lev][m];
            nc.box_ref[lev][m] = box[1][lev][m];
          }
        nc.atlas_limit = d_atlas_limit;
        nc.pred_count = d_pred_count;
        nc.pred_point = d_pred_point;
        nc.pred_dist2 = d_pred_dist2;
        nc.pt2pred = d_pt2pred;
        nc.indexes = d_indexes;
        if (n_ret > 0) {
          lod_ret_keys_kernel<<<grid_for(n_ret, 256), 256, 0, st>>>(
            n_ret, d_ret, d_code, nc.shift3, d_ret_key);
          build_boxes(0, d_ret, n_ret);
        }
        if (nc.intra)
          build_boxes(1, d_refine + start, n_ref);
        const long long inf = INT64_MAX;
        HIP_TRY(hipMemcpyAsync(d_atlas_limit, &inf, sizeof(inf), hipMemcpyHostToDevice, st));
        if (n_ret > 0)
          {
            Timer tm(ctx, "lod_atlas_limit");
            lod_atlas_limit_kernel<<<grid_for(n_ret, 256), 256, 0, st>>>(nc, d_atlas_limit);
          }
        {
          Timer tm(ctx, level_name("lod_nn_search", lod));
          if (frame)
            lod_nn_search_kernel<false, true><<<grid_for(n_ref, 256), 256, 0, st>>>(nc);
          else
            lod_nn_search_kernel<false><<<grid_for(n_ref, 256), 256, 0, st>>>(nc);
        }
      }
      if (n_ret > 0)
        npl.push_back(n_ret);
      std::swap(d_input, d_ret);
      n_in = n_ret;
    }
    {
      Timer tm(ctx, "lod_finalise");
      if (frame)
        lod_finalise_inter_kernel<<<grid_for(n, 256), 256, 0, st>>>(
          n, d_pred_count, d_pred_point, d_pt2pred, d_pred_dist2, d_neigh_index, d_inter_ref,
          frame->frame_distance);
      else
        lod_finalise_kernel<<<grid_for(n, 256), 256, 0, st>>>(
          n, 0, d_pred_count, d_pred_point, d_pt2pred, d_pred_dist2, d_neigh_index);
    }
    lod_compute_weights_kernel<<<grid_for(n, 256), 256, 0, st>>>(
      n, d_pred_count, d_pred_dist2, d_weight);
    if (lp->attr_encoding == 1 && lp->pred_weight_blending_enabled_flag) {
      if (frame)
        lod_blend_weights_inter_kernel<<<grid_for(n, 256), 256, 0, st>>>(
          n, d_pred_count, d_pred_point, d_xyz, d_fxyz, d_weight);
      else
        lod_blend_weights_kernel<<<grid_for(n, 256), 256, 0, st>>>(
          n, d_pred_count, d_pred_point, d_xyz, d_weight);
    }
    HIP_TRY(hipGetLastError());
    out->count = d_pred_count;
    out->xyz = d_xyz;
    out->neigh_index = d_neigh_index;
    out->weight = d_weight;
    out->indexes = d_indexes;
    out->error = d_error;
    out->inter_ref = frame ? d_inter_ref : nullptr;
    out->npl.assign(npl.rbegin(), npl.rend());
    out->arena_end = ar_used;
    return GPCC_OK;
  };
  return run();
}

}  // namespace

extern "C" {

static int
gpcc_lod_build_impl(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, const int32_t* xyz, int32_t n,
  int32_t* neigh_count, int32_t* neigh_index, int32_t* neigh_weight,
  int32_t* indexes, int32_t* num_points_in_lod, int32_t* num_lods)
{
  if (!neigh_count || !neigh_index || !neigh_weight || !indexes || !num_points_in_lod || !num_lods)
    return fail(GPCC_ERR_INVALID_ARG, "null output buffer");
  LodDeviceOut o;
  int r = lod_build_core(ctx, lp, xyz, n, 0, &o);
  if (r)
    return r;
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  int32_t h_err = 0;
  HIP_TRY(d2h_user(ctx, neigh_count, o.count, sizeof(int32_t) * N, st));
  HIP_TRY(d2h_user(ctx, neigh_index, o.neigh_index, sizeof(int32_t) * 3 * N, st));
  HIP_TRY(d2h_user(ctx, neigh_weight, o.weight, sizeof(int32_t) * 3 * N, st));
  HIP_TRY(d2h_user(ctx, indexes, o.indexes, sizeof(int32_t) * N, st));
  HIP_TRY(hipMemcpyAsync(&h_err, o.error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h_err)
    return fail(GPCC_ERR_HIP, "a dependency wait in the LoD sub-sampling kernel expired");
  *num_lods = (int)o.npl.size();
  for (size_t i = 0; i < o.npl.size(); i++)
    num_points_in_lod[i] = o.npl[i];
  return GPCC_OK;
}

static int
gpcc_lod_build_inter_impl(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, const int32_t* xyz, int32_t n, const int32_t* xyz_ref,
  int32_t n_ref, int32_t search_range, int32_t frame_distance, int32_t* neigh_count,
  int32_t* neigh_index, int32_t* neigh_weight, int32_t* indexes, int32_t* num_points_in_lod,
  int32_t* num_lods, int32_t* inter_ref)
{
  if (!neigh_count || !neigh_index || !neigh_weight || !indexes || !num_points_in_lod || !num_lods || !inter_ref)
    return fail(GPCC_ERR_INVALID_ARG, "null output buffer");
  LodInterFrame frame{xyz_ref, n_ref, search_range, frame_distance};
  LodDeviceOut o;
  int r = lod_build_core(ctx, lp, xyz, n, 0, &o, false, &frame);
  if (r)
    return r;
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  int32_t h_err = 0;
  HIP_TRY(d2h_user(ctx, neigh_count, o.count, sizeof(int32_t) * N, st));
  HIP_TRY(d2h_user(ctx, neigh_index, o.neigh_index, sizeof(int32_t) * 3 * N, st));
  HIP_TRY(d2h_user(ctx, neigh_weight, o.weight, sizeof(int32_t) * 3 * N, st));
  HIP_TRY(d2h_user(ctx, indexes, o.indexes, sizeof(int32_t) * N, st));
  HIP_TRY(d2h_user(ctx, inter_ref, o.inter_ref, sizeof(int32_t) * 3 * N, st));
  HIP_TRY(hipMemcpyAsync(&h_err, o.error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h_err)
    return fail(GPCC_ERR_HIP, "a dependency wait in the LoD sub-sampling kernel expired");
  *num_lods = (int)o.npl.size();
  for (size_t i = 0; i < o.npl.size(); i++)
    num_points_in_lod[i] = o.npl[i];
  return GPCC_OK;
}

// The predicting attribute coder of one slice minus the entropy loop --
// AttributeLods::generate + encodeColorsPred / encodeReflectancesPred
// (AttributeEncoder.cpp:575-579, 749-853, 1075-1210) resp. decode...Pred
// (AttributeDecoder.cpp:292-296, 328-523): the predictors never leave the
// device.
static int
pred_attr_driver(
  gpcc_ctx* ctx, bool encoder, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  const int32_t* xyz, int32_t* attrs, int32_t* values, int8_t* icp, int32_t* indexes,
  int32_t n, int32_t c)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!pred || !attrs || !values || (c != 1 && c != 3))
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or attribute count not 1 / 3");
  const bool icp_on = c == 3 && pred->inter_component_prediction_enabled_flag;
  if (icp_on && !icp)
    return fail(GPCC_ERR_INVALID_ARG, "icp_coeffs is null");
  const size_t N = (size_t)(n > 0 ? n : 0);
  const size_t extra = ((N * c * sizeof(int32_t) + 255) & ~size_t(255)) * 2 + 512
    + pred_scratch_bytes(n > 0 ? n : 1) + 1024 + N * 8 + 256;
  LodDeviceOut o;
  int r = lod_build_core(ctx, lod, xyz, n, extra, &o);
  if (r)
    return r;
  pred->scalable_lifting_enabled_flag = lod->scalable_lifting_enabled_flag != 0;
  pred->num_lods = (int)o.npl.size();
  for (size_t i = 0; i < o.npl.size(); i++)
    pred->num_points_in_lod[i] = o.npl[i];
  r = check_pred_params(pred, n, c, encoder);
  if (r)
    return r;
  hipStream_t st = ctx->stream;
  Arena ar = ctx->arena;  // carve behind the LoD workspace
  ar.used = o.arena_end;
  PredDev d{};
  d.nc = o.count;
  d.ni = o.neigh_index;
  d.nw = o.weight;
  d.indexes = o.indexes;
  r = qp_regions_to_points(pred, o.xyz, n, ar.take<int32_t>(N * 2), st, &d.qp_off);
  if (r)
    return r;
  d.attrs = ar.take<int32_t>(N * c);
  d.values = ar.take<int32_t>(N * c);
  int8_t* d_icp = ar.take<int8_t>(GPCC_MAX_LODS * 3);
  char* scratch = ar.base + ar.used;
  if (ar.used + pred_scratch_bytes(n) > ctx->arena.cap)
    return fail(GPCC_ERR_OUT_OF_MEMORY, "arena reservation too small");
  if (encoder) {
    HIP_TRY(h2d_user(ctx, d.attrs, attrs, sizeof(int32_t) * N * c, st));
  } else {
    HIP_TRY(h2d_user(ctx, d.values, values, sizeof(int32_t) * N * c, st));
    if (icp_on)
      HIP_TRY(hipMemcpyAsync(d_icp, icp, GPCC_MAX_LODS * 3, hipMemcpyHostToDevice, st));
  }
  r = c == 1 ? launch_pred<1>(ctx, encoder, pred, n, d, d_icp, scratch)
             : launch_pred<3>(ctx, encoder, pred, n, d, d_icp, scratch);
  if (r)
    return r;
  int32_t h_err = 0;
  HIP_TRY(d2h_user(ctx, attrs, d.attrs, sizeof(int32_t) * N * c, st));
  if (encoder) {
    HIP_TRY(d2h_user(ctx, values, d.values, sizeof(int32_t) * N * c, st));
    if (icp_on)
      HIP_TRY(hipMemcpyAsync(icp, d_icp, GPCC_MAX_LODS * 3, hipMemcpyDeviceToHost, st));
  }
  if (indexes)
    HIP_TRY(d2h_user(ctx, indexes, o.indexes, sizeof(int32_t) * N, st));
  HIP_TRY(hipMemcpyAsync(&h_err, o.error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h_err)
    return fail(GPCC_ERR_HIP, "a dependency wait in the LoD sub-sampling kernel expired");
  return pred_check_error(ctx);
}

// The lifting attribute coder of one slice minus the entropy loop --
// AttributeLods::generate + encodeColorsLift / encodeReflectancesLift
// (AttributeEncoder.cpp:575-579, 1379-1648) resp. decode...Lift
// (AttributeDecoder.cpp:292-296, 678-857): the predictors never leave the
// device.
static int
lift_attr_driver(
  gpcc_ctx* ctx, bool encoder, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  const int32_t* xyz, int32_t* attrs, int32_t* coeffs, int8_t* lcp, int32_t* indexes,
  int32_t n, int32_t c)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!lift || !attrs || !coeffs || c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or attribute count not 1..3");
  const bool lcp_on = c == 3 && lift->last_component_prediction_enabled_flag;
  if (lcp_on && !lcp)
    return fail(GPCC_ERR_INVALID_ARG, "lcp_coeffs is null");
  const size_t N = (size_t)(n > 0 ? n : 0);
  const size_t extra = ((N * c * sizeof(int32_t) + 255) & ~size_t(255)) * 2 + 256
    + lift_scratch_bytes(n > 0 ? n : 1, c) + 1024 + N * 8 + 256;
  LodDeviceOut o;
  int r = lod_build_core(ctx, lod, xyz, n, extra, &o);
  if (r)
    return r;
  lift->scalable_lifting_enabled_flag = lod->scalable_lifting_enabled_flag != 0;
  lift->num_lods = (int)o.npl.size();
  for (size_t i = 0; i < o.npl.size(); i++)
    lift->num_points_in_lod[i] = o.npl[i];
  r = check_lift_params(lift, n, c);
  if (r)
    return r;
  hipStream_t st = ctx->stream;
  Arena ar = ctx->arena;  // carve behind the LoD workspace
  ar.used = o.arena_end;
  LiftDev d{};
  d.nc = o.count;
  d.ni = o.neigh_index;
  d.nw = o.weight;
  d.indexes = o.indexes;
  r = qp_regions_to_points(lift, o.xyz, n, ar.take<int32_t>(N * 2), st, &d.qp_off);
  if (r)
    return r;
  d.attrs = ar.take<int32_t>(N * c);
  d.coeffs = ar.take<int32_t>(N * c);
  int8_t* d_lcp = ar.take<int8_t>(GPCC_MAX_LODS);
  char* scratch = ar.base + ar.used;
  if (ar.used + lift_scratch_bytes(n, c) > ctx->arena.cap)
    return fail(GPCC_ERR_OUT_OF_MEMORY, "arena reservation too small");
  if (encoder) {
    HIP_TRY(h2d_user(ctx, d.attrs, attrs, sizeof(int32_t) * N * c, st));
  } else {
    HIP_TRY(h2d_user(ctx, d.coeffs, coeffs, sizeof(int32_t) * N * c, st));
    if (lcp_on)
      HIP_TRY(hipMemcpyAsync(d_lcp, lcp, GPCC_MAX_LODS, hipMemcpyHostToDevice, st));
  }
  switch (c) {
  case 1: r = launch_lift<1>(ctx, encoder, lift, n, d, d_lcp, scratch); break;
  case 2: r = launch_lift<2>(ctx, encoder, lift, n, d, d_lcp, scratch); break;
  default: r = launch_lift<3>(ctx, encoder, lift, n, d, d_lcp, scratch); break;
  }
  if (r)
    return r;
  int32_t h_err = 0;
  HIP_TRY(d2h_user(ctx, attrs, d.attrs, sizeof(int32_t) * N * c, st));
  if (encoder) {
    HIP_TRY(d2h_user(ctx, coeffs, d.coeffs, sizeof(int32_t) * N * c, st));
    if (lcp_on)
      HIP_TRY(hipMemcpyAsync(lcp, d_lcp, GPCC_MAX_LODS, hipMemcpyDeviceToHost, st));
  }
  if (indexes)
    HIP_TRY(d2h_user(ctx, indexes, o.indexes, sizeof(int32_t) * N, st));
  HIP_TRY(hipMemcpyAsync(&h_err, o.error, sizeof(int32_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (h_err)
    return fail(GPCC_ERR_HIP, "a dependency wait in the LoD sub-sampling kernel expired");
  return GPCC_OK;
}

static int
gpcc_lift_encode_attr_impl(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs, int32_t* indexes, int32_t n, int32_t c)
{
  return lift_attr_driver(ctx, true, lod, lift, xyz, attrs, coeffs, lcp_coeffs, indexes, n, c);
}

static int
gpcc_lift_decode_attr_impl(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, const int8_t* lcp_coeffs, int32_t* indexes, int32_t n,
  int32_t c)
{
  return lift_attr_driver(
    ctx, false, lod, lift, xyz, attrs, const_cast<int32_t*>(coeffs),
    const_cast<int8_t*>(lcp_coeffs), indexes, n, c);
}

namespace {
int
slice_driver(
  gpcc_ctx* ctx, const gpcc_raht_params* params, bool encoder, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth,
  int32_t* runs = nullptr, int32_t* values = nullptr, int32_t* num_symbols = nullptr,
  int32_t* trailing_run = nullptr, const gpcc_qp_regions* regions = nullptr)
{
  // packed: the encoder hands back (zero run, values) symbols instead of the
  // coefficient array
  const bool packed = runs != nullptr;
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!xyz || !attrs || (!coeffs && !packed) || n <= 0 || bitdepth < 1 || bitdepth > 16
      || (packed && (!values || !num_symbols || !trailing_run || !encoder)))
    return fail(GPCC_ERR_INVALID_ARG, "null buffer, n <= 0 or bitdepth outside [1, 16]");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  int rcode = check_params(params, c, encoder);
  if (rcode)
    return rcode;
  int32_t mx = 0;
  for (int64_t i = 0; i < (int64_t)n * 3; i++) {
    if (xyz[i] < 0 || xyz[i] >= (1 << 21))
      return fail(GPCC_ERR_INVALID_ARG, "coordinate outside [0, 2^21)");
    mx = std::max(mx, xyz[i]);
  }
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  int32_t *d_xyz = nullptr, *d_order = nullptr, *d_pt = nullptr, *d_a = nullptr, *d_c = nullptr;
  int32_t *d_qu = nullptr, *d_q = nullptr;  // QP-region offsets per point: input order, Morton order
  int64_t* d_m = nullptr;
  char* d_pack = nullptr;
  auto cleanup = [&]() {
    pool_free(ctx, d_pack);
    pool_free(ctx, d_qu);
    pool_free(ctx, d_q);
    pool_free(ctx, d_xyz);
    pool_free(ctx, d_order);
    pool_free(ctx, d_pt);
    pool_free(ctx, d_a);
    pool_free(ctx, d_c);
    pool_free(ctx, d_m);
  };
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, (void**)&d_xyz, sizeof(int32_t) * 3 * N));
    HIP_TRY(pool_malloc(ctx, (void**)&d_order, sizeof(int32_t) * N));
    HIP_TRY(pool_malloc(ctx, (void**)&d_m, sizeof(int64_t) * N));
    HIP_TRY(pool_malloc(ctx, (void**)&d_pt, sizeof(int32_t) * N * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_a, sizeof(int32_t) * N * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_c, sizeof(int32_t) * N * c));
    HIP_TRY(h2d_user(ctx, d_xyz, xyz, sizeof(int32_t) * 3 * N, st));
    const int bits = std::max(1, 3 * bitlen64((uint64_t)mx));
    const int64_t offs[2] = {0, n};
    {
      const int saved = ctx->morton_bits;
      ctx->morton_bits = bits;
      int r = gpcc_dev_attr_morton_sort_impl(ctx, 1, offs, d_xyz, d_m, d_order);
      ctx->morton_bits = saved;
      if (r)
        return r;
    }
    if (encoder) {
      HIP_TRY(h2d_user(ctx, d_pt, attrs, sizeof(int32_t) * N * c, st));
      {
        Timer tm(ctx, "attr_gather");
        attr_gather_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, c, d_order, d_pt, d_a);
      }
      HIP_TRY(hipMemsetAsync(d_c, 0, sizeof(int32_t) * N * c, st));
    } else {
      HIP_TRY(h2d_user(ctx, d_c, coeffs, sizeof(int32_t) * N * c, st));
    }
    // QP regions: the offset of the first box that holds the point (qpSet.regionQpOffset), in the order of
    // the Morton codes like the attributes
    const int32_t* d_qp = nullptr;
    if (regions && regions->num_qp_regions != 0) {
      HIP_TRY(pool_malloc(ctx, (void**)&d_qu, sizeof(int32_t) * 2 * N));
      HIP_TRY(pool_malloc(ctx, (void**)&d_q, sizeof(int32_t) * 2 * N));
      const int32_t* filled = nullptr;
      int rq = qp_regions_to_points(regions, d_xyz, n, d_qu, st, &filled);
      if (rq)
        return rq;
      attr_gather_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, 2, d_order, d_qu, d_q);
      d_qp = d_q;
    }
    int r = dev_transform(ctx, params, encoder, 1, offs, d_m, d_qp, d_a, d_c, c, bits);
    if (r)
      return r;
    {
      Timer tm(ctx, "attr_clip_scatter");
      attr_clip_scatter_kernel<<<grid_for(n, 256), 256, 0, st>>>(
        n, c, (1 << bitdepth) - 1, d_order, d_a, d_pt);
    }
    if (packed) {
      // zero-run formation where the coefficients are: only symbols cross PCIe
      Arena ar;
      ar.take<int32_t>(N);          // positions
      ar.take<int32_t>(N + 1);      // non-zero positions
      ar.take<int32_t>(N);          // runs
      ar.take<int32_t>(N * c);      // values
      ar.take<uint8_t>(N);          // flags
      ar.take<int32_t>(64);
      ar.take<unsigned long long>(1024);
      HIP_TRY(pool_malloc(ctx, (void**)&d_pack, ar.used));
      ar.base = d_pack;
      ar.reset();
      int32_t* d_pos = ar.take<int32_t>(N);
      int32_t* d_nz = ar.take<int32_t>(N + 1);
      int32_t* d_runs = ar.take<int32_t>(N);
      int32_t* d_vals = ar.take<int32_t>(N * c);
      uint8_t* d_flags = ar.take<uint8_t>(N);
      int32_t* d_small = ar.take<int32_t>(64);
      unsigned long long* d_scan = ar.take<unsigned long long>(1024);
      HIP_TRY(hipMemsetAsync(d_small, 0, sizeof(int32_t) * 64, st));
      HIP_TRY(hipMemsetAsync(d_scan, 0, sizeof(unsigned long long) * 1024, st));
      {
        Timer tm(ctx, "zero_run_pack");
        zero_run_flags_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, c, 1, d_c, d_flags, d_pos);
        const int grid = (int)std::min<int64_t>(1024, ((int64_t)n + 1023) / 1024);
        lod_partition_kernel<<<std::max(grid, 1), 256, 0, st>>>(
          n, d_flags, d_pos, d_nz, nullptr, d_small, d_scan, 1);
        zero_run_emit_kernel<<<grid_for(n, 256), 256, 0, st>>>(
          n, c, 1, d_c, d_nz, d_small, d_runs, d_vals, d_small + 1);
      }
      int32_t h[2] = {0, 0};
      HIP_TRY(hipMemcpyAsync(h, d_small, sizeof(h), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (h[0] > 0) {
        HIP_TRY(d2h_user(ctx, runs, d_runs, sizeof(int32_t) * (size_t)h[0], st));
        HIP_TRY(d2h_user(ctx, values, d_vals, sizeof(int32_t) * (size_t)h[0] * c, st));
      }
      *num_symbols = h[0];
      *trailing_run = h[1];
    }
    // the caller's attributes are overwritten only after the device result is
    // known to be valid (on failure they still hold the source)
    HIP_TRY(hipStreamSynchronize(st));
    r = check_device_error(ctx);
    if (r)
      return r;
    HIP_TRY(d2h_user(ctx, attrs, d_pt, sizeof(int32_t) * N * c, st));
    if (encoder && !packed)
      HIP_TRY(d2h_user(ctx, coeffs, d_c, sizeof(int32_t) * N * c, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  int r = run();
  cleanup();
  return r;
}
}  // namespace

static int
gpcc_raht_encode_attr_packed_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz, int32_t* attrs,
  int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run, int32_t n,
  int32_t c, int32_t bitdepth)
{
  if (!runs)
    return fail(GPCC_ERR_INVALID_ARG, "runs is null");
  return slice_driver(
    ctx, params, true, xyz, attrs, nullptr, n, c, bitdepth, runs, values, num_symbols, trailing_run);
}

static int
gpcc_raht_encode_attr_packed_regions_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run, int32_t n,
  int32_t c, int32_t bitdepth)
{
  if (!runs)
    return fail(GPCC_ERR_INVALID_ARG, "runs is null");
  return slice_driver(
    ctx, params, true, xyz, attrs, nullptr, n, c, bitdepth, runs, values, num_symbols, trailing_run, regions);
}

static int
gpcc_raht_decode_attr_regions_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return slice_driver(
    ctx, params, false, xyz, attrs, const_cast<int32_t*>(coeffs), n, c, bitdepth, nullptr, nullptr, nullptr,
    nullptr, regions);
}

static int
gpcc_raht_encode_attr_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return slice_driver(ctx, params, true, xyz, attrs, coeffs, n, c, bitdepth);
}

static int
gpcc_raht_decode_attr_impl(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return slice_driver(
    ctx, params, false, xyz, attrs, const_cast<int32_t*>(coeffs), n, c, bitdepth);
}

static int
gpcc_zero_run_pack_impl(
  gpcc_ctx* ctx, const int32_t* coeffs, int32_t n, int32_t c, int32_t planar,
  int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!coeffs || !runs || !values || !num_symbols || !trailing_run || n <= 0 || c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer, n <= 0 or attribute count not 1..3");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t N = (size_t)n;
  Arena m;
  int32_t *d_co, *d_pos, *d_nz, *d_runs, *d_vals, *d_small;
  uint8_t* d_flags;
  unsigned long long* d_scan;
  auto carve = [&](Arena& ar) {
    ar.reset();
    d_co = ar.take<int32_t>(N * c);
    d_pos = ar.take<int32_t>(N);
    d_nz = ar.take<int32_t>(N + 1);
    d_runs = ar.take<int32_t>(N);
    d_vals = ar.take<int32_t>(N * c);
    d_flags = ar.take<uint8_t>(N);
    d_small = ar.take<int32_t>(64);
    d_scan = ar.take<unsigned long long>(1024);
  };
  carve(m);
  int rcode = ensure_arena(ctx, m.used);
  if (rcode)
    return rcode;
  carve(ctx->arena);
  HIP_TRY(h2d_user(ctx, d_co, coeffs, sizeof(int32_t) * N * c, st));
  HIP_TRY(hipMemsetAsync(d_small, 0, sizeof(int32_t) * 64, st));
  HIP_TRY(hipMemsetAsync(d_scan, 0, sizeof(unsigned long long) * 1024, st));
  {
    Timer tm(ctx, "zero_run_pack");
    zero_run_flags_kernel<<<grid_for(n, 256), 256, 0, st>>>(n, c, planar, d_co, d_flags, d_pos);
    const int grid = (int)std::min<int64_t>(1024, ((int64_t)n + 1023) / 1024);
    lod_partition_kernel<<<std::max(grid, 1), 256, 0, st>>>(
      n, d_flags, d_pos, d_nz, nullptr, d_small, d_scan, 1);
    zero_run_emit_kernel<<<grid_for(n, 256), 256, 0, st>>>(
      n, c, planar, d_co, d_nz, d_small, d_runs, d_vals, d_small + 1);
  }
  int32_t h[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(h, d_small, sizeof(h), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  // only the symbols cross PCIe
  if (h[0] > 0) {
    HIP_TRY(d2h_user(ctx, runs, d_runs, sizeof(int32_t) * (size_t)h[0], st));
    HIP_TRY(d2h_user(ctx, values, d_vals, sizeof(int32_t) * (size_t)h[0] * c, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  *num_symbols = h[0];
  *trailing_run = h[1];
  return GPCC_OK;
}

static int
gpcc_estimate_dist2_impl(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int32_t sampling_period,
  int32_t search_range, float percentile, int32_t* shift_bits)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!xyz || !shift_bits || n < 0 || sampling_period < 1 || search_range < 0
      || !(percentile >= 0.f && percentile < 1.f))
    return fail(GPCC_ERR_INVALID_ARG, "bad argument");
  if (n > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per call");
  *shift_bits = 0;
  if (n < 2)
    return GPCC_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int ns = (n + sampling_period - 1) / sampling_period;
  int32_t* d_xyz = nullptr;
  long long* d_dist = nullptr;
  std::vector<long long> dists((size_t)ns);
  int rc = GPCC_OK;
  do {
    if (pool_malloc(ctx, (void**)&d_xyz, sizeof(int32_t) * 3 * (size_t)n) != hipSuccess
        || pool_malloc(ctx, (void**)&d_dist, sizeof(long long) * (size_t)ns) != hipSuccess) {
      rc = fail(GPCC_ERR_OUT_OF_MEMORY, "hipMalloc(estimate_dist2)");
      break;
    }
    hipError_t e = h2d_user(ctx, d_xyz, xyz, sizeof(int32_t) * 3 * (size_t)n, st);
    if (e == hipSuccess) {
      Timer tm(ctx, "estimate_dist2");
      estimate_dist2_kernel<<<grid_for(ns * 64, 256), 256, 0, st>>>(
        n, d_xyz, sampling_period, search_range, ns, d_dist);
    }
    if (e == hipSuccess)
      e = hipMemcpyAsync(dists.data(), d_dist, sizeof(long long) * (size_t)ns, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess)
      e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
      rc = fail(GPCC_ERR_HIP, hipGetErrorString(e));
      break;
    }
    // int p = int(std::floor(dists.size() * percentileEstimate)), :1712
    const int p = int(std::floor(dists.size() * percentile));
    std::nth_element(dists.begin(), dists.begin() + p, dists.end());
    const long long dist2 = dists[p];
    int shift = 0;
    while ((int64_t(3) << (shift << 1)) < dist2 && shift < 20)
      ++shift;
    *shift_bits = shift;
  } while (0);
  pool_free(ctx, d_xyz);
  pool_free(ctx, d_dist);
  return rc;
}

}  // extern "C"

#if GPCC_FIN_VAR == 2
extern "C" int
gpcc_debug_fin(unsigned int* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpcc::g_fin_dbg), sizeof(gpcc::g_fin_dbg)) != hipSuccess)
    return -1;
  if (reset) {
    unsigned int z[4 + 4 * 60] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_fin_dbg), z, sizeof(z)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

#ifdef GPCC_SUB_PROF
extern "C" int
gpcc_debug_sub_prof(unsigned long long* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpcc::g_sub_prof), sizeof(gpcc::g_sub_prof)) != hipSuccess)
    return -1;
  if (reset) {
    static unsigned long long z[16 + 32 * 20] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_sub_prof), z, sizeof(z)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

#ifdef GPCC_PIPE_PROF
extern "C" int
gpcc_debug_pipe_prof(unsigned long long* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpcc::g_pipe_prof), sizeof(gpcc::g_pipe_prof)) != hipSuccess)
    return -1;
  if (reset) {
    static unsigned long long z[32 * 8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_pipe_prof), z, sizeof(z)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

#ifdef GPCC_TILE_PROF
extern "C" int
gpcc_debug_tile_prof(unsigned long long* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpcc::g_tile_prof), sizeof(gpcc::g_tile_prof)) != hipSuccess)
    return -1;
  if (reset) {
    static unsigned long long z[5 * 24 * 8] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_tile_prof), z, sizeof(z)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

#ifdef GPCC_CX_PROF
extern "C" int
gpcc_debug_cx_prof(unsigned long long* out, int reset)
{
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(gpcc::g_cx_prof), sizeof(gpcc::g_cx_prof)) != hipSuccess)
    return -1;
  if (hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(gpcc::g_cx_prof_n), sizeof(gpcc::g_cx_prof_n)) != hipSuccess)
    return -1;
  if (reset) {
    static unsigned long long z[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_cx_prof), z, sizeof(z)) != hipSuccess
        || hipMemcpyToSymbol(HIP_SYMBOL(gpcc::g_cx_prof_n), z, sizeof(gpcc::g_cx_prof_n)) != hipSuccess)
      return -1;
  }
  return 0;
}
#endif

// every entry reports its outcome to the context's counters (gpcc_ctx_stats)
static int
counted(gpcc_ctx* ctx, int rc, int64_t points)
{
  if (ctx) {
    if (rc == GPCC_OK) {
      ctx->stats.calls_ok++;
      ctx->stats.points_ok += points;
    } else if (rc == GPCC_ERR_UNSUPPORTED) {
      ctx->stats.calls_unsupported++;
    } else {
      ctx->stats.calls_failed++;
    }
  }
  return rc;
}

extern "C" {

int
gpcc_raht_forward(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c)
{
  return counted(ctx, gpcc_raht_forward_impl(ctx, params, morton, qp_off, attrs, coeffs, n, c), n);
}

int
gpcc_raht_inverse(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, const int32_t* coeffs, int32_t n,
  int32_t c)
{
  return counted(ctx, gpcc_raht_inverse_impl(ctx, params, morton, qp_off, attrs, coeffs, n, c), n);
}

int
gpcc_raht_forward_inter(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_raht_inter_params* inter, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref,
  int32_t n_ref, int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps)
{
  return counted(
    ctx,
    host_transform_inter(
      ctx, params, inter, true, morton, qp_off, attrs, coeffs, n, c, morton_ref, attrs_ref, n_ref, layer_modes, num_modes,
      filter_taps, num_taps),
    n);
}

int
gpcc_raht_inverse_inter(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_raht_inter_params* inter, const int64_t* morton,
  const int32_t* qp_off, int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref,
  int32_t n_ref, const int32_t* layer_modes, int32_t num_modes, const int32_t* filter_taps, int32_t num_taps)
{
  int32_t nm = num_modes, nt = num_taps;
  // (an empty std::vector hands over a null pointer)
  static const int32_t none[1] = {0};
  if (!layer_modes && num_modes == 0)
    layer_modes = none;
  if (!filter_taps && num_taps == 0)
    filter_taps = none;
  return counted(
    ctx,
    host_transform_inter(
      ctx, params, inter, false, morton, qp_off, attrs, const_cast<int32_t*>(coeffs), n, c, morton_ref, attrs_ref, n_ref,
      const_cast<int32_t*>(layer_modes), &nm, const_cast<int32_t*>(filter_taps), &nt),
    n);
}

int
gpcc_dev_raht_forward(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, void* d_coeffs, int32_t c)
{
  return counted(ctx, gpcc_dev_raht_forward_impl(ctx, params, num_slices, offsets, d_morton, d_qp_off, d_attrs, d_coeffs, c), (offsets && num_slices > 0 ? offsets[num_slices] : 0));
}

int
gpcc_dev_raht_inverse(
  gpcc_ctx* ctx, const gpcc_raht_params* params, int32_t num_slices,
  const int64_t* offsets, const void* d_morton, const void* d_qp_off,
  void* d_attrs, const void* d_coeffs, int32_t c)
{
  return counted(ctx, gpcc_dev_raht_inverse_impl(ctx, params, num_slices, offsets, d_morton, d_qp_off, d_attrs, d_coeffs, c), (offsets && num_slices > 0 ? offsets[num_slices] : 0));
}

int
gpcc_dev_attr_morton_sort(
  gpcc_ctx* ctx, int32_t num_slices, const int64_t* offsets, const void* d_xyz,
  void* d_morton, void* d_order)
{
  return counted(ctx, gpcc_dev_attr_morton_sort_impl(ctx, num_slices, offsets, d_xyz, d_morton, d_order), (offsets && num_slices > 0 ? offsets[num_slices] : 0));
}

int
gpcc_attr_morton_sort(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int64_t* morton, int32_t* order)
{
  return counted(ctx, gpcc_attr_morton_sort_impl(ctx, xyz, n, morton, order), n);
}

int
gpcc_lift_forward(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs)
{
  return counted(ctx, gpcc_lift_forward_impl(ctx, params, n, c, neigh_count, neigh_index, neigh_weight, indexes, qp_off, attrs, coeffs, lcp_coeffs), n);
}

int
gpcc_lift_inverse(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index,
  const int32_t* neigh_weight, const int32_t* indexes, const int32_t* qp_off,
  int32_t* attrs, const int32_t* coeffs, const int8_t* lcp_coeffs)
{
  return counted(ctx, gpcc_lift_inverse_impl(ctx, params, n, c, neigh_count, neigh_index, neigh_weight, indexes, qp_off, attrs, coeffs, lcp_coeffs), n);
}

int
gpcc_lift_forward_inter(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, const int32_t* neigh_count,
  const int32_t* neigh_index, const int32_t* neigh_weight, const int32_t* inter_ref,
  const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref, int32_t* coeffs)
{
  if (!inter_ref)
    return counted(ctx, fail(GPCC_ERR_INVALID_ARG, "inter_ref is null"), n);
  return counted(
    ctx,
    host_lift(
      ctx, true, params, n, 1, neigh_count, neigh_index, neigh_weight, indexes, nullptr, attrs, coeffs,
      nullptr, inter_ref, attrs_ref, n_ref),
    n);
}

int
gpcc_lift_inverse_inter(
  gpcc_ctx* ctx, const gpcc_lift_params* params, int32_t n, const int32_t* neigh_count,
  const int32_t* neigh_index, const int32_t* neigh_weight, const int32_t* inter_ref,
  const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref,
  const int32_t* coeffs)
{
  if (!inter_ref)
    return counted(ctx, fail(GPCC_ERR_INVALID_ARG, "inter_ref is null"), n);
  return counted(
    ctx,
    host_lift(
      ctx, false, params, n, 1, neigh_count, neigh_index, neigh_weight, indexes, nullptr, attrs,
      const_cast<int32_t*>(coeffs), nullptr, inter_ref, attrs_ref, n_ref),
    n);
}

int
gpcc_pred_forward_inter(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, const int32_t* neigh_count,
  const int32_t* neigh_index, const int32_t* neigh_weight, const int32_t* inter_ref,
  const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref, int32_t* values)
{
  if (!inter_ref)
    return counted(ctx, fail(GPCC_ERR_INVALID_ARG, "inter_ref is null"), n);
  return counted(
    ctx,
    host_pred(
      ctx, true, params, n, 1, neigh_count, neigh_index, neigh_weight, indexes, nullptr, attrs, values,
      nullptr, inter_ref, attrs_ref, n_ref),
    n);
}

int
gpcc_pred_inverse_inter(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, const int32_t* neigh_count,
  const int32_t* neigh_index, const int32_t* neigh_weight, const int32_t* inter_ref,
  const int32_t* indexes, int32_t* attrs, const int32_t* attrs_ref, int32_t n_ref,
  const int32_t* values)
{
  if (!inter_ref)
    return counted(ctx, fail(GPCC_ERR_INVALID_ARG, "inter_ref is null"), n);
  return counted(
    ctx,
    host_pred(
      ctx, false, params, n, 1, neigh_count, neigh_index, neigh_weight, indexes, nullptr, attrs,
      const_cast<int32_t*>(values), nullptr, inter_ref, attrs_ref, n_ref),
    n);
}

int
gpcc_lod_compute_weights(
  gpcc_ctx* ctx, int32_t n, int32_t* neigh_count, const uint64_t* dist2,
  int32_t* neigh_weight)
{
  return counted(ctx, gpcc_lod_compute_weights_impl(ctx, n, neigh_count, dist2, neigh_weight), n);
}

int
gpcc_lod_build(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, const int32_t* xyz, int32_t n,
  int32_t* neigh_count, int32_t* neigh_index, int32_t* neigh_weight,
  int32_t* indexes, int32_t* num_points_in_lod, int32_t* num_lods)
{
  return counted(ctx, gpcc_lod_build_impl(ctx, lp, xyz, n, neigh_count, neigh_index, neigh_weight, indexes, num_points_in_lod, num_lods), n);
}

int
gpcc_lod_build_inter(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, const int32_t* xyz, int32_t n, const int32_t* xyz_ref,
  int32_t n_ref, int32_t search_range, int32_t frame_distance, int32_t* neigh_count,
  int32_t* neigh_index, int32_t* neigh_weight, int32_t* indexes, int32_t* num_points_in_lod,
  int32_t* num_lods, int32_t* inter_ref)
{
  return counted(
    ctx,
    gpcc_lod_build_inter_impl(
      ctx, lp, xyz, n, xyz_ref, n_ref, search_range, frame_distance, neigh_count, neigh_index,
      neigh_weight, indexes, num_points_in_lod, num_lods, inter_ref),
    n);
}

int
gpcc_lift_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int8_t* lcp_coeffs, int32_t* indexes, int32_t n, int32_t c)
{
  return counted(ctx, gpcc_lift_encode_attr_impl(ctx, lod, lift, xyz, attrs, coeffs, lcp_coeffs, indexes, n, c), n);
}

int
gpcc_lift_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, const int8_t* lcp_coeffs, int32_t* indexes, int32_t n,
  int32_t c)
{
  return counted(ctx, gpcc_lift_decode_attr_impl(ctx, lod, lift, xyz, attrs, coeffs, lcp_coeffs, indexes, n, c), n);
}

int
gpcc_pred_forward(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index, const int32_t* neigh_weight,
  const int32_t* indexes, const int32_t* qp_off, int32_t* attrs, int32_t* values,
  int8_t* icp_coeffs)
{
  return counted(ctx, host_pred(ctx, true, params, n, c, neigh_count, neigh_index, neigh_weight, indexes, qp_off, attrs, values, icp_coeffs), n);
}

int
gpcc_pred_inverse(
  gpcc_ctx* ctx, const gpcc_pred_params* params, int32_t n, int32_t c,
  const int32_t* neigh_count, const int32_t* neigh_index, const int32_t* neigh_weight,
  const int32_t* indexes, const int32_t* qp_off, int32_t* attrs, const int32_t* values,
  const int8_t* icp_coeffs)
{
  return counted(ctx, host_pred(ctx, false, params, n, c, neigh_count, neigh_index, neigh_weight, indexes, qp_off, attrs, const_cast<int32_t*>(values), const_cast<int8_t*>(icp_coeffs)), n);
}

int
gpcc_pred_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred, const int32_t* xyz,
  int32_t* attrs, int32_t* values, int8_t* icp_coeffs, int32_t* indexes, int32_t n, int32_t c)
{
  return counted(ctx, pred_attr_driver(ctx, true, lod, pred, xyz, attrs, values, icp_coeffs, indexes, n, c), n);
}

int
gpcc_pred_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred, const int32_t* xyz,
  int32_t* attrs, const int32_t* values, const int8_t* icp_coeffs, int32_t* indexes, int32_t n,
  int32_t c)
{
  return counted(ctx, pred_attr_driver(ctx, false, lod, pred, xyz, attrs, const_cast<int32_t*>(values), const_cast<int8_t*>(icp_coeffs), indexes, n, c), n);
}

int
gpcc_raht_encode_attr_packed(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz, int32_t* attrs,
  int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run, int32_t n,
  int32_t c, int32_t bitdepth)
{
  return counted(ctx, gpcc_raht_encode_attr_packed_impl(ctx, params, xyz, attrs, runs, values, num_symbols, trailing_run, n, c, bitdepth), n);
}

int
gpcc_raht_encode_attr(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return counted(ctx, gpcc_raht_encode_attr_impl(ctx, params, xyz, attrs, coeffs, n, c, bitdepth), n);
}

int
gpcc_raht_decode_attr(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return counted(ctx, gpcc_raht_decode_attr_impl(ctx, params, xyz, attrs, coeffs, n, c, bitdepth), n);
}

int
gpcc_raht_encode_attr_packed_regions(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run, int32_t n,
  int32_t c, int32_t bitdepth)
{
  return counted(
    ctx,
    gpcc_raht_encode_attr_packed_regions_impl(
      ctx, params, regions, xyz, attrs, runs, values, num_symbols, trailing_run, n, c, bitdepth),
    n);
}

int
gpcc_raht_decode_attr_regions(
  gpcc_ctx* ctx, const gpcc_raht_params* params, const gpcc_qp_regions* regions, const int32_t* xyz,
  int32_t* attrs, const int32_t* coeffs, int32_t n, int32_t c, int32_t bitdepth)
{
  return counted(ctx, gpcc_raht_decode_attr_regions_impl(ctx, params, regions, xyz, attrs, coeffs, n, c, bitdepth), n);
}

int
gpcc_zero_run_pack(
  gpcc_ctx* ctx, const int32_t* coeffs, int32_t n, int32_t c, int32_t planar,
  int32_t* runs, int32_t* values, int32_t* num_symbols, int32_t* trailing_run)
{
  return counted(ctx, gpcc_zero_run_pack_impl(ctx, coeffs, n, c, planar, runs, values, num_symbols, trailing_run), n);
}

int
gpcc_estimate_dist2(
  gpcc_ctx* ctx, const int32_t* xyz, int32_t n, int32_t sampling_period,
  int32_t search_range, float percentile, int32_t* shift_bits)
{
  return counted(ctx, gpcc_estimate_dist2_impl(ctx, xyz, n, sampling_period, search_range, percentile, shift_bits), n);
}

}  // extern "C"

// ---- device tier of the LoD build and the lifting coder -----------------------------
// Slices back to back in HBM, results left in HBM, workspace from the context's
// arena: no allocation, no PCIe traffic but the few integers per level of detail
// the host needs to enqueue the next launch (the sizes of the retained lists).
namespace {

// The slices of a batch are independent and the LoD build of one slice is a
// chain of latency-bound launches with a host round trip per level of detail
// (the size of the retained list), so several slices run CONCURRENTLY: each on
// a lane -- a context of its own (stream, workspace arena) on the same device,
// driven by a host thread of its own.  Lane 0 is the caller's context and
// thread.  GPCC_LOD_LANES sets the number of lanes (default 4, 1 = one slice
// after the other); with profiling on the slices run in turn so that the
// per-kernel times stay attributable.
int
lod_lanes(gpcc_ctx* ctx, int num_slices)
{
  static const int want = [] {
    const char* e = getenv("GPCC_LOD_LANES");
    const int v = e ? atoi(e) : 4;
    return v < 1 ? 1 : (v > 16 ? 16 : v);
  }();
  if (ctx->profiling)
    return 1;
  return std::min(want, num_slices);
}

template<class Body>
int
run_slices(gpcc_ctx* ctx, int num_slices, Body&& body)
{
  const int K = lod_lanes(ctx, num_slices);
  if (K <= 1) {
    for (int s = 0; s < num_slices; s++) {
      const int r = body(ctx, s);
      if (r)
        return r;
    }
    return GPCC_OK;
  }
  HIP_TRY(hipSetDevice(ctx->device));
  while ((int)ctx->lanes.size() < K - 1) {
    gpcc_ctx* l = nullptr;
    const int r = gpcc_ctx_create(ctx->device, nullptr, &l);
    if (r)
      return r;
    ctx->lanes.push_back(l);
  }
  if (!ctx->ev_lanes)
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_lanes, hipEventDisableTiming));
  // the caller's inputs are ordered on the context's stream
  HIP_TRY(hipEventRecord(ctx->ev_lanes, ctx->stream));
  for (int w = 1; w < K; w++) {
    ctx->lanes[w - 1]->morton_bits = ctx->morton_bits;
    HIP_TRY(hipStreamWaitEvent(ctx->lanes[w - 1]->stream, ctx->ev_lanes, 0));
  }
  // the lanes share the device's resident-workgroup slots
  static const int lane_grid = [] {
    const char* e = getenv("GPCC_LOD_GRID");
    return e ? std::max(8, atoi(e)) : 256;
  }();
  struct GridGuard {
    gpcc_ctx* c;
    int saved;
    ~GridGuard() { c->lod_grid = saved; }
  } guard{ctx, ctx->lod_grid};
  ctx->lod_grid = lane_grid;
  for (int w = 1; w < K; w++)
    ctx->lanes[w - 1]->lod_grid = lane_grid;
  std::atomic<int> next{0};
  std::vector<int> rc(K, GPCC_OK);
  std::vector<std::string> msg(K);
  auto work = [&](int w) {
    gpcc_ctx* lane = w ? ctx->lanes[w - 1] : ctx;
    if (hipSetDevice(ctx->device) != hipSuccess) {
      rc[w] = GPCC_ERR_HIP;
      msg[w] = "hipSetDevice failed on a lane";
      return;
    }
    for (;;) {
      const int s = next.fetch_add(1);
      if (s >= num_slices)
        break;
      const int r = body(lane, s);
      if (r) {
        rc[w] = r;
        msg[w] = g_last_error;
        next.store(num_slices);  // the others stop at their next slice
        break;
      }
    }
  };
  std::vector<std::thread> th;
  for (int w = 1; w < K; w++)
    th.emplace_back(work, w);
  work(0);
  for (auto& t : th)
    t.join();
  for (int w = 0; w < K; w++)
    if (rc[w])
      return fail(rc[w], msg[w]);
  return GPCC_OK;
}

int
check_slices(gpcc_ctx* ctx, int32_t num_slices, const int64_t* offsets)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (num_slices < 1 || !offsets || offsets[0] != 0)
    return fail(GPCC_ERR_INVALID_ARG, "bad slice offsets");
  for (int i = 0; i < num_slices; i++)
    if (offsets[i + 1] <= offsets[i])
      return fail(GPCC_ERR_INVALID_ARG, "empty or unordered slice");
  if (offsets[num_slices] > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 points per batch");
  return GPCC_OK;
}

int
lod_error_word(gpcc_ctx* ctx, const LodDeviceOut& o)
{
  int32_t h_err = 0;
  HIP_TRY(hipMemcpyAsync(&h_err, o.error, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  if (h_err)
    return fail(GPCC_ERR_HIP, "a dependency wait in the LoD sub-sampling kernel expired");
  return GPCC_OK;
}

int
dev_lod_build(
  gpcc_ctx* ctx, const gpcc_lod_params* lp, int32_t num_slices, const int64_t* offsets,
  const int32_t* d_xyz, int32_t* d_count, int32_t* d_index, int32_t* d_weight,
  int32_t* d_indexes, int32_t* num_points_in_lod, int32_t* num_lods)
{
  int r = check_slices(ctx, num_slices, offsets);
  if (r)
    return r;
  if (!d_xyz || !d_count || !d_index || !d_weight || !d_indexes || !num_points_in_lod || !num_lods)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer");
  return run_slices(ctx, num_slices, [&](gpcc_ctx* lane, int s) -> int {
    hipStream_t st = lane->stream;
    const size_t b = (size_t)offsets[s], N = (size_t)(offsets[s + 1] - offsets[s]);
    LodDeviceOut o;
    int r = lod_build_core(lane, lp, d_xyz + 3 * b, (int32_t)N, 0, &o, true);
    if (r)
      return r;
    HIP_TRY(hipMemcpyAsync(d_count + b, o.count, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_index + 3 * b, o.neigh_index, sizeof(int32_t) * 3 * N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_weight + 3 * b, o.weight, sizeof(int32_t) * 3 * N, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(d_indexes + b, o.indexes, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
    r = lod_error_word(lane, o);  // (the arena is reused by the lane's next slice: the copies are done)
    if (r)
      return r;
    num_lods[s] = (int)o.npl.size();
    for (size_t i = 0; i < o.npl.size(); i++)
      num_points_in_lod[(size_t)s * GPCC_MAX_LODS + i] = o.npl[i];
    return GPCC_OK;
  });
}

int
dev_lift_attr(
  gpcc_ctx* ctx, bool encoder, const gpcc_lod_params* lod, gpcc_lift_params* lift,
  int32_t num_slices, const int64_t* offsets, const int32_t* d_xyz, int32_t* d_attrs,
  int32_t* d_coeffs, int8_t* lcp, int32_t* d_indexes, int32_t c)
{
  int r = check_slices(ctx, num_slices, offsets);
  if (r)
    return r;
  if (!lift || !d_xyz || !d_attrs || !d_coeffs || c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or attribute count not 1..3");
  for (int s = 0; s < num_slices; s++)
    if (c == 3 && lift[s].last_component_prediction_enabled_flag && !lcp)
      return fail(GPCC_ERR_INVALID_ARG, "lcp_coeffs is null");
  return run_slices(ctx, num_slices, [&](gpcc_ctx* lane, int s) -> int {
    hipStream_t st = lane->stream;
    int r = GPCC_OK;
    gpcc_lift_params* lf = lift + s;
    const bool lcp_on = c == 3 && lf->last_component_prediction_enabled_flag;
    const size_t b = (size_t)offsets[s], N = (size_t)(offsets[s + 1] - offsets[s]);
    const int32_t n = (int32_t)N;
    const size_t extra = 512 + lift_scratch_bytes(n, c) + 1024 + N * 8 + 256;
    LodDeviceOut o;
    r = lod_build_core(lane, lod, d_xyz + 3 * b, n, extra, &o, true);
    if (r)
      return r;
    lf->scalable_lifting_enabled_flag = lod->scalable_lifting_enabled_flag != 0;
    lf->num_lods = (int)o.npl.size();
    for (size_t i = 0; i < o.npl.size(); i++)
      lf->num_points_in_lod[i] = o.npl[i];
    r = check_lift_params(lf, n, c);
    if (r)
      return r;
    Arena ar = lane->arena;  // carve behind the LoD workspace
    ar.used = o.arena_end;
    LiftDev d{};
    d.nc = o.count;
    d.ni = o.neigh_index;
    d.nw = o.weight;
    d.indexes = o.indexes;
    r = qp_regions_to_points(lf, o.xyz, n, ar.take<int32_t>(N * 2), st, &d.qp_off);
    if (r)
      return r;
    d.attrs = d_attrs + b * c;    // the caller's buffers, in place
    d.coeffs = d_coeffs + b * c;
    int8_t* d_lcp = ar.take<int8_t>(GPCC_MAX_LODS);
    char* scratch = ar.base + ar.used;
    if (ar.used + lift_scratch_bytes(n, c) > lane->arena.cap)
      return fail(GPCC_ERR_OUT_OF_MEMORY, "arena reservation too small");
    int8_t* h_lcp = lcp ? lcp + (size_t)s * GPCC_MAX_LODS : nullptr;
    if (!encoder && lcp_on)
      HIP_TRY(hipMemcpyAsync(d_lcp, h_lcp, GPCC_MAX_LODS, hipMemcpyHostToDevice, st));
    switch (c) {
    case 1: r = launch_lift<1>(lane, encoder, lf, n, d, d_lcp, scratch); break;
    case 2: r = launch_lift<2>(lane, encoder, lf, n, d, d_lcp, scratch); break;
    default: r = launch_lift<3>(lane, encoder, lf, n, d, d_lcp, scratch); break;
    }
    if (r)
      return r;
    if (encoder && lcp_on)
      HIP_TRY(hipMemcpyAsync(h_lcp, d_lcp, GPCC_MAX_LODS, hipMemcpyDeviceToHost, st));
    if (d_indexes)
      HIP_TRY(hipMemcpyAsync(d_indexes + b, o.indexes, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
    r = lod_error_word(lane, o);
    if (r)
      return r;
    return GPCC_OK;
  });
}

int
dev_pred_attr(
  gpcc_ctx* ctx, bool encoder, const gpcc_lod_params* lod, gpcc_pred_params* pred,
  int32_t num_slices, const int64_t* offsets, const int32_t* d_xyz, int32_t* d_attrs,
  int32_t* d_values, int8_t* icp, int32_t* d_indexes, int32_t c)
{
  int r = check_slices(ctx, num_slices, offsets);
  if (r)
    return r;
  if (!pred || !d_xyz || !d_attrs || !d_values || (c != 1 && c != 3))
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or attribute count not 1 / 3");
  for (int s = 0; s < num_slices; s++) {
    if (c == 3 && pred[s].inter_component_prediction_enabled_flag && !icp)
      return fail(GPCC_ERR_INVALID_ARG, "icp_coeffs is null");
  }
  return run_slices(ctx, num_slices, [&](gpcc_ctx* lane, int s) -> int {
    hipStream_t st = lane->stream;
    gpcc_pred_params* pp = pred + s;
    const bool icp_on = c == 3 && pp->inter_component_prediction_enabled_flag;
    const size_t b = (size_t)offsets[s], N = (size_t)(offsets[s + 1] - offsets[s]);
    const int32_t n = (int32_t)N;
    const size_t extra = 1024 + pred_scratch_bytes(n) + 1024 + N * 8 + 256;
    LodDeviceOut o;
    int r = lod_build_core(lane, lod, d_xyz + 3 * b, n, extra, &o, true);
    if (r)
      return r;
    pp->scalable_lifting_enabled_flag = lod->scalable_lifting_enabled_flag != 0;
    pp->num_lods = (int)o.npl.size();
    for (size_t i = 0; i < o.npl.size(); i++)
      pp->num_points_in_lod[i] = o.npl[i];
    r = check_pred_params(pp, n, c, encoder);
    if (r)
      return r;
    Arena ar = lane->arena;  // carve behind the LoD workspace
    ar.used = o.arena_end;
    PredDev d{};
    d.nc = o.count;
    d.ni = o.neigh_index;
    d.nw = o.weight;
    d.indexes = o.indexes;
    r = qp_regions_to_points(pp, o.xyz, n, ar.take<int32_t>(N * 2), st, &d.qp_off);
    if (r)
      return r;
    d.attrs = d_attrs + b * c;  // the caller's buffers, in place
    d.values = d_values + b * c;
    int8_t* d_icp = ar.take<int8_t>(GPCC_MAX_LODS * 3);
    char* scratch = ar.base + ar.used;
    if (ar.used + pred_scratch_bytes(n) > lane->arena.cap)
      return fail(GPCC_ERR_OUT_OF_MEMORY, "arena reservation too small");
    int8_t* h_icp = icp ? icp + (size_t)s * GPCC_MAX_LODS * 3 : nullptr;
    if (!encoder && icp_on)
      HIP_TRY(hipMemcpyAsync(d_icp, h_icp, GPCC_MAX_LODS * 3, hipMemcpyHostToDevice, st));
    r = c == 1 ? launch_pred<1>(lane, encoder, pp, n, d, d_icp, scratch)
               : launch_pred<3>(lane, encoder, pp, n, d, d_icp, scratch);
    if (r)
      return r;
    if (encoder && icp_on)
      HIP_TRY(hipMemcpyAsync(h_icp, d_icp, GPCC_MAX_LODS * 3, hipMemcpyDeviceToHost, st));
    if (d_indexes)
      HIP_TRY(hipMemcpyAsync(d_indexes + b, o.indexes, sizeof(int32_t) * N, hipMemcpyDeviceToDevice, st));
    r = lod_error_word(lane, o);
    if (r)
      return r;
    return pred_check_error(lane);
  });
}

}  // namespace

extern "C" {

int
gpcc_dev_lod_build(
  gpcc_ctx* ctx, const gpcc_lod_params* params, int32_t num_slices, const int64_t* offsets,
  const void* d_xyz, void* d_neigh_count, void* d_neigh_index, void* d_neigh_weight,
  void* d_indexes, int32_t* num_points_in_lod, int32_t* num_lods)
{
  return counted(
    ctx,
    dev_lod_build(
      ctx, params, num_slices, offsets, (const int32_t*)d_xyz, (int32_t*)d_neigh_count,
      (int32_t*)d_neigh_index, (int32_t*)d_neigh_weight, (int32_t*)d_indexes,
      num_points_in_lod, num_lods),
    offsets && num_slices > 0 ? offsets[num_slices] : 0);
}

int
gpcc_dev_lift_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, int32_t num_slices,
  const int64_t* offsets, const void* d_xyz, void* d_attrs, void* d_coeffs,
  int8_t* lcp_coeffs, void* d_indexes, int32_t c)
{
  return counted(
    ctx,
    dev_lift_attr(
      ctx, true, lod, lift, num_slices, offsets, (const int32_t*)d_xyz, (int32_t*)d_attrs,
      (int32_t*)d_coeffs, lcp_coeffs, (int32_t*)d_indexes, c),
    offsets && num_slices > 0 ? offsets[num_slices] : 0);
}

int
gpcc_dev_lift_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_lift_params* lift, int32_t num_slices,
  const int64_t* offsets, const void* d_xyz, void* d_attrs, const void* d_coeffs,
  const int8_t* lcp_coeffs, void* d_indexes, int32_t c)
{
  return counted(
    ctx,
    dev_lift_attr(
      ctx, false, lod, lift, num_slices, offsets, (const int32_t*)d_xyz, (int32_t*)d_attrs,
      (int32_t*)const_cast<void*>(d_coeffs), const_cast<int8_t*>(lcp_coeffs),
      (int32_t*)d_indexes, c),
    offsets && num_slices > 0 ? offsets[num_slices] : 0);
}

int
gpcc_dev_pred_encode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred, int32_t num_slices,
  const int64_t* offsets, const void* d_xyz, void* d_attrs, void* d_values,
  int8_t* icp_coeffs, void* d_indexes, int32_t c)
{
  return counted(
    ctx,
    dev_pred_attr(
      ctx, true, lod, pred, num_slices, offsets, (const int32_t*)d_xyz, (int32_t*)d_attrs,
      (int32_t*)d_values, icp_coeffs, (int32_t*)d_indexes, c),
    offsets && num_slices > 0 ? offsets[num_slices] : 0);
}

int
gpcc_dev_pred_decode_attr(
  gpcc_ctx* ctx, const gpcc_lod_params* lod, gpcc_pred_params* pred, int32_t num_slices,
  const int64_t* offsets, const void* d_xyz, void* d_attrs, const void* d_values,
  const int8_t* icp_coeffs, void* d_indexes, int32_t c)
{
  return counted(
    ctx,
    dev_pred_attr(
      ctx, false, lod, pred, num_slices, offsets, (const int32_t*)d_xyz, (int32_t*)d_attrs,
      (int32_t*)const_cast<void*>(d_values), const_cast<int8_t*>(icp_coeffs),
      (int32_t*)d_indexes, c),
    offsets && num_slices > 0 ? offsets[num_slices] : 0);
}

}  // extern "C"

// ---- several GPUs from one host process ----------------------------------------------
// The reference is one single-threaded process that codes slice after slice;
// slices are independent units (tmc3/encoder.cpp:544-571).  gpcc_multi owns one
// context per device, gives every device a contiguous, size-balanced run of the
// batch's slices, runs the transforms concurrently (every device on its own
// stream; the host only enqueues), and brings the coefficient and
// reconstruction buffers together on the first device -- RCCL send / receive
// over xGMI when the devices are distinct -- from where one download hands
// them to the host's entropy coder.  RCCL is loaded on first use (dlopen), the
// rest of the library does not depend on it.
namespace {

typedef struct ncclComm* ncclComm_t;
struct RcclApi {
  void* handle = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool load()
  {
    if (handle)
      return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (handle)
        break;
    }
    if (!handle)
      return false;
#define RCCL_SYM(field, sym) \
  field = reinterpret_cast<decltype(field)>(dlsym(handle, sym)); \
  if (!field)                                                     \
    return false;
    RCCL_SYM(CommInitAll, "ncclCommInitAll")
    RCCL_SYM(CommDestroy, "ncclCommDestroy")
    RCCL_SYM(GroupStart, "ncclGroupStart")
    RCCL_SYM(GroupEnd, "ncclGroupEnd")
    RCCL_SYM(Send, "ncclSend")
    RCCL_SYM(Recv, "ncclRecv")
    RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
    return true;
  }
};
constexpr int kNcclInt32 = 2;  // ncclInt32 (rccl.h)

}  // namespace

struct gpcc_multi {
  std::vector<int> devices;
  std::vector<gpcc_ctx*> ctx;
  std::vector<ncclComm_t> comm;  // empty: the devices are not distinct (or one): plain copies
  RcclApi rccl;
  // per device: its part of the batch in HBM (pooled buffers of its context)
  struct Part {
    int s0 = 0, s1 = 0;  // slices [s0, s1)
    int64_t* d_m = nullptr;
    int32_t *d_a = nullptr, *d_c = nullptr;
  };
  std::vector<Part> part;
};

namespace {

// contiguous runs of slices: device d's run ends at the slice boundary nearest
// to its share (d + 1) / nd of the points (a run may be empty when there are
// fewer slices than devices)
void
shard_slices(int num_slices, const int64_t* offsets, int nd, std::vector<gpcc_multi::Part>* part)
{
  part->assign(nd, gpcc_multi::Part());
  const int64_t total = offsets[num_slices];
  int s = 0;
  for (int d = 0; d < nd; d++) {
    (*part)[d].s0 = s;
    if (d == nd - 1) {
      s = num_slices;
    } else {
      const int64_t target = total * (d + 1) / nd;
      while (s < num_slices
             && std::llabs(offsets[s + 1] - target) <= std::llabs(offsets[s] - target))
        s++;
    }
    (*part)[d].s1 = s;
  }
}

int
multi_transform(
  gpcc_multi* m, const gpcc_raht_params* params, bool encoder, int32_t num_slices,
  const int64_t* offsets, const int64_t* morton, int32_t* attrs, int32_t* coeffs, int32_t c)
{
  if (!m)
    return fail(GPCC_ERR_INVALID_ARG, "multi context is null");
  int r = check_slices(m->ctx[0], num_slices, offsets);
  if (r)
    return r;
  if (!morton || !attrs || !coeffs || c < 1 || c > 3)
    return fail(GPCC_ERR_INVALID_ARG, "null buffer or attribute count not 1..3");
  const int nd = (int)m->devices.size();
  shard_slices(num_slices, offsets, nd, &m->part);
  const int64_t n_total = offsets[num_slices];
  int bits = 1;
  for (int s = 0; s < num_slices; s++)
    bits = std::max(bits, bitlen64((uint64_t)(morton[offsets[s]] ^ morton[offsets[s + 1] - 1])));
  // The caller's buffers are pageable: an "asynchronous" copy from pageable memory is
  // staged synchronously, so the uploads of eight devices would run one after the other
  // in front of transforms that take less time than they do.  Pinned for the duration of
  // the call (when the runtime allows it), the copies of all devices overlap.
  std::vector<void*> pinned;
  auto pin = [&](const void* ptr, size_t bytes) {
    if (nd > 1 && bytes && hipHostRegister((void*)ptr, bytes, hipHostRegisterDefault) == hipSuccess)
      pinned.push_back((void*)ptr);
    else
      (void)hipGetLastError();  // (already registered, or not permitted: the copies still work)
  };
  pin(morton, sizeof(int64_t) * (size_t)n_total);
  pin(attrs, sizeof(int32_t) * (size_t)n_total * c);
  pin(coeffs, sizeof(int32_t) * (size_t)n_total * c);
  auto release = [&]() {
    for (void* q : pinned)
      hipHostUnregister(q);
    for (int d = 0; d < nd; d++) {
      auto& p = m->part[d];
      pool_free(m->ctx[d], p.d_m);
      pool_free(m->ctx[d], p.d_a);
      pool_free(m->ctx[d], p.d_c);
      p.d_m = nullptr;
      p.d_a = p.d_c = nullptr;
    }
  };
  auto run = [&]() -> int {
    // upload + enqueue, device after device: the host never waits for a kernel
    for (int d = 0; d < nd; d++) {
      auto& p = m->part[d];
      gpcc_ctx* ctx = m->ctx[d];
      HIP_TRY(hipSetDevice(ctx->device));
      // device 0 holds the gathered coefficients (the encoder's output for the
      // arithmetic coder), every device its own part of everything else
      const int64_t b = offsets[p.s0];
      const int64_t np = offsets[p.s1] - b;
      const int64_t nbuf = (d == 0 && encoder) ? n_total : np;
      if (nbuf == 0)
        continue;
      HIP_TRY(pool_malloc(ctx, (void**)&p.d_a, sizeof(int32_t) * std::max<int64_t>(np, 1) * c));
      HIP_TRY(pool_malloc(ctx, (void**)&p.d_c, sizeof(int32_t) * nbuf * c));
      if (np == 0)
        continue;
      HIP_TRY(pool_malloc(ctx, (void**)&p.d_m, sizeof(int64_t) * np));
      hipStream_t st = ctx->stream;
      // (device 0's part is the head of the batch: its buffers start at offset 0)
      HIP_TRY(hipMemcpyAsync(p.d_m, morton + b, sizeof(int64_t) * np, hipMemcpyHostToDevice, st));
      if (encoder) {
        HIP_TRY(hipMemcpyAsync(p.d_a, attrs + b * c, sizeof(int32_t) * np * c, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemsetAsync(p.d_c, 0, sizeof(int32_t) * np * c, st));
      } else {
        HIP_TRY(hipMemcpyAsync(p.d_c, coeffs + b * c, sizeof(int32_t) * np * c, hipMemcpyHostToDevice, st));
      }
      std::vector<int64_t> offs(p.s1 - p.s0 + 1);
      for (int s = p.s0; s <= p.s1; s++)
        offs[s - p.s0] = offsets[s] - b;
      r = dev_transform(ctx, params, encoder, p.s1 - p.s0, offs.data(), p.d_m, nullptr, p.d_a, p.d_c, c, bits);
      if (r)
        return r;
    }
    // gather on device 0 (behind each device's transform, on its stream)
    auto& root = m->part[0];
    hipStream_t st0 = m->ctx[0]->stream;
    // every ncclResult is kept: a failed enqueue (a communicator that went bad after a
    // device error) must not let the host download an ungathered buffer as a result
    int nccl_err = 0;
    auto nccl = [&](int r) {
      if (r && !nccl_err)
        nccl_err = r;
    };
    if (!m->comm.empty())
      nccl(m->rccl.GroupStart());
    for (int d = 1; d < nd; d++) {
      auto& p = m->part[d];
      const int64_t b = offsets[p.s0], np = offsets[p.s1] - b;
      if (np == 0)
        continue;
      hipStream_t st = m->ctx[d]->stream;
      // The COEFFICIENTS are gathered on device 0 over xGMI (they feed one arithmetic
      // coder); the reconstruction goes home from the device that made it.
      if (encoder) {
        if (!m->comm.empty()) {
          nccl(m->rccl.Send(p.d_c, (size_t)np * c, kNcclInt32, 0, m->comm[d], st));
          nccl(m->rccl.Recv(root.d_c + b * c, (size_t)np * c, kNcclInt32, d, m->comm[0], st0));
        } else {
          // one physical device behind several entries: a copy on the producer's stream
          HIP_TRY(hipSetDevice(m->ctx[d]->device));
          HIP_TRY(hipMemcpyAsync(root.d_c + b * c, p.d_c, sizeof(int32_t) * np * c, hipMemcpyDeviceToDevice, st));
        }
      }
    }
    if (!m->comm.empty()) {
      nccl(m->rccl.GroupEnd());  // (the group is closed even after a failed enqueue)
      if (nccl_err)
        return fail(GPCC_ERR_HIP, std::string("RCCL gather: ") + m->rccl.GetErrorString(nccl_err));
    }
    // every device done and error free, then one download from device 0
    for (int d = nd - 1; d >= 0; d--) {
      HIP_TRY(hipSetDevice(m->ctx[d]->device));
      HIP_TRY(hipStreamSynchronize(m->ctx[d]->stream));
      r = check_device_error(m->ctx[d]);
      if (r)
        return r;
    }
    // (outputs only now: a failed device leaves the caller's buffers as they were)
    for (int d = 0; d < nd; d++) {
      auto& p = m->part[d];
      const int64_t b = offsets[p.s0], np = offsets[p.s1] - b;
      if (np == 0)
        continue;
      HIP_TRY(hipSetDevice(m->ctx[d]->device));
      HIP_TRY(hipMemcpyAsync(attrs + b * c, p.d_a, sizeof(int32_t) * np * c, hipMemcpyDeviceToHost, m->ctx[d]->stream));
    }
    HIP_TRY(hipSetDevice(m->ctx[0]->device));
    if (encoder)
      HIP_TRY(hipMemcpyAsync(coeffs, root.d_c, sizeof(int32_t) * n_total * c, hipMemcpyDeviceToHost, st0));
    for (int d = 0; d < nd; d++) {
      HIP_TRY(hipSetDevice(m->ctx[d]->device));
      HIP_TRY(hipStreamSynchronize(m->ctx[d]->stream));
    }
    return GPCC_OK;
  };
  r = run();
  release();
  return r;
}

}  // namespace

namespace {

// The LoD-based coders of a batch: every device codes its run of slices with the
// host-tier one-call entry (upload, LoD build, transform, download -- the values go
// straight back into the caller's buffers, which is where the host's arithmetic coder
// reads them), one host thread per device so that the devices work concurrently.
// A failed slice fails the call; the first failure's message is the caller's.
template<class Slice>
int
multi_slices(gpcc_multi* m, int32_t num_slices, const int64_t* offsets, int32_t c, Slice&& slice)
{
  if (!m)
    return fail(GPCC_ERR_INVALID_ARG, "multi context is null");
  int r = check_slices(m->ctx[0], num_slices, offsets);
  if (r)
    return r;
  if (c != 1 && c != 3)
    return fail(GPCC_ERR_INVALID_ARG, "attribute count not 1 / 3");
  for (int s = 0; s < num_slices; s++)
    if (offsets[s + 1] - offsets[s] > INT32_MAX)
      return fail(GPCC_ERR_INVALID_ARG, "slice larger than 2^31 points");
  const int nd = (int)m->devices.size();
  shard_slices(num_slices, offsets, nd, &m->part);
  std::vector<int> rc(nd, GPCC_OK);
  std::vector<std::string> msg(nd);
  std::vector<std::thread> th;
  for (int d = 0; d < nd; d++)
    th.emplace_back([&, d]() {
      for (int s = m->part[d].s0; s < m->part[d].s1 && rc[d] == GPCC_OK; s++) {
        rc[d] = slice(m->ctx[d], s, offsets[s], (int32_t)(offsets[s + 1] - offsets[s]));
        if (rc[d])
          msg[d] = g_last_error;  // (thread local: carried over to the caller's thread)
      }
    });
  for (auto& t : th)
    t.join();
  for (int d = 0; d < nd; d++)
    if (rc[d])
      return fail(rc[d], msg[d]);
  return GPCC_OK;
}

}  // namespace

extern "C" {

int
gpcc_multi_create(const int32_t* devices, int32_t num_devices, gpcc_multi** out)
{
  if (!out || !devices || num_devices < 1 || num_devices > 64)
    return fail(GPCC_ERR_INVALID_ARG, "bad device list");
  *out = nullptr;
  gpcc_multi* m = new gpcc_multi();
  bool distinct = true;
  for (int i = 0; i < num_devices; i++) {
    for (int k = 0; k < i; k++)
      distinct = distinct && devices[i] != devices[k];
    gpcc_ctx* ctx = nullptr;
    int r = gpcc_ctx_create(devices[i], nullptr, &ctx);
    if (r) {
      gpcc_multi_destroy(m);
      return r;
    }
    m->devices.push_back(devices[i]);
    m->ctx.push_back(ctx);
  }
  if (num_devices > 1 && distinct) {
    if (!m->rccl.load()) {
      gpcc_multi_destroy(m);
      return fail(GPCC_ERR_NO_DEVICE, "librccl.so could not be loaded for the multi-GPU gather");
    }
    m->comm.resize(num_devices);
    std::vector<int> devs(devices, devices + num_devices);
    const int e = m->rccl.CommInitAll(m->comm.data(), num_devices, devs.data());
    if (e) {
      m->comm.clear();
      const std::string msg = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(e);
      gpcc_multi_destroy(m);
      return fail(GPCC_ERR_HIP, msg);
    }
  }
  *out = m;
  return GPCC_OK;
}

void
gpcc_multi_destroy(gpcc_multi* m)
{
  if (!m)
    return;
  for (auto c : m->comm)
    if (c)
      m->rccl.CommDestroy(c);
  for (auto ctx : m->ctx)
    gpcc_ctx_destroy(ctx);
  delete m;
}

int
gpcc_multi_num_devices(const gpcc_multi* m)
{
  return m ? (int)m->devices.size() : 0;
}

int
gpcc_multi_uses_rccl(const gpcc_multi* m)
{
  return m && !m->comm.empty();
}

// librccl loads, the seven symbols resolve, and a one-rank communicator moves a buffer
// through ncclSend / ncclRecv on `device`: what can be checked of the gather's transport
// on a box with a single GPU.  0 = fine.
int
gpcc_multi_rccl_selftest(int32_t device)
{
  static RcclApi api;
  if (!api.load())
    return fail(GPCC_ERR_NO_DEVICE, "librccl could not be loaded or lacks a symbol");
  HIP_TRY(hipSetDevice(device));
  ncclComm_t comm = nullptr;
  int dev = device;
  int e = api.CommInitAll(&comm, 1, &dev);
  if (e)
    return fail(GPCC_ERR_HIP, std::string("ncclCommInitAll: ") + api.GetErrorString(e));
  const int n = 1 << 16;
  int32_t *a = nullptr, *b = nullptr;
  hipStream_t st = nullptr;
  int rc = GPCC_OK;
  auto run = [&]() -> int {
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    HIP_TRY(hipMalloc((void**)&a, sizeof(int32_t) * n));
    HIP_TRY(hipMalloc((void**)&b, sizeof(int32_t) * n));
    std::vector<int32_t> h(n), g(n, 0);
    for (int i = 0; i < n; i++)
      h[i] = i * 2654435761u;
    HIP_TRY(hipMemcpyAsync(a, h.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(b, 0, sizeof(int32_t) * n, st));
    int err = api.GroupStart();
    err = err ? err : api.Send(a, n, kNcclInt32, 0, comm, st);
    err = err ? err : api.Recv(b, n, kNcclInt32, 0, comm, st);
    const int end = api.GroupEnd();
    err = err ? err : end;
    if (err)
      return fail(GPCC_ERR_HIP, std::string("RCCL send / receive: ") + api.GetErrorString(err));
    HIP_TRY(hipMemcpyAsync(g.data(), b, sizeof(int32_t) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (g != h)
      return fail(GPCC_ERR_HIP, "RCCL send / receive returned different data");
    return GPCC_OK;
  };
  rc = run();
  if (a)
    hipFree(a);
  if (b)
    hipFree(b);
  if (st)
    hipStreamDestroy(st);
  api.CommDestroy(comm);
  return rc;
}

int
gpcc_multi_lift_encode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_lift_params* lift, int32_t num_slices,
  const int64_t* offsets, const int32_t* xyz, int32_t* attrs, int32_t* coeffs,
  int8_t* lcp_coeffs, int32_t* indexes, int32_t c)
{
  if (!lod || !lift || !xyz || !attrs || !coeffs || !lcp_coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null argument");
  return multi_slices(m, num_slices, offsets, c, [&](gpcc_ctx* ctx, int s, int64_t o, int32_t n) {
    return gpcc_lift_encode_attr(
      ctx, lod, &lift[s], xyz + 3 * o, attrs + c * o, coeffs + c * o, lcp_coeffs + (size_t)s * GPCC_MAX_LODS,
      indexes ? indexes + o : nullptr, n, c);
  });
}

int
gpcc_multi_lift_decode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_lift_params* lift, int32_t num_slices,
  const int64_t* offsets, const int32_t* xyz, int32_t* attrs, const int32_t* coeffs,
  const int8_t* lcp_coeffs, int32_t* indexes, int32_t c)
{
  if (!lod || !lift || !xyz || !attrs || !coeffs || !lcp_coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null argument");
  return multi_slices(m, num_slices, offsets, c, [&](gpcc_ctx* ctx, int s, int64_t o, int32_t n) {
    return gpcc_lift_decode_attr(
      ctx, lod, &lift[s], xyz + 3 * o, attrs + c * o, coeffs + c * o, lcp_coeffs + (size_t)s * GPCC_MAX_LODS,
      indexes ? indexes + o : nullptr, n, c);
  });
}

int
gpcc_multi_pred_encode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_pred_params* pred, int32_t num_slices,
  const int64_t* offsets, const int32_t* xyz, int32_t* attrs, int32_t* values,
  int8_t* icp_coeffs, int32_t* indexes, int32_t c)
{
  if (!lod || !pred || !xyz || !attrs || !values || !icp_coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null argument");
  return multi_slices(m, num_slices, offsets, c, [&](gpcc_ctx* ctx, int s, int64_t o, int32_t n) {
    return gpcc_pred_encode_attr(
      ctx, lod, &pred[s], xyz + 3 * o, attrs + c * o, values + c * o, icp_coeffs + (size_t)s * GPCC_MAX_LODS * 3,
      indexes ? indexes + o : nullptr, n, c);
  });
}

int
gpcc_multi_pred_decode_attr(
  gpcc_multi* m, const gpcc_lod_params* lod, gpcc_pred_params* pred, int32_t num_slices,
  const int64_t* offsets, const int32_t* xyz, int32_t* attrs, const int32_t* values,
  const int8_t* icp_coeffs, int32_t* indexes, int32_t c)
{
  if (!lod || !pred || !xyz || !attrs || !values || !icp_coeffs)
    return fail(GPCC_ERR_INVALID_ARG, "null argument");
  return multi_slices(m, num_slices, offsets, c, [&](gpcc_ctx* ctx, int s, int64_t o, int32_t n) {
    return gpcc_pred_decode_attr(
      ctx, lod, &pred[s], xyz + 3 * o, attrs + c * o, values + c * o, icp_coeffs + (size_t)s * GPCC_MAX_LODS * 3,
      indexes ? indexes + o : nullptr, n, c);
  });
}

int
gpcc_multi_raht_forward(
  gpcc_multi* m, const gpcc_raht_params* params, int32_t num_slices, const int64_t* offsets,
  const int64_t* morton, int32_t* attrs, int32_t* coeffs, int32_t c)
{
  return multi_transform(m, params, true, num_slices, offsets, morton, attrs, coeffs, c);
}

int
gpcc_multi_raht_inverse(
  gpcc_multi* m, const gpcc_raht_params* params, int32_t num_slices, const int64_t* offsets,
  const int64_t* morton, int32_t* attrs, const int32_t* coeffs, int32_t c)
{
  return multi_transform(
    m, params, false, num_slices, offsets, morton, attrs, const_cast<int32_t*>(coeffs), c);
}

}  // extern "C"

// ---- binarisation of the residual symbols (residual_bins.hpp) -------------------------
namespace {

int
binarise_symbols(
  gpcc_ctx* ctx, const int32_t* runs, const int32_t* values, int32_t num_symbols,
  int32_t trailing_run, int32_t c, uint8_t* bins, int64_t cap, int64_t* num_bins)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (num_symbols < 0 || (c != 1 && c != 3) || trailing_run < 0 || !num_bins
      || (num_symbols > 0 && (!runs || !values)))
    return fail(GPCC_ERR_INVALID_ARG, "bad symbol stream (c must be 1 or 3)");
  if (num_symbols > kMaxPoints)
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^29 symbols");
  *num_bins = 0;
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const size_t m = (size_t)num_symbols;
  const int nblk = (num_symbols + 1 + kBinBlock - 1) / kBinBlock;
  int32_t *d_runs = nullptr, *d_vals = nullptr, *d_cnt = nullptr, *d_blk = nullptr;
  uint8_t* d_bins = nullptr;
  long long* d_base = nullptr;
  auto cleanup = [&]() {
    pool_free(ctx, d_runs);
    pool_free(ctx, d_vals);
    pool_free(ctx, d_cnt);
    pool_free(ctx, d_blk);
    pool_free(ctx, d_base);
    pool_free(ctx, d_bins);
  };
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, (void**)&d_runs, sizeof(int32_t) * (m + 1)));
    HIP_TRY(pool_malloc(ctx, (void**)&d_vals, sizeof(int32_t) * (m + 1) * c));
    HIP_TRY(pool_malloc(ctx, (void**)&d_cnt, sizeof(int32_t) * (m + 1)));
    HIP_TRY(pool_malloc(ctx, (void**)&d_blk, sizeof(int32_t) * ((size_t)nblk + 1)));
    HIP_TRY(pool_malloc(ctx, (void**)&d_base, sizeof(long long) * ((size_t)nblk + 1)));
    if (m) {
      HIP_TRY(h2d_user(ctx, d_runs, runs, sizeof(int32_t) * m, st));
      HIP_TRY(h2d_user(ctx, d_vals, values, sizeof(int32_t) * m * c, st));
    }
    {
      Timer t(ctx, "bins_count");
      bins_count_kernel<<<nblk, kBinBlock, 0, st>>>(num_symbols, d_runs, d_vals, trailing_run, c, d_cnt, d_blk);
      bins_scan_kernel<<<1, 1024, 0, st>>>(nblk, d_blk, d_base);
    }
    long long total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_base + nblk, sizeof(long long), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *num_bins = total;
    if (total > cap)
      return fail(GPCC_ERR_INVALID_ARG, "bins buffer too small (see *num_bins)");
    if (total == 0)
      return GPCC_OK;
    if (!bins)
      return fail(GPCC_ERR_INVALID_ARG, "bins is null");
    HIP_TRY(pool_malloc(ctx, (void**)&d_bins, (size_t)total));
    {
      Timer t(ctx, "bins_emit");
      bins_emit_kernel<<<nblk, kBinBlock, 0, st>>>(num_symbols, d_runs, d_vals, trailing_run, c, d_cnt, d_base, d_bins);
    }
    HIP_TRY(d2h_user(ctx, bins, d_bins, (size_t)total, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  int r = run();
  cleanup();
  return r;
}

}  // namespace

extern "C" int
gpcc_binarise_symbols(
  gpcc_ctx* ctx, const int32_t* runs, const int32_t* values, int32_t num_symbols,
  int32_t trailing_run, int32_t c, uint8_t* bins, int64_t cap, int64_t* num_bins)
{
  return counted(
    ctx, binarise_symbols(ctx, runs, values, num_symbols, trailing_run, c, bins, cap, num_bins),
    num_symbols);
}

// ---- attribute transfer onto a re-quantised geometry (recolour_kernels.hpp) -----------
namespace {

// inclusive scan of a[0 .. n) in place
int
rc_scan(gpcc_ctx* ctx, int32_t* a, size_t n, long long* sums)
{
  HIP_TRY(kd_scan(ctx->stream, a, n, sums));
  return GPCC_OK;
}

// the working set of one tree's build, from the context's pool (recolour_kdtree.hpp)
struct KdAlloc {
  KdBuild b{};
  std::vector<void*> blocks;
};

int
kd_alloc(gpcc_ctx* ctx, KdAlloc* ka, const int32_t* d_xyz, int n)
{
  const size_t N = (size_t)n, M = 2 * N + 2;
  auto take = [&](void** p, size_t bytes) -> hipError_t {
    hipError_t e = pool_malloc(ctx, p, bytes);
    if (e == hipSuccess)
      ka->blocks.push_back(*p);
    return e;
  };
  KdBuild& b = ka->b;
  b.t.xyz = d_xyz;
  b.t.n = n;
  HIP_TRY(take((void**)&b.t.vind, sizeof(int32_t) * N));
  b.node_cap = (int32_t)kd_node_capacity(N);
  HIP_TRY(take((void**)&b.t.nodes, sizeof(KdNode) * kd_node_capacity(N)));
  HIP_TRY(take((void**)&b.pnode, sizeof(int32_t) * N));
  HIP_TRY(take((void**)&b.rng, sizeof(int32_t) * 2 * M));
  HIP_TRY(take((void**)&b.parent, sizeof(int32_t) * M));
  HIP_TRY(take((void**)&b.box, sizeof(double) * 6 * M));
  HIP_TRY(take((void**)&b.mm, sizeof(int32_t) * 6 * M));
  HIP_TRY(take((void**)&b.cut, sizeof(double) * M));
  HIP_TRY(take((void**)&b.lim, sizeof(int32_t) * 2 * M));
  HIP_TRY(take((void**)&b.split, sizeof(int32_t) * M));
  HIP_TRY(take((void**)&b.flag, sizeof(int32_t) * (N + 1)));
  HIP_TRY(take((void**)&b.tmp_l, sizeof(int32_t) * N));
  HIP_TRY(take((void**)&b.tmp_r, sizeof(int32_t) * N));
  HIP_TRY(take((void**)&b.sums, sizeof(long long) * ((N + 1) / kKdScanBlock + 2)));
  HIP_TRY(take((void**)&b.counters, sizeof(int32_t) * 4));
  HIP_TRY(take((void**)&b.sub_list, sizeof(int32_t) * 2 * (N / 11 + 2)));
  return GPCC_OK;
}

void
kd_release(gpcc_ctx* ctx, KdAlloc* ka, bool keep_tree)
{
  for (void* q : ka->blocks)
    if (!(keep_tree && (q == (void*)ka->b.t.vind || q == (void*)ka->b.t.nodes)))
      pool_free(ctx, q);
  ka->blocks.clear();
}

int
recolour_impl(
  gpcc_ctx* ctx, const gpcc_recolour_params* p, const int32_t* src_xyz, const int32_t* src_attrs,
  int32_t ns, const int32_t* tgt_xyz, int32_t nt, int32_t c, float scale, const int32_t* offset,
  int32_t* tgt_attrs)
{
  if (!ctx)
    return fail(GPCC_ERR_INVALID_ARG, "ctx is null");
  if (!p || !src_xyz || !src_attrs || !tgt_xyz || !tgt_attrs || !offset || ns <= 0 || nt <= 0
      || (c != 1 && c != 3))
    return fail(GPCC_ERR_INVALID_ARG, "null buffer, empty cloud or attribute count not 1 / 3");
  if (ns > (1 << 27) || nt > (1 << 27))
    return fail(GPCC_ERR_INVALID_ARG, "more than 2^27 points");
  const int kf = p->num_neighbours_fwd, kb = p->num_neighbours_bwd;
  if (kf < 1 || kf > kRcMaxK || kb < 1 || kb > kRcMaxK || p->bitdepth < 1 || p->bitdepth > 16
      || p->search_range < 0 || !(scale > 0))
    return fail(GPCC_ERR_INVALID_ARG, "neighbour counts must be 1..8, bitdepth 1..16, scale > 0");
  if (ns < kf || nt < kb)
    return fail(GPCC_ERR_UNSUPPORTED, "fewer points than neighbours asked for");
  HIP_TRY(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;

  int32_t *d_sx = nullptr, *d_sa = nullptr, *d_tx = nullptr, *d_out = nullptr, *d_box = nullptr;
  int32_t *d_ref1 = nullptr, *d_bt = nullptr, *d_lstart = nullptr, *d_lcur = nullptr, *d_lsrc = nullptr;
  int32_t* d_near = nullptr;  // finite forward geometry limit: nearest source point per target, then the first limited target
  double *d_bd = nullptr, *d_ldist = nullptr;
  long long* d_sums = nullptr;
  KdAlloc ks, kt;
  auto cleanup = [&]() {
    pool_free(ctx, d_near);
    for (void* q : {(void*)d_sx, (void*)d_sa, (void*)d_tx, (void*)d_out, (void*)d_box, (void*)d_ref1,
                    (void*)d_bt, (void*)d_lstart, (void*)d_lcur, (void*)d_lsrc, (void*)d_bd, (void*)d_ldist,
                    (void*)d_sums})
      pool_free(ctx, q);
    kd_release(ctx, &ks, false);
    kd_release(ctx, &kt, false);
  };
  auto run = [&]() -> int {
    HIP_TRY(pool_malloc(ctx, (void**)&d_sx, sizeof(int32_t) * 3 * (size_t)ns));
    HIP_TRY(pool_malloc(ctx, (void**)&d_sa, sizeof(int32_t) * (size_t)c * ns));
    HIP_TRY(pool_malloc(ctx, (void**)&d_tx, sizeof(int32_t) * 3 * (size_t)nt));
    HIP_TRY(pool_malloc(ctx, (void**)&d_out, sizeof(int32_t) * (size_t)c * nt));
    HIP_TRY(pool_malloc(ctx, (void**)&d_box, sizeof(int32_t) * 12));
    HIP_TRY(h2d_user(ctx, d_sx, src_xyz, sizeof(int32_t) * 3 * (size_t)ns, st));
    HIP_TRY(h2d_user(ctx, d_sa, src_attrs, sizeof(int32_t) * (size_t)c * ns, st));
    HIP_TRY(h2d_user(ctx, d_tx, tgt_xyz, sizeof(int32_t) * 3 * (size_t)nt, st));
    int32_t h_box[12];
    for (int k = 0; k < 3; k++) {
      h_box[k] = h_box[6 + k] = 0x7fffffff;
      h_box[3 + k] = h_box[9 + k] = -0x7fffffff;
    }
    HIP_TRY(hipMemcpyAsync(d_box, h_box, sizeof(h_box), hipMemcpyHostToDevice, st));
    {
      Timer t(ctx, "rc_bbox");
      // (few workgroups: every wavefront ends with six atomics on the same words)
      rc_bbox_kernel<<<std::min(grid_for(ns, 256), 128), 256, 0, st>>>(d_sx, ns, d_box);
      rc_bbox_kernel<<<std::min(grid_for(nt, 256), 128), 256, 0, st>>>(d_tx, nt, d_box + 6);
    }
    HIP_TRY(hipMemcpyAsync(h_box, d_box, sizeof(h_box), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    for (int k = 0; k < 12; k++)
      if (h_box[k] <= -(1 << 30) || h_box[k] >= (1 << 30))
        return fail(GPCC_ERR_INVALID_ARG, "coordinates outside (-2^30, 2^30)");

    RcCtx cx{};
    int src_depth = 0, tgt_depth = 0;  // levels of the two k-d trees (GPCC_RC_DEBUG prints them)
    cx.p = *p;
    cx.c = c;
    cx.s2t = (double)scale;
    cx.t2s = 1.0 / (double)scale;
    for (int k = 0; k < 3; k++)
      cx.off[k] = offset[k];
    cx.src_attrs = d_sa;

    // ---- the two k-d trees, built together: the source's on the context's stream, the target's on
    //      a second one (a level is a handful of small launches and a look at the node counter) ------
    {
      Timer t(ctx, "rc_kdtree");
      // (the stream and the event are shared with inter-frame RAHT's second candidate)
      if (!ctx->kd_stream)
        HIP_TRY(hipStreamCreateWithFlags(&ctx->kd_stream, hipStreamNonBlocking));
      if (!ctx->kd_event)
        HIP_TRY(hipEventCreateWithFlags(&ctx->kd_event, hipEventDisableTiming));
      if (!ctx->h_kd)
        HIP_TRY(hipHostMalloc((void**)&ctx->h_kd, 8 * sizeof(int32_t)));
      int r = kd_alloc(ctx, &ks, d_sx, ns);
      if (!r)
        r = kd_alloc(ctx, &kt, d_tx, nt);
      if (r)
        return r;
      // (the second stream starts behind the uploads and the bounding boxes)
      HIP_TRY(hipEventRecord(ctx->kd_event, st));
      HIP_TRY(hipStreamWaitEvent(ctx->kd_stream, ctx->kd_event, 0));
      KdLevelLoop loop[2];
      HIP_TRY(loop[0].begin(ks.b, d_box, st, ctx->h_kd));
      HIP_TRY(loop[1].begin(kt.b, d_box + 6, ctx->kd_stream, ctx->h_kd + 4));
      while (!loop[0].done() || !loop[1].done()) {
        for (int w = 0; w < 2; w++)
          if (!loop[w].done())
            HIP_TRY(loop[w].launch());
        for (int w = 0; w < 2; w++)
          if (!loop[w].done())
            HIP_TRY(loop[w].finish());
      }
      // (the context's stream goes on behind the second one)
      HIP_TRY(hipEventRecord(ctx->kd_event, ctx->kd_stream));
      HIP_TRY(hipStreamWaitEvent(st, ctx->kd_event, 0));
      for (int which = 0; which < 2; which++) {
        if (loop[which].tree_depth() > kKdMaxDepth)
          return fail(GPCC_ERR_UNSUPPORTED, "k-d tree deeper than 64 levels: it stays on the reference CPU path");
        (which ? tgt_depth : src_depth) = loop[which].tree_depth();
        KdAlloc& ka = which ? kt : ks;
        KdTree& tr = which ? cx.tgt : cx.src;
        tr = ka.b.t;
        for (int k = 0; k < 3; k++) {
          tr.root_lo[k] = (double)h_box[6 * which + k];
          tr.root_hi[k] = (double)h_box[6 * which + 3 + k];
        }
        kd_release(ctx, &ka, true);  // (the build's working set goes back to the pool; index array and nodes stay)
        ka.blocks.push_back((void*)tr.vind);
        ka.blocks.push_back((void*)tr.nodes);
      }
    }
    static const bool rc_debug = [] {
      const char* e = getenv("GPCC_RC_DEBUG");
      return e && e[0] == '1';
    }();
    // (GPCC_RC_DEBUG=1: the trees' depths and a synchronisation + error check behind every stage)
    auto stage_check = [&](const char* what) -> int {
      if (!rc_debug)
        return GPCC_OK;
      hipError_t e = hipStreamSynchronize(st);
      if (e == hipSuccess)
        e = hipDeviceSynchronize();
      if (e == hipSuccess)
        e = hipGetLastError();
      fprintf(stderr, "gpcc recolour: %s: %s (ns %d nt %d depths %d %d)\n", what, hipGetErrorString(e), ns, nt, src_depth, tgt_depth);
      return e == hipSuccess ? GPCC_OK : fail(GPCC_ERR_HIP, std::string("recolour stage ") + what + ": " + hipGetErrorString(e));
    };
    {
      int rd = stage_check("trees");
      if (rd)
        return rd;
    }
    const size_t total_cap = (size_t)ns * kb;
    HIP_TRY(pool_malloc(ctx, (void**)&d_ref1, sizeof(int32_t) * (size_t)c * nt));
    HIP_TRY(pool_malloc(ctx, (void**)&d_bt, sizeof(int32_t) * total_cap));
    HIP_TRY(pool_malloc(ctx, (void**)&d_bd, sizeof(double) * total_cap));
    HIP_TRY(pool_malloc(ctx, (void**)&d_lstart, sizeof(int32_t) * ((size_t)nt + 1)));
    HIP_TRY(pool_malloc(ctx, (void**)&d_lcur, sizeof(int32_t) * (size_t)nt));
    HIP_TRY(pool_malloc(ctx, (void**)&d_ldist, sizeof(double) * total_cap));
    HIP_TRY(pool_malloc(ctx, (void**)&d_lsrc, sizeof(int32_t) * total_cap));
    HIP_TRY(pool_malloc(ctx, (void**)&d_sums, sizeof(long long) * (((size_t)nt + 1) / kKdScanBlock + 2)));
    cx.ref1 = d_ref1;
    if (p->max_geometry_dist2_fwd < 512) {
      // (round 5: the reference's result vectors shrink for good at the first target beyond the limit)
      HIP_TRY(pool_malloc(ctx, (void**)&d_near, sizeof(int32_t) * ((size_t)nt + 1)));
      cx.nearest = d_near;
      cx.fwd_first = d_near + nt;
      HIP_TRY(hipMemsetAsync(cx.fwd_first, 0x7f, sizeof(int32_t), st));
    }
    cx.bt = d_bt;
    cx.bd = d_bd;
    cx.lstart = d_lstart;
    cx.lcur = d_lcur;
    cx.ldist = d_ldist;
    cx.lsrc = d_lsrc;
    cx.out = d_out;

    // ---- forward, backward, lists, blend ------------------------------------------------
    {
      Timer t(ctx, "rc_forward");
      // list capacity 1 / 2 / 4 / 8 and with / without the attribute limit: 16 variants
      const bool alimit = p->max_attribute_dist2_fwd < 512;
      const int kcap = kf <= 1 ? 1 : kf <= 2 ? 2 : kf <= 4 ? 4 : 8;
      const int fgrid = (nt + 255) / 256;
#define GPCC_RC_FWD(CC, KK)                                                \
  do {                                                                     \
    if (alimit)                                                            \
      rc_forward_kernel<CC, KK, true><<<fgrid, 256, 0, st>>>(cx);          \
    else                                                                   \
      rc_forward_kernel<CC, KK, false><<<fgrid, 256, 0, st>>>(cx);         \
  } while (0)
#define GPCC_RC_FWD_K(CC)                                                  \
  do {                                                                     \
    switch (kcap) {                                                        \
    case 1: GPCC_RC_FWD(CC, 1); break;                                     \
    case 2: GPCC_RC_FWD(CC, 2); break;                                     \
    case 4: GPCC_RC_FWD(CC, 4); break;                                     \
    default: GPCC_RC_FWD(CC, 8); break;                                    \
    }                                                                      \
  } while (0)
      if (c == 3)
        GPCC_RC_FWD_K(3);
      else
        GPCC_RC_FWD_K(1);
#undef GPCC_RC_FWD_K
#undef GPCC_RC_FWD
      if (cx.nearest)
        rc_forward_limit_kernel<<<fgrid, 256, 0, st>>>(cx);
    }
    {
      int rd = stage_check("forward");
      if (rd)
        return rd;
    }
    HIP_TRY(hipMemsetAsync(d_lstart, 0, sizeof(int32_t) * ((size_t)nt + 1), st));
    HIP_TRY(hipMemsetAsync(d_lcur, 0, sizeof(int32_t) * (size_t)nt, st));
    {
      Timer t(ctx, "rc_backward");
      const int bgrid = (ns + 255) / 256;
      switch (kb <= 1 ? 1 : kb <= 2 ? 2 : kb <= 4 ? 4 : 8) {
      case 1: rc_backward_kernel<1><<<bgrid, 256, 0, st>>>(cx); break;
      case 2: rc_backward_kernel<2><<<bgrid, 256, 0, st>>>(cx); break;
      case 4: rc_backward_kernel<4><<<bgrid, 256, 0, st>>>(cx); break;
      default: rc_backward_kernel<8><<<bgrid, 256, 0, st>>>(cx); break;
      }
    }
    {
      int rd = stage_check("backward");
      if (rd)
        return rd;
    }
    {
      Timer t(ctx, "rc_lists");
      int r = rc_scan(ctx, d_lstart, (size_t)nt + 1, d_sums);
      if (r)
        return r;
      rc_list_fill_kernel<<<(ns + 255) / 256, 256, 0, st>>>(cx);
    }
    {
      Timer t(ctx, "rc_blend");
      if (c == 3)
        rc_blend_kernel<3><<<(nt + 255) / 256, 256, 0, st>>>(cx);
      else
        rc_blend_kernel<1><<<(nt + 255) / 256, 256, 0, st>>>(cx);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(d2h_user(ctx, tgt_attrs, d_out, sizeof(int32_t) * (size_t)c * nt, st));
    HIP_TRY(hipStreamSynchronize(st));
    return GPCC_OK;
  };
  const int r = run();
  cleanup();
  return r;
}

}  // namespace

extern "C" int
gpcc_recolour(
  gpcc_ctx* ctx, const gpcc_recolour_params* params, const int32_t* src_xyz,
  const int32_t* src_attrs, int32_t ns, const int32_t* tgt_xyz, int32_t nt, int32_t c,
  float source_to_target_scale, const int32_t target_to_source_offset[3], int32_t* tgt_attrs)
{
  return counted(
    ctx,
    recolour_impl(
      ctx, params, src_xyz, src_attrs, ns, tgt_xyz, nt, c, source_to_target_scale,
      target_to_source_offset, tgt_attrs),
    nt);
}
