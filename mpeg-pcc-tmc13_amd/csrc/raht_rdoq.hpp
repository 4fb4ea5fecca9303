// raht_rdoq.hpp -- resolves the encoder's RDOQ zero-run state in parallel.
//
// The reference (tmc3/RAHT.cpp:1154, 1618-1669) walks ALL coefficients of a
// slice in coding order with one counter, trainZeros: a coefficient whose
// quantised magnitudes sum to < 3 is zeroed iff
// (Dist2 << 26) < lambda * Rate(trainZeros), and the counter is incremented
// on a zeroed / all-zero coefficient and reset otherwise.
//
// Restated without the loop-carried counter: every coefficient either
// increments or resets, so trainZeros before coefficient i equals
// i - 1 - L(i) with L(i) the index of the last reset before i.  Rate() is
// a non-decreasing step function, hence each coefficient carries one
// threshold thr (computed by the analyze pass) and
//     coefficient i resets  <=>  it is a definite reset (magnitude sum >= 3,
//                                or thr unreachable), or
//                                0 < sum < 3 and a reset lies in [i-thr, i-1].
// Resetting is monotone in the set of earlier resets, so the least fixed
// point -- which, by causality, is the sequential answer -- is reached by
// iterating "mark every candidate that sees a known reset in its window";
// each iteration marks ALL such candidates of a 64-coefficient chunk at
// once (ballot), and only chains of failed candidates cost extra rounds.
//
// Across a slice the only carried state is L.  Tiles of 2048 coefficients
// are first evaluated under the two extreme hypotheses (L as old / as
// recent as possible); when both agree on the tile's outgoing L the tile
// is "closed", when neither produces a reset it is "transparent", and only
// the rare remaining ("open") tiles wait for their predecessor
// (rdoq_resolve_kernel: one launch per level).
#pragma once

#include "raht_common.hpp"
#include "raht_levels.hpp"

namespace gpcc {

constexpr int kRdoqTile = 2048;

struct RdoqCtx {
  TreeView tv;
  const SliceSched* sched;
  const int32_t* tile_base;  // [S+1] first global tile of each slice (host built)
  int32_t num_tiles;
  const uint32_t* desc;      // [N], slice s at pt_off[s]
  int32_t* coeffs;
  int32_t* slice_l;          // [S] L carried from level to level
  unsigned long long* state; // [num_tiles] look-back words of rdoq_resolve_kernel
  int32_t li;
  int32_t c;
};

enum { kTileTransparent = 0, kTileClosed = 1, kTileOpen = 2 };

// coefficient range of level li in slice s, slice relative
__device__ __forceinline__ bool
level_coeff_range(const RdoqCtx& cx, int s, int* a, int* b)
{
  const LevelSched e = cx.sched[s].lvl[cx.li];
  if (!e.processed || e.coarse)  // coarse levels resolve their state inside raht_coarse_kernel
    return false;
  const int m = cx.tv.soff[cx.li][s + 1] - cx.tv.soff[cx.li][s];
  const int mp = cx.tv.soff[cx.li + 1][s + 1] - cx.tv.soff[cx.li + 1][s];
  *a = e.coeff_base;
  *b = e.coeff_base + (e.is_root ? m : m - mp);
  return true;
}

// One 64-coefficient chunk.  i = index of this lane's coefficient (slice
// relative), valid = inside the range.  Returns the updated last-reset
// index; *tz_out = zero-run length seen by this lane's coefficient.
__device__ __forceinline__ int
rdoq_chunk(uint32_t d, int i, bool valid, int l_in, int i0, int* tz_out)
{
  const int lane = lane_id();
  const unsigned long long lt = (1ull << lane) - 1;
  const bool z = d >> 31;
  const uint32_t thr = d & kDescNever;
  const bool isdef = valid && !z && thr == kDescNever;
  const bool isthr = valid && !z && thr != kDescNever && thr != 0;
  unsigned long long resets = __ballot(isdef);
  int lhat;
  for (;;) {
    const unsigned long long below = resets & lt;
    lhat = below ? i0 + 63 - __clzll((long long)below) : l_in;
    const bool fail = isthr && !((resets >> lane) & 1)
      && (uint32_t)(i - lhat) <= thr;
    const unsigned long long m = __ballot(fail);
    if (!m)
      break;
    resets |= m;
  }
  *tz_out = i - 1 - lhat;
  return resets ? i0 + 63 - __clzll((long long)resets) : l_in;
}

__device__ __forceinline__ bool
tile_range(const RdoqCtx& cx, int gt, int* s_out, int* a, int* b)
{
  const int s = find_slice(cx.tile_base, cx.tv.num_slices, gt);
  int la, lb;
  if (!level_coeff_range(cx, s, &la, &lb))
    return false;
  const int t0 = (gt - cx.tile_base[s]) * kRdoqTile;
  const int ta = t0 > la ? t0 : la;
  const int tb = t0 + kRdoqTile < lb ? t0 + kRdoqTile : lb;
  *s_out = s;
  *a = ta;
  *b = tb;
  return ta < tb;
}

// ---- classify, carry and apply in ONE launch: decoupled look-back over the tiles ----
// One wavefront per tile, tiles in launch order.  A tile first evaluates itself
// under the two extreme hypotheses (as rdoq_classify_kernel does) and publishes
// what it can: "transparent" (no reset either way: L passes through) or
// "closed" (its outgoing L is the same either way).  Its own incoming L is the
// outgoing L of the nearest predecessor that is not transparent -- found by
// reading 64 predecessors' words at a time -- or what the previous level left
// (slice_l) when the walk reaches the slice's first tile of the level.  Only
// an OPEN predecessor (hypotheses disagree: rare) has to be waited for; it
// publishes its outgoing L once it has its own incoming one.  Words carry the
// level as an epoch, data and flag in one 8-byte relaxed agent-scope access.
// Predecessors have lower wave indices and workgroups are dispatched in
// order, so the tile being waited for is always resident or done.
__global__ __launch_bounds__(256) void
rdoq_resolve_kernel(RdoqCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = lane_id();
  const int gt = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  if (gt >= cx.num_tiles)
    return;
  int s, a, b;
  if (!tile_range(cx, gt, &s, &a, &b))
    return;
  const unsigned long long ep = (unsigned long long)(cx.li + 1) << 48;
  const int pt0 = cx.tv.pt_off[s];
  const int n_s = cx.tv.pt_off[s + 1] - pt0;
  const uint32_t* __restrict__ desc = cx.desc + pt0;
  int32_t* __restrict__ co = cx.coeffs + (size_t)pt0 * cx.c;
  int la_lvl, lb_lvl;
  level_coeff_range(cx, s, &la_lvl, &lb_lvl);
  const int gt_first = cx.tile_base[s] + la_lvl / kRdoqTile;  // the slice's first tile of this level
  const int gt_last = cx.tile_base[s] + (lb_lvl - 1) / kRdoqTile;

  // 1. the tile under the two hypotheses; its descriptors stay in registers
  // (2048 / 64 = 32 per lane)
  uint32_t dreg[kRdoqTile / kWave];
  const int la0 = -1, lb0 = a - 1;
  int la = la0, lb = lb0;
#pragma unroll
  for (int r = 0; r < kRdoqTile / kWave; r++) {
    const int i0 = a + r * kWave;
    const int i = i0 + lane;
    const bool valid = i < b;
    dreg[r] = valid ? desc[i] : kDescZero;
  }
#pragma unroll
  for (int r = 0; r < kRdoqTile / kWave; r++) {
    const int i0 = a + r * kWave;
    if (i0 >= b)
      break;
    const int i = i0 + lane;
    int tz;
    la = rdoq_chunk(dreg[r], i, i < b, la, i0, &tz);
    lb = rdoq_chunk(dreg[r], i, i < b, lb, i0, &tz);
  }
  int status, l_out = 0;
  if (lb == lb0) {
    status = kTileTransparent;
  } else if (la == lb) {
    status = kTileClosed;
    l_out = la;
  } else {
    status = kTileOpen;
  }
  auto replay = [&](int l) -> int {
#pragma unroll
    for (int r = 0; r < kRdoqTile / kWave; r++) {
      const int i0 = a + r * kWave;
      if (i0 >= b)
        break;
      int tz;
      l = rdoq_chunk(dreg[r], i0 + lane, i0 + lane < b, l, i0, &tz);
    }
    return l;
  };

  // 2. incoming L.  The slice's FIRST tile of the level is the only one that
  // reads what the previous level left (slice_l) -- it knows its incoming L at
  // once and always publishes a deciding word, so every walk ends there at the
  // latest; the LAST tile overwrites slice_l for the next level, and only after
  // it has seen the first tile's word (i.e. after that read).
  int l_in = 0;
  if (gt == gt_first) {
    l_in = cx.slice_l[s];
    // (every lane has read the word before lane 0 of a tile that is also the level's last overwrites it:
    // lock step on the hardware, a rendezvous for the CPU emulator of the test tier)
    __builtin_amdgcn_wave_barrier();
    l_out = status == kTileTransparent ? l_in : (status == kTileClosed ? l_out : replay(l_in));
    if (lane == 0)
      __hip_atomic_store(
        &cx.state[gt], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (lane == 0 && status != kTileOpen)
      __hip_atomic_store(
        &cx.state[gt], ep | ((unsigned long long)(status == kTileClosed ? 2 : 1) << 32) | (uint32_t)l_out,
        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool have = false;
    int k0 = gt - 1;  // lane u looks at tile k0 - u
    unsigned spins = 0;
    while (!have) {
      const int k = k0 - lane;
      const bool in = k >= gt_first;
      unsigned long long w = 0;
      if (in)
        w = __hip_atomic_load(&cx.state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool ready = in && (w >> 48) == (unsigned long long)(cx.li + 1);
      const int kind = ready ? (int)((w >> 32) & 0xffff) : 0;
      // the nearest lane that stops the walk: a deciding word, or a tile that
      // has not published yet (everything nearer is transparent)
      const unsigned long long decides = __ballot(ready && kind >= 2);
      const unsigned long long notready = __ballot(in && !ready);
      const unsigned long long stop = decides | notready;
      if (!stop) {
        k0 -= kWave;  // 64 transparent tiles: further back (the first tile always decides)
        continue;
      }
      const int first = __ffsll((long long)stop) - 1;
      if ((decides >> first) & 1) {
        l_in = (int)(uint32_t)__shfl((int)(uint32_t)w, first);
        have = true;
      } else {
        k0 -= first;
        if (++spins > (1u << 22)) {
          if (lane == 0)
            atomicExch(cx.tv.error, 1);
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    // 3. an open tile now knows its outgoing L
    if (status == kTileOpen) {
      l_out = replay(l_in);
      if (lane == 0)
        __hip_atomic_store(
          &cx.state[gt], ep | (3ull << 32) | (uint32_t)l_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (status == kTileTransparent) {
      l_out = l_in;
    }
  }
  if (gt == gt_last) {
    if (gt != gt_first) {
      unsigned spins = 0;
      while ((__hip_atomic_load(&cx.state[gt_first], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 48)
             != (unsigned long long)(cx.li + 1)) {
        if (++spins > (1u << 22)) {
          if (lane == 0)
            atomicExch(cx.tv.error, 1);
          return;
        }
        __builtin_amdgcn_s_sleep(2);
      }
    }
    if (lane == 0)
      cx.slice_l[s] = l_out;  // carried to the next level's launch
  }

  // 4. ... and every tile applies its decisions
  {
    int l = l_in;
#pragma unroll
    for (int r = 0; r < kRdoqTile / kWave; r++) {
      const int i0 = a + r * kWave;
      if (i0 >= b)
        break;
      const int i = i0 + lane;
      const bool valid = i < b;
      int tz;
      l = rdoq_chunk(dreg[r], i, valid, l, i0, &tz);
      const uint32_t thr = dreg[r] & kDescNever;
      if (valid && thr != kDescNever && (uint32_t)tz >= thr) {
        for (int k = 0; k < cx.c; k++)
          co[(size_t)k * n_s + i] = 0;
      }
    }
  }
}

}  // namespace gpcc
