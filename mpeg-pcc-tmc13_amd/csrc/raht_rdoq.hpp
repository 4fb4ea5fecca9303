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
// the rare remaining tiles are re-evaluated in order by the per-slice
// carry pass.
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
  int2* tile_sum;            // [num_tiles] {status, l_out}
  int32_t* tile_lin;         // [num_tiles] incoming L of each tile
  int32_t* slice_l;          // [S] L carried from level to level
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

// pass 1: classify tiles
__global__ __launch_bounds__(256) void
rdoq_classify_kernel(RdoqCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  for (int gt = wave; gt < cx.num_tiles; gt += nwaves) {
    int s, a, b;
    if (!tile_range(cx, gt, &s, &a, &b))
      continue;
    const uint32_t* __restrict__ desc = cx.desc + cx.tv.pt_off[s];
    // hypothesis A: no reset since the slice began; B: reset just before
    const int la0 = -1, lb0 = a - 1;
    int la = la0, lb = lb0;
    for (int i0 = a; i0 < b; i0 += kWave) {
      const int i = i0 + lane;
      const bool valid = i < b;
      const uint32_t d = valid ? desc[i] : kDescZero;
      int tz;
      la = rdoq_chunk(d, i, valid, la, i0, &tz);
      lb = rdoq_chunk(d, i, valid, lb, i0, &tz);
    }
    if (lane == 0) {
      int2 r;
      if (lb == lb0) {
        r.x = kTileTransparent;  // no reset even in the most-reset case
        r.y = 0;
      } else if (la == lb) {
        r.x = kTileClosed;
        r.y = la;
      } else {
        r.x = kTileOpen;
        r.y = 0;
      }
      cx.tile_sum[gt] = r;
    }
  }
}

// pass 2: one wave per slice carries L through the tile summaries
__global__ __launch_bounds__(64) void
rdoq_carry_kernel(RdoqCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = lane_id();
  for (int s = blockIdx.x; s < cx.tv.num_slices; s += gridDim.x) {
    int la, lb;
    if (!level_coeff_range(cx, s, &la, &lb) || la >= lb)
      continue;
    const uint32_t* __restrict__ desc = cx.desc + cx.tv.pt_off[s];
    const int gt0 = cx.tile_base[s] + la / kRdoqTile;
    const int gt1 = cx.tile_base[s] + (lb - 1) / kRdoqTile;
    int l = cx.slice_l[s];
    for (int g0 = gt0; g0 <= gt1; g0 += kWave) {
      const int g = g0 + lane;
      int2 sum = make_int2(kTileTransparent, 0);
      if (g <= gt1)
        sum = cx.tile_sum[g];
      int lin_mine = 0;
      const int cnt = gt1 - g0 + 1 < kWave ? gt1 - g0 + 1 : kWave;
      for (int u = 0; u < cnt; u++) {
        const int st = __shfl(sum.x, u);
        const int lo = __shfl(sum.y, u);
        if (lane == u)
          lin_mine = l;
        if (st == kTileClosed) {
          l = lo;
        } else if (st == kTileOpen) {
          // rare: evaluate the tile with its real incoming state
          const int t0 = (g0 + u - cx.tile_base[s]) * kRdoqTile;
          const int a = t0 > la ? t0 : la;
          const int b = t0 + kRdoqTile < lb ? t0 + kRdoqTile : lb;
          for (int i0 = a; i0 < b; i0 += kWave) {
            const int i = i0 + lane;
            const bool valid = i < b;
            const uint32_t d = valid ? desc[i] : kDescZero;
            int tz;
            l = rdoq_chunk(d, i, valid, l, i0, &tz);
          }
        }
      }
      if (g <= gt1)
        cx.tile_lin[g] = lin_mine;
    }
    if (lane == 0)
      cx.slice_l[s] = l;
  }
}

// pass 3: apply -- zero the coefficients RDOQ drops
__global__ __launch_bounds__(256) void
rdoq_apply_kernel(RdoqCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  for (int gt = wave; gt < cx.num_tiles; gt += nwaves) {
    int s, a, b;
    if (!tile_range(cx, gt, &s, &a, &b))
      continue;
    const int pt0 = cx.tv.pt_off[s];
    const int n_s = cx.tv.pt_off[s + 1] - pt0;
    const uint32_t* __restrict__ desc = cx.desc + pt0;
    int32_t* __restrict__ co = cx.coeffs + (size_t)pt0 * cx.c;
    int l = cx.tile_lin[gt];
    for (int i0 = a; i0 < b; i0 += kWave) {
      const int i = i0 + lane;
      const bool valid = i < b;
      const uint32_t d = valid ? desc[i] : kDescZero;
      int tz;
      l = rdoq_chunk(d, i, valid, l, i0, &tz);
      const uint32_t thr = d & kDescNever;
      if (valid && thr != kDescNever && (uint32_t)tz >= thr) {
        for (int k = 0; k < cx.c; k++)
          co[(size_t)k * n_s + i] = 0;
      }
    }
  }
}

}  // namespace gpcc
