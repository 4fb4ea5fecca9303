// raht_tile.hpp -- the level pass of RAHT without sub-node prediction as a
// TILE kernel: one workgroup owns a contiguous range of parents of one level
// (Morton order), stages everything that range needs with coalesced loads,
// and searches the parent-level neighbours in LDS.
//
// Why.  The blocks of a level are independent (tmc3/RAHT.cpp:1306-1808 with
// raht_subnode_prediction_enabled_flag = 0), so the pass ought to stream.
// The first design (raht_level_kernel) gave every 8-lane group its own chain
// of ~25 dependent global gathers -- worklist, child range, child keys, first
// points, a 12-step lower_bound per neighbour in the parent key array -- and
// a wavefront waits for the SLOWEST of its 64 lanes at every step, so nearly
// every step paid an L2 / HBM miss: 12 % VALU activity, 0.7 G blocks/s.
// Here the node arrays of a level are what they are -- sorted -- so a tile of
// consecutive parents reads CONSECUTIVE ranges of every array:
//
//   stage    fc[j0..j1] (child ranges), key[j0-W .. j1+W) (the parents a
//            neighbour can be, raht_prediction_search_range permitting), the
//            slices' plans, the children's first points / octants and the
//            encoder's source sums: coalesced loads into LDS, every load of a
//            phase issued before the first is consumed (clamped indices, not
//            predicated loads); a hash table over the key window (open
//            addressing, 16-bit window indices, load <= 0.5);
//   classify one thread per parent: single-child parents are finished on the
//            spot (the copy the old prepass kernel made), the others are
//            appended to the tile's block list;
//   blocks   8 lanes per block as before (DPP butterflies); the neighbour
//            look-ups are one or two probes of the LDS table; a neighbour
//            whose allowed index range leaves the staged window falls back to
//            the global lower_bound (rare: Morton neighbours are index
//            neighbours except across large octant boundaries).
//
// What is left per block in global memory: the parent's and the neighbours'
// values (one round trip) and the stores.  The tile size is a template
// parameter: levels where almost every parent has a single child (<= 1.125
// nodes per parent) run kTileTSparse-parent tiles, their work being the
// per-tile fixed phases.
//
// The lossy encoder still needs two passes around the RDOQ resolution
// (raht_rdoq.hpp), but the second one no longer repeats the search and the
// prediction: kAnalyze leaves the transformed prediction of every coefficient
// position in `ptrans`, kSynthRec reads it back (streaming: no key window, no
// neighbour gathers).
//
// The same tile routine, called level after level by ONE workgroup per
// slice, is the coarse kernel: every level whose parents fit a few tiles
// (schedule_kernel decides, LevelSched::coarse) is processed without leaving
// the launch -- analyze, the RDOQ state walked sequentially (it is a few
// thousand coefficients), synthesis -- instead of 4-6 launches per level of a
// few microseconds of work each.
#pragma once

#include "raht_inter.hpp"
#include "raht_levels.hpp"
#include "raht_rdoq.hpp"

namespace gpcc {

// 4 waves/SIMD = 128 registers: no spills in any instantiation (at 5 the C = 3
// kernels keep 60-110 B of scratch); LDS (30 KB per workgroup) allows 5
#ifndef GPCC_TILE_WAVES
#define GPCC_TILE_WAVES 4
#endif
#ifndef GPCC_TILE_T
#define GPCC_TILE_T 256
#endif
#ifndef GPCC_TILE_W
#define GPCC_TILE_W 512
#endif
#ifndef GPCC_TILE_THREADS
#define GPCC_TILE_THREADS 256
#endif
constexpr int kTileT = GPCC_TILE_T;  // parents per tile
constexpr int kTileW = GPCC_TILE_W;  // staged key window on either side
constexpr int kTileThreads = GPCC_TILE_THREADS;
// Two tile sizes: kTileT parents for the levels where blocks are many, and
// kTileTSparse for the levels where almost every parent has one child (the
// host picks per level from the node counts): there a tile's time is its
// fixed phases -- staging rounds, classification, ONE block pass for a
// handful of blocks -- and four times the parents cost about the same.
#ifndef GPCC_TILE_T_SPARSE
#define GPCC_TILE_T_SPARSE 1024
#endif
constexpr int kTileTSparse = GPCC_TILE_T_SPARSE;
constexpr uint32_t kTabEmpty = 0xffffu;
constexpr int kTileSlices = 8;     // slices a tile may span with their info in LDS
#ifndef GPCC_COARSE_PARENTS
#define GPCC_COARSE_PARENTS 1024
#endif
constexpr int kCoarseParents = GPCC_COARSE_PARENTS;  // a level is coarse while a slice has at most this many parents
constexpr int kSynthRec = 4;       // LevelMode: synthesis from the analyze pass's record

// Where a tile's time goes (experiment builds only, -DGPCC_TILE_PROF: s_memtime
// at the phase boundaries, thread 0, summed per mode and level into g_tile_prof;
// read back with gpcc_debug_tile_prof).  Empty otherwise.
#ifdef GPCC_TILE_PROF
__device__ unsigned long long g_tile_prof[5 * 24 * 8];
struct TileProf {
  unsigned long long last;
  int mode, li;
  __device__ TileProf(int m, int l) : mode(m), li(l) { last = __builtin_amdgcn_s_memtime(); }
  __device__ void mark(int phase)
  {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0)
      atomicAdd(&g_tile_prof[(mode * 24 + li) * 8 + phase], t - last);
    last = t;
  }
  __device__ void count(int slot, int n)
  {
    if (threadIdx.x == 0)
      atomicAdd(&g_tile_prof[(mode * 24 + li) * 8 + slot], (unsigned long long)n);
  }
};
#else
struct TileProf {
  __device__ TileProf(int, int) {}
  __device__ void mark(int) {}
  __device__ void count(int, int) {}
};
#endif

struct TileSlice {
  int32_t sp0, sp1;  // the slice's parents  [sp0, sp1) in level li+1
  int32_t sc0;       // first child of the slice in level li
  int32_t pt0, n_s;  // first point, number of points
  LevelSched e;
};

// kSearch: the parent key window (every mode but kSynthRec); SUMC: components
// of the children's source sums (encoder passes), 0 = none
template<bool kSearch, int SUMC, int T = kTileT>
struct TileSmem {
  static constexpr int kT = T;  // parents per tile
  // children staged at a time (a tile with more is worked in parts): a sparse
  // level has hardly more children than parents
  static constexpr int kCC = T >= 1024 ? T + T / 4 : 4 * T;
  // hash table over the staged key window: 16-bit window indices, load <= 0.5
  static constexpr int kWin = T + 2 * kTileW;
  static constexpr int kSlots = kWin * 2 <= 2048 ? 2048 : (kWin * 2 <= 4096 ? 4096 : 8192);
  static constexpr int kHashShift = 32 - (kSlots == 2048 ? 11 : (kSlots == 4096 ? 12 : 13));
  static_assert(kWin < 0xffff && kWin * 2 <= 8192, "window indices are 16 bit, table load <= 0.5");
  SharedLut lut;
  int64_t key[kSearch ? kWin : 1];
  uint32_t tab[kSearch ? kSlots / 2 : 1];  // two 16-bit entries per word
  int32_t fc[T + 1];
  uint16_t blocks[T];
  // the children of the parents being worked on: first points (weights are
  // differences), octants, and what the encoder's source sums come from --
  // the attribute prefix sum at every first point, or the Haar low-pass value
  int32_t cfp[kCC + 1];
  uint8_t coct[kCC];
  int32_t cpre[SUMC ? (kCC + 1) * SUMC : 1];
  TileSlice sl[kTileSlices];
  int32_t nblocks, s_lo, ns, nsl;
};

__device__ __forceinline__ TileSlice
load_tile_slice(const LevelCtx& ctx, int li, int s)
{
  const TreeView& tv = ctx.tv;
  TileSlice r;
  r.sp0 = tv.soff[li + 1][s];
  r.sp1 = tv.soff[li + 1][s + 1];
  r.sc0 = tv.soff[li][s];
  r.pt0 = tv.pt_off[s];
  r.n_s = tv.pt_off[s + 1] - r.pt0;
  r.e = ctx.sched[s].lvl[li];
  return r;
}

// the slice of parent j among the (at most kTileSlices) slices staged in LDS
template<typename Smem>
__device__ __forceinline__ TileSlice
tile_slice(const Smem& sm, int j)
{
  int u = 0;
  while (u + 1 < sm.nsl && j >= sm.sl[u].sp1)
    u++;
  return sm.sl[u];
}

// ---- neighbour lookup in the staged key window -----------------------------------
// findNeighbour (tmc3/RAHT.cpp:272-293) is a lower_bound in the parent key
// array limited to raht_prediction_search_range entries either side: it finds
// the parent with key `want` iff that parent exists at an index in [ga, gb).
// The window's keys sit in LDS behind a hash table (open addressing, 16-bit
// window indices): one or two probes instead of eleven dependent bisection
// steps -- the bisection was a quarter of the block pass's instructions.
template<typename Smem>
__device__ __forceinline__ uint32_t
tile_hash(int64_t key)
{
  const uint32_t x = (uint32_t)key ^ (uint32_t)((uint64_t)key >> 29);
  return (x * 0x9E3779B1u) >> Smem::kHashShift;
}

template<typename Smem>
__device__ __forceinline__ uint32_t
tile_tab_get(const Smem& sm, uint32_t h)
{
  return (sm.tab[h >> 1] >> ((h & 1) * 16)) & 0xffffu;
}

// all threads: clear (before the barrier that also covers the key staging) ...
template<typename Smem>
__device__ __forceinline__ void
tile_tab_clear(Smem& sm)
{
  for (int i = threadIdx.x; i < Smem::kSlots / 2; i += blockDim.x)
    sm.tab[i] = 0xffffffffu;
}
// ... and insert the window's keys (after it)
template<typename Smem>
__device__ __forceinline__ void
tile_tab_build(Smem& sm, int nwin)
{
  for (int i = threadIdx.x; i < nwin; i += blockDim.x) {
    uint32_t h = tile_hash<Smem>(sm.key[i]);
    for (;;) {
      const uint32_t sh = (h & 1) * 16;
      const uint32_t old = sm.tab[h >> 1];
      if (((old >> sh) & 0xffffu) == kTabEmpty) {
        const uint32_t nw = (old & ~(0xffffu << sh)) | ((uint32_t)i << sh);
        if (atomicCAS(&sm.tab[h >> 1], old, nw) == old)
          break;
      } else {
        h = (h + 1) & (Smem::kSlots - 1);
      }
    }
  }
}

// index of the parent with key `want` in [ga, gb), -1 if there is none;
// *outside = the staged window [wlo, whi) cannot tell (the key may lie in the
// part of [ga, gb) that is not staged)
template<typename Smem>
__device__ __forceinline__ int
window_lookup(const Smem& sm, int64_t want, int ga, int gb, int wlo, int whi, bool* outside)
{
  *outside = false;
  uint32_t h = tile_hash<Smem>(want);
  for (;;) {
    const uint32_t i = tile_tab_get(sm, h);
    if (i == kTabEmpty)
      break;
    if (sm.key[i] == want) {
      // (the window may hold other slices of the batch with equal keys)
      const int q = wlo + (int)i;
      if (q >= ga && q < gb)
        return q;
    }
    h = (h + 1) & (Smem::kSlots - 1);
  }
  // ga < wlo implies wlo is inside the searching block's slice, likewise whi
  const bool below = ga < wlo && want < sm.key[0];
  const bool above = gb > whi && want > sm.key[whi - wlo - 1];
  *outside = below || above;
  return -1;
}

// One tile [j0, j1) of the parents of level li + 1 (children in level li).
// All threads of the workgroup call it together.  `honor_coarse`: skip the
// (slice, level) pairs the coarse kernel owns.  INTER: attribute inter prediction (raht_inter.hpp) -- the
// blocks of a level are matched against the reference frame where ctx.inter says so, and the lossy encoder
// writes a second, intra-only candidate of the level (ctx.inter.dual).
template<int C, int MODE, bool INTER = false, typename Smem>
__device__ __forceinline__ void
tile_process(
  const LevelCtx& ctx, const int li, const int j0, const int j1, Smem& sm,
  const bool honor_coarse)
{
  constexpr bool kSearch = MODE != kSynthRec;
  constexpr bool kEnc = MODE == kAnalyze || MODE == kFused;
  constexpr bool kRecon = MODE != kAnalyze;
  constexpr bool kCopy = MODE != kSynthRec;  // this pass finishes the single-child parents
  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const SharedLut& lut = sm.lut;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = lane_id();
  const int t = tid & 7;
  const int gbase = lane & 56;
  const bool haar = prm->integer_haar_enable_flag != 0;
  const bool ext = prm->raht_extension != 0;
  const int S = tv.num_slices;
  const int32_t* __restrict__ soffP = tv.soff[li + 1];
  const int nt = j1 - j0;

  // ---- stage: everything the tile needs that is known up front, in ONE round
  //      of coalesced loads -- child ranges, the parent key window, the slices
  //      the tile lies in, and (when they fit) the children's first points and
  //      octants; the encoder's source sums follow (they need the first points).
  //      A tile's fixed cost is what most tiles of a sparse level consist of. ------
  __syncthreads();  // the previous tile's readers are done with the LDS arrays
  TileProf prof(MODE, li);
  const int32_t* __restrict__ pfc = tv.fc[li + 1];
  const int c_lo = pfc[j0], c_hi = pfc[j1];  // (uniform addresses: scalar loads)
  const bool one_part = c_hi - c_lo <= Smem::kCC;
  TileSlice mine;
  bool have = false;
  if (S <= nthr) {
    // one thread per slice: which slices hold the tile's first / last parent
    if (tid < S) {
      const int a = soffP[tid], b = soffP[tid + 1];
      if (a <= j0 && j0 < b)
        sm.s_lo = tid;
      if (a < j1 && j1 <= b)
        sm.ns = tid;  // last slice, turned into a count below
      if (b > j0 && a < j1) {
        have = true;
        mine = load_tile_slice(ctx, li, tid);
      }
    }
  } else if (tid == 0) {
    sm.s_lo = find_slice(soffP, S, j0);
    sm.ns = find_slice(soffP, S, j1 - 1);
  }
  // Every load of the round is issued before the first result is used: a loop
  // "load, store to LDS, next" makes each of its iterations a round trip of its
  // own (five for the key window), and the staging rounds are what most tiles
  // of a sparse level consist of.  (Trip counts for 256 threads; the coarse
  // kernel's 1 024 threads cover the ranges in fewer.)
  constexpr int kRfc = (Smem::kT + 1 + 255) / 256;
  constexpr int kRkey = (Smem::kWin + 255) / 256;
  constexpr int kRcc = (Smem::kCC + 1 + 255) / 256;
  int wlo = 0, whi = 0;
  const int32_t* __restrict__ pfp = tv.fp[li];
  const int64_t* __restrict__ pck = tv.key[li];
  {
    int32_t r_fc[kRfc];
#pragma unroll
    for (int r = 0; r < kRfc; r++) {
      const int i = tid + r * nthr;
      r_fc[r] = pfc[j0 + (i <= nt ? i : nt)];  // (clamped, not predicated: a select of addresses would be a flat load)
    }
    int64_t r_key[kSearch ? kRkey : 1];
    if (kSearch) {
      const int np_all = soffP[S];
      wlo = j0 - kTileW > 0 ? j0 - kTileW : 0;
      whi = j1 + kTileW < np_all ? j1 + kTileW : np_all;
      const int64_t* __restrict__ pk = tv.key[li + 1];
#pragma unroll
      for (int r = 0; r < kRkey; r++) {
        const int i = tid + r * nthr;
        r_key[r] = pk[wlo + (i < whi - wlo ? i : whi - wlo - 1)];
      }
    }
    int32_t r_cfp[kRcc];
    int64_t r_ck[kRcc];
    const int cn = one_part ? c_hi - c_lo : -1;
    const int cnl = cn > 0 ? cn : 0, ckl = cn > 1 ? cn - 1 : 0;
#pragma unroll
    for (int r = 0; r < kRcc; r++) {
      const int i = tid + r * nthr;
      r_cfp[r] = pfp[c_lo + (i <= cnl ? i : cnl)];
      r_ck[r] = pck[c_lo + (i <= ckl ? i : ckl)];
    }
    if (kSearch)
      tile_tab_clear(sm);
#pragma unroll
    for (int r = 0; r < kRfc; r++) {
      const int i = tid + r * nthr;
      if (i <= nt)
        sm.fc[i] = r_fc[r];
    }
    if (kSearch) {
#pragma unroll
      for (int r = 0; r < kRkey; r++) {
        const int i = tid + r * nthr;
        if (i < whi - wlo)
          sm.key[i] = r_key[r];
      }
    }
#pragma unroll
    for (int r = 0; r < kRcc; r++) {
      const int i = tid + r * nthr;
      if (i <= cn)
        sm.cfp[i] = r_cfp[r];
      if (i < cn)
        sm.coct[i] = (uint8_t)(r_ck[r] & 7);
    }
  }
  __syncthreads();
  prof.mark(0);  // first round of staging
  const int s_lo = sm.s_lo, ns = sm.ns - s_lo + 1;
  // second round: the slices' plans into LDS, the source sums
  auto stage_sums = [&](int cb, int cn) {
    if (!kEnc)
      return;
    if (haar) {
      const int32_t* __restrict__ lf = ctx.haar_lf[li];
      constexpr int kR = (Smem::kCC * C + 255) / 256;
      int32_t v[kR];
#pragma unroll
      for (int r = 0; r < kR; r++) {
        const int i = tid + r * nthr;
        v[r] = lf[(size_t)cb * C + (i < cn * C ? i : 0)];
      }
#pragma unroll
      for (int r = 0; r < kR; r++) {
        const int i = tid + r * nthr;
        if (i < cn * C)
          sm.cpre[i] = v[r];
      }
    } else {
      int32_t v[kRcc][C];
#pragma unroll
      for (int r = 0; r < kRcc; r++) {
        const int i = tid + r * nthr;
        const size_t f = (size_t)sm.cfp[i <= cn ? i : 0];
#pragma unroll
        for (int k = 0; k < C; k++)
          v[r][k] = ctx.attr_prefix[f * C + k];
      }
#pragma unroll
      for (int r = 0; r < kRcc; r++) {
        const int i = tid + r * nthr;
        if (i <= cn) {
#pragma unroll
          for (int k = 0; k < C; k++)
            sm.cpre[i * C + k] = v[r][k];
        }
      }
    }
  };
  if (S <= nthr) {
    if (have && tid - s_lo < kTileSlices)
      sm.sl[tid - s_lo] = mine;
  } else if (tid < ns && tid < kTileSlices) {
    sm.sl[tid] = load_tile_slice(ctx, li, s_lo + tid);
  }
  if (one_part)
    stage_sums(c_lo, c_hi - c_lo);
  if (kSearch)
    tile_tab_build(sm, whi - wlo);
  if (tid == 0) {
    sm.nsl = ns < kTileSlices ? ns : kTileSlices;
    sm.nblocks = 0;
  }
  __syncthreads();
  prof.mark(1);  // second round (plans, source sums)

  // A tile normally lies inside one slice; one over many small slices is
  // worked through kTileSlices slices at a time (their offsets and level
  // plan sit in LDS).
  for (int sb = 0; sb < ns; sb += kTileSlices) {
  const int nsl = ns - sb < kTileSlices ? ns - sb : kTileSlices;
  if (sb) {
    __syncthreads();  // the previous window has been consumed
    if (tid < nsl)
      sm.sl[tid] = load_tile_slice(ctx, li, s_lo + sb + tid);
    if (tid == 0) {
      sm.nsl = nsl;
      sm.nblocks = 0;
    }
    __syncthreads();
  }
  // parents of this window of slices inside the tile
  const int ja = sm.sl[0].sp0 > j0 ? sm.sl[0].sp0 : j0;
  const int jz = sm.sl[nsl - 1].sp1 < j1 ? sm.sl[nsl - 1].sp1 : j1;

  // The children of a run of parents are a contiguous range of level li; when
  // a tile has more than Smem::kCC of them they are staged a part at a time.
  for (int pa = ja; pa < jz;) {
  int pb = jz, cb = c_lo;
  if (!one_part) {
    const int cb0 = sm.fc[pa - j0];
    int lo = pa + 1, hi = jz;  // a parent has at most 8 children: at least one fits
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (sm.fc[mid - j0] - cb0 <= Smem::kCC)
        lo = mid;
      else
        hi = mid - 1;
    }
    pb = lo;
    cb = cb0;
    const int cn = sm.fc[pb - j0] - cb;
    __syncthreads();  // whatever used the child arrays and the block lists is done
    if (tid == 0) {
      sm.nblocks = 0;
    }
    {
      int32_t r_cfp[kRcc];
      int64_t r_ck[kRcc];
#pragma unroll
      for (int r = 0; r < kRcc; r++) {
        const int i = tid + r * nthr;
        r_cfp[r] = pfp[cb + (i <= cn ? i : cn)];
        r_ck[r] = pck[cb + (i < cn ? i : cn - 1)];
      }
#pragma unroll
      for (int r = 0; r < kRcc; r++) {
        const int i = tid + r * nthr;
        if (i <= cn)
          sm.cfp[i] = r_cfp[r];
        if (i < cn)
          sm.coct[i] = (uint8_t)(r_ck[r] & 7);
      }
    }
    __syncthreads();
    if (kEnc) {
      stage_sums(cb, cn);
      __syncthreads();
    }
  }

  // ---- classify: one thread per parent ----------------------------------------
  for (int jb = pa; jb < pb; jb += nthr) {
    const int j = jb + tid;
    const int jl = j - j0;
    bool real = false;
    if (j < pb) {
      const TileSlice sl = tile_slice(sm, j);
      const LevelSched e = sl.e;
      if (e.processed && !(honor_coarse && e.coarse)) {
        const int c0 = sm.fc[jl];
        const int nchild = sm.fc[jl + 1] - c0;
        if (ext && nchild == 1) {
          // no prediction, no coefficient: the inherited DC moves to the
          // child's position and the child has its parent's weight, so its
          // reconstruction IS the parent's (tmc3/RAHT.cpp:1382-1401)
          if (kCopy) {
            const int64_t prow = (int64_t)sl.pt0 + (j - sl.sp0);
            const int64_t crow = (int64_t)sl.pt0 + (c0 - sl.sc0);
            const int pp = e.parity ^ 1, cp = e.parity;
#pragma unroll
            for (int k = 0; k < C; k++) {
              par2(ctx.rec_us, cp)[crow * C + k] = par2(ctx.rec_us, pp)[prow * C + k];
              par2(ctx.rec, cp)[crow * C + k] = par2(ctx.rec, pp)[prow * C + k];
            }
            par2(ctx.nneigh, cp)[crow] = 19;
            if (ctx.asc_qp) {
              par2(ctx.dqp, cp)[crow * 2] = par2(ctx.dqp, pp)[prow * 2];
              par2(ctx.dqp, cp)[crow * 2 + 1] = par2(ctx.dqp, pp)[prow * 2 + 1];
            }
          }
        } else {
          real = true;
        }
      }
    }
    const unsigned long long m = __ballot(real);
    int at = 0;
    if (lane == 0 && m)
      at = atomicAdd(&sm.nblocks, __popcll(m));
    at = __shfl(at, 0);
    if (real)
      sm.blocks[at + __popcll(m & ((1ull << lane) - 1))] = (uint16_t)jl;
  }
  __syncthreads();
  prof.mark(2);  // classification + single-child copies

  // ---- blocks: 8 lanes each ------------------------------------------------------
  const int nblocks = sm.nblocks;
  prof.count(7, nblocks);
  prof.count(5, 1);
  const int gpp = nthr >> 3;  // groups per pass
  for (int b0 = 0; b0 < nblocks; b0 += gpp) {
    const int bi = b0 + (tid >> 3);
    const bool on = bi < nblocks;
    // a wavefront none of whose eight groups has a block skips the pass (no
    // barrier and no cross-wave exchange inside it): at the sparse levels a
    // tile has fewer blocks than one wavefront takes
    if (!__any(on))
      continue;
    const int jl = on ? sm.blocks[bi] : 0;
    const int j = j0 + jl;
    // (lanes of idle groups read the slice of j0: harmless, they store nothing)
    const TileSlice sl = tile_slice(sm, j);
    const LevelSched e = sl.e;
    const int sp0 = sl.sp0, sp1 = sl.sp1, sc0 = sl.sc0, pt0 = sl.pt0, n_s = sl.n_s;
    const int c0 = on ? sm.fc[jl] : 0;
    const int nchild = on ? sm.fc[jl + 1] - c0 : 0;
    const int pj = j - sp0;
    const int par_par = e.parity ^ 1, cur_par = e.parity;
    const int64_t prow = (int64_t)pt0 + pj;  // parent row in the rec buffers

    // ---- children -> positions, from the staged arrays ------------------------
    const int lc0 = c0 - cb;  // the block's first child in the staged range
    const uint32_t occ = group8_or(t < nchild ? 1u << sm.coct[lc0 + t] : 0u);
    const bool has = (occ >> t) & 1;
    const int cu = popc32(occ & ((1u << t) - 1));
    const int child = c0 + cu;
    const int64_t crow = (int64_t)pt0 + (child - sc0);
    int32_t w = 0;
    int64_t src[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      src[k] = 0;
    if (has) {
      w = sm.cfp[lc0 + cu + 1] - sm.cfp[lc0 + cu];
      if (kEnc) {
        if (haar) {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = fp_from_int(sm.cpre[(lc0 + cu) * C + k]);
        } else {
          // node sum = difference of the modular prefix sums: the reference
          // accumulates these sums in `int` as well (tmc3/RAHT.cpp:131,196)
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] = fp_from_int((int32_t)(
              (uint32_t)sm.cpre[(lc0 + cu + 1) * C + k] - (uint32_t)sm.cpre[(lc0 + cu) * C + k]));
        }
      }
    }
    // parent values every mode needs
    const bool inherit_dc = !e.is_root;
    int pneigh = 0;
    if (kSearch && on && inherit_dc)
      pneigh = par2(ctx.nneigh, par_par)[prow];
    int64_t dcv[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      dcv[k] = 0;
    if (kRecon && on && inherit_dc && t == 0) {
#pragma unroll
      for (int k = 0; k < C; k++)
        dcv[k] = par2(ctx.rec_us, par_par)[prow * C + k];
    }

    // ---- neighbour search in the staged window (tmc3/RAHT.cpp:272-293,
    //      299-368): the neighbours some occupied child position uses (:340) ------
    int pn[3] = {-1, -1, -1};  // neighbour i = 1 + t + 8*slot
    int fb_ga[3] = {0, 0, 0}, fb_gb[3] = {0, 0, 0};
    int64_t want[3] = {0, 0, 0};
    bool fb[3] = {false, false, false};
    if (kSearch) {
      const int64_t cur_pos = on ? sm.key[j - wlo] : 0;
      const uint64_t base = morton3d_add((uint64_t)cur_pos, ~0ull);
      const int64_t range = prm->raht_prediction_search_range;
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
        const int i = 1 + t + 8 * slot;
        if (on && i < 19 && (occ & neigh_mask(i))) {
          const int64_t np = (int64_t)morton3d_add(base, neigh_offset(i));
          int64_t d = np - cur_pos;
          int ga, gb;  // global index range findNeighbour may look at
          if (d >= 0) {
            d = d >= range ? range : d;
            ga = j;
            gb = (d + 1 < (int64_t)(sp1 - j)) ? j + (int)(d + 1) : sp1;
          } else {
            d = (-d) >= range ? range : -d;
            gb = j;
            ga = (d < (int64_t)(j - sp0)) ? j - (int)d : sp0;
          }
          fb_ga[slot] = ga;
          fb_gb[slot] = gb;
          want[slot] = np;
          if (ga < gb)
            pn[slot] = window_lookup(sm, np, ga, gb, wlo, whi, &fb[slot]);
        }
      }
    }

    // ---- inter-level prediction gating (tmc3/RAHT.cpp:1391-1432) --------------
    const bool pred_in_level =
      kSearch && on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
    bool enable_pred = pred_in_level;
    int neigh_count = 0;
    bool do_search = false;
    if (pred_in_level) {
      if (ext && nchild == 1) {
        enable_pred = false;
        neigh_count = 19;
      } else if (pneigh < prm->raht_prediction_threshold0) {
        enable_pred = false;
      } else {
        do_search = true;
      }
    }
    if (kSearch) {
      // only the neighbours some occupied child position uses are searched
      // (tmc3/RAHT.cpp:340)
      bool any_fb = false;
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
        const int i = 1 + t + 8 * slot;
        const bool needed = do_search && i < 19 && (occ & neigh_mask(i));
        if (!needed)
          pn[slot] = -1;
        fb[slot] = fb[slot] && needed;
        any_fb |= fb[slot];
      }
      if (__any(any_fb)) {
        // the global lower_bound, for the few that left the window
        int lo[3], hi[3];
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          lo[slot] = fb[slot] ? fb_ga[slot] : 0;
          hi[slot] = fb[slot] ? fb_gb[slot] : 0;
        }
        const int64_t* __restrict__ pkey = tv.key[li + 1];
        while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]))) {
          int mid[3];
          int64_t kv[3];
#pragma unroll
          for (int slot = 0; slot < 3; slot++) {
            mid[slot] = lo[slot] + ((hi[slot] - lo[slot]) >> 1);
            kv[slot] = lo[slot] < hi[slot] ? pkey[mid[slot]] : 0;
          }
#pragma unroll
          for (int slot = 0; slot < 3; slot++) {
            if (lo[slot] < hi[slot]) {
              if (kv[slot] < want[slot])
                lo[slot] = mid[slot] + 1;
              else
                hi[slot] = mid[slot];
            }
          }
        }
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          if (fb[slot] && lo[slot] < fb_gb[slot] && pkey[lo[slot]] == want[slot])
            pn[slot] = lo[slot];
        }
      }
      int found = (pn[0] >= 0) + (pn[1] >= 0) + (pn[2] >= 0);
      found = group8_sum(found);
      if (do_search) {
        neigh_count = found + 1;
        if (neigh_count < prm->raht_prediction_threshold1)
          enable_pred = false;
      }
    }

    // ---- node qp on the way down (see oracle/raht_oracle.c,
    //      descend_block_qp; tmc3/RAHT.cpp:185-189 vs :246-253) ----------------
    int32_t nq0 = 0, nq1 = 0;
    if (ctx.asc_qp) {
      int32_t a0 = 0, a1 = 0;
      if (has) {
        a0 = ctx.asc_qp[li][(size_t)child * 2];
        a1 = ctx.asc_qp[li][(size_t)child * 2 + 1];
      }
      int32_t wa = w, b0q = a0, b1q = a1;  // current sub-tree weight, average
      int32_t st_w[3], st_a0[3], st_a1[3], st_pw[3];
#pragma unroll
      for (int st = 0; st < 3; st++) {
        const int bit = 1 << st;
        const int32_t pw = lane_xor8(wa, bit);
        const int32_t p0 = lane_xor8(b0q, bit), p1 = lane_xor8(b1q, bit);
        st_w[st] = wa;
        st_a0[st] = b0q;
        st_a1[st] = b1q;
        st_pw[st] = pw;
        if (wa && pw) {
          b0q = (b0q + p0) >> 1;
          b1q = (b1q + p1) >> 1;
        } else if (pw) {
          b0q = p0;
          b1q = p1;
        }
        wa += pw;
      }
      int32_t d0 = on ? par2(ctx.dqp, par_par)[prow * 2] : 0;
      int32_t d1 = on ? par2(ctx.dqp, par_par)[prow * 2 + 1] : 0;
#pragma unroll
      for (int st = 2; st >= 0; st--) {
        const int bit = 1 << st;
        if ((t & bit) && st_w[st] && st_pw[st]) {
          d0 = st_a0[st];
          d1 = st_a1[st];
        }
      }
      if (has) {
        nq0 = d0 >> 4;
        nq1 = d1 >> 4;
        if (kRecon) {
          par2(ctx.dqp, cur_par)[crow * 2] = d0;
          par2(ctx.dqp, cur_par)[crow * 2 + 1] = d1;
        }
      }
    }

    // ---- butterfly weights + coefficients (mkWeightTree :742) -------------------
    int32_t wl[3], wr[3];
    int64_t ca[3], cb[3];
    int32_t cw = w;
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int bit = 1 << st;
      const int32_t pw = lane_xor8(cw, bit);
      const bool left = !(t & bit);
      wl[st] = left ? cw : pw;
      wr[st] = left ? pw : cw;
      ca[st] = cb[st] = 0;
      if (wl[st] && wr[st]) {
        if (!haar)
          raht_coeffs(wl[st], wr[st], lut, &ca[st], &cb[st]);
        cw = wl[st] + wr[st];
      } else {
        cw = left ? wl[st] + wr[st] : 0;
      }
    }

    // ---- coefficient slot of this position (scanBlock :776-791) ---------------
    const uint32_t present = group8_bits(on && cw != 0) | (on ? 1u : 0u);
    const int spos = (0x74516230u >> (4 * t)) & 7;  // scan order 0,4,2,1,6,5,3,7
    const uint32_t pscan = ((present >> 0) & 1) | (((present >> 4) & 1) << 1)
      | (((present >> 2) & 1) << 2) | (((present >> 1) & 1) << 3)
      | (((present >> 6) & 1) << 4) | (((present >> 5) & 1) << 5)
      | (((present >> 3) & 1) << 6) | (((present >> 7) & 1) << 7);
    const int rank = popc32(pscan & ((1u << spos) - 1));
    const bool is_present = on && ((present >> t) & 1);
    const bool coded = is_present && (t != 0 || !inherit_dc);
    const int cidx = e.coeff_base
      + (inherit_dc ? (c0 - sc0) - pj + rank - 1 : rank);
    int32_t* __restrict__ cplane = ctx.coeffs + (size_t)pt0 * C + cidx;
    // the analyze pass's record: one entry per coefficient position, in
    // scan order, at the rows of the block's children
    const int64_t trow = (int64_t)pt0 + (c0 - sc0) + rank;

    // ---- intraDcPred (tmc3/RAHT.cpp:421-589), parent-level neighbours -----------
    int64_t pred[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = 0;
    if (kSearch) {
      const bool run = do_search && enable_pred;
      int wsum = 0;
      if (__any(run)) {
        int64_t lim_lo = 0, lim_hi = 0;
        const int64_t* __restrict__ prec = par2(ctx.rec, par_par);
        const int64_t rbase = (int64_t)pt0 - sp0;
        // every lane fetches the values of the neighbours it searched (and
        // of the parent itself) in ONE round trip
        int64_t nb_v[3][C], own_v[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          own_v[k] = run ? prec[(rbase + j) * C + k] : 0;
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
#pragma unroll
          for (int k = 0; k < C; k++)
            nb_v[slot][k] = (run && pn[slot] >= 0) ? prec[(rbase + pn[slot]) * C + k] : 0;
#pragma unroll
        for (int i = 0; i < 19; i++) {
          int q;
          int64_t v[C];
          if (i == 0) {
            q = j;
#pragma unroll
            for (int k = 0; k < C; k++)
              v[k] = own_v[k];
          } else {
            const int owner = gbase | ((i - 1) & 7);
            q = __shfl(pn[(i - 1) >> 3], owner);
#pragma unroll
            for (int k = 0; k < C; k++)
              v[k] = shfl_i64(nb_v[(i - 1) >> 3][k], owner);
          }
          if (!run || q < 0)
            continue;
          if (i) {
            if (10 * v[0] <= lim_lo || 10 * v[0] >= lim_hi)
              continue;
          } else {
            lim_lo = 2 * v[0];
            lim_hi = 25 * v[0];
          }
          if (has && ((neigh_mask(i) >> t) & 1)) {
            const int64_t pw = prm->pred_weight_parent[i];
            wsum += (int)pw;
            const int64_t mul = ext ? pw : (pw << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++)
              pred[k] += v[k] * mul;
          }
        }
      }
      if (run && has) {
        const int64_t div = pred_divisor(wsum);
#pragma unroll
        for (int k = 0; k < C; k++) {
          pred[k] = fp_mul_c(pred[k], div);
          if (haar)
            pred[k] = (pred[k] >> kFpFrac) << kFpFrac;
        }
      }
    }

    // ---- normalise (tmc3/RAHT.cpp:1445-1499) ---------------------------------------
    if (!haar && w > 1) {
      if (kEnc) {
#pragma unroll
        for (int k = 0; k < C; k++)
          src[k] = scale_rsqrt(src[k], w, lut);
      }
      if (kSearch && enable_pred) {
        const int64_t sq = sqrt_weight(w, lut);
#pragma unroll
        for (int k = 0; k < C; k++)
          pred[k] = fp_mul_c(pred[k], sq);
      }
    }

    // ---- forward butterflies (tmc3/RAHT.cpp:671-701) ------------------------------
    if (kSearch) {
#pragma unroll
      for (int st = 0; st < 3; st++) {
        const int bit = 1 << st;
        const bool left = !(t & bit);
        const bool both = wl[st] && wr[st];
        const bool swap = !wl[st] && wr[st];
#pragma unroll
        for (int k = 0; k < C; k++) {
          if (kEnc) {
            const int64_t own = src[k], oth = shfl_xor_i64(own, bit);
            if (both) {
              if (haar) {
                const int64_t hf = left ? oth - own : own - oth;
                src[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
              } else {
                src[k] = left ? fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st])
                              : fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st]);
              }
            } else if (swap) {
              src[k] = oth;
            }
          }
          {
            const int64_t own = pred[k], oth = shfl_xor_i64(own, bit);
            if (enable_pred) {
              if (both) {
                if (haar) {
                  const int64_t hf = left ? oth - own : own - oth;
                  pred[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
                } else {
                  pred[k] = left ? fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st])
                                 : fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st]);
                }
              } else if (swap) {
                pred[k] = oth;
              }
            }
          }
        }
      }
    } else if (is_present) {
      // kSynthRec: the transformed prediction as the analyze pass left it
#pragma unroll
      for (int k = 0; k < C; k++)
        pred[k] = ctx.ptrans[trow * C + k];
    }
    // ---- the reference frame's block (tmc3/RAHT.cpp:1322-1347, 1533-1545): where it exists it
    //      predicts every coefficient of the block in place of the intra prediction; the intra
    //      prediction stays the second candidate's (`ipred`, ctx.inter.dual) -------------------------
    int64_t ipred[C];
    bool enable_intra = false;
    if (INTER && kSearch) {
      enable_intra = enable_pred;
#pragma unroll
      for (int k = 0; k < C; k++)
        ipred[k] = pred[k];
      if (ctx.inter.blocks) {
        bool inter_node;
        int64_t pin[C];
        if (ctx.inter.hkey)
          inter_block_haar<C>(ctx.inter, on ? sm.key[j - wlo] : 0, t, on, &inter_node, pin);
        else
          inter_block<C>(ctx.inter, on ? sm.key[j - wlo] : 0, t, on, lut, &inter_node, pin);
        if (inter_node) {
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] = pin[k];
          enable_pred = on;
        }
      }
    }
    const bool dual = INTER && MODE == kAnalyze && ctx.inter.dual;
    if (MODE == kAnalyze && is_present) {
#pragma unroll
      for (int k = 0; k < C; k++)
        ctx.ptrans[trow * C + k] = enable_pred ? pred[k] : 0;
      if (dual) {
#pragma unroll
        for (int k = 0; k < C; k++)
          ctx.inter.iptrans[trow * C + k] = enable_intra ? ipred[k] : 0;
      }
    }

    if (coded) {
      int ac0 = 0, ac1 = 0;
      if (e.ac_layer < prm->num_ac_qp_layers && t) {
        ac0 = prm->ac_qp_offset[e.ac_layer][t - 1][0];
        ac1 = prm->ac_qp_offset[e.ac_layer][t - 1][1];
      }
      Quantizer qa[2];
      qpset_quantizers(prm, e.qp_layer, nq0 + ac0, nq1 + ac1, qa);

      if (kEnc) {
        if (dual) {
          // the intra candidate of the level: residual, RDOQ statistics, coefficients (:1560-1616)
          int64_t isrc[C];
#pragma unroll
          for (int k = 0; k < C; k++)
            isrc[k] = enable_intra ? src[k] - ipred[k] : src[k];
          Quantizer qr[2];
          qpset_quantizers(prm, e.qp_layer, nq0, nq1, qr);
          int64_t sum_coeff = 0, dist2 = 0;
          int rate_coeff = 0;
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int64_t co = fp_round(isrc[k]);
            dist2 += co * co;
            int64_t aq = quantize(qr[k ? 1 : 0], co * 256);
            aq = aq < 0 ? -aq : aq;
            sum_coeff += aq;
            rate_coeff += rate_log_small(aq);
          }
          uint32_t d = kDescNever;
          if (sum_coeff < 3) {
            const int64_t l0 = qr[0].step;
            const int64_t lambda = l0 * l0 * (C == 1 ? 25 : 35);
            d = rdoq_threshold(dist2, lambda, rate_coeff, (uint32_t)n_s);
            if (sum_coeff == 0)
              d |= kDescZero;
          }
          ctx.inter.idesc[(size_t)pt0 + cidx] = d;
          int32_t* __restrict__ iplane = ctx.inter.icoeffs + (size_t)pt0 * C + cidx;
#pragma unroll
          for (int k = 0; k < C; k++)
            iplane[(size_t)k * n_s] = (int32_t)quantize(qa[k ? 1 : 0], fp_round(isrc[k]) * 256);
        }
        if (enable_pred) {
#pragma unroll
          for (int k = 0; k < C; k++)
            src[k] -= pred[k];
        }
        if (MODE == kAnalyze) {
          // RDOQ statistics (tmc3/RAHT.cpp:1584-1616)
          Quantizer qr[2];
          qpset_quantizers(prm, e.qp_layer, nq0, nq1, qr);
          int64_t sum_coeff = 0, dist2 = 0;
          int rate_coeff = 0;
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int64_t co = fp_round(src[k]);
            dist2 += co * co;
            int64_t aq = quantize(qr[k ? 1 : 0], co * 256);
            aq = aq < 0 ? -aq : aq;
            sum_coeff += aq;
            rate_coeff += rate_log_small(aq);
          }
          uint32_t d = kDescNever;
          if (sum_coeff < 3) {
            const int64_t l0 = qr[0].step;
            const int64_t lambda = l0 * l0 * (C == 1 ? 25 : 35);
            d = rdoq_threshold(dist2, lambda, rate_coeff, (uint32_t)n_s);
            if (sum_coeff == 0)
              d |= kDescZero;
          }
          ctx.desc[(size_t)pt0 + cidx] = d;
        }
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int64_t co = quantize(qa[k ? 1 : 0], fp_round(src[k]) * 256);
          cplane[(size_t)k * n_s] = (int32_t)co;
          if (MODE == kFused)
            pred[k] += fp_from_int(dequantize(qa[k ? 1 : 0], co));
        }
      } else {
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int64_t co = cplane[(size_t)k * n_s];
          pred[k] += fp_from_int(dequantize(qa[k ? 1 : 0], co));
        }
      }
    }

    if (MODE == kAnalyze) {
      // the children's neighbour count is final here already
      if (has)
        par2(ctx.nneigh, cur_par)[crow] = inherit_dc ? neigh_count : 19;
      continue;
    }

    // ---- DC inheritance (tmc3/RAHT.cpp:1727-1742) --------------------------------
    if (on && inherit_dc && t == 0) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int64_t val = dcv[k];
        if (ext)
          pred[k] = val;
        else
          pred[k] = val > 0 ? val << (kFpFrac - 2) : -((-val) << (kFpFrac - 2));
      }
    }

    // ---- inverse butterflies (tmc3/RAHT.cpp:707-737) ------------------------------
#pragma unroll
    for (int st = 2; st >= 0; st--) {
      const int bit = 1 << st;
      const bool left = !(t & bit);
      const bool both = wl[st] && wr[st];
      const bool swap = !wl[st] && wr[st];
#pragma unroll
      for (int k = 0; k < C; k++) {
        const int64_t own = pred[k], oth = shfl_xor_i64(own, bit);
        if (both) {
          if (haar) {
            const int64_t lf = left ? own : oth, hf = left ? oth : own;
            const int64_t lv = lf - ((hf >> (1 + kFpFrac)) << kFpFrac);
            pred[k] = left ? lv : hf + lv;
          } else {
            pred[k] = left ? fp_mul_c(own, ca[st]) - fp_mul_c(oth, cb[st])
                           : fp_mul_c(oth, cb[st]) + fp_mul_c(own, ca[st]);
          }
        } else if (swap) {
          pred[k] = oth;
        }
      }
    }

    // ---- store the children's reconstruction (:1754-1806) -----------------------
    if (has) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        int64_t v = pred[k];
        par2(ctx.rec_us, cur_par)[crow * C + k] = ext ? v : fp_round(v * 4);
        if (!haar && w > 1)
          v = scale_rsqrt(v, w, lut);
        par2(ctx.rec, cur_par)[crow * C + k] = ext ? v : fp_round(v);
      }
      if (MODE != kSynthRec)
        par2(ctx.nneigh, cur_par)[crow] = inherit_dc ? neigh_count : 19;
    }
  }
  prof.mark(3);  // 8-lane blocks
  pa = pb;
  }  // part of the window's parents
  }  // slice window
}

// ---- one level, all slices: a grid of tiles ------------------------------------------
// Workgroup b runs on XCD b % 8 (observed dispatch; locality only): every
// XCD gets one contiguous eighth of the tiles, so the key windows of
// neighbouring tiles meet in the same L2.
template<int C, int MODE, int T, bool INTER = false>
__global__ __launch_bounds__(kTileThreads, T > kTileT ? 3 : (MODE == kSynthRec ? 6 : GPCC_TILE_WAVES)) void
raht_tile_kernel(LevelCtx ctx)
{
  __shared__ TileSmem<MODE != kSynthRec, (MODE == kAnalyze || MODE == kFused) ? C : 0, T> sm;
  if (tree_failed(ctx.tv))
    return;
  const int li = ctx.li;
  const int np = ctx.tv.soff[li + 1][ctx.tv.num_slices];
  const int ntiles = (np + T - 1) / T;
  const int per = ((int)gridDim.x + 7) >> 3;
  const int slot = ((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
  const int stride = per * 8;
  if (slot >= ntiles)
    return;
  load_lut(&sm.lut, ctx.lut);
  // (contiguous chunk per workgroup when the grid is smaller than the level)
  const int chunk = (ntiles + stride - 1) / stride;
  for (int tile = slot * chunk; tile < (slot + 1) * chunk && tile < ntiles; tile++) {
    const int j0 = tile * T;
    const int j1 = j0 + T < np ? j0 + T : np;
    tile_process<C, MODE, INTER>(ctx, li, j0, j1, sm, true);
  }
}

// ---- the coarse levels of a slice in one workgroup ---------------------------------
enum CoarseKind { kCoarseLossy = 0, kCoarseDecode = 1, kCoarseHaar = 2 };

template<int C, int KIND>
__global__ __launch_bounds__(1024) void
raht_coarse_kernel(LevelCtx ctx)
{
  __shared__ TileSmem<true, KIND == kCoarseDecode ? 0 : C> sm;
  const TreeView& tv = ctx.tv;
  if (tree_failed(tv))
    return;
  load_lut(&sm.lut, ctx.lut);
  for (int s = blockIdx.x; s < tv.num_slices; s += gridDim.x) {
    const SliceSched* __restrict__ sc = &ctx.sched[s];
    const int top = sc->top_level;
    int l = -1;  // RDOQ: index of the last reset (raht_rdoq.hpp), none yet
    for (int li = top - 1; li >= 0; li--) {
      const LevelSched e = sc->lvl[li];
      if (!e.coarse)
        break;  // coarse levels are the top ones
      if (!e.processed)
        continue;
      const int p0 = tv.soff[li + 1][s], p1 = tv.soff[li + 1][s + 1];
      if (KIND == kCoarseLossy) {
        for (int j0 = p0; j0 < p1; j0 += kTileT)
          tile_process<C, kAnalyze>(ctx, li, j0, j0 + kTileT < p1 ? j0 + kTileT : p1, sm, false);
        __syncthreads();
        // the zero-run state, walked in coding order by one wavefront: with
        // the incoming L known no hypothesis is needed (tmc3/RAHT.cpp:1618-1669).
        // The descriptors are brought into LDS by the whole workgroup first (the
        // key window is free between the passes): the walk is a chain of
        // dependent steps and must not wait for memory at each of them.
        {
          const int m = tv.soff[li][s + 1] - tv.soff[li][s];
          const int a = e.coeff_base;
          const int b = a + (e.is_root ? m : m - (p1 - p0));
          const int pt0 = tv.pt_off[s];
          const int n_s = tv.pt_off[s + 1] - pt0;
          const uint32_t* __restrict__ desc = ctx.desc + pt0;
          int32_t* __restrict__ co = ctx.coeffs + (size_t)pt0 * C;
          uint32_t* dl = reinterpret_cast<uint32_t*>(sm.key);
          constexpr int kChunk = (int)(sizeof(sm.key) / sizeof(uint32_t)) / kWave * kWave;
          for (int c0 = a; c0 < b; c0 += kChunk) {
            const int c1 = c0 + kChunk < b ? c0 + kChunk : b;
            __syncthreads();
            for (int i = c0 + (int)threadIdx.x; i < c1; i += blockDim.x)
              dl[i - c0] = desc[i];
            __syncthreads();
            if (threadIdx.x < kWave) {
              const int lane = lane_id();
              for (int i0 = c0; i0 < c1; i0 += kWave) {
                const int i = i0 + lane;
                const bool valid = i < c1;
                const uint32_t d = valid ? dl[i - c0] : kDescZero;
                int tz;
                l = rdoq_chunk(d, i, valid, l, i0, &tz);
                const uint32_t thr = d & kDescNever;
                if (valid && thr != kDescNever && (uint32_t)tz >= thr) {
#pragma unroll
                  for (int k = 0; k < C; k++)
                    co[(size_t)k * n_s + i] = 0;
                }
              }
            }
          }
        }
        __syncthreads();
        for (int j0 = p0; j0 < p1; j0 += kTileT)
          tile_process<C, kSynthRec>(ctx, li, j0, j0 + kTileT < p1 ? j0 + kTileT : p1, sm, false);
      } else {
        for (int j0 = p0; j0 < p1; j0 += kTileT)
          tile_process<C, KIND == kCoarseDecode ? kSynth : kFused>(
            ctx, li, j0, j0 + kTileT < p1 ? j0 + kTileT : p1, sm, false);
      }
      __syncthreads();  // the level is in memory before the next one reads it
    }
    if (KIND == kCoarseLossy && threadIdx.x == 0)
      ctx.slice_l[s] = l;  // carried into the per-level RDOQ passes
  }
}

}  // namespace gpcc
