// raht_links.hpp -- O(1) NEIGHBOUR LINKS of the octree levels (round 5).
//
// findNeighbours (tmc3/RAHT.cpp:299-368) looks the 18 face / edge neighbours of a
// parent up with one lower_bound each over the level's positions
// (findNeighbour, tmc3/RAHT.cpp:272-293: limited to raht_prediction_search_range
// entries either side).  On the device that was, per lane, three (sub-node
// kernels) or six (compact level pass) 12-step bisections of 8-byte keys in global
// memory -- a dozen DEPENDENT round trips in front of everything else a block
// does, 40 % of the compact pass's time and most of the 43 % of a round the
// sub-node kernels spend in their prologue.
//
// The neighbours are a pure function of the geometry, and the octree gives them
// without a search: the neighbour of node c (child octant o of parent p) in
// direction d is a child of p itself or of p's neighbour in the direction d'
// that keeps the axes on which c + d leaves p -- and moving by one along an
// axis flips that axis' bit of the octant.  So, top-down, level by level:
//
//   link(c, d) = child(o ^ axes(d)) of (d' == 0 ? p : link(p, d'))
//
// where child(q, o') = fc[q] + popc(occ[q] & ((1 << o') - 1)) if occ[q] has bit
// o'.  Only nodes with more than one point get links (a node with one point has
// no block below it): 0.8 M of the 5.2 M node-levels of a 1 M-point lidar
// frame.  A level's records are produced from the level above by ONE launch at
// full occupancy (three dependent round trips per thread), kept for two levels
// (ping-pong by level parity: 80 bytes per point) and consumed by the level
// kernels with one load per neighbour; the search window of findNeighbour is
// the test |link - j| <= range on the consumer's side (an index distance: keys
// are distinct and ascending, so the distance in keys is never smaller).
//
// Records are 20 ints, 16-byte aligned: links of the reference's neighbour ids
// 1..18 (tmc3/RAHT.cpp:314-326, the order of neigh_offset() in raht_levels.hpp),
// then the node itself, then a spare.
#pragma once

#include <cstdlib>

#include <algorithm>

#include "raht_common.hpp"

namespace gpcc {

constexpr int kLinkRec = 20;

struct LinkView {
  uint8_t* occ[kMaxLevels];  // [cap + 1] child octants of every node (levels >= 1)
  int32_t* lrec[2];          // [n + 1] by level parity: node -> its record (nodes with > 1 point)
  int32_t* rec[2];           // [cap_rec][kLinkRec] by level parity
  int32_t* cnt;              // [kMaxLevels] records per level
  int32_t cap_rec;
};

struct alignas(16) LinkQuad {
  int32_t v[4];
};

// offsets of the reference's neighbours 1..18 from (x - 1, y - 1, z - 1), Morton-interleaved two bits
// per axis (tmc3/RAHT.cpp:314-326; the same numbers as neigh_offset() of raht_levels.hpp)
constexpr uint8_t kLinkOff[19] = {0, 35, 21, 14, 49, 42, 28, 1, 2, 3, 4, 5, 6, 10, 12, 17, 20, 33, 34};

constexpr int
link_axis(int id, int axis)  // -1 / 0 / +1 of neighbour id along x (0), y (1), z (2); id 0 = the node itself
{
  if (id == 0)
    return 0;
  const int o = kLinkOff[id];
  const int hi = (o >> (5 - axis)) & 1, lo = (o >> (2 - axis)) & 1;
  return hi * 2 + lo - 1;
}

constexpr int
link_id(int dx, int dy, int dz)
{
  for (int i = 1; i < 19; i++)
    if (link_axis(i, 0) == dx && link_axis(i, 1) == dy && link_axis(i, 2) == dz)
      return i;
  return 0;
}

// the parent's 19 candidate nodes (0 = the parent, i = its neighbour i or -1), their first children and
// occupancies; indexed with compile-time constants only
struct LinkTab {
  int32_t q[19];
  int32_t fc[19];
  uint32_t occ[19];
};

// neighbour I of the child at octant o (bits x y z) of the parent T describes
template<int I>
__device__ __forceinline__ int32_t
link_of_child(const LinkTab& T, int o)
{
  constexpr int dx = link_axis(I, 0), dy = link_axis(I, 1), dz = link_axis(I, 2);
  // the axes d moves along: A, and B for an edge
  constexpr int a_axis = dx ? 0 : (dy ? 1 : 2);
  constexpr int b_axis = dx ? (dy ? 1 : (dz ? 2 : -1)) : (dy && dz ? 2 : -1);
  constexpr int da = a_axis == 0 ? dx : (a_axis == 1 ? dy : dz);
  constexpr int db = b_axis < 0 ? 0 : (b_axis == 1 ? dy : dz);
  constexpr int id_a = link_id(a_axis == 0 ? da : 0, a_axis == 1 ? da : 0, a_axis == 2 ? da : 0);
  constexpr int id_b = b_axis < 0 ? 0 : link_id(0, b_axis == 1 ? db : 0, b_axis == 2 ? db : 0);
  static_assert(id_a != 0 && (b_axis < 0 || id_b != 0), "direction table");
  // c + d leaves the parent along an axis iff the child sits on that side
  const bool cross_a = ((o >> (2 - a_axis)) & 1) == (da > 0);
  const bool cross_b = b_axis >= 0 && ((o >> (2 - (b_axis < 0 ? 0 : b_axis))) & 1) == (db > 0);
  int32_t q, f;
  uint32_t oc;
  if (b_axis < 0) {
    q = cross_a ? T.q[I] : T.q[0];
    f = cross_a ? T.fc[I] : T.fc[0];
    oc = cross_a ? T.occ[I] : T.occ[0];
  } else {
    q = cross_a ? (cross_b ? T.q[I] : T.q[id_a]) : (cross_b ? T.q[id_b] : T.q[0]);
    f = cross_a ? (cross_b ? T.fc[I] : T.fc[id_a]) : (cross_b ? T.fc[id_b] : T.fc[0]);
    oc = cross_a ? (cross_b ? T.occ[I] : T.occ[id_a]) : (cross_b ? T.occ[id_b] : T.occ[0]);
  }
  constexpr int flip = (dx ? 4 : 0) | (dy ? 2 : 0) | (dz ? 1 : 0);
  const int o2 = o ^ flip;
  const bool there = q >= 0 && ((oc >> o2) & 1u);
  return there ? f + __popc(oc & ((1u << o2) - 1u)) : -1;
}

// ---- occupancy of every node of every level >= 1 (one launch) ---------------------------------
__global__ __launch_bounds__(256) void
link_occ_kernel(TreeView tv, LinkView lv)
{
  if (tree_failed(tv))
    return;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int nth = gridDim.x * blockDim.x;
  for (int l = 1; l < tv.nlev; l++) {
    const int m = tv.soff[l][tv.num_slices];
    const int32_t* __restrict__ fc = tv.fc[l];
    const int64_t* __restrict__ ck = tv.key[l - 1];
    for (int j = tid; j < m; j += nth) {
      const int c0 = fc[j], c1 = fc[j + 1];
      uint32_t occ = 0;
      for (int c = c0; c < c1; c++)
        occ |= 1u << (int)(ck[c] & 7);
      lv.occ[l][j] = (uint8_t)occ;
    }
  }
}

// ---- records of level L from those of level L + 1 ----------------------------------------------
// Threads tid, tid + nth, ... take the records of level L + 1; whole wavefronts run the loop
// together (the slots of a wavefront's new records are one atomic add).
__device__ __forceinline__ void
link_level(const TreeView& tv, const LinkView& lv, int L, int tid, int nth)
{
  const int P = L + 1;
  const int np = lv.cnt[P];
  const int32_t* __restrict__ recP = lv.rec[P & 1];
  int32_t* __restrict__ recL = lv.rec[L & 1];
  int32_t* __restrict__ lrecL = lv.lrec[L & 1];
  const int32_t* __restrict__ fcP = tv.fc[P];
  const uint8_t* __restrict__ occP = lv.occ[P];
  const int32_t* __restrict__ fpL = tv.fp[L];
  const int lane = lane_id();
  const int first = tid - lane;  // the wavefront's first thread
  for (int r0 = first; r0 < np; r0 += nth) {
    const int r = r0 + lane;
    const bool live = r < np;
    LinkTab T;
    {
      LinkQuad v[5];
#pragma unroll
      for (int u = 0; u < 5; u++) {
        if (live) {
          v[u] = *reinterpret_cast<const LinkQuad*>(recP + (size_t)r * kLinkRec + 4 * u);
        } else {
          v[u].v[0] = v[u].v[1] = v[u].v[2] = v[u].v[3] = -1;
        }
      }
      T.q[0] = live ? v[4].v[2] : -1;
#pragma unroll
      for (int i = 1; i < 19; i++)
        T.q[i] = v[(i - 1) >> 2].v[(i - 1) & 3];
    }
    // first children and occupancies of the candidates: one batch of loads
#pragma unroll
    for (int i = 0; i < 19; i++) {
      const int qq = T.q[i] < 0 ? 0 : T.q[i];
      T.fc[i] = fcP[qq];
      T.occ[i] = T.q[i] < 0 ? 0u : (uint32_t)occP[qq];
    }
    // children with more than one point get a record
    const uint32_t pocc = T.occ[0];
    const int c0 = T.fc[0];
    const int nchild = __popc(pocc);
    uint32_t multi = 0;  // bit u: the u-th child has more than one point
    {
      int prev = live && nchild ? fpL[c0] : 0;
      for (int u = 0; u < nchild; u++) {
        const int nxt = fpL[c0 + u + 1];
        multi |= (nxt - prev > 1) ? 1u << u : 0u;
        prev = nxt;
      }
    }
    const uint32_t k = (uint32_t)__popc(multi);
    const uint32_t incl = wave_incl_scan_u32(k);
    const uint32_t total = (uint32_t)__shfl((int)incl, kWave - 1);
    int base = 0;
    if (lane == 0 && total)
      base = atomicAdd(&lv.cnt[L], (int)total);
    base = __shfl(base, 0);
    if (base + (int)total > lv.cap_rec) {
      // cannot happen: the nodes with more than one point of a level are disjoint (<= n / 2 of them)
      if (lane == 0)
        atomicExch(tv.error, 4);
      return;
    }
    int slot = base + (int)(incl - k);
    uint32_t occ_left = pocc;
    for (int u = 0; u < nchild; u++) {
      const int o = __ffs((int)occ_left) - 1;
      occ_left &= occ_left - 1;
      if (!((multi >> u) & 1))
        continue;
      const int c = c0 + u;
      LinkQuad w[5];
      w[0].v[0] = link_of_child<1>(T, o);
      w[0].v[1] = link_of_child<2>(T, o);
      w[0].v[2] = link_of_child<3>(T, o);
      w[0].v[3] = link_of_child<4>(T, o);
      w[1].v[0] = link_of_child<5>(T, o);
      w[1].v[1] = link_of_child<6>(T, o);
      w[1].v[2] = link_of_child<7>(T, o);
      w[1].v[3] = link_of_child<8>(T, o);
      w[2].v[0] = link_of_child<9>(T, o);
      w[2].v[1] = link_of_child<10>(T, o);
      w[2].v[2] = link_of_child<11>(T, o);
      w[2].v[3] = link_of_child<12>(T, o);
      w[3].v[0] = link_of_child<13>(T, o);
      w[3].v[1] = link_of_child<14>(T, o);
      w[3].v[2] = link_of_child<15>(T, o);
      w[3].v[3] = link_of_child<16>(T, o);
      w[4].v[0] = link_of_child<17>(T, o);
      w[4].v[1] = link_of_child<18>(T, o);
      w[4].v[2] = c;
      w[4].v[3] = -1;
#pragma unroll
      for (int x = 0; x < 5; x++)
        *reinterpret_cast<LinkQuad*>(recL + (size_t)slot * kLinkRec + 4 * x) = w[x];
      lrecL[c] = slot;
      slot++;
    }
  }
}

// the top level: one node per slice, no neighbours (a slice's tree is searched inside the slice only)
__device__ __forceinline__ void
link_seed(const TreeView& tv, const LinkView& lv, int tid, int nth)
{
  const int top = tv.nlev - 1;
  for (int s = tid; s < tv.num_slices; s += nth) {
    const int j = tv.soff[top][s];
    int32_t* r = lv.rec[top & 1] + (size_t)s * kLinkRec;
    for (int i = 0; i < 18; i++)
      r[i] = -1;
    r[18] = j;
    r[19] = -1;
    lv.lrec[top & 1][j] = s;
  }
  if (tid == 0)
    lv.cnt[top] = tv.num_slices;
}

// The seed and the levels L_hi .. L_lo in ONE launch of one workgroup (the top levels hold a few nodes each
// and a level needs the one above it; cnt[] has been cleared).  A level's records are made visible to the
// whole workgroup before the next level reads them.
__global__ __launch_bounds__(1024) void
link_top_kernel(TreeView tv, LinkView lv, int L_lo)
{
  if (tree_failed(tv))
    return;
  link_seed(tv, lv, (int)threadIdx.x, (int)blockDim.x);
  __threadfence();
  __syncthreads();
  for (int L = tv.nlev - 2; L >= L_lo; L--) {
    link_level(tv, lv, L, (int)threadIdx.x, (int)blockDim.x);
    __threadfence();
    __syncthreads();
  }
}

// one level, as many workgroups as the level above can have records
__global__ __launch_bounds__(256) void
link_level_kernel(TreeView tv, LinkView lv, int L)
{
  if (tree_failed(tv))
    return;
  link_level(tv, lv, L, (int)(blockIdx.x * blockDim.x + threadIdx.x), (int)(gridDim.x * blockDim.x));
}

// ---- the consumer's side -----------------------------------------------------------------------
// neighbour `id` (1..18) of node j of level L whose record is rj, inside findNeighbour's window
__device__ __forceinline__ int32_t
link_lookup(const int32_t* __restrict__ rec, int rj, int id, int j, int64_t range)
{
  const int32_t q = rec[(size_t)rj * kLinkRec + (id - 1)];
  const int64_t d = (int64_t)q - j;
  return (q >= 0 && d <= range && -d <= range) ? q : -1;
}

// ---- host side: storage and launch order --------------------------------------------------------

// GPCC_LINKS=1 turns the links on (read at every call) in a build with the experiments compiled in (below).  OFF
// by default: measured on the MI355X
// (profiles/r05_links_ab.txt) the consumers gain less than the passes cost -- compact level pass of 10 x 1 M
// points 2.40 -> 2.02 ms, sub-node encoder 25.7 -> 24.7 ms, against 2.4 ms for the link passes (0.9 for a
// single frame): the bisections they replace run in L2-resident keys that neighbouring lanes share, and
// the level kernels are bound by instruction issue and by their dependency chains, not by those round trips.
// The consumers' side of both round-5 experiments (the link look-ups in the level kernels, the multi-round claims of
// raht_subnode.hpp) is compiled in only with -DGPCC_EXPERIMENTS=1 (the CPU emulator builds of tests/emu always; the
// gfx950 library on request: GPCC_EXTRA_FLAGS=-DGPCC_EXPERIMENTS=1 python -m ... build): left in unconditionally the
// dead branches cost the headline kernel 28 bytes of scratch, 350 instructions and 1.6 % of its time.
#ifndef GPCC_EXPERIMENTS
#define GPCC_EXPERIMENTS 0
#endif
inline bool
links_enabled()
{
#if GPCC_EXPERIMENTS
  const char* e = getenv("GPCC_LINKS");
  return e && e[0] == '1';
#else
  return false;
#endif
}

// storage of the neighbour links of a batch of n points in s slices: `take` as in cx_carve
template<class Take>
void
link_carve(Take&& take, LinkView& lv, const TreeView& tv, int n, int s, int nlev)
{
  for (int li = 0; li < kMaxLevels; li++)
    lv.occ[li] = nullptr;
  for (int li = 1; li < nlev; li++)
    lv.occ[li] = (uint8_t*)take((size_t)tv.cap[li] + 1);
  lv.cap_rec = n / 2 + s + 1;
  for (int i = 0; i < 2; i++) {
    lv.lrec[i] = (int32_t*)take(((size_t)n + 1) * 4);
    lv.rec[i] = (int32_t*)take((size_t)lv.cap_rec * kLinkRec * 4);
  }
  lv.cnt = (int32_t*)take(kMaxLevels * 4);
}

// The link passes of a call: the occupancy pass and the top levels at once, then level by level in step
// with the level kernels that consume them -- the records of a level live in the buffers of its parity,
// so level L may only be produced once the consumer of level L + 2 has been launched.
struct LinkSchedule {
  TreeView tv;
  LinkView lv;
  int next = -1;  // next level to produce

  // `nodes[l]`: nodes per level as the host knows them; `first_need`: level of the first consumer's parents
  template<class Prof>
  void begin(hipStream_t st, const int32_t* nodes, int first_need, Prof&& prof)
  {
    auto t = prof("link_top", -1);
    const int grid = std::min(std::max((nodes[1] + 255) / 256, 1), 2048);
    hipMemsetAsync(lv.cnt, 0, kMaxLevels * sizeof(int32_t), st);
    hipLaunchKernelGGL(link_occ_kernel, dim3(grid), dim3(256), 0, st, tv, lv);
    // the single workgroup takes the levels whose parents number at most 2048, down to the first consumer's
    int lk = tv.nlev - 1;
    while (lk - 1 >= 1 && lk - 1 >= first_need && nodes[lk] <= 2048)
      lk--;
    hipLaunchKernelGGL(link_top_kernel, dim3(1), dim3(1024), 0, st, tv, lv, lk);
    next = lk - 1;
    this->nodes_ = nodes;
  }
  // everything the consumer of level `need` reads is enqueued when this returns
  template<class Prof>
  void produce(hipStream_t st, int need, Prof&& prof)
  {
    while (next >= need && next >= 1) {
      auto t = prof("link_level", next);
      const int parents = std::min(nodes_[next + 1], lv.cap_rec);
      const int grid = std::min(std::max((parents + 255) / 256, 1), 1 << 16);
      hipLaunchKernelGGL(link_level_kernel, dim3(grid), dim3(256), 0, st, tv, lv, next);
      next--;
    }
  }
  const int32_t* nodes_ = nullptr;
};

}  // namespace gpcc
