// raht_tree.hpp -- builds every octree level of the RAHT tree in three
// launches, independent of the tree depth.
//
// The reference ascends one BINARY level at a time, compacting the node
// list 3*depth times (reduceUnique/reduceLevel, tmc3/RAHT.cpp:108-205).
// Here the whole structure follows from one observation on the
// Morton-sorted point list: point i starts a new node at octree level li
// iff bitlength(pos[i] ^ pos[i-1]) > 3*li.  So a point is a node head at
// levels 0 .. h(i)-1 with h(i) = ceil(bitlength / 3), and the index of its
// node at level li is the number of heads before it -- `nlev` simultaneous
// prefix counts.  Weights are differences of first-point indices and
// attribute sums (non-Haar) are differences of one modular prefix sum, so
// no per-level reduction pass exists at all.
//
//   tree_count : per tile (one wave, 1024 points) head counts per level
//                and attribute tile sums          -- reads pos, attrs
//   tree_scan  : one workgroup scans the tile table -- tiny
//   tree_emit  : per tile, writes key/fp/fc of every level and the
//                attribute prefix array P        -- reads pos, attrs
#pragma once

#include "raht_common.hpp"

namespace gpcc {

// levels at which point i heads a node
__device__ __forceinline__ int
head_levels(const TreeView& tv, int i)
{
  const int s = find_slice(tv.pt_off, tv.num_slices, i);
  if (tv.pt_off[s] == i)
    return tv.nlev;
  const uint64_t x = (uint64_t)(tv.pos[i] ^ tv.pos[i - 1]);
  if (!x)
    return 0;
  const int h = (bitlen64(x) + 2) / 3;
  return h < tv.nlev ? h : tv.nlev;
}

// tile table layout: cnt[tile][nlev] then attr[tile][C]
template<int C>
__global__ __launch_bounds__(256) void
tree_count_kernel(
  TreeView tv, const int32_t* __restrict__ attrs, uint32_t* tile_cnt,
  int32_t* tile_attr)
{
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  for (int tile = wave; tile < tv.num_tiles; tile += nwaves) {
    const int base = tile * kTilePoints;
    uint32_t acc = 0;  // lane li counts heads of level li
    int32_t asum[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      asum[k] = 0;
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = base + r * kWave + lane;
      int h = 0;
      if (i < tv.n_total) {
        h = head_levels(tv, i);
        if (attrs) {
#pragma unroll
          for (int k = 0; k < C; k++)
            asum[k] += attrs[(size_t)i * C + k];
        }
      }
      for (int li = 0; li < tv.nlev; li++) {
        const unsigned long long b = __ballot(h > li);
        if (!b)
          break;
        if (lane == li)
          acc += __popcll(b);
      }
    }
    if (lane < tv.nlev)
      tile_cnt[(size_t)tile * tv.nlev + lane] = acc;
    if (attrs) {
#pragma unroll
      for (int k = 0; k < C; k++) {
        int32_t v = asum[k];
#pragma unroll
        for (int d = 1; d < kWave; d <<= 1)
          v += __shfl_xor(v, d);
        if (lane == 0)
          tile_attr[(size_t)tile * C + k] = v;
      }
    }
  }
}

// Exclusive scan of every column of the tile table; one wave per column.
// Also writes the per-level sentinels and node totals.
template<int C>
__global__ __launch_bounds__(1024) void
tree_scan_kernel(
  TreeView tv, uint32_t* tile_cnt, int32_t* tile_attr, int32_t* attr_prefix,
  int has_attrs)
{
  const int lane = lane_id();
  const int wave = threadIdx.x / kWave;
  const int nwaves = blockDim.x / kWave;
  const int ncol = tv.nlev + (has_attrs ? C : 0);
  for (int col = wave; col < ncol; col += nwaves) {
    uint32_t running = 0;
    if (col < tv.nlev) {
      for (int t0 = 0; t0 < tv.num_tiles; t0 += kWave) {
        const int t = t0 + lane;
        uint32_t v = t < tv.num_tiles ? tile_cnt[(size_t)t * tv.nlev + col] : 0;
        uint32_t inc = wave_incl_scan_u32(v);
        if (t < tv.num_tiles)
          tile_cnt[(size_t)t * tv.nlev + col] = running + inc - v;
        running += __shfl(inc, kWave - 1);
      }
      if (lane == 0) {
        int m = (int)running;
        // more nodes than the level's arrays hold: the Morton-bits hint was
        // smaller than the codes' width (the top levels are sized from it).
        // tree_emit and everything behind it leave on the error word.
        if (m > tv.cap[col]) {
          atomicExch(tv.error, 2);
          m = tv.cap[col];
        }
        tv.soff[col][tv.num_slices] = m;
        tv.fp[col][m] = tv.n_total;
      }
    } else {
      const int k = col - tv.nlev;
      for (int t0 = 0; t0 < tv.num_tiles; t0 += kWave) {
        const int t = t0 + lane;
        uint32_t v = t < tv.num_tiles ? (uint32_t)tile_attr[(size_t)t * C + k] : 0;
        uint32_t inc = wave_incl_scan_u32(v);
        if (t < tv.num_tiles)
          tile_attr[(size_t)t * C + k] = (int32_t)(running + inc - v);
        running += __shfl(inc, kWave - 1);
      }
      if (lane == 0)
        attr_prefix[(size_t)tv.n_total * C + k] = (int32_t)running;
    }
  }
  __syncthreads();
  // fc sentinel of level li = node count of level li-1
  if (threadIdx.x >= 1 && (int)threadIdx.x < tv.nlev) {
    const int li = threadIdx.x;
    tv.fc[li][tv.soff[li][tv.num_slices]] = tv.soff[li - 1][tv.num_slices];
  }
}

template<int C>
__global__ __launch_bounds__(256) void
tree_emit_kernel(
  TreeView tv, const int32_t* __restrict__ attrs,
  const uint32_t* __restrict__ tile_cnt, const int32_t* __restrict__ tile_attr,
  int32_t* attr_prefix)
{
  if (tree_failed(tv))
    return;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  const unsigned long long lt = (1ull << lane) - 1;
  for (int tile = wave; tile < tv.num_tiles; tile += nwaves) {
    const int base = tile * kTilePoints;
    uint32_t acc = lane < tv.nlev ? tile_cnt[(size_t)tile * tv.nlev + lane] : 0;
    int32_t run[C];
    if (attrs) {
#pragma unroll
      for (int k = 0; k < C; k++)
        run[k] = tile_attr[(size_t)tile * C + k];
    }
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = base + r * kWave + lane;
      const bool in = i < tv.n_total;
      int h = 0;
      int64_t p = 0;
      bool start = false;
      int s = 0;
      if (in) {
        h = head_levels(tv, i);
        p = tv.pos[i];
        if (h == tv.nlev) {
          s = find_slice(tv.pt_off, tv.num_slices, i);
          start = tv.pt_off[s] == i;
        }
      }
      if (attrs) {
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int32_t a = in ? attrs[(size_t)i * C + k] : 0;
          const int32_t inc = (int32_t)wave_incl_scan_u32((uint32_t)a);
          if (in)
            attr_prefix[(size_t)i * C + k] = run[k] + inc - a;
          run[k] += __shfl(inc, kWave - 1);
        }
      }
      int prev_idx = 0;
      for (int li = 0; li < tv.nlev; li++) {
        const unsigned long long b = __ballot(h > li);
        if (!b)
          break;
        const int idx = (int)__shfl(acc, li) + __popcll(b & lt);
        if (h > li) {
          tv.fp[li][idx] = i;
          tv.key[li][idx] = p >> (3 * li);
          if (li)
            tv.fc[li][idx] = prev_idx;
          if (start)
            tv.soff[li][s] = idx;
        }
        prev_idx = idx;
        if (lane == li)
          acc += __popcll(b);
      }
    }
  }
}

// One thread per slice: which levels run, their layers and coefficient
// bases (tmc3/RAHT.cpp:1165-1217,1264-1265), and which of them the coarse
// kernel takes: the top levels of the slice down to the last one whose
// parents number at most `coarse_max_parents`.  One workgroup; the summary the
// host sizes its launches from goes to `stats` (pinned host memory) when given.
__global__ __launch_bounds__(256) void
schedule_kernel(
  TreeView tv, SliceSched* sched, int num_qp_layers, int coarse_max_parents,
  TreeStats* stats)
{
  __shared__ int s_fine, s_top;
  if (tree_failed(tv)) {
    if (stats && threadIdx.x == 0) {
      stats->fine_levels = 0;
      stats->max_top = 0;
    }
    return;
  }
  if (threadIdx.x == 0)
    s_fine = s_top = 0;
  __syncthreads();
  // (no per-thread arrays: the plan is written straight into sched[s] and the level sizes are re-read from the
  // slice offsets, so the kernel needs no scratch memory -- it used 284 B per lane and was the only dispatch with
  // scratch in a fixed-point sub-node-off call)
  for (int s = threadIdx.x; s < tv.num_slices; s += blockDim.x) {
    auto nodes = [&](int li) { return tv.soff[li][s + 1] - tv.soff[li][s]; };
    SliceSched* sc = &sched[s];
    int top = tv.nlev - 1;
    // the level arrays were sized for ONE node per slice at the top level: a
    // Morton-bits hint smaller than the codes' real width breaks that (and
    // tree_emit has then written past the top levels' arrays, inside the
    // workspace) -- report it instead of returning a wrong result
    if (nodes(top) != 1)
      atomicExch(tv.error, 2);
    while (top > 0 && nodes(top - 1) == 1)
      top--;
    sc->num_unique = nodes(0);
    sc->top_level = top;
    int coarse_from = top;  // levels li >= coarse_from belong to the coarse kernel
    for (int li = top - 1; li >= 0 && nodes(li + 1) <= coarse_max_parents; li--)
      coarse_from = li;
    int qp_layer = 0, ac_layer = -1, parity = 1, coeff = 0;
    LevelSched none;
    none.processed = 0;
    none.is_root = 0;
    none.qp_layer = 0;
    none.ac_layer = -1;
    none.parity = 0;
    none.coarse = 0;
    none.pad[0] = none.pad[1] = 0;
    none.coeff_base = 0;
    for (int li = kMaxLevels - 1; li >= (top > 0 ? top : 0); li--)
      sc->lvl[li] = none;
    int m_up = nodes(top);
    for (int li = top - 1; li >= 0; li--) {
      const bool root = li == top - 1;
      const int m_li = nodes(li);
      LevelSched e = none;
      e.coarse = li >= coarse_from;
      if (root || m_li != m_up) {
        qp_layer = qp_layer + 1 < num_qp_layers ? qp_layer + 1 : num_qp_layers - 1;
        ac_layer++;
        parity ^= 1;
        e.processed = 1;
        e.is_root = root;
        e.qp_layer = (uint8_t)qp_layer;
        e.ac_layer = (int8_t)(ac_layer > 127 ? 127 : ac_layer);
        e.parity = (uint8_t)parity;
        e.coeff_base = coeff;
        coeff += root ? m_li : m_li - m_up;
      }
      sc->lvl[li] = e;
      m_up = m_li;
    }
    sc->final_qp_layer = qp_layer;
    sc->final_parity = parity;
    atomicMax(&s_fine, coarse_from);
    atomicMax(&s_top, top);
  }
  __syncthreads();
  if (stats) {
    if (threadIdx.x == 0) {
      stats->fine_levels = s_fine;
      stats->max_top = s_top;
    }
    if ((int)threadIdx.x < kMaxLevels)
      stats->nodes[threadIdx.x] =
        (int)threadIdx.x < tv.nlev ? tv.soff[threadIdx.x][tv.num_slices] : 0;
  }
}

}  // namespace gpcc
