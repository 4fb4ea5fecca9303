// cx_tree.hpp -- tree build of the COMPACT level pass (cx_level.hpp): the level
// arrays of raht_tree.hpp (key / fp / fc / soff of every octree level, from the
// Morton-sorted points in a fixed number of launches) plus what lets a level be
// processed by its BRANCHING blocks only:
//
//   h[i]        head levels of point i: it starts a node at levels 0 .. h-1
//               (h = ceil(bitlength(pos[i] ^ pos[i-1]) / 3), nlev at a slice start,
//               0 for a duplicate of the previous point);
//   hold[l][q]  for node q of level l: WHERE ITS VALUE LIVES.  With the RAHT
//               extension a parent with a single child hands its values to that
//               child unchanged (tmc3/RAHT.cpp:1382-1401), so a chain of such
//               nodes is one value.  A node [a, b) (first point a, b = next head
//               of its level) has the same points as its ancestors up to level
//               T = min(h(a), h(b)) - 1; at level T it is a child of a block with
//               >= 2 children ("real child") or the root.  Real children that do
//               not start their block correspond one to one to the points with
//               1 <= h < nlev (child at level h - 1): value slot = a.  A first
//               child is identified by the start b of its right sibling: slot
//               N + b.  So slot(node) = h(a) <= h(b) ? a : N + b, and the level
//               kernels never copy a value down a chain.  hold = slot | T << 27;
//   bp / bc / bq / rb  the blocks of every children level l in Morton order: parent
//               index in level l + 1, first child (node of level l), rank of the
//               block's first real child, and the block of every real-child rank --
//               a wavefront of the level pass takes ~56 consecutive ranks (whole
//               blocks), lane = child.  (The first-child array fc of raht_tree.hpp is
//               written as well since round 5: the neighbour links of raht_links.hpp
//               descend through it.)
//
// A point i with h(i) == l + 1 is a non-first child at level l; it is the SECOND
// child of its block iff the head of level l before it also heads level l + 1.
// The previous head of a level is found inside the 64-point row with a ballot,
// carried in registers from row to row of a tile, and located with one bisection
// per level over the points for the first row of a tile.
#pragma once

#include "raht_common.hpp"
#include "raht_tree.hpp"

namespace gpcc {

// per-level offsets into the all-level lists, written by cx_scan_fin_kernel
struct CxLevelTab {
  int32_t boff[kMaxLevels + 1];  // first block of children level l in bp; bq has one sentinel per level: bqoff = boff + l
  int32_t roff[kMaxLevels + 1];  // first rank of level l in rb
  int32_t nb[kMaxLevels];        // blocks of level l
  int32_t nr[kMaxLevels];        // real children of level l
  int32_t nodes[kMaxLevels];     // nodes of level l (the host sizes the link passes with it, raht_links.hpp)
};

struct CxLists {
  uint8_t* h;                  // [N + 1]
  uint32_t* hold[kMaxLevels];  // [cap + 1]
  int32_t* bp;                 // [N]         all levels, level l at tab->boff[l]
  int32_t* bq;                 // [N + nlev]  level l at tab->boff[l] + l, nb[l] + 1 entries
  int32_t* rb;                 // [2 N]       level l at tab->roff[l]
  int32_t* bc;                 // [N]         first child (node of level l) of every block, at tab->boff[l]
  CxLevelTab* tab;
  // tile tables of the count / scan / emit passes: [tile][ncol], columns
  //   [0, nlev) heads per level, [nlev, 2 nlev) non-first children per level,
  //   [2 nlev, 3 nlev) blocks per level, then C attribute sums
  uint32_t* tile_tab;
  uint32_t* col_total;         // [ncol]
};

__device__ __forceinline__ int cx_ncol(int nlev, int c) { return 3 * nlev + c; }

__device__ __forceinline__ uint32_t
cx_hold_value(int a, int ha, int b, int hb, int n_total)
{
  const int slot = ha <= hb ? a : n_total + b;
  const int t = (ha < hb ? ha : hb) - 1;
  return (uint32_t)slot | ((uint32_t)t << kCxSlotBits);
}

// the head of level l that precedes point `base` (base is not a slice start):
// first point of the level-l node that holds point base - 1
__device__ __forceinline__ int
cx_prev_head(const TreeView& tv, int l, int base)
{
  const int s = find_slice(tv.pt_off, tv.num_slices, base - 1);
  int lo = tv.pt_off[s], hi = base - 1;
  const int sh = 3 * l;
  const int64_t target = tv.pos[base - 1] >> sh;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if ((tv.pos[mid] >> sh) < target)
      lo = mid + 1;
    else
      hi = mid;
  }
  return lo;
}

// State a wavefront carries from row to row of its tile: lane l holds, for
// level l, the previous head (point, head levels) and the running counters.
struct CxRowState {
  int prev_a, prev_h;           // previous head of level `lane`
  uint32_t heads, nf, blk;      // running counts of level `lane`
};

// One 64-point row: calls visit(l, is_head, idx, prev_a, prev_h, is_nf, is_second,
// nf_before, blocks_incl, parent_idx) for every level some lane of the row heads.
// All lanes call it together (wave-uniform l).
template<class Visit>
__device__ __forceinline__ void
cx_row_levels(const TreeView& tv, int i, int h, CxRowState& st, Visit&& visit)
{
  const int lane = lane_id();
  const unsigned long long lt = (1ull << lane) - 1;
  const unsigned long long le = lt | (1ull << lane);
  unsigned long long mask = __ballot(h > 0);
  for (int l = 0; l < tv.nlev; l++) {
    if (!mask)
      break;
    const unsigned long long mask1 = l + 1 < tv.nlev ? __ballot(h > l + 1) : 0ull;
    // the head of level l before this lane: inside the row, or the carry
    const unsigned long long below = mask & lt;
    const int p = below ? 63 - __clzll((long long)below) : 0;
    const int row_a = i - lane + p;
    const int row_h = __shfl(h, p);
    const int car_a = wave_bcast(st.prev_a, l);
    const int car_h = wave_bcast(st.prev_h, l);
    const int prev_a = below ? row_a : car_a;
    const int prev_h = below ? row_h : car_h;
    const bool is_head = h > l;
    // non-first child of level l (a parent level exists above it)
    const bool is_nf = l + 1 < tv.nlev && h == l + 1;
    const bool is_second = is_nf && prev_h > l + 1;
    const unsigned long long nfm = l + 1 < tv.nlev ? (mask & ~mask1) : 0ull;
    const unsigned long long secm = __ballot(is_second);
    const uint32_t heads0 = (uint32_t)wave_bcast((int)st.heads, l);
    const uint32_t heads1 = l + 1 < tv.nlev ? (uint32_t)wave_bcast((int)st.heads, l + 1) : 0u;
    const uint32_t nf0 = (uint32_t)wave_bcast((int)st.nf, l);
    const uint32_t blk0 = (uint32_t)wave_bcast((int)st.blk, l);
    const int idx = (int)heads0 + __popcll(mask & lt);
    const int nf_before = (int)nf0 + __popcll(nfm & lt);
    const int blocks_incl = (int)blk0 + __popcll(secm & le);
    // node of level l + 1 that holds this (non-head there) point
    const int parent_idx = (int)heads1 + __popcll(mask1 & lt) - 1;
    visit(l, is_head, idx, prev_a, prev_h, is_nf, is_second, nf_before, blocks_incl, parent_idx);
    if (lane == l) {
      const int last = 63 - __clzll((long long)mask);
      st.prev_a = i - lane + last;
      st.heads += __popcll(mask);
      st.nf += __popcll(nfm);
      st.blk += __popcll(secm);
    }
    // (every lane needs the last head's h: one more exchange, wave-uniform source)
    const int last_h = wave_bcast(h, 63 - __clzll((long long)mask));
    if (lane == l)
      st.prev_h = last_h;
    mask = mask1;
  }
}

// Which points of a tile start a slice: normally none, or the tile's first point.
// Only a tile that holds a slice boundary in its middle looks every point up.
struct CxTileSlices {
  bool simple;
  int s0;       // slice of the tile's first point
  int start0;   // that slice's first point (a slice start inside the tile iff == base)
};

__device__ __forceinline__ CxTileSlices
cx_tile_slices(const TreeView& tv, int base)
{
  CxTileSlices ts;
  const int b = base < tv.n_total ? base : tv.n_total - 1;
  ts.s0 = find_slice(tv.pt_off, tv.num_slices, b);
  ts.start0 = tv.pt_off[ts.s0];
  ts.simple = tv.pt_off[ts.s0 + 1] >= base + kTilePoints;
  return ts;
}

// head_levels() of raht_tree.hpp without a slice search per point
__device__ __forceinline__ int
cx_head_levels(const TreeView& tv, const CxTileSlices& ts, int i)
{
  if (!ts.simple)
    return head_levels(tv, i);
  if (i == ts.start0)
    return tv.nlev;
  const uint64_t x = (uint64_t)(tv.pos[i] ^ tv.pos[i - 1]);
  if (!x)
    return 0;
  const int h = (bitlen64(x) + 2) / 3;
  return h < tv.nlev ? h : tv.nlev;
}

// carry of a tile's first row: lane l looks the previous head of level l up
__device__ __forceinline__ void
cx_tile_carry(const TreeView& tv, int base, CxRowState& st)
{
  const int lane = lane_id();
  st.prev_a = 0;
  st.prev_h = tv.nlev;
  // (also when the tile starts a slice: the node before it -- the previous
  // slice's last of every level -- gets its value slot from this carry)
  if (base > 0 && base < tv.n_total && lane < tv.nlev) {
    st.prev_a = cx_prev_head(tv, lane, base);
    st.prev_h = head_levels(tv, st.prev_a);
  }
}

// ---- count: per tile (one wavefront, kTilePoints points) -------------------------
template<int C>
__global__ __launch_bounds__(256) void
cx_count_kernel(TreeView tv, const int32_t* __restrict__ attrs, CxLists cl)
{
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  const int ncol = cx_ncol(tv.nlev, C);
  for (int tile = wave; tile < tv.num_tiles; tile += nwaves) {
    const int base = tile * kTilePoints;
    CxRowState st;
    st.heads = st.nf = st.blk = 0;
    cx_tile_carry(tv, base, st);
    const CxTileSlices ts = cx_tile_slices(tv, base);
    int32_t asum[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      asum[k] = 0;
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = base + r * kWave + lane;
      int h = 0;
      if (i < tv.n_total) {
        h = cx_head_levels(tv, ts, i);
        if (attrs) {
#pragma unroll
          for (int k = 0; k < C; k++)
            asum[k] += attrs[(size_t)i * C + k];
        }
      }
      cx_row_levels(tv, i, h, st, [](int, bool, int, int, int, bool, bool, int, int, int) {});
    }
    uint32_t* row = cl.tile_tab + (size_t)tile * ncol;
    if (lane < tv.nlev) {
      row[lane] = st.heads;
      row[tv.nlev + lane] = st.nf;
      row[2 * tv.nlev + lane] = st.blk;
    }
#pragma unroll
    for (int k = 0; k < C; k++) {
      int32_t v = attrs ? asum[k] : 0;
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1)
        v += __shfl_xor(v, d);
      if (lane == 0)
        row[3 * tv.nlev + k] = (uint32_t)v;
    }
  }
}

// ---- scan: one workgroup per column of the tile table -----------------------------
__global__ __launch_bounds__(256) void
cx_scan_kernel(TreeView tv, CxLists cl, int ncol)
{
  __shared__ uint32_t part[256];
  const int col = blockIdx.x;
  const int tid = threadIdx.x;
  const int per = (tv.num_tiles + 255) / 256;
  const int t0 = tid * per;
  const int t1 = t0 + per < tv.num_tiles ? t0 + per : tv.num_tiles;
  uint32_t sum = 0;
  for (int t = t0; t < t1; t++)
    sum += cl.tile_tab[(size_t)t * ncol + col];
  part[tid] = sum;
  __syncthreads();
  // exclusive prefix of the 256 partial sums (one wavefront, four per lane)
  if (tid < kWave) {
    uint32_t v[4], s = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      v[u] = part[tid * 4 + u];
      s += v[u];
    }
    const uint32_t inc = wave_incl_scan_u32(s);
    uint32_t run = inc - s;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      part[tid * 4 + u] = run;
      run += v[u];
    }
    if (tid == kWave - 1)
      cl.col_total[col] = inc;
  }
  __syncthreads();
  uint32_t run = part[tid];
  for (int t = t0; t < t1; t++) {
    const uint32_t v = cl.tile_tab[(size_t)t * ncol + col];
    cl.tile_tab[(size_t)t * ncol + col] = run;
    run += v;
  }
}

// ---- sentinels and the per-level list offsets --------------------------------------
// tab_host: a second copy of the level table in pinned HOST memory (or null) -- the host sizes the level launches
// from it after an event behind this kernel.  (Until round 5 the table came back with a hipMemcpyAsync: one step in
// ten of a long-running process then took 20-80 ms with every kernel at its usual time -- the copy engine's path, not
// the kernels'; tools/archive/r05_stall_probe3.py.  The sub-node path never had it: its schedule kernel writes its statistics
// to pinned memory itself.)
template<int C>
__global__ __launch_bounds__(64) void
cx_scan_fin_kernel(TreeView tv, CxLists cl, int32_t* attr_prefix, int has_attrs, CxLevelTab* tab_host = nullptr)
{
  const int lane = threadIdx.x;
  const int nlev = tv.nlev;
  int m = 0, nf = 0, nb = 0;
  if (lane < nlev) {
    m = (int)cl.col_total[lane];
    nf = (int)cl.col_total[nlev + lane];
    nb = (int)cl.col_total[2 * nlev + lane];
    // more nodes than the level's arrays hold: the Morton-bits hint was smaller
    // than the codes' width; everything behind leaves on the error word
    if (m > tv.cap[lane]) {
      atomicExch(tv.error, 2);
      m = tv.cap[lane];
    }
    tv.soff[lane][tv.num_slices] = m;
    tv.fp[lane][m] = tv.n_total;
  }
  const int below = __shfl_up(m, 1u);
  if (lane >= 1 && lane < nlev && tv.fc[lane])
    tv.fc[lane][m] = below;
  // exclusive prefixes over the levels
  const uint32_t nbi = wave_incl_scan_u32((uint32_t)nb);
  const uint32_t nri = wave_incl_scan_u32((uint32_t)(nb + nf));
  if (lane < nlev) {
    cl.tab->boff[lane] = (int32_t)(nbi - nb);
    cl.tab->roff[lane] = (int32_t)(nri - (nb + nf));
    cl.tab->nb[lane] = nb;
    cl.tab->nr[lane] = nb + nf;
    cl.tab->nodes[lane] = m;
    if (tab_host) {
      tab_host->boff[lane] = (int32_t)(nbi - nb);
      tab_host->roff[lane] = (int32_t)(nri - (nb + nf));
      tab_host->nb[lane] = nb;
      tab_host->nr[lane] = nb + nf;
      tab_host->nodes[lane] = m;
    }
    cl.bq[(int)(nbi - nb) + lane + nb] = nb + nf;  // the level's sentinel
  }
  if (lane == 0)
    cl.h[tv.n_total] = (uint8_t)nlev;
  if (has_attrs && lane < C)
    attr_prefix[(size_t)tv.n_total * C + lane] = (int32_t)cl.col_total[3 * nlev + lane];
}

// ---- emit -------------------------------------------------------------------------
template<int C>
__global__ __launch_bounds__(256) void
cx_emit_kernel(
  TreeView tv, const int32_t* __restrict__ attrs, CxLists cl, int32_t* attr_prefix)
{
  GPCC_VGPR_FLOOR_64();
  if (tree_failed(tv))
    return;
  const int lane = lane_id();
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) / kWave;
  const int nwaves = gridDim.x * blockDim.x / kWave;
  const int ncol = cx_ncol(tv.nlev, C);
  const int n = tv.n_total;
  for (int tile = wave; tile < tv.num_tiles; tile += nwaves) {
    const int base = tile * kTilePoints;
    const uint32_t* row = cl.tile_tab + (size_t)tile * ncol;
    CxRowState st;
    st.heads = lane < tv.nlev ? row[lane] : 0;
    st.nf = lane < tv.nlev ? row[tv.nlev + lane] : 0;
    st.blk = lane < tv.nlev ? row[2 * tv.nlev + lane] : 0;
    cx_tile_carry(tv, base, st);
    const CxTileSlices ts = cx_tile_slices(tv, base);
    int32_t run[C];
    if (attrs) {
#pragma unroll
      for (int k = 0; k < C; k++)
        run[k] = (int32_t)row[3 * tv.nlev + k];
    }
    for (int r = 0; r < kTilePoints / kWave; r++) {
      const int i = base + r * kWave + lane;
      const bool in = i < n;
      int h = 0;
      int64_t p = 0;
      bool start = false;
      int s = 0;
      if (in) {
        h = cx_head_levels(tv, ts, i);
        p = tv.pos[i];
        cl.h[i] = (uint8_t)h;
        if (h == tv.nlev) {
          s = ts.simple ? ts.s0 : find_slice(tv.pt_off, tv.num_slices, i);
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
      cx_row_levels(
        tv, i, h, st,
        [&](int l, bool is_head, int idx, int prev_a, int prev_h, bool is_nf, bool is_second,
            int nf_before, int blocks_incl, int parent_idx) {
          if (is_head) {
            tv.fp[l][idx] = i;
            tv.key[l][idx] = p >> (3 * l);
            if (l && tv.fc[l])
              tv.fc[l][idx] = prev_idx;
            if (start)
              tv.soff[l][s] = idx;
            // the node before this one ends here: its value slot is known now
            if (idx > 0)
              cl.hold[l][idx - 1] = cx_hold_value(prev_a, prev_h, i, h, n);
            prev_idx = idx;
          }
          if (is_nf) {
            const int b = blocks_incl - 1;
            const int rank = nf_before + blocks_incl;
            const int boff = cl.tab->boff[l], roff = cl.tab->roff[l];
            cl.rb[roff + rank] = b;
            if (is_second) {
              cl.rb[roff + rank - 1] = b;
              cl.bp[boff + b] = parent_idx;
              cl.bc[boff + b] = idx - 1;  // (the node before this one starts the block)
              cl.bq[boff + l + b] = rank - 1;
            }
          }
        });
    }
    // the last node of every level ends with the batch
    if (tile == tv.num_tiles - 1 && lane < tv.nlev && st.heads > 0)
      cl.hold[lane][st.heads - 1] = cx_hold_value(st.prev_a, st.prev_h, n, tv.nlev, n);
  }
}

}  // namespace gpcc
