// lod_kernels.hpp -- level-of-detail generation on gfx950
// (buildPredictorsFast, tmc3/PCCTMC3Common.h:2300-2469) for intra,
// non-scalable attribute coding.
//
// The reference walks the Morton-sorted point list once per LoD with two
// pieces of sliding state (a 128^3-cell "atlas" and a window cursor).  Here:
//
//  * sub-sampling by distance (subsampleByDistance :1984-2085) is a greedy
//    over points, but its state collapses per CELL of size 2^(shift+1): at
//    most one point of a cell is retained (the first without a retained
//    point within the radius in the 19 neighbour cells of its atlas block),
//    and a cell only depends on neighbour cells that PRECEDE it in Morton
//    order.  Cells are processed by a dependency-ordered kernel (tickets +
//    done flags, the scheme of raht_subnode.hpp), one lane per cell.
//  * the nearest-neighbour search of a refinement point
//    (computeNearestNeighbors :1147-1953) is a pure function of the point,
//    the sorted retained list, its three-level bounding boxes and one scalar
//    per LoD (`atlas_limit`, see lod_atlas_limit_kernel): one thread per
//    point replays the reference's exact visit order (27 atlas cells, then
//    the +-range window with box pruning, then the same-LoD window, then the
//    distribution-aware replacement of the third neighbour), with the atlas
//    look-ups replaced by binary searches in the retained list.
#pragma once

#include "raht_common.hpp"
#include "raht_subnode.hpp"
#include "lift_kernels.hpp"

namespace gpcc {

constexpr int kAtlasBits = 21;  // 3 * log2(128)

struct LodCtx {
  int32_t n;                 // points of the slice
  const int64_t* code;       // [n] sorted Morton codes
  const int32_t* order;      // [n] point index of each sorted entry
  const int32_t* pos;        // [n][3] positions, sorted order
  const int32_t* bpos;       // [n][3] positions * lodNeighBias, sorted order
  // the LoD being built
  const int32_t* input;      // [n_in] packed (sorted-order) indices, ascending
  int32_t n_in;
  int32_t shift3;            // 3 * (shift bits) of this step
  int32_t boundary;          // min(63, shift3 + 21)
  int64_t radius2;
  // cells of the input list (sub-sampling by distance)
  int32_t* cell_first;       // [ncell + 1] position in `input`
  int32_t ncell;
  int64_t* cell_key;         // [ncell] code >> shift3 of each cell
  uint32_t* cell_state;      // [ncell][4] {x, y, z, tag} of the retained point (16-B granule)
  int32_t* ticket;           // [8]
  int32_t* error;
  int32_t epoch;
  uint8_t* flags;            // [n_in] 1 = retained
};

__device__ __forceinline__ int64_t
norm2_i3(const int32_t* a, const int32_t* b)
{
  const int64_t dx = (int64_t)a[0] - b[0], dy = (int64_t)a[1] - b[1], dz = (int64_t)a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;
}
__device__ __forceinline__ int32_t
norm1_i3(const int32_t* a, const int32_t* b)
{
  return abs(a[0] - b[0]) + abs(a[1] - b[1]) + abs(a[2] - b[2]);
}

// positions the sort moved (canonical point order: none may have)
__global__ __launch_bounds__(256) void
lod_count_moved_kernel(int n, const int32_t* __restrict__ order, int32_t* moved)
{
  int cnt = 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    cnt += order[i] != i;
  if (cnt)
    atomicAdd(moved, cnt);
}

// ---- gather sorted positions ---------------------------------------------
__global__ __launch_bounds__(256) void
lod_gather_pos_kernel(
  int n, const int32_t* __restrict__ xyz, const int32_t* __restrict__ order,
  int b0, int b1, int b2, int32_t* pos, int32_t* bpos, int32_t* list)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    const int p = order[i];
    const int32_t x = xyz[3 * (size_t)p], y = xyz[3 * (size_t)p + 1], z = xyz[3 * (size_t)p + 2];
    pos[3 * (size_t)i] = x;
    pos[3 * (size_t)i + 1] = y;
    pos[3 * (size_t)i + 2] = z;
    bpos[3 * (size_t)i] = x * b0;
    bpos[3 * (size_t)i + 1] = y * b1;
    bpos[3 * (size_t)i + 2] = z * b2;
    list[i] = i;
  }
}

// scalable lifting: the search of LoD `l` sees every point at the corner of its
// octree node of size 2^l (clacIntermediatePosition :925-940), then biased
__global__ __launch_bounds__(256) void
lod_node_corner_bpos_kernel(
  int n, const int32_t* __restrict__ pos, uint32_t mask, int b0, int b1, int b2, int32_t* bpos)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bpos[3 * (size_t)i] = (int32_t)((uint32_t)pos[3 * (size_t)i] & mask) * b0;
    bpos[3 * (size_t)i + 1] = (int32_t)((uint32_t)pos[3 * (size_t)i + 1] & mask) * b1;
    bpos[3 * (size_t)i + 2] = (int32_t)((uint32_t)pos[3 * (size_t)i + 2] & mask) * b2;
  }
}

// ---- stable partition of positions 0..n-1 by a flag ----------------------
// out_true / out_false receive list[t] of the flagged / unflagged positions
// in order; counts[0] = number flagged.  Decoupled look-back over <= 1024
// resident workgroups (cf. raht_level_prepass_kernel).
__global__ __launch_bounds__(256) void
lod_partition_kernel(
  int n, const uint8_t* __restrict__ flags, const int32_t* __restrict__ list,
  int32_t* out_true, int32_t* out_false, int32_t* counts,
  unsigned long long* scan_state, int epoch)
{
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int64_t chunks = ((int64_t)n + 255) >> 8;
  const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const int64_t gbeg = min((int64_t)blockIdx.x * per, chunks);
  const int64_t gend = min(gbeg + per, chunks);
  int mine = 0;
  for (int64_t ch = gbeg; ch < gend; ch++) {
    const int t = (int)(ch * 256) + threadIdx.x;
    if (t < n)
      mine += flags[t] != 0;
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1)
    mine += __shfl_xor(mine, d);
  if (lane == 0)
    wave_cnt[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ep = (unsigned long long)epoch << 48;
    const unsigned total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    unsigned excl = 0;
    if (blockIdx.x > 0) {
      __hip_atomic_store(&scan_state[blockIdx.x], ep | (1ull << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int k = (int)blockIdx.x - 1;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(&scan_state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 48) != (unsigned long long)epoch) {
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        excl += (unsigned)v;
        if (((v >> 32) & 0xffff) == 2)
          break;
        k--;
      }
    }
    __hip_atomic_store(&scan_state[blockIdx.x], ep | (2ull << 32) | (excl + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == gridDim.x - 1)
      counts[0] = (int)(excl + total);
    base_s = (int)excl;
  }
  __syncthreads();
  int running = base_s;
  for (int64_t ch = gbeg; ch < gend; ch++) {
    const int t = (int)(ch * 256) + threadIdx.x;
    const bool f = t < n && flags[t] != 0;
    const unsigned long long m = __ballot(f);
    __syncthreads();
    if (lane == 0)
      wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; w++)
      off += wave_cnt[w];
    const int rank_true = off + __popcll(m & ((1ull << lane) - 1));
    if (t < n) {
      if (f)
        out_true[rank_true] = list[t];
      else if (out_false)
        out_false[t - rank_true] = list[t];
    }
    running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
}

// ---- sub-sampling ----------------------------------------------------------
// periodic decimation (subsampleByDecimation :2198-2214)
__global__ __launch_bounds__(256) void
lod_flag_periodic_kernel(int n_in, int period, uint8_t* flags)
{
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n_in;
       t += gridDim.x * blockDim.x)
    flags[t] = (t % period) == 0;
}

// ---- subsampleByOctreeWithCentroid (lodDecimator 2, :2089-2194) ------------
// The reference walks the list once: points accumulate until a cell boundary
// is reached with at least `period` points, then the point closest to the
// (masked) centroid of the group is retained.  The greedy grouping restated:
// a group starting at g ends at the end of the cell that holds position
// g + period - 1, so the group starts are the orbit of 0 under
// next(g) = cell_end(min(g + period - 1, n - 1)) + 1 -- marked by pointer
// doubling (log2 n launches), no serial walk.
__global__ __launch_bounds__(256) void
lod_centroid_next_kernel(LodCtx cx, int period, int32_t* __restrict__ nxt)
{
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g <= cx.n_in;
       g += gridDim.x * blockDim.x) {
    if (g == cx.n_in) {
      nxt[g] = cx.n_in;
      continue;
    }
    const int x = min(g + period - 1, cx.n_in - 1);
    const int64_t key = cx.code[cx.input[x]] >> cx.shift3;
    int lo = x, hi = cx.n_in;  // first position whose cell key is larger
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((cx.code[cx.input[mid]] >> cx.shift3) <= key)
        lo = mid + 1;
      else
        hi = mid;
    }
    nxt[g] = lo;
  }
}

// one doubling round: marked starts mark their 2^k-th successor, every
// pointer is squared
__global__ __launch_bounds__(256) void
lod_centroid_jump_kernel(
  int n_in, const int32_t* __restrict__ nxt_in, int32_t* __restrict__ nxt_out,
  uint8_t* mark)
{
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g <= n_in;
       g += gridDim.x * blockDim.x) {
    const int n1 = nxt_in[g];
    if (g < n_in && n1 < n_in && mark[g])
      mark[n1] = 1;
    nxt_out[g] = n1 < n_in ? nxt_in[n1] : n_in;
  }
}

// one thread per group: centroid, nearest member (ties: the first one met walking
// from the back -- lodDecimator 2 and the odd levels of scalable lifting -- or
// from the front -- its even levels, subsample :2230-2235), flags
__global__ __launch_bounds__(256) void
lod_centroid_pick_kernel(
  LodCtx cx, int node_log2, const int32_t* __restrict__ nxt0,
  const uint8_t* __restrict__ mark, int backward)
{
  const uint32_t mask = node_log2 ? 0xffffffffu << node_log2 : 0xffffffffu;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < cx.n_in;
       g += gridDim.x * blockDim.x) {
    if (!mark[g])
      continue;
    const int e = nxt0[g];  // one past the group's last position
    const int nv = e - g;
    uint32_t cen[3] = {0, 0, 0};  // Vec3<int32_t> sums wrap like the reference's
    for (int v = g; v < e; v++) {
      const int idx = cx.input[v];
#pragma unroll
      for (int d = 0; d < 3; d++)
        cen[d] += (uint32_t)cx.pos[3 * (size_t)idx + d] & mask;
    }
    int best = backward ? e - 1 : g;
    int64_t best_m = INT64_MAX;
    for (int w = 0; w < nv; w++) {
      const int v = backward ? e - 1 - w : g + w;
      const int idx = cx.input[v];
      int64_t m = 0;
#pragma unroll
      for (int d = 0; d < 3; d++) {
        const uint32_t p = ((uint32_t)cx.pos[3 * (size_t)idx + d] & mask) * (uint32_t)nv;
        const int32_t df = (int32_t)(p - cen[d]);
        m += df < 0 ? -(int64_t)df : (int64_t)df;
      }
      m = (int32_t)m;  // getNorm1 on Vec3<int32_t>
      if (best_m > m) {
        best_m = m;
        best = v;
      }
    }
    for (int v = g; v < e; v++)
      cx.flags[v] = v == best;
  }
}

// cell heads of the input list
__global__ __launch_bounds__(256) void
lod_flag_cell_heads_kernel(LodCtx cx, uint8_t* heads, int32_t* positions)
{
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < cx.n_in;
       t += gridDim.x * blockDim.x) {
    const int64_t c = cx.code[cx.input[t]] >> cx.shift3;
    heads[t] = t == 0 || (cx.code[cx.input[t - 1]] >> cx.shift3) != c;
    positions[t] = t;
    cx.flags[t] = 0;
  }
}

// keys of the cells (one binary-search probe = one load)
__global__ __launch_bounds__(256) void
lod_cell_keys_kernel(LodCtx cx)
{
  for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < cx.ncell;
       x += gridDim.x * blockDim.x)
    cx.cell_key[x] = cx.code[cx.input[cx.cell_first[x]]] >> cx.shift3;
}


// subsampleByDistance (:1984-2085).  At most one point per cell is retained
// and the decision of a cell depends on the retained points of up to 19
// neighbour cells that precede it (same 128^3-cell atlas block): a wavefront
// of dependencies, one lane per cell, cells claimed in Morton order.  A cell
// publishes ONE 16-byte write-through granule {x, y, z, tag} (tag = epoch
// and "a point was retained"); consumers poll the granules of their
// neighbours directly -- the data is the flag (cdna_hip_programming.md G16,
// form R2), one memory round trip per dependency hop.  The hop is one
// ITERATION of the polling loop, so the loop body is what counts: an arriving
// neighbour is folded into a bit mask of eliminated points at once and nothing
// else is kept of it (no per-lane neighbour lists in LDS; 168 registers, three
// waves per SIMD -- at four the prologue's 19 lock-step bisections spill).
#ifndef GPCC_LOD_SUB_WAVES
#define GPCC_LOD_SUB_WAVES 3
#endif
// granules polled side by side per memory round trip of the loop (19 = all of a lane's neighbours at once)
#ifndef GPCC_LOD_BATCH
#define GPCC_LOD_BATCH 10
#endif
#ifndef GPCC_LOD_IDLE_FAST
#define GPCC_LOD_IDLE_FAST 1
#endif
__global__ __launch_bounds__(256, GPCC_LOD_SUB_WAVES) void
lod_subsample_distance_kernel(LodCtx cx)
{
  constexpr uint8_t kOff[19] = {3, 5, 6, 12, 10, 17, 20, 34, 33, 4,
                                2, 1, 24, 40, 48, 32, 16, 8, 0};
  const int lane = lane_id();
  const int cls = blockIdx.x & 7;
  const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(
    cx.cell_state, 0, (int)(((size_t)cx.ncell + 1) * 16), 0x00020000);
  const uint32_t tag_none = (uint32_t)cx.epoch << 1, tag_has = tag_none | 1;
  constexpr int kCellCache = 8;
  // cell edge 2^(shift3/3): neighbour points are < 2 edges away per axis
  const bool small = cx.shift3 <= 3 * 13;
  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&cx.ticket[cls], 1);
    tk = __shfl(tk, 0);
    const int64_t wround = (int64_t)tk * 8 + cls;
    if (wround * 64 >= cx.ncell)
      break;
    // a bounded wait has expired somewhere: the result is discarded anyway,
    // leave at once instead of spinning through every remaining round
    if (__hip_atomic_load(cx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
    const int x = (int)(wround * 64) + lane;
    const bool live = x < cx.ncell;
    int t0 = 0, t1 = 0;
    // ---- the 19 neighbour cells: lower_bound in cell_key[0, x), all
    //      searches of a lane advance in lock step (loads in flight together)
    int lo[19], hi[19];
    int64_t want[19];
    uint32_t vmask = 0;
#pragma unroll
    for (int k = 0; k < 19; k++) {
      lo[k] = hi[k] = 0;
      want[k] = 0;
    }
    if (live) {
      t0 = cx.cell_first[x];
      t1 = cx.cell_first[x + 1];
      const int64_t cell = cx.cell_key[x];
      const int64_t atlas = cell >> kAtlasBits;
      const uint64_t base = morton3d_add((uint64_t)cell, ~0ull);
#pragma unroll
      for (int k = 0; k < 19; k++) {
        const int64_t nb = (int64_t)morton3d_add(base, kOff[k]);
        // only cells that precede this one hold retained points when it is
        // examined; later cells never matter
        if ((nb >> kAtlasBits) == atlas && nb >= 0 && nb < cell) {
          want[k] = nb;
          hi[k] = x;
          vmask |= 1u << k;
        }
      }
    }
    for (;;) {
      bool active = false;
#pragma unroll
      for (int k = 0; k < 19; k++)
        active |= lo[k] < hi[k];
      if (!__any(active))
        break;
      int64_t kv[19];
#pragma unroll
      for (int k = 0; k < 19; k++)
        kv[k] = lo[k] < hi[k] ? cx.cell_key[lo[k] + ((hi[k] - lo[k]) >> 1)] : 0;
#pragma unroll
      for (int k = 0; k < 19; k++)
        if (lo[k] < hi[k]) {
          const int mid = lo[k] + ((hi[k] - lo[k]) >> 1);
          if (kv[k] < want[k])
            lo[k] = mid + 1;
          else
            hi[k] = mid;
        }
    }
    uint32_t pend = 0;
    if (live) {
#pragma unroll
      for (int k = 0; k < 19; k++)
        if (((vmask >> k) & 1) && lo[k] < x && cx.cell_key[lo[k]] == want[k])
          pend |= 1u << k;
    }

    // the cell's first points, fetched while the neighbours are pending
    int32_t cpx[kCellCache], cpy[kCellCache], cpz[kCellCache];
#pragma unroll
    for (int u = 0; u < kCellCache; u++) {
      cpx[u] = cpy[u] = cpz[u] = 0;
      if (t0 + u < t1) {
        const int idx = cx.input[t0 + u];
        cpx[u] = cx.pos[3 * (size_t)idx];
        cpy[u] = cx.pos[3 * (size_t)idx + 1];
        cpz[u] = cx.pos[3 * (size_t)idx + 2];
      }
    }
    // Which neighbour cells can matter to which point: a cell none of whose
    // voxels lies within the radius of a point cannot eliminate it, wherever
    // its retained point is (exact).  The points are examined in order and
    // the first one that survives is retained, so point u is decidable as
    // soon as the cells that can reach IT have arrived; cells that reach no
    // point at all are dropped from the dependency set.  Shortens the chains.
    const bool small_cell = t1 - t0 <= kCellCache;
    uint32_t rm[kCellCache];
#pragma unroll
    for (int u = 0; u < kCellCache; u++)
      rm[u] = 0;
    if (live && pend && small_cell) {
      const int sh = cx.shift3 / 3;  // cell edge 2^sh
      const int32_t edge = 1 << sh;
      const int32_t ox = (cpx[0] >> sh) << sh, oy = (cpy[0] >> sh) << sh, oz = (cpz[0] >> sh) << sh;
#pragma unroll
      for (int k = 0; k < 19; k++) {
        if (!((pend >> k) & 1))
          continue;
        // kOff = Morton code of (dx+1, dy+1, dz+1); x is the top bit of a triple
        const int off = kOff[k];
        const int dx = ((off >> 2) & 1) + 2 * ((off >> 5) & 1) - 1;
        const int dy = ((off >> 1) & 1) + 2 * ((off >> 4) & 1) - 1;
        const int dz = (off & 1) + 2 * ((off >> 3) & 1) - 1;
        const int32_t lx = ox + dx * edge, ly = oy + dy * edge, lz = oz + dz * edge;
        bool reach = false;
#pragma unroll
        for (int u = 0; u < kCellCache; u++) {
          if (t0 + u >= t1)
            continue;
          const int64_t gx = max(max(lx - cpx[u], cpx[u] - (lx + edge - 1)), 0);
          const int64_t gy = max(max(ly - cpy[u], cpy[u] - (ly + edge - 1)), 0);
          const int64_t gz = max(max(lz - cpz[u], cpz[u] - (lz + edge - 1)), 0);
          if (gx * gx + gy * gy + gz * gz <= cx.radius2) {
            rm[u] |= 1u << k;
            reach = true;
          }
        }
        if (!reach)
          pend &= ~(1u << k);
      }
    }
    // ---- the wait.  What an arriving neighbour can change is which of the
    //      cell's first points lie within the radius of a retained point: one
    //      bit per cached point (`elim`), set when the neighbour ARRIVES (eight
    //      distance tests, once), so the per-iteration decision is a few bit
    //      operations: the first point not eliminated is retained as soon as no
    //      cell that can reach it is still out.  The neighbours' points are not
    //      kept (a cell with more than kCellCache points re-reads the granules
    //      of its neighbours when it scans the rest: they have all arrived).
    const int ncache = live ? (t1 - t0 < kCellCache ? t1 - t0 : kCellCache) : 0;
    const uint32_t valid = (1u << ncache) - 1;
    uint32_t elim = 0;
    uint32_t hasmask = 0;  // neighbours that arrived with a retained point
    bool pending = live;
    unsigned spins = 0;
    bool decided_once = false;  // the decision below has been taken with the current `pend` / `elim`
    while (__any(pending)) {
      // poll every pending neighbour; the loads of a batch are issued together
      // BEFORE any result is looked at (two batches bound the live registers)
      const uint32_t todo = pending ? pend : 0;
#pragma unroll
      for (int k0 = 0; k0 < 19; k0 += GPCC_LOD_BATCH) {
        u32x4 vv[GPCC_LOD_BATCH];
#pragma unroll
        for (int q = 0; q < GPCC_LOD_BATCH; q++) {
          const int k = k0 + q;
          vv[q] = u32x4{0, 0, 0, 0};
          if (k < 19 && ((todo >> k) & 1))
            vv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lo[k] * 16, 0, /*sc1*/ 16);
        }
#pragma unroll
        for (int q = 0; q < GPCC_LOD_BATCH; q++) {
          const int k = k0 + q;
          if (k < 19 && ((todo >> k) & 1) && (vv[q].w >> 1) == (uint32_t)cx.epoch) {
            pend &= ~(1u << k);
            if (vv[q].w & 1) {
              hasmask |= 1u << k;
              const int32_t nx = (int32_t)vv[q].x, ny = (int32_t)vv[q].y, nz = (int32_t)vv[q].z;
#pragma unroll
              for (int u = 0; u < kCellCache; u++) {
                bool in;
                if (small) {
                  // neighbours lie in adjacent cells: |d| < 2^14, squares fit 32 bits
                  const int32_t dx = nx - cpx[u], dy = ny - cpy[u], dz = nz - cpz[u];
                  in = (int64_t)(dx * dx + dy * dy + dz * dz) <= cx.radius2;
                } else {
                  const int64_t dx = (int64_t)nx - cpx[u], dy = (int64_t)ny - cpy[u],
                                dz = (int64_t)nz - cpz[u];
                  in = dx * dx + dy * dy + dz * dz <= cx.radius2;
                }
                elim |= (uint32_t)in << u;
              }
            }
          }
        }
      }
      // (no neighbour of any lane has arrived since the last decision: it stands -- the waiting iteration, of
      // which a cell spends tens, is the polls and this test; GPCC_LOD_IDLE_FAST, profiles/r06_idle_ab.txt)
      if (GPCC_LOD_IDLE_FAST && decided_once && !__any(pending && pend != todo)) {
        if (++spins > (1u << 20)) {
          if (lane == 0)
            atomicExch(cx.error, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      decided_once = true;
      // ---- decide as far as the arrived neighbours allow -----------------
      int kept = -1;
      int32_t kp[3] = {0, 0, 0};
      int kt = 0;
      bool ready = false, scan = false;
      if (pending) {
        const uint32_t cand = ~elim & valid;
        if (small_cell) {
          uint32_t blockmask = 0;
#pragma unroll
          for (int u = 0; u < kCellCache; u++)
            blockmask |= (uint32_t)((rm[u] & pend) != 0) << u;
          if (!cand) {
            ready = true;  // every point eliminated: nothing retained
          } else {
            const int u1 = __ffs((int)cand) - 1;
            ready = !((blockmask >> u1) & 1);
            kept = ready ? u1 : -1;
          }
        } else if (pend == 0) {
          ready = true;
          if (cand)
            kept = __ffs((int)cand) - 1;
          else
            scan = true;  // the first points are eliminated: the rest of the cell
        }
      }
      if (!__any(ready)) {
        if (++spins > (1u << 20)) {
          if (lane == 0)
            atomicExch(cx.error, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        continue;
      }
      if (kept >= 0) {
        // the cached point `kept` (select chain: register arrays take no dynamic index)
#pragma unroll
        for (int u = 0; u < kCellCache; u++)
          if (u == kept) {
            kp[0] = cpx[u];
            kp[1] = cpy[u];
            kp[2] = cpz[u];
          }
        kt = t0 + kept;
        cx.flags[kt] = 1;
      }
      // cells with more points: the whole wavefront scans 64 points per round
      // trip for one such cell at a time (the scan sits on the dependency
      // chain of every later cell)
      for (;;) {
        const unsigned long long big = __ballot(scan);
        if (!big)
          break;
        const int owner = __ffsll((long long)big) - 1;
        // lane k < 19 fetches the retained point of the owner's neighbour k
        const uint32_t o_has = __shfl(hasmask, owner);
        int o_lo = 0;
#pragma unroll
        for (int k = 0; k < 19; k++) {
          const int v = __shfl(lo[k], owner);
          if (lane == k)
            o_lo = v;
        }
        u32x4 g = u32x4{0, 0, 0, 0};
        if (lane < 19 && ((o_has >> lane) & 1))
          g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, o_lo * 16, 0, /*sc1*/ 16);
        const int o_t1 = __shfl(t1, owner);
        int o_kept = -1, o_t = 0;
        int32_t okp[3] = {0, 0, 0};
        for (int tb = __shfl(t0, owner) + kCellCache; tb < o_t1 && o_kept < 0; tb += 64) {
          const int t = tb + lane;
          bool cand = false;
          int idx = 0;
          int32_t px = 0, py = 0, pz = 0;
          if (t < o_t1) {
            idx = cx.input[t];
            px = cx.pos[3 * (size_t)idx];
            py = cx.pos[3 * (size_t)idx + 1];
            pz = cx.pos[3 * (size_t)idx + 2];
          }
          bool found = false;
#pragma unroll
          for (int k = 0; k < 19; k++) {
            const int64_t dx = (int64_t)(int32_t)__shfl((int)g.x, k) - px,
                          dy = (int64_t)(int32_t)__shfl((int)g.y, k) - py,
                          dz = (int64_t)(int32_t)__shfl((int)g.z, k) - pz;
            found |= ((o_has >> k) & 1) && dx * dx + dy * dy + dz * dz <= cx.radius2;
          }
          cand = t < o_t1 && !found;
          const unsigned long long m = __ballot(cand);
          if (m) {
            const int first = __ffsll((long long)m) - 1;
            o_kept = __shfl(idx, first);
            o_t = tb + first;
            okp[0] = __shfl(px, first);
            okp[1] = __shfl(py, first);
            okp[2] = __shfl(pz, first);
          }
        }
        if (lane == owner) {
          scan = false;
          if (o_kept >= 0) {
            kept = o_kept;
            kp[0] = okp[0];
            kp[1] = okp[1];
            kp[2] = okp[2];
            cx.flags[o_t] = 1;
          }
        }
      }
      if (ready) {
        const u32x4 st = {(uint32_t)kp[0], (uint32_t)kp[1], (uint32_t)kp[2], kept >= 0 ? tag_has : tag_none};
        __builtin_amdgcn_raw_buffer_store_b128(st, rsrc, x * 16, 0, /*sc1*/ 16);
        pending = false;
      }
    }
  }
}

// ---- nearest-neighbour search ----------------------------------------------
struct NnCtx {
  int32_t n;
  const int64_t* code;
  const int32_t* order;
  const int32_t* bpos;       // [n][3] biased positions, sorted order
  const int32_t* retained;   // [n_ret] packed indices (ascending codes)
  const int64_t* ret_key;    // [n_ret] code >> shift3 of the retained entries
  int32_t n_ret;
  const int32_t* refine;     // [n_ref] packed indices of this LoD's points
  int32_t n_ref;
  int32_t start;             // position of refine[0] in the coding-order list
  int32_t shift3, boundary;
  int32_t distribution, range_inter, range_intra, intra, max_neigh;
  const int32_t* box_ret[3][2];  // [level][min/max] -> [buckets][3]
  const int32_t* box_ref[3][2];
  const long long* atlas_limit;
  // outputs (predictor p = n - 1 - position in coding-order list)
  int32_t* pred_count;       // [n]
  int32_t* pred_point;       // [n][3] neighbour POINT index
  uint64_t* pred_dist2;      // [n][3]
  int32_t* pt2pred;          // [n]
  int32_t* indexes;          // [n] predictor order -> point index
  // scalable lifting only (lod_nn_search_kernel<true>): neighbours beyond the
  // range are dropped with all that follow (:1918-1939)
  const int32_t* pos;        // [n][3] positions, sorted order
  uint32_t node_mask;        // clears the bits below the octree level of this LoD
  int32_t unit_bias;         // lodNeighBias == 1: the squared distances ARE the test values
  int64_t prune_dist;        // 3 * (max_neigh_range_minus1 + 1) << 2 * lod
  // attribute inter prediction only (lod_nn_search_kernel<.., true>): the reference
  // frame in Morton order (computeNearestNeighbors :1270-1292, :1606-1796)
  const int64_t* frame_code;     // [n_frame] sorted Morton codes
  const int32_t* frame_order;    // [n_frame] point index (in the reference frame) of each entry
  const int32_t* frame_bpos;     // [n_frame][3] biased positions, sorted order
  const int32_t* frame_identity; // [n_frame] 0, 1, 2, ... (the list the window scan walks)
  int32_t n_frame;
  int32_t frame_range;           // abh.attrInterPredSearchRange
  int32_t frame_boundary;        // min(63, shift3 + 9): the 8^3-cell inter atlas
  const int32_t* box_frame[3][2];
};

// A candidate of the reference frame carries this bit in its index from the moment it is
// met: the pair (index, localRef) of the reference in one word, so the duplicate tests,
// the sorting and the replacement rules need no second array (indices are < 2^29).  In
// pred_point it marks a neighbour whose value is a point index of the reference frame.
constexpr int32_t kFrameTag = 1 << 30;

struct NnState {
  int32_t idx[6];
  int64_t dist[6];
  int idx2;
};

__device__ __forceinline__ void
nn_update(NnState& s, int32_t d, int32_t index)
{
  if (d >= s.dist[2])
    return;
  if (d < s.dist[0]) {
    s.dist[2] = s.dist[1];
    s.dist[1] = s.dist[0];
    s.dist[0] = d;
    s.idx[2] = s.idx[1];
    s.idx[1] = s.idx[0];
    s.idx[0] = index;
  } else if (d < s.dist[1]) {
    s.dist[2] = s.dist[1];
    s.dist[1] = d;
    s.idx[2] = s.idx[1];
    s.idx[1] = index;
  } else {
    s.dist[2] = d;
    s.idx[2] = index;
  }
}

// slots 3..5 form a small ring written at s.idx2; the register-array form
// avoids dynamic indexing
__device__ __forceinline__ void
nn_push_extra(NnState& s, int32_t v)
{
  if (s.idx2 == 3)
    s.idx[3] = v;
  else if (s.idx2 == 4)
    s.idx[4] = v;
  else
    s.idx[5] = v;
  s.idx2++;
}

__device__ __forceinline__ void
nn_update_dist(NnState& s, int32_t d, int32_t index)
{
  if (d > s.dist[2]) {
  } else if (d < s.dist[0]) {
    if (s.idx[2] != -1)
      nn_push_extra(s, s.idx[2]);
    s.dist[2] = s.dist[1];
    s.dist[1] = s.dist[0];
    s.dist[0] = d;
    s.idx[2] = s.idx[1];
    s.idx[1] = s.idx[0];
    s.idx[0] = index;
  } else if (d < s.dist[1]) {
    if (s.idx[2] != -1)
      nn_push_extra(s, s.idx[2]);
    s.dist[2] = s.dist[1];
    s.dist[1] = d;
    s.idx[2] = s.idx[1];
    s.idx[1] = index;
  } else if (d < s.dist[2]) {
    if (s.idx[2] != -1)
      nn_push_extra(s, s.idx[2]);
    s.dist[2] = d;
    s.idx[2] = index;
  } else if (s.idx[5] == -1) {
    nn_push_extra(s, index);
  }
  if (s.idx2 == 6)
    s.idx2 = 3;
}

__device__ __forceinline__ void
nn_visit(NnState& s, bool distribution, bool check, int32_t d, int32_t index)
{
  if (check) {
    if (index == s.idx[0] || index == s.idx[1] || index == s.idx[2])
      return;
    if (distribution && (index == s.idx[3] || index == s.idx[4] || index == s.idx[5]))
      return;
  }
  if (distribution)
    nn_update_dist(s, d, index);
  else
    nn_update(s, d, index);
}

__device__ __forceinline__ int32_t
box_dist1(const int32_t* const box[2], int b, const int32_t* p)
{
  int32_t s = 0;
#pragma unroll
  for (int d = 0; d < 3; d++) {
    const int32_t a = box[0][3 * (size_t)b + d] - p[d];
    const int32_t c = p[d] - box[1][3 * (size_t)b + d];
    const int32_t m = a > 0 ? a : 0;
    s += m > c ? m : c;
  }
  return s;
}

// the bucketed window scans of :1436-1522 / :1551-1604 over a list whose
// biased positions are bpos[list[k]]; candidates are reported as `k` (list
// position) or list[k] (packed index)
template<int TAG = 0>
__device__ __forceinline__ void
window_scan(
  NnState& s, const int32_t* const box[3][2], const int32_t* __restrict__ bpos,
  const int32_t* __restrict__ list, const int32_t* bp, int k0, int k1, int dir,
  bool distribution, bool check, bool report_packed)
{
  if (k0 > k1)
    return;
  if (dir > 0) {
    for (int b2 = k0 >> 15; b2 <= (k1 >> 15); b2++) {
      if (s.idx[2] != -1 && box_dist1(box[2], b2, bp) >= s.dist[2])
        continue;
      const int s1 = max(k0 >> 10, b2 << 5), e1 = min(k1 >> 10, (b2 << 5) + 31);
      for (int b1 = s1; b1 <= e1; b1++) {
        if (s.idx[2] != -1 && box_dist1(box[1], b1, bp) >= s.dist[2])
          continue;
        const int s0 = max(k0 >> 5, b1 << 5), e0 = min(k1 >> 5, (b1 << 5) + 31);
        for (int b0 = s0; b0 <= e0; b0++) {
          if (s.idx[2] != -1 && box_dist1(box[0], b0, bp) >= s.dist[2])
            continue;
          const int h0 = max(k0, b0 << 5), h1 = min(k1, (b0 << 5) + 31);
          // candidates in chunks of eight: indices, then positions, fetched
          // together (two round trips per chunk instead of two per
          // candidate); the visiting order is unchanged
          for (int kb = h0; kb <= h1; kb += 8) {
            int pk8[8];
            int32_t d8[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
              pk8[u] = list[min(kb + u, h1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
              d8[u] = norm1_i3(bp, &bpos[3 * (size_t)pk8[u]]);
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (kb + u <= h1)
                nn_visit(s, distribution, check, d8[u], (report_packed ? pk8[u] : kb + u) | TAG);
          }
        }
      }
    }
  } else {
    for (int c2 = k1 >> 15; c2 >= (k0 >> 15); c2--) {
      if (s.idx[2] != -1 && box_dist1(box[2], c2, bp) >= s.dist[2])
        continue;
      const int s1 = max(k0 >> 10, c2 << 5), e1 = min(k1 >> 10, (c2 << 5) + 31);
      for (int c1 = e1; c1 >= s1; c1--) {
        if (s.idx[2] != -1 && box_dist1(box[1], c1, bp) >= s.dist[2])
          continue;
        const int s0 = max(k0 >> 5, c1 << 5), e0 = min(k1 >> 5, (c1 << 5) + 31);
        for (int c0 = e0; c0 >= s0; c0--) {
          if (s.idx[2] != -1 && box_dist1(box[0], c0, bp) >= s.dist[2])
            continue;
          const int h0 = max(k0, c0 << 5), h1 = min(k1, (c0 << 5) + 31);
          for (int kb = h1; kb >= h0; kb -= 8) {
            int pk8[8];
            int32_t d8[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
              pk8[u] = list[max(kb - u, h0)];
#pragma unroll
            for (int u = 0; u < 8; u++)
              d8[u] = norm1_i3(bp, &bpos[3 * (size_t)pk8[u]]);
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (kb - u >= h0)
                nn_visit(s, distribution, check, d8[u], (report_packed ? pk8[u] : kb - u) | TAG);
          }
        }
      }
    }
  }
}

// level-0 boxes: one thread per bucket of 32 list entries
__global__ __launch_bounds__(256) void
lod_box0_kernel(
  int n_list, const int32_t* __restrict__ list, const int32_t* __restrict__ bpos,
  int32_t* bmin, int32_t* bmax)
{
  const int nb = (n_list + 31) >> 5;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb;
       b += gridDim.x * blockDim.x) {
    int32_t mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    const int e = min(n_list, (b + 1) << 5);
    for (int i = b << 5; i < e; i++)
      for (int d = 0; d < 3; d++) {
        const int32_t v = bpos[3 * (size_t)list[i] + d];
        mn[d] = min(mn[d], v);
        mx[d] = max(mx[d], v);
      }
    for (int d = 0; d < 3; d++) {
      bmin[3 * (size_t)b + d] = mn[d];
      bmax[3 * (size_t)b + d] = mx[d];
    }
  }
}

// upper levels: merge 32 boxes
__global__ __launch_bounds__(256) void
lod_box_up_kernel(
  int n_lo, const int32_t* __restrict__ lo_min, const int32_t* __restrict__ lo_max,
  int32_t* bmin, int32_t* bmax)
{
  const int nb = (n_lo + 31) >> 5;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nb;
       b += gridDim.x * blockDim.x) {
    int32_t mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    const int e = min(n_lo, (b + 1) << 5);
    for (int i = b << 5; i < e; i++)
      for (int d = 0; d < 3; d++) {
        mn[d] = min(mn[d], lo_min[3 * (size_t)i + d]);
        mx[d] = max(mx[d], lo_max[3 * (size_t)i + d]);
      }
    for (int d = 0; d < 3; d++) {
      bmin[3 * (size_t)b + d] = mn[d];
      bmax[3 * (size_t)b + d] = mx[d];
    }
  }
}

// cell keys of the retained list (one probe of a neighbour-cell search = one load)
__global__ __launch_bounds__(256) void
lod_ret_keys_kernel(
  int n_ret, const int32_t* __restrict__ retained, const int64_t* __restrict__ code,
  int shift3, int64_t* __restrict__ ret_key)
{
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n_ret; k += gridDim.x * blockDim.x)
    ret_key[k] = code[retained[k]] >> shift3;
}

// The reference fills its atlas block by block from a cursor that only
// advances over retained entries of the block being entered (:1349-1363): a
// retained entry whose block holds no refinement point is never passed and
// from that block on the atlas stays empty.  atlas_limit = smallest block id
// of a retained entry without a refinement point in the same block.
__global__ __launch_bounds__(256) void
lod_atlas_limit_kernel(NnCtx cx, long long* atlas_limit)
{
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < cx.n_ret;
       r += gridDim.x * blockDim.x) {
    const int64_t id = cx.code[cx.retained[r]] >> cx.boundary;
    int lo = 0, hi = cx.n_ref;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((cx.code[cx.refine[mid]] >> cx.boundary) < id)
        lo = mid + 1;
      else
        hi = mid;
    }
    const bool present = lo < cx.n_ref && (cx.code[cx.refine[lo]] >> cx.boundary) == id;
    if (!present)
      atomicMin(atlas_limit, (long long)id);
  }
}

// (measured: 5 waves/SIMD 3.3 / 10.5 ms on the 1M dense / lidar clouds against
// 3.6 / 10.0 ms at 4; 6 and 8 spill and lose)
#ifndef GPCC_NN_WAVES
#define GPCC_NN_WAVES 4
#endif
template<bool SCALABLE, bool INTER = false>
__global__ __launch_bounds__(256, GPCC_NN_WAVES) void
lod_nn_search_kernel(NnCtx cx)
{
  constexpr uint8_t kNeigh[27] = {7,  3,  5,  6,  35, 21, 14, 28, 42,
                                  49, 12, 10, 17, 20, 34, 33, 4,  2,
                                  1,  56, 24, 40, 48, 32, 16, 8,  0};
  const bool distribution = cx.distribution != 0;
  const int64_t atlas_limit = *cx.atlas_limit;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cx.n_ref;
       q += gridDim.x * blockDim.x) {
    NnState s;
#pragma unroll
    for (int h = 0; h < 6; h++) {
      s.idx[h] = -1;
      s.dist[h] = INT64_MAX;
    }
    s.idx2 = 3;
    const int index = cx.refine[q];
    const int64_t code = cx.code[index];
    const int64_t atlas_id = code >> cx.boundary;
    const int64_t cell = code >> cx.shift3;
    const int32_t bp[3] = {cx.bpos[3 * (size_t)index], cx.bpos[3 * (size_t)index + 1], cx.bpos[3 * (size_t)index + 2]};

    if (cx.n_ret) {
      int lo = 0, hi = cx.n_ret;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cx.code[cx.retained[mid]] <= code)
          lo = mid + 1;
        else
          hi = mid;
      }
      const int j = min(lo, cx.n_ret - 1);
      if (atlas_id < atlas_limit) {
        const uint64_t base = morton3d_add((uint64_t)cell, ~0ull);
        // the 27 neighbour cells, nine lower_bound searches in lock step at a
        // time over the keys of the retained list (one load per probe, the
        // nine probes of a step in flight together); cells are visited in the
        // reference's order afterwards
#pragma unroll
        for (int nb0 = 0; nb0 < 27; nb0 += 9) {
          int lo9[9], hi9[9];
          int64_t want9[9];
#pragma unroll
          for (int u = 0; u < 9; u++) {
            want9[u] = (int64_t)morton3d_add(base, kNeigh[nb0 + u]);
            lo9[u] = 0;
            hi9[u] = (want9[u] >> kAtlasBits) == atlas_id ? cx.n_ret : 0;
          }
          for (;;) {
            bool active = false;
#pragma unroll
            for (int u = 0; u < 9; u++)
              active |= lo9[u] < hi9[u];
            if (!active)
              break;
            int64_t kv[9];
#pragma unroll
            for (int u = 0; u < 9; u++)
              kv[u] = lo9[u] < hi9[u] ? cx.ret_key[lo9[u] + ((hi9[u] - lo9[u]) >> 1)] : 0;
#pragma unroll
            for (int u = 0; u < 9; u++)
              if (lo9[u] < hi9[u]) {
                const int mid = lo9[u] + ((hi9[u] - lo9[u]) >> 1);
                if (kv[u] < want9[u])
                  lo9[u] = mid + 1;
                else
                  hi9[u] = mid;
              }
          }
#pragma unroll
          for (int u = 0; u < 9; u++) {
            if ((want9[u] >> kAtlasBits) != atlas_id)
              continue;
            for (int k = lo9[u]; k < cx.n_ret && cx.ret_key[k] == want9[u]; k++)
              nn_visit(s, distribution, false, norm1_i3(bp, &cx.bpos[3 * (size_t)cx.retained[k]]), k);
          }
        }
      }
      if (s.idx[2] == -1) {
        const int center = s.idx[0] == -1 ? j : s.idx[0];
        const int k0 = max(0, center - cx.range_inter);
        const int k1 = min(cx.n_ret - 1, center + cx.range_inter);
        nn_visit(s, distribution, true, norm1_i3(bp, &cx.bpos[3 * (size_t)cx.retained[center]]), center);
        for (int nn = 1; nn <= 2; nn++) {
          const int kp = center + nn;
          if (kp <= k1)
            nn_visit(s, distribution, true, norm1_i3(bp, &cx.bpos[3 * (size_t)cx.retained[kp]]), kp);
          const int kn = center - nn;
          if (kn >= k0)
            nn_visit(s, distribution, true, norm1_i3(bp, &cx.bpos[3 * (size_t)cx.retained[kn]]), kn);
        }
        const int p1 = min(cx.n_ret - 1, center + 3);
        const int p0 = max(0, center - 3);
        window_scan(s, cx.box_ret, cx.bpos, cx.retained, bp, p1, k1, +1, distribution, true, false);
        window_scan(s, cx.box_ret, cx.bpos, cx.retained, bp, k0, p0, -1, distribution, true, false);
      }
      // retained-list positions -> packed indices
#pragma unroll
      for (int h = 0; h < 6; h++)
        if ((h < 3 || distribution) && s.idx[h] != -1)
          s.idx[h] = cx.retained[s.idx[h]];
    }

    if (cx.intra) {
      const int k00 = q + 1;
      const int k01 = min(cx.n_ref - 1, k00 + 2);
      for (int k = k00; k <= k01; k++)
        nn_visit(s, distribution, false, norm1_i3(bp, &cx.bpos[3 * (size_t)cx.refine[k]]), cx.refine[k]);
      const int w0 = k01 + 1;
      const int w1 = min(cx.n_ref - 1, k00 + cx.range_intra);
      window_scan(s, cx.box_ref, cx.bpos, cx.refine, bp, w0, w1, +1, distribution, false, true);
    }

    if (INTER) {
      // candidates of the reference frame, no duplicate tests (:1606-1796).
      // (a) The inter-frame atlas.  Its block test shifts a neighbour CELL by the intra
      // atlas' 21 bits (:1627) although the block id was formed with 9: the two agree
      // in block 0 only, so the atlas answers for cells of the first 8^3 block alone,
      // and a neighbour cell outside that block aliases into it (the address is
      // masked, :155-158).
      if ((code >> cx.frame_boundary) == 0) {
        const uint64_t base = morton3d_add((uint64_t)cell, ~0ull);
        for (int nb0 = 0; nb0 < 27; nb0++) {
          const int64_t nb = (int64_t)morton3d_add(base, kNeigh[nb0]);
          if ((nb >> kAtlasBits) != 0)
            continue;
          const int64_t want = nb & 0x1ff;
          int lo = 0, hi = cx.n_frame;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((cx.frame_code[mid] >> cx.shift3) < want)
              lo = mid + 1;
            else
              hi = mid;
          }
          for (int k = lo; k < cx.n_frame && (cx.frame_code[k] >> cx.shift3) == want; k++)
            nn_visit(s, distribution, false, norm1_i3(bp, &cx.frame_bpos[3 * (size_t)k]), k | kFrameTag);
        }
      }
      // (b) the window around the first entry that does not precede the point; the
      // left part is walked upwards too
      if (cx.n_frame > 0) {
        int lo = 0, hi = cx.n_frame;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (cx.frame_code[mid] < code)
            lo = mid + 1;
          else
            hi = mid;
        }
        const int jr = min(lo, cx.n_frame - 1);
        const int k1 = min(cx.n_frame - 1, max(0, jr + cx.frame_range));
        window_scan<kFrameTag>(
          s, cx.box_frame, cx.frame_bpos, cx.frame_identity, bp, jr, k1, +1, distribution, false, false);
        const int l0 = min(cx.n_frame - 1, max(0, jr - 1));
        const int l1 = min(cx.n_frame - 1, max(0, l0 - cx.frame_range));
        window_scan<kFrameTag>(
          s, cx.box_frame, cx.frame_bpos, cx.frame_identity, bp, l1, l0, +1, distribution, false, false);
      }
    }
    // biased position of a candidate: the reference frame's for a tagged one
    auto pos_of = [&](int32_t v) -> const int32_t* {
      return INTER && (v & kFrameTag) ? &cx.frame_bpos[3 * (size_t)(v & ~kFrameTag)] : &cx.bpos[3 * (size_t)v];
    };

    int count = (s.idx[0] != -1) + (s.idx[1] != -1) + (s.idx[2] != -1);
    count = min(cx.max_neigh, count);
    if (distribution) {
      const int c1 = 3 + (s.idx[3] != -1) + (s.idx[4] != -1) + (s.idx[5] != -1);
#pragma unroll
      for (int m = 3; m < 6; m++)
        if (m < c1 && s.dist[m] == INT64_MAX)
          s.dist[m] = norm1_i3(bp, pos_of(s.idx[m]));
#pragma unroll
      for (int m = 3; m < 6; m++)
#pragma unroll
        for (int l = m + 1; l < 6; l++)
          if (l < c1 && s.dist[l] < s.dist[m]) {
            const int32_t ti = s.idx[l];
            s.idx[l] = s.idx[m];
            s.idx[m] = ti;
            const int64_t td = s.dist[l];
            s.dist[l] = s.dist[m];
            s.dist[m] = td;
          }
      if (count >= 3) {
        // every index below is a compile-time constant after unrolling, so
        // the candidate state stays in registers
        constexpr int8_t loose[8][3] = {{3, 5, 6}, {2, 4, 7}, {1, 4, 7}, {0, 5, 6},
                                        {1, 2, 7}, {0, 3, 6}, {0, 3, 5}, {1, 2, 4}};
        int numend = 3;
        bool open_end = true;
#pragma unroll
        for (int m = 3; m < 6; m++) {
          open_end = open_end && m < c1 && (s.dist[m] << 5) < s.dist[2] * 54;
          numend += open_end;
        }
        int dir[6];
#pragma unroll
        for (int h = 0; h < 6; h++) {
          dir[h] = -1;
          if (h < numend) {
            const int32_t* o = pos_of(s.idx[h]);
            dir[h] = ((o[0] - bp[0] >= 0) << 2) + ((o[1] - bp[1] >= 0) << 1) + (o[2] - bp[2] >= 0);
          }
        }
        bool replace = true;
        int32_t rep = -1;
        if (dir[1] == 7 - dir[0] || dir[2] == 7 - dir[0] || dir[2] == 7 - dir[1])
          replace = false;
#pragma unroll
        for (int h = 3; h < 6; h++)
          if (replace && h < numend && (dir[h] == 7 - dir[0] || dir[h] == 7 - dir[1])) {
            replace = false;
            rep = s.idx[h];
          }
        const bool e01 = dir[0] == dir[1], e02 = dir[0] == dir[2], e12 = dir[1] == dir[2];
        int l0 = 0, l1 = 0, l2 = 0;
#pragma unroll
        for (int v = 0; v < 8; v++)
          if (dir[0] == v) {
            l0 = loose[v][0];
            l1 = loose[v][1];
            l2 = loose[v][2];
          }
        // which of the three candidate filters applies (:1861-1899)
        int filter = 0;  // 1: in loose set, 2: differs from dir[0] and dir[1]
        if ((e02 || e12) && e01)
          filter = 1;
        else if ((e02 || e12) && !e01)
          filter = (dir[1] == l0 || dir[1] == l1 || dir[1] == l2) ? 0 : 2;
        else if (e01)
          filter = (dir[2] == l0 || dir[2] == l1 || dir[2] == l2) ? 0 : 1;
#pragma unroll
        for (int h = 3; h < 6; h++) {
          const bool hit = filter == 1 ? (dir[h] == l0 || dir[h] == l1 || dir[h] == l2)
                                       : (dir[h] != dir[0] && dir[h] != dir[1]);
          if (replace && filter != 0 && h < numend && hit) {
            replace = false;
            rep = s.idx[h];
          }
        }
        if (rep >= 0)
          s.idx[2] = rep;
      }
    }
    int32_t pp[3] = {0, 0, 0};
    uint64_t pw[3] = {0, 0, 0};
    for (int h = 0; h < count; h++) {
      pp[h] = INTER && (s.idx[h] & kFrameTag) ? (cx.frame_order[s.idx[h] & ~kFrameTag] | kFrameTag) : cx.order[s.idx[h]];
      pw[h] = (uint64_t)norm2_i3(pos_of(s.idx[h]), bp);
    }
    if (SCALABLE) {
      bool cut = false;
#pragma unroll
      for (int h = 1; h < 3; h++) {
        if (h >= count || cut)
          continue;
        int64_t d2 = (int64_t)pw[h];
        if (!cx.unit_bias) {
          d2 = 0;
#pragma unroll
          for (int d = 0; d < 3; d++) {
            const int64_t a = (int64_t)(int32_t)((uint32_t)cx.pos[3 * (size_t)index + d] & cx.node_mask)
              - (int64_t)(int32_t)((uint32_t)cx.pos[3 * (size_t)s.idx[h] + d] & cx.node_mask);
            d2 += a * a;
          }
        }
        if (d2 > cx.prune_dist) {
          count = h;
          cut = true;
        }
      }
    }
    if (count > 1) {
      if (pw[0] > pw[1]) {
        const int32_t ti = pp[0]; pp[0] = pp[1]; pp[1] = ti;
        const uint64_t tw = pw[0]; pw[0] = pw[1]; pw[1] = tw;
      }
      if (count == 3 && pw[1] > pw[2]) {
        int32_t ti = pp[1]; pp[1] = pp[2]; pp[2] = ti;
        uint64_t tw = pw[1]; pw[1] = pw[2]; pw[2] = tw;
        if (pw[0] > pw[1]) {
          ti = pp[0]; pp[0] = pp[1]; pp[1] = ti;
          tw = pw[0]; pw[0] = pw[1]; pw[1] = tw;
        }
      }
    }
    const int pred = cx.n - 1 - (cx.start + q);
    const int point = cx.order[index];
    cx.pred_count[pred] = count;
    for (int h = 0; h < 3; h++) {
      cx.pred_point[3 * (size_t)pred + h] = pp[h];
      cx.pred_dist2[3 * (size_t)pred + h] = pw[h];
    }
    cx.pt2pred[point] = pred;
    cx.indexes[pred] = point;
  }
}

// updatePredictors (:2273-2296) + optional computeWeights
__global__ __launch_bounds__(256) void
lod_finalise_kernel(
  int n, int raw, int32_t* count, const int32_t* __restrict__ pred_point,
  const int32_t* __restrict__ pt2pred, uint64_t* dist2, int32_t* neigh_index)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    int c = count[i];
    uint64_t w0 = dist2[3 * (size_t)i];
    if (c < 2) {
      w0 = 1;
    } else if (w0 == 0) {
      c = 1;
      w0 = 1;
    }
    dist2[3 * (size_t)i] = w0;
    count[i] = c;
    for (int k = 0; k < 3; k++) {
      const int p = pred_point[3 * (size_t)i + k];
      neigh_index[3 * (size_t)i + k] = k < c ? pt2pred[p] : p;
    }
  }
}

// ... with attribute inter prediction: a neighbour in the reference frame (tagged in
// pred_point) keeps its point index there, is flagged in inter_ref [n][3] and moves away
// by the frame distance (updatePredictors :2286-2293)
__global__ __launch_bounds__(256) void
lod_finalise_inter_kernel(
  int n, int32_t* count, const int32_t* __restrict__ pred_point,
  const int32_t* __restrict__ pt2pred, uint64_t* dist2, int32_t* neigh_index, int32_t* inter_ref,
  int frame_distance)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    int c = count[i];
    uint64_t w0 = dist2[3 * (size_t)i];
    if (c < 2) {
      w0 = 1;
    } else if (w0 == 0) {
      c = 1;
      w0 = 1;
    }
    dist2[3 * (size_t)i] = w0;
    count[i] = c;
    for (int k = 0; k < 3; k++) {
      const int p = pred_point[3 * (size_t)i + k];
      const bool frame = (p & kFrameTag) != 0;
      const int pi = p & ~kFrameTag;
      neigh_index[3 * (size_t)i + k] = k < c && !frame ? pt2pred[pi] : pi;
      inter_ref[3 * (size_t)i + k] = frame;
      if (k < c && frame)
        dist2[3 * (size_t)i + k] += (uint64_t)(int64_t)frame_distance;
    }
  }
}

// blendWeights with neighbours in the reference frame: their positions are that
// frame's (:654-656)
__global__ __launch_bounds__(256) void
lod_blend_weights_inter_kernel(
  int n, const int32_t* __restrict__ neigh_count, const int32_t* __restrict__ neigh_point,
  const int32_t* __restrict__ xyz, const int32_t* __restrict__ xyz_frame, int32_t* weight)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x) {
    if (neigh_count[i] != 3)
      continue;
    const int32_t* q[3];
    for (int k = 0; k < 3; k++) {
      const int p = neigh_point[3 * (size_t)i + k];
      q[k] = p & kFrameTag ? &xyz_frame[3 * (size_t)(p & ~kFrameTag)] : &xyz[3 * (size_t)p];
    }
    int64_t d01 = 0, d02 = 0, d12 = 0;
    for (int c = 0; c < 3; c++) {
      const int64_t a = (int64_t)q[0][c] - q[1][c], b = (int64_t)q[0][c] - q[2][c], e = (int64_t)q[1][c] - q[2][c];
      d01 += a * a;
      d02 += b * b;
      d12 += e * e;
    }
    constexpr int dd = 10, bb = 1, cc = 5;
    const int b1 = d01 <= d02 ? bb : cc;
    const int b2 = d01 <= d12 ? cc : bb;
    const int b3 = d02 <= d12 ? bb : cc;
    const int w0 = weight[3 * (size_t)i], w1 = weight[3 * (size_t)i + 1], w2 = weight[3 * (size_t)i + 2];
    const int v0 = (w0 * dd + w1 * (16 - dd - b2) + w2 * b3) >> 4;
    const int v1 = (w0 * b1 + w1 * dd + w2 * (16 - dd - b3)) >> 4;
    weight[3 * (size_t)i] = v0;
    weight[3 * (size_t)i + 1] = v1;
    weight[3 * (size_t)i + 2] = 256 - v0 - v1;
  }
}

// estimateDist2 (tmc3/AttributeEncoder.cpp:1698-1709): one wavefront per
// sampled point, lanes stride the +-range window
__global__ __launch_bounds__(256) void
estimate_dist2_kernel(
  int n, const int32_t* __restrict__ xyz, int period, int range, int num_samples,
  long long* __restrict__ dists)
{
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int smp = wave; smp < num_samples; smp += nwaves) {
    const int index = smp * period;
    const int k0 = max(0, index - range), k1 = min(n - 1, index + range);
    const int32_t px = xyz[3 * (size_t)index], py = xyz[3 * (size_t)index + 1], pz = xyz[3 * (size_t)index + 2];
    long long best = INT64_MAX;
    for (int k = k0 + lane; k <= k1; k += 64) {
      if (k == index)
        continue;
      const long long dx = (long long)px - xyz[3 * (size_t)k], dy = (long long)py - xyz[3 * (size_t)k + 1],
                      dz = (long long)pz - xyz[3 * (size_t)k + 2];
      const long long d = dx * dx + dy * dy + dz * dz;
      best = d < best ? d : best;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const long long o = __shfl_xor(best, d);
      best = o < best ? o : best;
    }
    if (lane == 0)
      dists[smp] = best;
  }
}

// ---- zero-run formation of a coefficient stream ------------------------------
// (the non-arithmetic part of the entropy loops, AttributeEncoder.cpp:1279-1291
// / 1347-1362 RAHT, :1458-1474 / 1617-1633 lifting): a position whose c values
// are all zero extends the run, any other emits (run, values).  Flags ->
// stable compaction of the non-zero positions (lod_partition_kernel) -> runs
// are differences of consecutive positions.
__global__ __launch_bounds__(256) void
zero_run_flags_kernel(
  int n, int c, int planar, const int32_t* __restrict__ coeffs, uint8_t* flags, int32_t* positions)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    bool any = false;
    for (int d = 0; d < c; d++)
      any |= (planar ? coeffs[(size_t)n * d + i] : coeffs[(size_t)i * c + d]) != 0;
    flags[i] = any;
    positions[i] = i;
  }
}

__global__ __launch_bounds__(256) void
zero_run_emit_kernel(
  int n, int c, int planar, const int32_t* __restrict__ coeffs,
  const int32_t* __restrict__ nzpos, const int32_t* __restrict__ count, int32_t* runs,
  int32_t* values, int32_t* trailing)
{
  const int m = count[0];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
    const int pos = nzpos[k];
    runs[k] = pos - (k ? nzpos[k - 1] : -1) - 1;
    for (int d = 0; d < c; d++)
      values[(size_t)k * c + d] = planar ? coeffs[(size_t)n * d + pos] : coeffs[(size_t)pos * c + d];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
    trailing[0] = m ? n - 1 - nzpos[m - 1] : n;
}

}  // namespace gpcc
