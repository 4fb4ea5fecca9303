// recolour_kdtree.hpp -- the k-d tree of pcc::recolour on the device.
//
// The reference searches two nanoflann trees (tmc3/pointset_processing.cpp:269-272,
// dependencies/nanoflann/nanoflann.hpp: KDTreeSingleIndexAdaptor, leaf size 10), and wherever
// candidates are equidistant -- on a voxelised cloud at a dyadic scale: nearly everywhere -- its
// result is the order in which THAT tree hands them over.  So the tree itself is rebuilt here,
// index permutation included, and searched in the reference's order:
//
//   build   divideTree :872 / middleSplit_ :922 / planeSplit :972, LEVEL BY LEVEL: all nodes of a
//           level are split by the same few launches over the index array.
//             kd_minmax    tight bounds of every node's points (atomics; they are also what the
//                          parent's divlow / divhigh become, :910-911)
//             kd_split     per node: leaf (<= 10 points) or cut dimension and cut value
//             kd_count     lim1 = #(v < cut), lim2 = #(v <= cut) per node
//             kd_flag / scan / kd_scatter / kd_gather, twice
//                          planeSplit's two two-pointer loops as rank arithmetic: the i-th index
//                          of the left zone that does not belong there changes places with the
//                          i-th index of the right zone (counted from the END) that belongs left
//             kd_children  split index (:958-960), the two children with the box cut at the plane
//             kd_assign    node of every position for the next level
//   search  findNeighbors :1200 / searchLevel :1308 / KNNResultSet::addPoint :175 as a walk with
//           an explicit stack, one query per thread: nearer child first, the other one iff
//           mindistsq <= worst distance, candidates of equal distance in visiting order.
//
// Doubles in the reference's order, no contraction.  The same source runs under the CPU
// wavefront emulator (tests/emu) against oracle/recolour_oracle.c.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

namespace gpcc {

constexpr int kKdLeaf = 10;       // KDTreeVectorOfVectorsAdaptor(3, cloud, 10)
constexpr int kKdMaxDepth = 64;   // deeper trees are declined (the search stack lives in scratch)
constexpr int kKdScanBlock = 2048;
// A node of at most this many points leaves the level-by-level build: ONE WAVEFRONT finishes its
// whole subtree in LDS (kd_subtree_kernel) -- the lower two thirds of a tree's levels in one launch.
// (2 048: 53 KB of LDS per wavefront, three wavefronts per CU, 3.1 ms per launch for a 1 M-point cloud --
// a subtree is a chain of dependent LDS accesses, so what counts is how many run side by side.)
constexpr int kKdSub = 1024;

// what the search reads: 32 bytes per node
struct KdNode {
  double divlow, divhigh;
  int32_t a, b;   // leaf: its range [a, b) of vind; internal: a = the first child, a + 1 the other
  int32_t feat;   // cut dimension, -1 = leaf
  int32_t pad;
};

struct KdTree {
  const int32_t* xyz;  // [n][3]
  int32_t n;
  int32_t* vind;       // [n] nanoflann's index permutation
  KdNode* nodes;       // [kd_node_capacity(n)]
  double root_lo[3], root_hi[3];
};

// A tree of n points has at most 2n - 1 nodes.  The subtrees take their node ids in one block each (2 per
// point, one atomic per subtree instead of one per split -- the counter is a single address every
// wavefront of the launch would queue on), so the ids have gaps: the levels above the subtrees hold
// fewer than n / 8 nodes as long as the depth stays within kKdMaxDepth (each has more than kKdSub points).
inline size_t
kd_node_capacity(size_t n)
{
  return 2 * n + n / 4 + 64;
}

// the build's working set (the arrays of the level loop hold 2n + 2 nodes; node ids in order of creation)
struct KdBuild {
  KdTree t;
  int32_t node_cap;  // entries of t.nodes
  int32_t* pnode;    // [n] node of every POSITION of vind at the level being split, -1 = in a leaf
  int32_t* rng;      // [nodes][2] left, right
  int32_t* parent;   // [nodes]
  double* box;       // [nodes][6] the box handed down: lo[3], hi[3]
  int32_t* mm;       // [nodes][6] tight min[3], max[3]
  double* cut;       // [nodes]
  int32_t* lim;      // [nodes][2] lim1, lim2
  int32_t* split;    // [nodes] absolute position where the right child starts
  int32_t* flag;     // [n + 1] scanned: flag[x] = misplaced positions before x
  int32_t* tmp_l;    // [n]
  int32_t* tmp_r;    // [n]
  long long* sums;   // [n / kKdScanBlock + 2]
  int32_t* counters; // [0] nodes created, [1] internal nodes of the level just split, [2] subtree roots,
                     // [3] the deepest level a subtree reached
  int32_t* sub_list; // [2][n / 11 + 2] roots of the subtrees kd_subtree_kernel finishes, then their levels
};

__device__ __forceinline__ int
kd_lane()
{
  return (int)(threadIdx.x & 63);
}

__device__ __forceinline__ int
kd_wave_min(int x)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_xor(x, d);
    x = o < x ? o : x;
  }
  return x;
}
__device__ __forceinline__ int
kd_wave_max(int x)
{
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_xor(x, d);
    x = o > x ? o : x;
  }
  return x;
}

// ---- scan (inclusive, in place; the arrays carry a leading zero) -----------------------
__global__ __launch_bounds__(256) void
kd_scan_sums_kernel(const int32_t* __restrict__ a, size_t n, long long* __restrict__ sums)
{
  __shared__ long long w[4];
  const size_t base = (size_t)blockIdx.x * kKdScanBlock;
  long long s = 0;
  for (int k = 0; k < kKdScanBlock / 256; k++) {
    const size_t i = base + (size_t)k * 256 + threadIdx.x;
    s += i < n ? a[i] : 0;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    s += __shfl_xor(s, d);
  if ((threadIdx.x & 63) == 0)
    w[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
    sums[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// (exclusive scan of the block sums, one workgroup: a thread sums its share, the shares are scanned inside the
// wavefronts by shuffles and across the sixteen wavefronts through LDS.  Until round 5 thread 0 walked the 1024
// shares through LDS alone: 84 us per scan, ~120 scans per recolour call -- more than the rest of the tree build)
__global__ __launch_bounds__(1024) void
kd_scan_blocks_kernel(long long* sums, int nblocks)
{
  __shared__ long long wtot[16];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  long long s = 0;
  for (int b = b0; b < b1; b++)
    s += sums[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  long long inc = s;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const long long o = __shfl_up(inc, d);
    if (lane >= d)
      inc += o;
  }
  if (lane == 63)
    wtot[wave] = inc;
  __syncthreads();
  long long run = inc - s;
  for (int w = 0; w < wave; w++)
    run += wtot[w];
  for (int b = b0; b < b1; b++) {
    const long long v = sums[b];
    sums[b] = run;
    run += v;
  }
}

__global__ __launch_bounds__(256) void
kd_scan_apply_kernel(int32_t* a, size_t n, const long long* __restrict__ sums)
{
  __shared__ int wsum[4];
  const size_t base = (size_t)blockIdx.x * kKdScanBlock;
  int run = (int)sums[blockIdx.x];
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < kKdScanBlock / 256; k++) {
    const size_t i = base + (size_t)k * 256 + threadIdx.x;
    const int v = i < n ? a[i] : 0;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(inc, d);
      if (lane >= d)
        inc += o;
    }
    __syncthreads();
    if (lane == 63)
      wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    int off = run;
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++)
      off += wsum[w];
    if (i < n)
      a[i] = off + inc;
    run += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
}

inline hipError_t
kd_scan(hipStream_t st, int32_t* a, size_t n, long long* sums)
{
  const int nblk = (int)((n + kKdScanBlock - 1) / kKdScanBlock);
  hipLaunchKernelGGL(kd_scan_sums_kernel, dim3(nblk), dim3(256), 0, st, (const int32_t*)a, n, sums);
  hipLaunchKernelGGL(kd_scan_blocks_kernel, dim3(1), dim3(1024), 0, st, sums, nblk);
  hipLaunchKernelGGL(kd_scan_apply_kernel, dim3(nblk), dim3(256), 0, st, a, n, (const long long*)sums);
  return hipGetLastError();
}

// ---- build -------------------------------------------------------------------------------
// box = the cloud's tight bounds (rc_bbox_kernel: min[3], max[3])
__global__ __launch_bounds__(256) void
kd_init_kernel(KdBuild b, const int32_t* __restrict__ bbox)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < b.t.n) {
    b.t.vind[i] = i;
    b.pnode[i] = 0;
  }
  if (i == 0) {
    b.rng[0] = 0;
    b.rng[1] = b.t.n;
    b.parent[0] = -1;
    for (int a = 0; a < 3; a++) {
      b.box[a] = (double)bbox[a];
      b.box[3 + a] = (double)bbox[3 + a];
      b.mm[a] = 0x7fffffff;
      b.mm[3 + a] = -0x7fffffff;
    }
    b.counters[0] = 1;
    b.counters[1] = 0;
    b.counters[2] = 0;
    b.counters[3] = 0;
    b.flag[0] = 0;
  }
}

// kd_minmax / kd_count: a wavefront takes a CONTIGUOUS chunk of kKdChunk positions and keeps what it has
// gathered for the node it is in across its rounds -- the nodes of a level are contiguous ranges, so at
// the upper levels a wavefront issues ONE set of atomics (with a round of positions per wavefront and
// launch it was 16 000 wavefronts on the same six words: 0.4-1.3 ms per launch for a 1 M-point cloud,
// half of the build).  A round whose lanes sit in different nodes falls back to atomics per lane.
constexpr int kKdChunk = 1024;

__global__ __launch_bounds__(256) void
kd_minmax_kernel(KdBuild b)
{
  const int n = b.t.n;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = kd_lane();
  const int x0 = wave * kKdChunk;
  int run_node = -1;  // the node the running bounds belong to (wave-uniform)
  int rmn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, rmx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
  auto flush = [&]() {
    if (run_node >= 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const int mn = kd_wave_min(rmn[a]), mx = kd_wave_max(rmx[a]);
        if (lane == 0) {
          atomicMin(&b.mm[6 * run_node + a], mn);
          atomicMax(&b.mm[6 * run_node + 3 + a], mx);
        }
        rmn[a] = 0x7fffffff;
        rmx[a] = -0x7fffffff;
      }
    }
    run_node = -1;
  };
  for (int r = 0; r < kKdChunk / 64; r++) {
    const int x = x0 + r * 64 + lane;
    const int node = x < n ? b.pnode[x] : -1;
    int v[3] = {0, 0, 0};
    if (node >= 0) {
      const int p = b.t.vind[x];
#pragma unroll
      for (int a = 0; a < 3; a++)
        v[a] = b.t.xyz[3 * p + a];
    }
    const unsigned long long live = __ballot(node >= 0);
    if (!live)
      continue;
    const int first = __ffsll((long long)live) - 1;
    const int node0 = __shfl(node, first);
    if (__all(node < 0 || node == node0)) {
      if (node0 != run_node)
        flush();
      run_node = node0;
      if (node >= 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          rmn[a] = v[a] < rmn[a] ? v[a] : rmn[a];
          rmx[a] = v[a] > rmx[a] ? v[a] : rmx[a];
        }
      }
    } else {
      flush();
      if (node >= 0) {
#pragma unroll
        for (int a = 0; a < 3; a++) {
          atomicMin(&b.mm[6 * node + a], v[a]);
          atomicMax(&b.mm[6 * node + 3 + a], v[a]);
        }
      }
    }
  }
  flush();
}

// nodes [nb, ne) of the level: the parent's divlow / divhigh (:910-911: the bounds the recursion
// RETURNS, i.e. the children's tight ones), leaf or the cut of middleSplit_ (:922-954)
__global__ __launch_bounds__(256) void
kd_split_kernel(KdBuild b, int nb, int ne, int level)
{
#pragma clang fp contract(off)
  const int k = nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ne)
    return;
  const int left = b.rng[2 * k], right = b.rng[2 * k + 1];
  const int32_t* mm = b.mm + 6 * k;
  const int par = b.parent[k];
  if (par >= 0) {
    const int f = b.t.nodes[par].feat;
    if (k == b.t.nodes[par].a)
      b.t.nodes[par].divlow = (double)mm[3 + f];
    else
      b.t.nodes[par].divhigh = (double)mm[f];
  }
  KdNode nd;
  nd.divlow = nd.divhigh = 0.0;
  nd.pad = 0;
  if (right - left <= kKdSub) {
    // a wavefront finishes this subtree (kd_subtree_kernel); the level loop sees a leaf
    nd.a = left;
    nd.b = right;
    nd.feat = -1;
    b.t.nodes[k] = nd;
    const int slot = atomicAdd(&b.counters[2], 1);
    b.sub_list[slot] = k;
    b.sub_list[b.t.n / 11 + 2 + slot] = level;
    return;
  }
  const double* bx = b.box + 6 * k;
  const double EPS = 0.00001;
  double max_span = bx[3] - bx[0];
#pragma unroll
  for (int a = 1; a < 3; a++) {
    const double span = bx[3 + a] - bx[a];
    max_span = span > max_span ? span : max_span;
  }
  double max_spread = -1;
  int feat = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const double span = bx[3 + a] - bx[a];
    if (span >= (1 - EPS) * max_span) {
      const double spread = (double)mm[3 + a] - (double)mm[a];
      if (spread > max_spread) {
        feat = a;
        max_spread = spread;
      }
    }
  }
  double lo_f = bx[0], hi_f = bx[3];
  int mn_f = mm[0], mx_f = mm[3];
#pragma unroll
  for (int a = 1; a < 3; a++) {
    if (feat == a) {
      lo_f = bx[a];
      hi_f = bx[3 + a];
      mn_f = mm[a];
      mx_f = mm[3 + a];
    }
  }
  const double split = (lo_f + hi_f) / 2;
  const double cut = split < (double)mn_f ? (double)mn_f : (split > (double)mx_f ? (double)mx_f : split);
  nd.a = -1;  // (its children: kd_children)
  nd.b = 0;
  nd.feat = feat;
  b.t.nodes[k] = nd;
  b.cut[k] = cut;
  b.lim[2 * k] = 0;
  b.lim[2 * k + 1] = 0;
}

__device__ __forceinline__ double
kd_coord(const KdBuild& b, int x, int feat)
{
  return (double)b.t.xyz[3 * b.t.vind[x] + feat];
}

__global__ __launch_bounds__(256) void
kd_count_kernel(KdBuild b)
{
  const int n = b.t.n;
  const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  const int lane = kd_lane();
  const int x0 = wave * kKdChunk;
  int run_node = -1, c1 = 0, c2 = 0;  // (wave-uniform)
  auto flush = [&]() {
    if (run_node >= 0 && lane == 0) {
      atomicAdd(&b.lim[2 * run_node], c1);
      atomicAdd(&b.lim[2 * run_node + 1], c2);
    }
    run_node = -1;
    c1 = c2 = 0;
  };
  for (int r = 0; r < kKdChunk / 64; r++) {
    const int x = x0 + r * 64 + lane;
    int node = x < n ? b.pnode[x] : -1;
    int feat = -1;
    if (node >= 0)
      feat = b.t.nodes[node].feat;
    if (feat < 0)
      node = -1;
    bool lt = false, le = false;
    if (node >= 0) {
      const double v = kd_coord(b, x, feat), cut = b.cut[node];
      lt = v < cut;
      le = v <= cut;
    }
    const unsigned long long live = __ballot(node >= 0);
    if (!live)
      continue;
    const int first = __ffsll((long long)live) - 1;
    const int node0 = __shfl(node, first);
    const bool uniform = __all(node < 0 || node == node0);
    const unsigned long long blt = __ballot(lt), ble = __ballot(le);
    if (uniform) {
      if (node0 != run_node)
        flush();
      run_node = node0;
      c1 += __popcll(blt);
      c2 += __popcll(ble);
    } else {
      flush();
      if (node >= 0) {
        if (lt)
          atomicAdd(&b.lim[2 * node], 1);
        if (le)
          atomicAdd(&b.lim[2 * node + 1], 1);
      }
    }
  }
  flush();
}

// One of planeSplit's two loops over a node's range [lo, right): the zone [lo, zb) is where the
// indices that "go left" end up.  PASS 0: lo = left, zb = left + lim1, goes left = v < cut;
// PASS 1: lo = left + lim1, zb = left + lim2, goes left = v <= cut.
struct KdZone {
  int lo, zb, right;
  bool goes_left, in_range;
};

template<int PASS>
__device__ __forceinline__ KdZone
kd_zone(const KdBuild& b, int x)
{
  KdZone z;
  z.lo = z.zb = z.right = 0;
  z.goes_left = z.in_range = false;
  const int node = x < b.t.n ? b.pnode[x] : -1;
  if (node < 0)
    return z;
  const int feat = b.t.nodes[node].feat;
  if (feat < 0)
    return z;
  const int left = b.rng[2 * node];
  z.right = b.rng[2 * node + 1];
  const int lim1 = b.lim[2 * node], lim2 = b.lim[2 * node + 1];
  const double v = kd_coord(b, x, feat), cut = b.cut[node];
  if (PASS == 0) {
    z.lo = left;
    z.zb = left + lim1;
    z.goes_left = v < cut;
  } else {
    z.lo = left + lim1;
    z.zb = left + lim2;
    z.goes_left = v <= cut;
  }
  z.in_range = x >= z.lo;
  return z;
}

template<int PASS>
__global__ __launch_bounds__(256) void
kd_flag_kernel(KdBuild b)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= b.t.n)
    return;
  const KdZone z = kd_zone<PASS>(b, x);
  b.flag[x + 1] = (z.in_range && ((x < z.zb) != z.goes_left)) ? 1 : 0;
}

// a misplaced index leaves its value where its partner finds it: pair i of the node = the i-th
// misplaced position of the left zone (ascending) and the i-th of the right zone counted from the end
template<int PASS, bool GATHER>
__global__ __launch_bounds__(256) void
kd_exchange_kernel(KdBuild b)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= b.t.n)
    return;
  if (b.flag[x + 1] == b.flag[x])
    return;  // in place
  const KdZone z = kd_zone<PASS>(b, x);
  const int base = b.flag[z.lo];
  const int nl = b.flag[z.zb] - base;  // pairs of the node
  const bool in_left = x < z.zb;
  const int i = in_left ? b.flag[x] - base : nl - 1 - (b.flag[x] - b.flag[z.zb]);
  if (!GATHER) {
    (in_left ? b.tmp_l : b.tmp_r)[base + i] = b.t.vind[x];
  } else {
    b.t.vind[x] = (in_left ? b.tmp_r : b.tmp_l)[base + i];
  }
}

// the split index (:958-960) and the children (divideTree :900-908)
__global__ __launch_bounds__(256) void
kd_children_kernel(KdBuild b, int nb, int ne)
{
  const int k = nb + blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= ne)
    return;
  const int feat = b.t.nodes[k].feat;
  if (feat < 0)
    return;
  const int left = b.rng[2 * k], right = b.rng[2 * k + 1];
  const int count = right - left, half = count / 2;
  const int lim1 = b.lim[2 * k], lim2 = b.lim[2 * k + 1];
  const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
  const int c1 = atomicAdd(&b.counters[0], 2);
  atomicAdd(&b.counters[1], 1);
  b.t.nodes[k].a = c1;
  b.split[k] = left + idx;
  const double cut = b.cut[k];
#pragma unroll
  for (int c = 0; c < 2; c++) {
    const int q = c1 + c;
    b.rng[2 * q] = c ? left + idx : left;
    b.rng[2 * q + 1] = c ? right : left + idx;
    b.parent[q] = k;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      double lo = b.box[6 * k + a], hi = b.box[6 * k + 3 + a];
      if (a == feat) {
        if (c)
          lo = cut;
        else
          hi = cut;
      }
      b.box[6 * q + a] = lo;
      b.box[6 * q + 3 + a] = hi;
      b.mm[6 * q + a] = 0x7fffffff;
      b.mm[6 * q + 3 + a] = -0x7fffffff;
    }
  }
}

__global__ __launch_bounds__(256) void
kd_assign_kernel(KdBuild b)
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= b.t.n)
    return;
  const int node = b.pnode[x];
  if (node < 0)
    return;
  const KdNode nd = b.t.nodes[node];
  b.pnode[x] = nd.feat < 0 ? -1 : (x < b.split[node] ? nd.a : nd.a + 1);
}

// ---- a whole subtree by one wavefront ----------------------------------------------------
// The root's points (<= kKdSub) are taken into LDS under LOCAL ids: coordinates by local id, `v` the
// permutation of local ids that planeSplit shuffles.  The wavefront walks the subtree depth first
// (the right child waits on a stack); every node is the reference's middleSplit_ / planeSplit with
// the lanes over the node's range: bounds by reduction, lim1 / lim2 by ballots, a plane split as two
// lists -- the misplaced positions of the left zone ascending, those of the right zone descending --
// whose i-th entries change places.  Node ids come from a block taken from the build's counter.
struct KdSubSmem {
  int32_t v[kKdSub];
  int32_t gid[kKdSub];
  int32_t c[3][kKdSub];
  int32_t ml[kKdSub / 2 + 64], mr[kKdSub / 2 + 64];
  int32_t st_node[kKdMaxDepth + 2], st_l[kKdMaxDepth + 2], st_r[kKdMaxDepth + 2], st_lv[kKdMaxDepth + 2];
  double st_box[kKdMaxDepth + 2][6];
};

// one of planeSplit's loops on v[lo, hi): goes_left(coordinate) decides, m = entries that go left
template<class Pred>
__device__ __forceinline__ void
kd_sub_split(KdSubSmem& sm, int lo, int hi, int m, const int32_t* __restrict__ cf, Pred goes_left)
{
  const int lane = kd_lane();
  const int zb = lo + m;
  // misplaced positions of the left zone, ascending
  int nl = 0;
  for (int base = lo; base < zb; base += 64) {
    const int x = base + lane;
    const bool mis = x < zb && !goes_left(cf[sm.v[x]]);
    const unsigned long long bal = __ballot(mis);
    if (mis)
      sm.ml[nl + __popcll(bal & ((1ull << lane) - 1ull))] = x;
    nl += __popcll(bal);
  }
  // ... of the right zone, descending
  int nr = 0;
  for (int top = hi; top > zb; top -= 64) {
    const int x = top - 1 - lane;
    const bool mis = x >= zb && goes_left(cf[sm.v[x]]);
    const unsigned long long bal = __ballot(mis);
    if (mis)
      sm.mr[nr + __popcll(bal & ((1ull << lane) - 1ull))] = x;
    nr += __popcll(bal);
  }
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < nl; i += 64) {
    const int a = sm.ml[i], b = sm.mr[i];
    const int32_t t = sm.v[a];
    sm.v[a] = sm.v[b];
    sm.v[b] = t;
  }
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64) void
kd_subtree_kernel(KdBuild b, int nsub)
{
#pragma clang fp contract(off)
  __shared__ KdSubSmem sm;
  const int lane = kd_lane();
  if ((int)blockIdx.x >= nsub)
    return;
  const int root = b.sub_list[blockIdx.x];
  const int L = b.rng[2 * root], R = b.rng[2 * root + 1];
  const int cnt = R - L;
  for (int i = lane; i < cnt; i += 64) {
    const int p = b.t.vind[L + i];
    sm.v[i] = i;
    sm.gid[i] = p;
    sm.c[0][i] = b.t.xyz[3 * p];
    sm.c[1][i] = b.t.xyz[3 * p + 1];
    sm.c[2][i] = b.t.xyz[3 * p + 2];
  }
  if (lane < 6)
    sm.st_box[0][lane] = b.box[6 * root + lane];
  if (lane == 0) {
    sm.st_node[0] = root;
    sm.st_l[0] = 0;
    sm.st_r[0] = cnt;
    sm.st_lv[0] = b.sub_list[b.t.n / 11 + 2 + blockIdx.x];
  }
  __builtin_amdgcn_wave_barrier();
  // the subtree's node ids: one block, two per point
  int next_id = 0;
  if (cnt > kKdLeaf) {
    if (lane == 0)
      next_id = atomicAdd(&b.counters[0], 2 * cnt);
    next_id = __shfl(next_id, 0);
    if (next_id + 2 * cnt > b.node_cap) {
      if (lane == 0)
        atomicMax(&b.counters[3], kKdMaxDepth + 1);  // (the caller declines)
      return;
    }
  }
  int sp = 0, deepest = 0;
  while (sp >= 0) {
    const int node = sm.st_node[sp], l = sm.st_l[sp], r = sm.st_r[sp], lv = sm.st_lv[sp];
    deepest = lv > deepest ? lv : deepest;
    double bx[6];
#pragma unroll
    for (int a = 0; a < 6; a++)
      bx[a] = sm.st_box[sp][a];
    __builtin_amdgcn_wave_barrier();  // (the entry is read by every lane before it is overwritten)
    sp--;
    const int count = r - l;
    if (count <= kKdLeaf) {
      if (lane == 0) {
        KdNode nd;
        nd.divlow = nd.divhigh = 0.0;
        nd.a = L + l;
        nd.b = L + r;
        nd.feat = -1;
        nd.pad = 0;
        b.t.nodes[node] = nd;
      }
      continue;
    }
    // middleSplit_ (:922-961): tight bounds of the node's points
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
    for (int x = l + lane; x < r; x += 64) {
      const int id = sm.v[x];
#pragma unroll
      for (int a = 0; a < 3; a++) {
        const int cv = sm.c[a][id];
        mn[a] = cv < mn[a] ? cv : mn[a];
        mx[a] = cv > mx[a] ? cv : mx[a];
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
      mn[a] = kd_wave_min(mn[a]);
      mx[a] = kd_wave_max(mx[a]);
    }
    const double EPS = 0.00001;
    double max_span = bx[3] - bx[0];
#pragma unroll
    for (int a = 1; a < 3; a++) {
      const double span = bx[3 + a] - bx[a];
      max_span = span > max_span ? span : max_span;
    }
    double max_spread = -1;
    int feat = 0;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      const double span = bx[3 + a] - bx[a];
      if (span >= (1 - EPS) * max_span) {
        const double spread = (double)mx[a] - (double)mn[a];
        if (spread > max_spread) {
          feat = a;
          max_spread = spread;
        }
      }
    }
    const double lo_f = feat == 0 ? bx[0] : (feat == 1 ? bx[1] : bx[2]);
    const double hi_f = feat == 0 ? bx[3] : (feat == 1 ? bx[4] : bx[5]);
    const int mn_f = feat == 0 ? mn[0] : (feat == 1 ? mn[1] : mn[2]);
    const int mx_f = feat == 0 ? mx[0] : (feat == 1 ? mx[1] : mx[2]);
    const double split = (lo_f + hi_f) / 2;
    const double cut = split < (double)mn_f ? (double)mn_f : (split > (double)mx_f ? (double)mx_f : split);
    const int32_t* __restrict__ cf = sm.c[feat];
    int lim1 = 0, lim2 = 0;
    for (int base = l; base < r; base += 64) {
      const int x = base + lane;
      const double cv = x < r ? (double)cf[sm.v[x]] : 0.0;
      lim1 += __popcll(__ballot(x < r && cv < cut));
      lim2 += __popcll(__ballot(x < r && cv <= cut));
    }
    // planeSplit (:972-998): < cut to the front, then == cut in front of > cut
    kd_sub_split(sm, l, r, lim1, cf, [cut](int32_t cv) { return (double)cv < cut; });
    kd_sub_split(sm, l + lim1, r, lim2 - lim1, cf, [cut](int32_t cv) { return (double)cv <= cut; });
    const int half = count / 2;
    const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    // divlow / divhigh (:910-911): what the recursion returns -- the children's tight bounds along feat
    int lmax = -0x7fffffff, rmin = 0x7fffffff;
    for (int x = l + lane; x < r; x += 64) {
      const int cv = cf[sm.v[x]];
      if (x < l + idx)
        lmax = cv > lmax ? cv : lmax;
      else
        rmin = cv < rmin ? cv : rmin;
    }
    lmax = kd_wave_max(lmax);
    rmin = kd_wave_min(rmin);
    const int c1 = next_id;
    next_id += 2;
    if (lane == 0) {
      KdNode nd;
      nd.divlow = (double)lmax;
      nd.divhigh = (double)rmin;
      nd.a = c1;
      nd.b = 0;
      nd.feat = feat;
      nd.pad = 0;
      b.t.nodes[node] = nd;
      // the right child waits, the left one is next (divideTree :900-908: the box cut at the plane)
      sm.st_node[sp + 1] = c1 + 1;
      sm.st_l[sp + 1] = l + idx;
      sm.st_r[sp + 1] = r;
      sm.st_lv[sp + 1] = lv + 1;
      sm.st_node[sp + 2] = c1;
      sm.st_l[sp + 2] = l;
      sm.st_r[sp + 2] = l + idx;
      sm.st_lv[sp + 2] = lv + 1;
    }
    if (lane < 6) {
      // lane a < 3: lo[a], lane 3 + a: hi[a] of the two children's boxes
      const bool is_lo = lane < 3;
      const int a = is_lo ? lane : lane - 3;
      double vr = lane == 0 ? bx[0] : lane == 1 ? bx[1] : lane == 2 ? bx[2] : lane == 3 ? bx[3] : lane == 4 ? bx[4] : bx[5];
      double vl = vr;
      if (a == feat) {
        if (is_lo)
          vr = cut;  // right child: lo[feat] = cut
        else
          vl = cut;  // left child: hi[feat] = cut
      }
      sm.st_box[sp + 1][lane] = vr;
      sm.st_box[sp + 2][lane] = vl;
    }
    __builtin_amdgcn_wave_barrier();
    sp += 2;
    if (sp >= kKdMaxDepth || lv + 1 > kKdMaxDepth) {
      // deeper than the search's stack allows: the caller declines
      if (lane == 0)
        atomicMax(&b.counters[3], kKdMaxDepth + 1);
      return;
    }
  }
  if (lane == 0)
    atomicMax(&b.counters[3], deepest);
  for (int i = lane; i < cnt; i += 64)
    b.t.vind[L + i] = sm.gid[sm.v[i]];
}

// The level loop as a stepper, so that the builds of the two trees of a recolour call advance together
// on two streams (each level is a handful of small launches: one build alone leaves the device idle).
// `bbox` = min[3], max[3] of the cloud on the device; `h` = four ints of PINNED host memory through which
// the host learns the number of nodes after every level.
struct KdLevelLoop {
  KdBuild b{};
  hipStream_t st = nullptr;
  volatile int32_t* h = nullptr;
  int nb = 0, ne = 1, depth = 0, phase = 0;  // phase 0: levels, 1: subtrees launched, 2: done

  hipError_t begin(const KdBuild& build, const int32_t* bbox, hipStream_t stream, int32_t* pinned4)
  {
    b = build;
    st = stream;
    h = pinned4;
    nb = 0;
    ne = 1;
    depth = 0;
    phase = 0;
    hipLaunchKernelGGL(kd_init_kernel, dim3((b.t.n + 255) / 256), dim3(256), 0, st, b, bbox);
    return hipGetLastError();
  }
  bool done() const { return phase == 2; }

  // enqueue the next piece of work and the copy of the counters behind it
  hipError_t launch()
  {
    const int n = b.t.n;
    const int pgrid = (n + 255) / 256;
    const int sgrid = ((n + kKdChunk - 1) / kKdChunk + 3) / 4;  // one wavefront per chunk of positions
    if (phase == 0) {
      depth++;
      const int ngrid = (ne - nb + 255) / 256;
      hipLaunchKernelGGL(kd_minmax_kernel, dim3(sgrid), dim3(256), 0, st, b);
      hipLaunchKernelGGL(kd_split_kernel, dim3(ngrid), dim3(256), 0, st, b, nb, ne, depth);
      if (depth <= kKdMaxDepth) {
        hipLaunchKernelGGL(kd_count_kernel, dim3(sgrid), dim3(256), 0, st, b);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_flag_kernel<0>), dim3(pgrid), dim3(256), 0, st, b);
        hipError_t e = kd_scan(st, b.flag, (size_t)n + 1, b.sums);
        if (e != hipSuccess)
          return e;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_exchange_kernel<0, false>), dim3(pgrid), dim3(256), 0, st, b);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_exchange_kernel<0, true>), dim3(pgrid), dim3(256), 0, st, b);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_flag_kernel<1>), dim3(pgrid), dim3(256), 0, st, b);
        e = kd_scan(st, b.flag, (size_t)n + 1, b.sums);
        if (e != hipSuccess)
          return e;
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_exchange_kernel<1, false>), dim3(pgrid), dim3(256), 0, st, b);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(kd_exchange_kernel<1, true>), dim3(pgrid), dim3(256), 0, st, b);
        hipLaunchKernelGGL(kd_children_kernel, dim3(ngrid), dim3(256), 0, st, b, nb, ne);
        hipLaunchKernelGGL(kd_assign_kernel, dim3(pgrid), dim3(256), 0, st, b);
      }
    } else if (phase == 1) {
      // the subtrees: one wavefront each
      hipLaunchKernelGGL(kd_subtree_kernel, dim3((int)h[2]), dim3(64), 0, st, b, (int)h[2]);
    }
    return hipMemcpyAsync((void*)h, b.counters, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
  }

  // wait for what launch() enqueued and decide what comes next
  hipError_t finish()
  {
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess)
      return e;
    if (phase == 0) {
      const int created = h[0];
      if (created == ne || depth > kKdMaxDepth) {
        // every node of the level was a leaf (or went to the subtree list)
        phase = h[2] > 0 && depth <= kKdMaxDepth ? 1 : 2;
      } else {
        nb = ne;
        ne = created;
      }
    } else if (phase == 1) {
      phase = 2;
    }
    return hipGetLastError();
  }
  // levels counted from 1 at the root: what the search's stack must hold
  int tree_depth() const { return std::max(depth, (int)h[3]); }
  int nodes() const { return (int)h[0]; }
};

// one tree alone (tests, the emulator harness)
inline hipError_t
kd_build_levels(const KdBuild& b, const int32_t* bbox, hipStream_t st, int* depth_out, int* nodes_out)
{
  int32_t h[4] = {0, 0, 0, 0};
  KdLevelLoop loop;
  hipError_t e = loop.begin(b, bbox, st, h);
  while (e == hipSuccess && !loop.done()) {
    e = loop.launch();
    if (e == hipSuccess)
      e = loop.finish();
  }
  *depth_out = loop.tree_depth();
  *nodes_out = loop.nodes();
  return e;
}

// ---- search -----------------------------------------------------------------------------
template<int K>
struct RcKnn {
  double d[K];
  int32_t i[K];
  int count;
};

// KNNResultSet::addPoint (:175-199): behind every entry that is not farther; what falls beyond
// the k-th place is dropped.  Static register indices only.
template<int K>
__device__ __forceinline__ void
rc_add(RcKnn<K>& r, int k, double d, int32_t idx)
{
  int pos = 0;
#pragma unroll
  for (int p = 0; p < K; p++)
    pos += (p < r.count && r.d[p] <= d) ? 1 : 0;
  if (pos < k) {
#pragma unroll
    for (int p = K - 1; p > 0; p--) {
      if (p > pos) {
        r.d[p] = r.d[p - 1];
        r.i[p] = r.i[p - 1];
      }
    }
#pragma unroll
    for (int p = 0; p < K; p++) {
      if (p == pos) {
        r.d[p] = d;
        r.i[p] = idx;
      }
    }
  }
  r.count = r.count < k ? r.count + 1 : k;
}

// worstDist(): the k-th distance, the largest double while fewer than k are held (init :150-157)
template<int K>
__device__ __forceinline__ double
rc_worst(const RcKnn<K>& r, int k)
{
  double v = r.d[0];
#pragma unroll
  for (int p = 1; p < K; p++)
    v = p == k - 1 ? r.d[p] : v;
  return r.count < k ? 1.7976931348623157e308 : v;
}

// findNeighbors (:1200-1215) + searchLevel (:1308-1365).  The recursion's frames: the node, the
// mindistsq it was entered with, and -- once its nearer child has returned -- the entry of
// dists[] that the second descent replaced.  phase 0 = entered, 1 = nearer child done, 2 = both.
template<int K>
__device__ __forceinline__ void
rc_kd_search(const KdTree& t, const double q[3], int k, RcKnn<K>& r)
{
#pragma clang fp contract(off)
  r.count = 0;
#pragma unroll
  for (int p = 0; p < K; p++) {
    r.d[p] = 1.7976931348623157e308;
    r.i[p] = 0;
  }
  double dists[3] = {0.0, 0.0, 0.0};
  double distsq = 0.0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (q[a] < t.root_lo[a]) {
      dists[a] = (q[a] - t.root_lo[a]) * (q[a] - t.root_lo[a]);
      distsq += dists[a];
    }
    if (q[a] > t.root_hi[a]) {
      dists[a] = (q[a] - t.root_hi[a]) * (q[a] - t.root_hi[a]);
      distsq += dists[a];
    }
  }
  int32_t st_node[kKdMaxDepth + 1];  // node << 2 | phase
  double st_mind[kKdMaxDepth + 1], st_dst[kKdMaxDepth + 1];
  int sp = 0;
  st_node[0] = 0;
  st_mind[0] = distsq;
  st_dst[0] = 0.0;
  while (sp >= 0) {
    const int32_t word = st_node[sp];
    const int phase = word & 3;
    const KdNode nd = t.nodes[word >> 2];
    if (nd.feat < 0) {
      const double worst = rc_worst(r, k);
      for (int e = nd.a; e < nd.b; e++) {
        const int32_t i = t.vind[e];
        // L2 adaptor: result += diff * diff, x then y then z
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
          const double diff = q[a] - (double)t.xyz[3 * i + a];
          s += diff * diff;
        }
        if (s < worst)
          rc_add(r, k, s, i);
      }
      sp--;
      continue;
    }
    const int f = nd.feat;
    const double val = f == 0 ? q[0] : (f == 1 ? q[1] : q[2]);
    const double diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
    const bool first_is_1 = (diff1 + diff2) < 0;
    if (phase == 0) {
      st_node[sp] = word | 1;
      st_node[sp + 1] = (first_is_1 ? nd.a : nd.a + 1) << 2;
      st_mind[sp + 1] = st_mind[sp];
      sp++;
    } else if (phase == 1) {
      const double cut_dist = first_is_1 ? (val - nd.divhigh) * (val - nd.divhigh)
                                         : (val - nd.divlow) * (val - nd.divlow);
      const double dst = f == 0 ? dists[0] : (f == 1 ? dists[1] : dists[2]);
      const double mind = st_mind[sp] + cut_dist - dst;
#pragma unroll
      for (int a = 0; a < 3; a++)
        dists[a] = f == a ? cut_dist : dists[a];
      st_dst[sp] = dst;
      st_node[sp] = (word & ~3) | 2;
      if (mind <= rc_worst(r, k)) {
        st_node[sp + 1] = (first_is_1 ? nd.a + 1 : nd.a) << 2;
        st_mind[sp + 1] = mind;
        sp++;
      }
    } else {
      const double dst = st_dst[sp];
#pragma unroll
      for (int a = 0; a < 3; a++)
        dists[a] = f == a ? dst : dists[a];
      sp--;
    }
  }
}

}  // namespace gpcc
