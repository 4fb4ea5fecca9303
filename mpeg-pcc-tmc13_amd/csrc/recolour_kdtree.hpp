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
  KdNode* nodes;       // [2n + 1]
  double root_lo[3], root_hi[3];
};

// the build's working set (capacity 2n + 1 nodes; node ids in order of creation)
struct KdBuild {
  KdTree t;
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
  int32_t* counters; // [0] nodes created, [1] internal nodes of the level just split
};

__device__ __forceinline__ int
kd_lane()
{
  return (int)(threadIdx.x & 63);
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

__global__ __launch_bounds__(1024) void
kd_scan_blocks_kernel(long long* sums, int nblocks)
{
  __shared__ long long part[1024];
  const int per = (nblocks + 1023) / 1024;
  const int b0 = threadIdx.x * per, b1 = b0 + per < nblocks ? b0 + per : nblocks;
  long long s = 0;
  for (int b = b0; b < b1; b++)
    s += sums[b];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = 0;
    for (int i = 0; i < 1024; i++) {
      const long long v = part[i];
      part[i] = run;
      run += v;
    }
  }
  __syncthreads();
  long long run = part[threadIdx.x];
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
    b.flag[0] = 0;
  }
}

// Every launch over positions runs whole wavefronts through a loop of uniform length: lanes
// beyond n carry node -1.  A wavefront whose lanes all sit in ONE node (the upper levels: always)
// reduces in registers and issues one set of atomics.
__global__ __launch_bounds__(256) void
kd_minmax_kernel(KdBuild b)
{
  const int n = b.t.n;
  const int stride = gridDim.x * blockDim.x;
  const int rounds = (n + stride - 1) / stride;
  for (int r = 0; r < rounds; r++) {
    const int x = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
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
#pragma unroll
      for (int a = 0; a < 3; a++) {
        int mn = node >= 0 ? v[a] : 0x7fffffff, mx = node >= 0 ? v[a] : -0x7fffffff;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o0 = __shfl_xor(mn, d), o1 = __shfl_xor(mx, d);
          mn = o0 < mn ? o0 : mn;
          mx = o1 > mx ? o1 : mx;
        }
        if (kd_lane() == first) {
          atomicMin(&b.mm[6 * node0 + a], mn);
          atomicMax(&b.mm[6 * node0 + 3 + a], mx);
        }
      }
    } else if (node >= 0) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        atomicMin(&b.mm[6 * node + a], v[a]);
        atomicMax(&b.mm[6 * node + 3 + a], v[a]);
      }
    }
  }
}

// nodes [nb, ne) of the level: the parent's divlow / divhigh (:910-911: the bounds the recursion
// RETURNS, i.e. the children's tight ones), leaf or the cut of middleSplit_ (:922-954)
__global__ __launch_bounds__(256) void
kd_split_kernel(KdBuild b, int nb, int ne)
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
  if (right - left <= kKdLeaf) {
    nd.a = left;
    nd.b = right;
    nd.feat = -1;
    b.t.nodes[k] = nd;
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
  const int stride = gridDim.x * blockDim.x;
  const int rounds = (n + stride - 1) / stride;
  for (int r = 0; r < rounds; r++) {
    const int x = r * stride + blockIdx.x * blockDim.x + threadIdx.x;
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
      if (kd_lane() == first) {
        atomicAdd(&b.lim[2 * node0], __popcll(blt));
        atomicAdd(&b.lim[2 * node0 + 1], __popcll(ble));
      }
    } else if (node >= 0) {
      if (lt)
        atomicAdd(&b.lim[2 * node], 1);
      if (le)
        atomicAdd(&b.lim[2 * node + 1], 1);
    }
  }
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

// The level loop.  `bbox` = min[3], max[3] of the cloud on the device; the host learns the number
// of nodes after every level (one small copy + synchronisation per level).
// Returns hipSuccess and *depth_out = levels (> kKdMaxDepth: the caller declines).
inline hipError_t
kd_build_levels(const KdBuild& b, const int32_t* bbox, hipStream_t st, int* depth_out, int* nodes_out)
{
  const int n = b.t.n;
  const int pgrid = (n + 255) / 256;
  const int sgrid = std::min(pgrid, 2048);
  hipLaunchKernelGGL(kd_init_kernel, dim3(pgrid), dim3(256), 0, st, b, bbox);
  int nb = 0, ne = 1, depth = 0;
  for (;;) {
    depth++;
    const int ngrid = (ne - nb + 255) / 256;
    hipLaunchKernelGGL(kd_minmax_kernel, dim3(sgrid), dim3(256), 0, st, b);
    hipLaunchKernelGGL(kd_split_kernel, dim3(ngrid), dim3(256), 0, st, b, nb, ne);
    if (depth > kKdMaxDepth)
      break;
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
    int32_t created = 0;
    e = hipMemcpyAsync(&created, b.counters, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess)
      return e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess)
      return e;
    if (created == ne)
      break;  // every node of the level was a leaf
    nb = ne;
    ne = created;
  }
  *depth_out = depth;
  *nodes_out = ne;
  return hipGetLastError();
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
