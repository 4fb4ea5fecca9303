// recolour_kernels.hpp -- attribute transfer onto a re-quantised geometry
// (pcc::recolour, tmc3/pointset_processing.cpp:926-957: recolourColour :253-594,
// recolourReflectance :618-916) on the device.
//
// The reference searches two nanoflann k-d trees point by point.  Here both clouds
// get a DENSE CELL TABLE over their bounding box (cell side 2^shift: at most 32 cells per
// point, refined for sparse clouds while an occupied cell holds more than 24 points;
// count / scan / fill, three launches),
// and a thread finds the exact K nearest points of its query by visiting the cells
// ring after ring around the query's cell until the K-th distance is not larger
// than anything an unvisited cell can hold.  Distances, weights, centroids and the
// +-search_range refinement are the reference's double-precision expressions in the
// reference's order (no contraction into fused multiply-adds); equidistant
// candidates are ordered by point index (the one place where the reference's
// outcome depends on its containers: see include/gpcc_attr_mi355.h).
//
//   rc_bbox          bounding box of a cloud                          (atomics)
//   rc_cell_count / rc_scan_* / rc_cell_fill     the cell table
//   rc_forward       per target point: K nearest source points, blended colour
//   rc_backward      per source point: its nearest target points -> (target, dist)
//   rc_list_fill     the backward lists, one contiguous range per target
//   rc_blend         per target point: list sorted by (distance, source), centroid,
//                    refinement
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpcc_attr_mi355.h"

namespace gpcc {

constexpr int kRcMaxK = 8;
// refine the cell side while the cell an average point sits in holds more than this
constexpr double kRcMaxCellLoad = 12.0;
constexpr int kRcScanBlock = 2048;  // elements per workgroup of the scan

struct RcGrid {
  const int32_t* xyz;   // [n][3]
  int32_t n;
  int32_t shift;
  int32_t lo[3], dim[3];
  int32_t* start;       // [cells + 1]
  int32_t* items;       // [n]
};

struct RcCtx {
  gpcc_recolour_params p;
  RcGrid src, tgt;
  const int32_t* src_attrs;  // [ns][c]
  int32_t c;
  double s2t, t2s;
  int32_t off[3];
  int32_t* ref1;       // [nt][c] forward colours
  int32_t* bt;         // [ns][kb] backward: target of every (source, place), -1 = none
  double* bd;          // [ns][kb] its squared distance
  int32_t* lstart;     // [nt + 1] backward lists (counts, then offsets)
  int32_t* lcur;       // [nt] fill cursors
  double* ldist;       // [total]
  int32_t* lsrc;       // [total]
  int32_t* out;        // [nt][c]
};

__device__ __forceinline__ size_t
rc_cell(const RcGrid& g, int x, int y, int z)
{
  return ((size_t)((x >> g.shift) - g.lo[0]) * g.dim[1] + (size_t)((y >> g.shift) - g.lo[1])) * g.dim[2]
    + (size_t)((z >> g.shift) - g.lo[2]);
}

// ---- bounding box ---------------------------------------------------------------
__global__ __launch_bounds__(256) void
rc_bbox_kernel(const int32_t* __restrict__ xyz, int n, int32_t* box /* min[3], max[3] */)
{
  int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {-0x7fffffff, -0x7fffffff, -0x7fffffff};
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int v = xyz[3 * i + k];
      mn[k] = v < mn[k] ? v : mn[k];
      mx[k] = v > mx[k] ? v : mx[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int a = __shfl_xor(mn[k], d), b = __shfl_xor(mx[k], d);
      mn[k] = a < mn[k] ? a : mn[k];
      mx[k] = b > mx[k] ? b : mx[k];
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&box[k], mn[k]);
      atomicMax(&box[3 + k], mx[k]);
    }
  }
}

// ---- cell table --------------------------------------------------------------------
__global__ __launch_bounds__(256) void
rc_cell_count_kernel(RcGrid g)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x)
    atomicAdd(&g.start[rc_cell(g, g.xyz[3 * i], g.xyz[3 * i + 1], g.xyz[3 * i + 2]) + 1], 1);
}

// sum over cells of count^2 (start[c + 1] still holds the cell's count here): divided by
// the number of points it is the load of the cell an average POINT sits in, which is what
// a query pays per cell it opens -- the mean over cells hides the crowded ones
__global__ __launch_bounds__(256) void
rc_cell_load_kernel(const int32_t* __restrict__ start, size_t cells, unsigned long long* load)
{
  unsigned long long acc = 0;
  for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < cells; c += (size_t)gridDim.x * blockDim.x) {
    const unsigned long long k = (unsigned)start[c + 1];
    acc += k * k;
  }
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
    acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0 && acc)
    atomicAdd(load, acc);
}

__global__ __launch_bounds__(256) void
rc_cell_fill_kernel(RcGrid g, int32_t* cursor)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n; i += gridDim.x * blockDim.x) {
    const size_t c = rc_cell(g, g.xyz[3 * i], g.xyz[3 * i + 1], g.xyz[3 * i + 2]);
    g.items[g.start[c] + atomicAdd(&cursor[c], 1)] = i;
  }
}

// inclusive scan of a[0..n) in place (a[0] stays: the arrays carry a leading zero),
// three launches: block sums, their scan by one workgroup, the blocks again
__global__ __launch_bounds__(256) void
rc_scan_sums_kernel(const int32_t* __restrict__ a, size_t n, long long* __restrict__ sums)
{
  __shared__ long long w[4];
  const size_t base = (size_t)blockIdx.x * kRcScanBlock;
  long long s = 0;
  for (int k = 0; k < kRcScanBlock / 256; k++) {
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
rc_scan_blocks_kernel(long long* sums, int nblocks)
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
rc_scan_apply_kernel(int32_t* a, size_t n, const long long* __restrict__ sums)
{
  __shared__ int wsum[4];
  const size_t base = (size_t)blockIdx.x * kRcScanBlock;
  int run = (int)sums[blockIdx.x];
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < kRcScanBlock / 256; k++) {
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

// ---- exact K nearest neighbours -----------------------------------------------------
// (K = the list's capacity, a template parameter: the kernels are built for 1, 2, 4 and 8 -- with
// the one list of 8 the forward kernel took 256 registers and 1.9 KB of scratch per lane, one
// wavefront per SIMD, and the backward kernel, which keeps ONE neighbour, carried seven idle ones)
template<int K>
struct RcKnn {
  double d[K];
  int32_t i[K];
  int count;
};

// (d2, idx) into the ascending list; equal distances by index; entries beyond the
// k-th fall off.  Static register indices only.
template<int K>
__device__ __forceinline__ void
rc_insert(RcKnn<K>& r, int k, double d, int32_t idx)
{
  // where it goes: the number of entries that come before it
  int pos = 0;
#pragma unroll
  for (int p = 0; p < K; p++)
    pos += (p < r.count && (r.d[p] < d || (r.d[p] == d && r.i[p] < idx))) ? 1 : 0;
  if (pos >= k)
    return;
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
  r.count = r.count < k ? r.count + 1 : k;
}

template<int K>
__device__ __forceinline__ double
rc_kth(const RcKnn<K>& r, int k)
{
  double v = r.d[0];
#pragma unroll
  for (int p = 1; p < K; p++)
    v = p == k - 1 ? r.d[p] : v;
  return v;
}

template<int K>
__device__ __forceinline__ void
rc_knn(const RcGrid& g, const double q[3], int k, RcKnn<K>& r)
{
#pragma clang fp contract(off)
  const int cs = 1 << g.shift;
  int cq[3];
#pragma unroll
  for (int a = 0; a < 3; a++)
    cq[a] = (int)floor(q[a] / cs) - g.lo[a];
  r.count = 0;
#pragma unroll
  for (int p = 0; p < K; p++) {
    r.d[p] = 0.0;
    r.i[p] = 0;
  }
  int maxr = 0;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int far = cq[a] > g.dim[a] - 1 - cq[a] ? cq[a] : g.dim[a] - 1 - cq[a];
    far = far < 0 ? -far : far;
    maxr = far > maxr ? far : maxr;
  }
  for (int rr = 0; rr <= maxr; rr++) {
    for (int dx = -rr; dx <= rr; dx++) {
      const int cx = cq[0] + dx;
      if (cx < 0 || cx >= g.dim[0])
        continue;
      for (int dy = -rr; dy <= rr; dy++) {
        const int cy = cq[1] + dy;
        if (cy < 0 || cy >= g.dim[1])
          continue;
        const bool shell = dx == -rr || dx == rr || dy == -rr || dy == rr;
        const int step = (shell || rr == 0) ? 1 : 2 * rr;
        for (int dz = -rr; dz <= rr; dz += step) {
          const int cz = cq[2] + dz;
          if (cz < 0 || cz >= g.dim[2])
            continue;
          const size_t c = ((size_t)cx * g.dim[1] + (size_t)cy) * g.dim[2] + (size_t)cz;
          const int e1 = g.start[c + 1];
          for (int e = g.start[c]; e < e1; e++) {
            const int32_t i = g.items[e];
            // nanoflann's L2 adaptor: result += diff * diff, x then y then z
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 3; a++) {
              const double diff = q[a] - (double)g.xyz[3 * i + a];
              s += diff * diff;
            }
            rc_insert(r, k, s, i);
          }
        }
      }
    }
    // a point of ring rr + 1 or beyond differs by more than rr * cs in some axis
    const double bound = (double)rr * cs;
    if (r.count == k && rc_kth(r, k) <= bound * bound)
      break;
  }
}

__device__ __forceinline__ double
rc_clip(double v, double lo, double hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}

__device__ __forceinline__ double
rc_limit(double v)
{
  return v < 512 ? v : 1.7976931348623157e308;
}

// ---- forward (pointset_processing.cpp:296-384 / 659-728) ---------------------------
// (ALIMIT: a finite max_attribute_dist2_fwd; without one the k x k comparison of the neighbours'
// attributes decides nothing and is left out)
template<int C, int K, bool ALIMIT>
__global__ __launch_bounds__(256) void
rc_forward_kernel(RcCtx cx)
{
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cx.tgt.n)
    return;
  const gpcc_recolour_params& p = cx.p;
  const int kf = p.num_neighbours_fwd;
  const double max_a = rc_limit(p.max_attribute_dist2_fwd);
  const double clip_max = (double)((1 << p.bitdepth) - 1);
  double q[3];
#pragma unroll
  for (int a = 0; a < 3; a++)
    q[a] = (double)(cx.tgt.xyz[3 * t + a] + cx.off[a]) * cx.t2s;
  RcKnn<K> r;
  rc_knn<K>(cx.src, q, kf, r);
  // the neighbours' attributes, nearest first
  int32_t col[K][C];
#pragma unroll
  for (int i = 0; i < K; i++)
#pragma unroll
    for (int k = 0; k < C; k++)
      col[i][k] = i < r.count ? cx.src_attrs[(size_t)r.i[i] * C + k] : 0;
  int32_t* out = cx.ref1 + (size_t)t * C;
  if (p.skip_avg_if_identical_fwd && r.d[0] < 0.0001) {
#pragma unroll
    for (int k = 0; k < C; k++)
      out[k] = col[0][k];
    return;
  }
  for (int nn = r.count; nn > 0; nn--) {
    if (nn == 1) {
#pragma unroll
      for (int k = 0; k < C; k++)
        out[k] = col[0][k];
      return;
    }
    // (colour differences wrap in 16 bits there: Vec3<attr_t> - Vec3<attr_t>,
    // tmc3/PCCMath.h:280; reflectances are subtracted as int)
    double maxa = 2.2250738585072014e-308;
    if (ALIMIT)
#pragma unroll
    for (int i = 0; i < K; i++)
#pragma unroll
      for (int j = 0; j < K; j++) {
        if (i < nn && j < nn) {
          double s = 0.0;
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int32_t di = col[i][k] - col[j][k];
            const double d = C == 3 ? (double)(uint16_t)di : (double)di;
            s += d * d;
          }
          maxa = s > maxa ? s : maxa;
        }
      }
    if (maxa > max_a)
      continue;
    double acc[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      acc[k] = 0.0;
    if (p.use_dist_weighted_avg_fwd) {
      double sumw = 0.0;
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (i < nn) {
          const double w = 1 / (r.d[i] + p.dist_offset_fwd);
#pragma unroll
          for (int k = 0; k < C; k++)
            acc[k] += (double)col[i][k] * w;
          sumw += w;
        }
      }
#pragma unroll
      for (int k = 0; k < C; k++)
        acc[k] /= sumw;
    } else {
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (i < nn) {
#pragma unroll
          for (int k = 0; k < C; k++)
            acc[k] += (double)col[i][k];
        }
      }
#pragma unroll
      for (int k = 0; k < C; k++)
        acc[k] /= nn;
    }
#pragma unroll
    for (int k = 0; k < C; k++)
      out[k] = (int32_t)rc_clip(round(acc[k]), 0.0, clip_max);
    return;
  }
}

// ---- backward (:386-424 / 730-766): nearest targets of every source point ------------
template<int K>
__global__ __launch_bounds__(256) void
rc_backward_kernel(RcCtx cx)
{
#pragma clang fp contract(off)
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cx.src.n)
    return;
  const int kb = cx.p.num_neighbours_bwd;
  const double max_g = rc_limit(cx.p.max_geometry_dist2_bwd);
  double q[3];
#pragma unroll
  for (int a = 0; a < 3; a++)
    q[a] = (double)cx.src.xyz[3 * s + a] * cx.s2t - (double)cx.off[a];
  RcKnn<K> r;
  rc_knn<K>(cx.tgt, q, kb, r);
#pragma unroll
  for (int i = 0; i < K; i++) {
    if (i < kb) {
      const bool ok = i < r.count && r.d[i] <= max_g;
      cx.bt[(size_t)s * kb + i] = ok ? r.i[i] : -1;
      cx.bd[(size_t)s * kb + i] = ok ? r.d[i] : 0.0;
      if (ok)
        atomicAdd(&cx.lstart[r.i[i] + 1], 1);
    }
  }
}

__global__ __launch_bounds__(256) void
rc_list_fill_kernel(RcCtx cx)
{
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= cx.src.n)
    return;
  const int kb = cx.p.num_neighbours_bwd;
  for (int i = 0; i < kb; i++) {
    const int t = cx.bt[(size_t)s * kb + i];
    if (t >= 0) {
      const int pos = cx.lstart[t] + atomicAdd(&cx.lcur[t], 1);
      cx.ldist[pos] = cx.bd[(size_t)s * kb + i];
      cx.lsrc[pos] = s;
    }
  }
}

// ---- blend and refinement (:426-592 / 768-914) ---------------------------------------
template<int C>
__global__ __launch_bounds__(256) void
rc_blend_kernel(RcCtx cx)
{
#pragma clang fp contract(off)
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cx.tgt.n)
    return;
  const gpcc_recolour_params& p = cx.p;
  const int32_t* __restrict__ sa = cx.src_attrs;
  const int l0 = cx.lstart[t];
  int n = cx.lstart[t + 1] - l0;
  double* ld = cx.ldist + l0;
  int32_t* ls = cx.lsrc + l0;
  const int32_t* c1 = cx.ref1 + (size_t)t * C;
  int32_t* out = cx.out + (size_t)t * C;
  if (n == 0) {
#pragma unroll
    for (int k = 0; k < C; k++)
      out[k] = c1[k];
    return;
  }
  // the list by (distance, source index): insertion sort in place (the fill order is
  // whatever the atomics made it)
  for (int i = 1; i < n; i++) {
    const double d = ld[i];
    const int32_t s = ls[i];
    int j = i;
    while (j > 0 && (ld[j - 1] > d || (ld[j - 1] == d && ls[j - 1] > s))) {
      ld[j] = ld[j - 1];
      ls[j] = ls[j - 1];
      j--;
    }
    ld[j] = d;
    ls[j] = s;
  }
  const double max_a = rc_limit(p.max_attribute_dist2_bwd);
  const double clip_max = (double)((1 << p.bitdepth) - 1);
  double cen2[C];
#pragma unroll
  for (int k = 0; k < C; k++)
    cen2[k] = 0.0;
  bool done = false;
  if (p.skip_avg_if_identical_bwd && ld[0] < 0.0001) {
    n = 1;
#pragma unroll
    for (int k = 0; k < C; k++)
      cen2[k] = (double)sa[(size_t)ls[0] * C + k];
    done = true;
  }
  while (!done) {
    if (n == 1) {
#pragma unroll
      for (int k = 0; k < C; k++)
        cen2[k] = (double)sa[(size_t)ls[0] * C + k];
      break;
    }
    double maxa = 2.2250738585072014e-308;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < C; k++) {
          const double d = (double)sa[(size_t)ls[i] * C + k] - (double)sa[(size_t)ls[j] * C + k];
          s += d * d;
        }
        maxa = s > maxa ? s : maxa;
      }
    if (maxa <= max_a) {
#pragma unroll
      for (int k = 0; k < C; k++)
        cen2[k] = 0.0;
      if (p.use_dist_weighted_avg_bwd) {
        double sumw = 0.0;
        for (int i = 0; i < n; i++) {
          const double w = 1 / (sqrt(ld[i]) + p.dist_offset_bwd);
#pragma unroll
          for (int k = 0; k < C; k++)
            cen2[k] += (double)sa[(size_t)ls[i] * C + k] * w;
          sumw += w;
        }
#pragma unroll
        for (int k = 0; k < C; k++)
          cen2[k] /= sumw;
      } else {
        for (int i = 0; i < n; i++)
#pragma unroll
          for (int k = 0; k < C; k++)
            cen2[k] += (double)sa[(size_t)ls[i] * C + k];
#pragma unroll
        for (int k = 0; k < C; k++)
          cen2[k] /= (double)n;
      }
      break;
    }
    n--;  // the farthest entry leaves
  }
  // fixWeight (m42538): w = 0, the start value is the backward centroid
  double c0[C], best[C], col[C];
#pragma unroll
  for (int k = 0; k < C; k++) {
    c0[k] = rc_clip(round(0.0 * (double)c1[k] + 1.0 * cen2[k]), 0.0, clip_max);
    best[k] = col[k] = c0[k];
  }
  const double r_source = 1.0 / (double)cx.src.n;
  const double r_target = 1.0 / (double)cx.tgt.n;
  double min_err = 1.7976931348623157e308;
  const int sr = p.search_range;
  const int n1 = C == 3 ? sr : 0;
  for (int s1 = -sr; s1 <= sr; s1++) {
    col[0] = rc_clip(c0[0] + s1, 0.0, clip_max);
    for (int s2 = -n1; s2 <= n1; s2++) {
      if (C == 3)
        col[C == 3 ? 1 : 0] = rc_clip(c0[C == 3 ? 1 : 0] + s2, 0.0, clip_max);
      for (int s3 = -n1; s3 <= n1; s3++) {
        if (C == 3)
          col[C == 3 ? 2 : 0] = rc_clip(c0[C == 3 ? 2 : 0] + s3, 0.0, clip_max);
        double e1 = 0.0;
#pragma unroll
        for (int k = 0; k < C; k++) {
          const double d = col[k] - (double)c1[k];
          e1 += d * d;
        }
        e1 *= r_target;
        double e2 = 0.0;
        for (int i = 0; i < n; i++)
#pragma unroll
          for (int k = 0; k < C; k++) {
            const double d = col[k] - (double)sa[(size_t)ls[i] * C + k];
            e2 += d * d;
          }
        e2 *= r_source;
        const double err = e1 > e2 ? e1 : e2;
        if (err < min_err) {
          min_err = err;
#pragma unroll
          for (int k = 0; k < C; k++)
            best[k] = col[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < C; k++)
    out[k] = (int32_t)best[k];
}

}  // namespace gpcc
