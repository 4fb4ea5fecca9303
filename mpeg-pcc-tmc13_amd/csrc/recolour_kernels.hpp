// recolour_kernels.hpp -- attribute transfer onto a re-quantised geometry
// (pcc::recolour, tmc3/pointset_processing.cpp:926-957: recolourColour :253-594,
// recolourReflectance :618-916) on the device.
//
// The reference searches two nanoflann k-d trees point by point and sorts its backward
// lists with std::sort; where candidates are equidistant (a voxelised cloud at a dyadic
// scale: at nearly every point) its result is the ORDER those containers produce.  Both
// are therefore rebuilt: recolour_kdtree.hpp builds nanoflann's trees level by level and
// walks them in nanoflann's order; the backward lists are put into the reference's
// insertion order (source index) and sorted by libstdc++'s algorithm (introsort with a
// median-of-three pivot, heap sort at the depth limit, insertion sort up to 16 entries --
// not stable beyond 16).  Distances, weights, centroids and the +-search_range refinement
// are the reference's double-precision expressions in the reference's order (no
// contraction into fused multiply-adds).  Identical to the reference, ties included.
//
//   rc_bbox          bounding box of a cloud                          (atomics)
//   kd_*             the two trees (recolour_kdtree.hpp)
//   rc_forward       per target point: K nearest source points, blended colour
//   rc_backward      per source point: its nearest target points -> (target, dist)
//   rc_list_fill     the backward lists, one contiguous range per target
//   rc_blend         per target point: list in the reference's order, centroid, refinement
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpcc_attr_mi355.h"
#include "gpcc_primitives.hpp"
#include "recolour_kdtree.hpp"

namespace gpcc {

constexpr int kRcMaxK = 8;

struct RcCtx {
  gpcc_recolour_params p;
  KdTree src, tgt;
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
  // a finite max_geometry_dist2_fwd (round 5): every target's nearest source point and the first target whose
  // k-th neighbour lies beyond the limit -- from there on the reference's shrunk result vectors hold ONE entry
  // (pointset_processing.cpp:292-313); null without the limit
  int32_t* nearest;    // [nt]
  int32_t* fwd_first;  // [1], starts at INT32_MAX
};

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

// wavefronts per SIMD the two search kernels are compiled for (0: the compiler's choice); an experiment knob
#ifndef GPCC_RC_WAVES
#define GPCC_RC_WAVES 0
#endif
#if GPCC_RC_WAVES > 0
#define GPCC_RC_OCCUPANCY __attribute__((amdgpu_waves_per_eu(GPCC_RC_WAVES)))
#else
#define GPCC_RC_OCCUPANCY
#endif

// ---- forward (pointset_processing.cpp:296-384 / 659-728) ---------------------------
// (ALIMIT: a finite max_attribute_dist2_fwd; without one the k x k comparison of the neighbours'
// attributes decides nothing and is left out)
template<int C, int K, bool ALIMIT>
__global__ __launch_bounds__(256) GPCC_RC_OCCUPANCY void
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
  rc_kd_search<K>(cx.src, q, kf, r);
  if (cx.nearest) {
    cx.nearest[t] = r.i[0];
    if (rc_worst(r, kf) > rc_limit(p.max_geometry_dist2_fwd))
      atomicMin(cx.fwd_first, t);
  }
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
  // the largest attribute distance among the first m neighbours, for every m, computed ONCE (the
  // reference recomputes all pairs for every candidate count, :341-349 / 692-699): pm[m] = max over
  // i, j < m, both orders (colour differences wrap in 16 bits there: Vec3<attr_t> - Vec3<attr_t>,
  // tmc3/PCCMath.h:280; reflectances are subtracted as int)
  double pm[K + 1];
#pragma unroll
  for (int m = 0; m <= K; m++)
    pm[m] = 2.2250738585072014e-308;
  if (ALIMIT) {
#pragma unroll
    for (int m = 2; m <= K; m++) {
      double best = pm[m - 1];
#pragma unroll
      for (int j = 0; j < m - 1; j++) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int k = 0; k < C; k++) {
          const int32_t di = col[m - 1][k] - col[j][k];
          const double d1 = C == 3 ? (double)(uint16_t)di : (double)di;
          const double d2 = C == 3 ? (double)(uint16_t)(-di) : (double)(-di);
          s1 += d1 * d1;
          s2 += d2 * d2;
        }
        best = s1 > best ? s1 : best;
        best = s2 > best ? s2 : best;
      }
      pm[m] = best;
    }
  }
  for (int nn = r.count; nn > 0; nn--) {
    if (nn == 1) {
#pragma unroll
      for (int k = 0; k < C; k++)
        out[k] = col[0][k];
      return;
    }
    double maxa = pm[1];
#pragma unroll
    for (int m = 2; m <= K; m++)
      maxa = nn == m ? pm[m] : maxa;
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

// the targets from the first one beyond a finite forward geometry limit on: the colour of the nearest source
// point (the reference's result vectors hold one entry from there, pointset_processing.cpp:304-313, 329-334)
__global__ __launch_bounds__(256) void
rc_forward_limit_kernel(RcCtx cx)
{
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cx.tgt.n || t < *cx.fwd_first)
    return;
  for (int k = 0; k < cx.c; k++)
    cx.ref1[(size_t)t * cx.c + k] = cx.src_attrs[(size_t)cx.nearest[t] * cx.c + k];
}

// ---- backward (:386-424 / 730-766): nearest targets of every source point ------------
template<int K>
__global__ __launch_bounds__(256) GPCC_RC_OCCUPANCY void
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
  rc_kd_search<K>(cx.tgt, q, kb, r);
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

// ---- std::sort(first, last, by distance) as libstdc++ does it (bits/stl_algo.h: __introsort_loop,
//      __unguarded_partition_pivot, __final_insertion_sort; bits/stl_heap.h), on the pair of arrays
//      (distance, source) of one list.  Checked against std::sort itself: tests/test_oracle_recolour.py
//      pins the C restatement this follows, tests/test_gpu_recolour.py the device against both.
__device__ __forceinline__ void
rc_ss_swap(double* d, int32_t* s, int a, int b)
{
  const double td = d[a];
  const int32_t ts = s[a];
  d[a] = d[b];
  s[a] = s[b];
  d[b] = td;
  s[b] = ts;
}

__device__ inline void
rc_ss_adjust_heap(double* d, int32_t* s, int hole, int len, double vd, int32_t vs)
{
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (d[child] < d[child - 1])
      child--;
    d[hole] = d[child];
    s[hole] = s[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    d[hole] = d[child - 1];
    s[hole] = s[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && d[parent] < vd) {
    d[hole] = d[parent];
    s[hole] = s[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  d[hole] = vd;
  s[hole] = vs;
}

__device__ inline void
rc_ss_heap_sort(double* d, int32_t* s, int len)
{
  if (len >= 2)
    for (int parent = (len - 2) / 2;; parent--) {
      rc_ss_adjust_heap(d, s, parent, len, d[parent], s[parent]);
      if (parent == 0)
        break;
    }
  for (int last = len - 1; last >= 1; last--) {
    const double vd = d[last];
    const int32_t vs = s[last];
    d[last] = d[0];
    s[last] = s[0];
    rc_ss_adjust_heap(d, s, 0, last, vd, vs);
  }
}

// entries [from, n) into the sorted prefix; `guarded`: an entry below the first one goes to the front
__device__ inline void
rc_ss_insertion(double* d, int32_t* s, int from, int n, bool guarded)
{
  for (int i = from; i < n; i++) {
    const double vd = d[i];
    const int32_t vs = s[i];
    int j = i;
    if (guarded && vd < d[0]) {
      for (; j > 0; j--) {
        d[j] = d[j - 1];
        s[j] = s[j - 1];
      }
    } else {
      while (vd < d[j - 1]) {
        d[j] = d[j - 1];
        s[j] = s[j - 1];
        j--;
      }
    }
    d[j] = vd;
    s[j] = vs;
  }
}

__device__ inline void
rc_std_sort(double* d, int32_t* s, int n)
{
  if (n < 2)
    return;
  if (n > 16) {
    int depth = 0;
    for (int m = n; m > 1; m >>= 1)
      depth++;
    depth *= 2;
    // the recursive call on [cut, last) as a stack; the two parts are disjoint ranges, their order is free
    int st_first[64], st_last[64], st_depth[64];
    int sp = 0;
    st_first[0] = 0;
    st_last[0] = n;
    st_depth[0] = depth;
    while (sp >= 0) {
      const int first = st_first[sp];
      int last = st_last[sp], dl = st_depth[sp];
      sp--;
      while (last - first > 16) {
        if (dl == 0) {
          rc_ss_heap_sort(d + first, s + first, last - first);
          break;
        }
        dl--;
        // median of first + 1, middle, last - 1 to the front
        const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
        if (d[x] < d[y]) {
          if (d[y] < d[z])
            rc_ss_swap(d, s, first, y);
          else if (d[x] < d[z])
            rc_ss_swap(d, s, first, z);
          else
            rc_ss_swap(d, s, first, x);
        } else if (d[x] < d[z])
          rc_ss_swap(d, s, first, x);
        else if (d[y] < d[z])
          rc_ss_swap(d, s, first, z);
        else
          rc_ss_swap(d, s, first, y);
        int i = first + 1, j = last;
        for (;;) {
          while (d[i] < d[first])
            i++;
          j--;
          while (d[first] < d[j])
            j--;
          if (!(i < j))
            break;
          rc_ss_swap(d, s, i, j);
          i++;
        }
        sp++;
        st_first[sp] = i;
        st_last[sp] = last;
        st_depth[sp] = dl;
        last = i;
      }
    }
    rc_ss_insertion(d, s, 1, 16, true);
    rc_ss_insertion(d, s, 16, n, false);
  } else {
    rc_ss_insertion(d, s, 1, n, true);
  }
}

// ---- blend and refinement (:426-592 / 768-914) ---------------------------------------
template<int C>
__global__ __launch_bounds__(256) void
rc_blend_kernel(RcCtx cx)
{
#pragma clang fp contract(off)
  GPCC_VGPR_FLOOR_64();
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
  // the reference pushes a target's entries in source order (:397-412; the fill order here is
  // whatever the atomics made it), then std::sort orders them by distance alone (:416-422)
  for (int i = 1; i < n; i++) {
    const double d = ld[i];
    const int32_t s = ls[i];
    int j = i;
    while (j > 0 && ls[j - 1] > s) {
      ld[j] = ld[j - 1];
      ls[j] = ls[j - 1];
      j--;
    }
    ld[j] = d;
    ls[j] = s;
  }
  rc_std_sort(ld, ls, n);
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
