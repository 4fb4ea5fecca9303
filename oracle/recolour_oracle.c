/* recolour_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of pcc::recolour (tmc3/pointset_processing.cpp:926-957:
 * recolourColour :253-594, recolourReflectance :618-916) in the form the device
 * kernels use.  The reference's outcome depends on the ORDER in which its containers
 * hand over equidistant candidates, so the containers are restated too:
 *   - the nanoflann k-d tree (dependencies/nanoflann/nanoflann.hpp: divideTree :872,
 *     middleSplit_ :922, planeSplit :972, leaf size 10) built LEVEL BY LEVEL, every
 *     plane split as rank arithmetic (the i-th misplaced index from the left changes
 *     places with the i-th from the right: what the two-pointer loop does, without
 *     the loop) -- the form a GPU can run;
 *   - its search (searchLevel :1308, KNNResultSet::addPoint :175) as an explicit
 *     stack walk: the same child first, the same pruning expression, candidates of
 *     equal distance in visiting order;
 *   - the backward lists in source order, sorted by std::sort's algorithm (libstdc++
 *     11: introsort with a median-of-three pivot, heap sort at the depth limit,
 *     insertion sort for ranges of at most 16 -- NOT stable beyond 16 entries);
 * and every floating-point expression of the reference evaluated in double in the
 * reference's order.  The device path has to match this file bit for bit; this file
 * matches the compiled reference (oracle/_ref) everywhere, ties included --
 * tests/test_oracle_recolour.py.
 *
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use it. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"

#define RC_MAXK 8
#define KD_LEAF 10
#define KD_MAX_DEPTH 120

/* ---- the k-d tree ------------------------------------------------------------------ */
typedef struct {
  int32_t left, right;   /* its range of vind */
  int32_t child1, child2; /* node indices, -1 for a leaf */
  int32_t feat;
  double divlow, divhigh; /* nanoflann.hpp:910-911: the children's TIGHT bounds along feat */
  double lo[3], hi[3];    /* the box handed down (build only: cut planes of the ancestors) */
} KdNode;

typedef struct {
  const int32_t* xyz;
  int n;
  int32_t* vind;
  KdNode* nodes;
  int nnodes;
  double root_lo[3], root_hi[3];
} KdTree;

/* planeSplit's two-pointer loop (nanoflann.hpp:972-998) as rank arithmetic: with m entries
 * that belong left, the i-th entry of ind[0, m) that does not belong there (ascending) and
 * the i-th entry of ind[m, count) that belongs left (DESCENDING) change places. */
static void
hoare_by_rank(int32_t* ind, int count, const uint8_t* goes_left, int32_t* tmp)
{
  int m = 0;
  for (int i = 0; i < count; i++)
    m += goes_left[i];
  int nl = 0, nr = 0;
  int32_t* lpos = tmp;
  int32_t* rpos = tmp + count;
  for (int i = 0; i < m; i++)
    if (!goes_left[i])
      lpos[nl++] = i;
  for (int i = count - 1; i >= m; i--)
    if (goes_left[i])
      rpos[nr++] = i;
  for (int i = 0; i < nl; i++) {
    const int32_t t = ind[lpos[i]];
    ind[lpos[i]] = ind[rpos[i]];
    ind[rpos[i]] = t;
  }
}

static int
kd_build(KdTree* kt, const int32_t* xyz, int n)
{
  kt->xyz = xyz;
  kt->n = n;
  kt->vind = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  kt->nodes = (KdNode*)malloc(sizeof(KdNode) * (2 * (size_t)n + 1));
  int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
  uint8_t* flag = (uint8_t*)malloc((size_t)n);
  if (!kt->vind || !kt->nodes || !tmp || !flag)
    return -1;
  for (int i = 0; i < n; i++)
    kt->vind[i] = i;
  KdNode* root = &kt->nodes[0];
  root->left = 0;
  root->right = n;
  for (int a = 0; a < 3; a++) {
    int mn = xyz[a], mx = xyz[a];
    for (int i = 1; i < n; i++) {
      const int v = xyz[3 * i + a];
      mn = v < mn ? v : mn;
      mx = v > mx ? v : mx;
    }
    root->lo[a] = kt->root_lo[a] = (double)mn;
    root->hi[a] = kt->root_hi[a] = (double)mx;
  }
  kt->nnodes = 1;
  /* breadth first: node k's children are appended while k is visited */
  for (int k = 0; k < kt->nnodes; k++) {
    KdNode* nd = &kt->nodes[k];
    const int count = nd->right - nd->left;
    int32_t* ind = kt->vind + nd->left;
    nd->child1 = nd->child2 = -1;
    nd->feat = 0;
    nd->divlow = nd->divhigh = 0.0;
    if (count <= KD_LEAF)
      continue;
    /* middleSplit_ (:922-961) */
    double mn[3], mx[3];
    for (int a = 0; a < 3; a++) {
      int lo = xyz[3 * ind[0] + a], hi = lo;
      for (int i = 1; i < count; i++) {
        const int v = xyz[3 * ind[i] + a];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
      mn[a] = (double)lo;
      mx[a] = (double)hi;
    }
    const double EPS = 0.00001;
    double max_span = nd->hi[0] - nd->lo[0];
    for (int a = 1; a < 3; a++) {
      const double span = nd->hi[a] - nd->lo[a];
      if (span > max_span)
        max_span = span;
    }
    double max_spread = -1;
    int feat = 0;
    for (int a = 0; a < 3; a++) {
      const double span = nd->hi[a] - nd->lo[a];
      if (span >= (1 - EPS) * max_span) {
        const double spread = mx[a] - mn[a];
        if (spread > max_spread) {
          feat = a;
          max_spread = spread;
        }
      }
    }
    const double split = (nd->lo[feat] + nd->hi[feat]) / 2;
    const double cut = split < mn[feat] ? mn[feat] : (split > mx[feat] ? mx[feat] : split);
    int lim1 = 0, lim2 = 0;
    for (int i = 0; i < count; i++) {
      const double v = (double)xyz[3 * ind[i] + feat];
      flag[i] = v < cut;
      lim1 += v < cut;
      lim2 += v <= cut;
    }
    hoare_by_rank(ind, count, flag, tmp);
    for (int i = lim1; i < count; i++)
      flag[i - lim1] = (double)xyz[3 * ind[i] + feat] <= cut;
    hoare_by_rank(ind + lim1, count - lim1, flag, tmp);
    const int half = count / 2;
    const int idx = lim1 > half ? lim1 : (lim2 < half ? lim2 : half);
    /* children: the box cut at the plane (divideTree :900-908) */
    KdNode* c1 = &kt->nodes[kt->nnodes];
    KdNode* c2 = &kt->nodes[kt->nnodes + 1];
    nd->child1 = kt->nnodes;
    nd->child2 = kt->nnodes + 1;
    kt->nnodes += 2;
    nd->feat = feat;
    for (int a = 0; a < 3; a++) {
      c1->lo[a] = c2->lo[a] = nd->lo[a];
      c1->hi[a] = c2->hi[a] = nd->hi[a];
    }
    c1->hi[feat] = cut;
    c2->lo[feat] = cut;
    c1->left = nd->left;
    c1->right = c2->left = nd->left + idx;
    c2->right = nd->right;
    /* the recursion returns the children's tight boxes in place of the ones handed down */
    int lmax = xyz[3 * ind[0] + feat], rmin = xyz[3 * ind[idx] + feat];
    for (int i = 1; i < idx; i++) {
      const int v = xyz[3 * ind[i] + feat];
      lmax = v > lmax ? v : lmax;
    }
    for (int i = idx + 1; i < count; i++) {
      const int v = xyz[3 * ind[i] + feat];
      rmin = v < rmin ? v : rmin;
    }
    nd->divlow = (double)lmax;
    nd->divhigh = (double)rmin;
  }
  free(flag);
  free(tmp);
  return 0;
}

static void
kd_free(KdTree* kt)
{
  free(kt->vind);
  free(kt->nodes);
}

/* KNNResultSet::addPoint (:175-199): behind every entry that is not farther */
static void
knn_add(double* d2, int32_t* idx, int* count, int k, double d, int32_t i)
{
  int pos = *count;
  while (pos > 0 && d2[pos - 1] > d) {
    if (pos < k) {
      d2[pos] = d2[pos - 1];
      idx[pos] = idx[pos - 1];
    }
    pos--;
  }
  if (pos < k) {
    d2[pos] = d;
    idx[pos] = i;
  }
  if (*count < k)
    (*count)++;
}

/* findNeighbors (:1200-1215) + searchLevel (:1308-1365) with the recursion as a stack:
 * phase 0 = the node is entered, 1 = the nearer child has returned, 2 = the other one has */
static int
kd_search(const KdTree* kt, const double q[3], int k, double* d2, int32_t* idx)
{
  struct {
    int32_t node, phase;
    double mind, dst;
  } st[KD_MAX_DEPTH];
  double dists[3] = {0.0, 0.0, 0.0};
  double distsq = 0.0;
  for (int a = 0; a < 3; a++) {
    if (q[a] < kt->root_lo[a]) {
      dists[a] = (q[a] - kt->root_lo[a]) * (q[a] - kt->root_lo[a]);
      distsq += dists[a];
    }
    if (q[a] > kt->root_hi[a]) {
      dists[a] = (q[a] - kt->root_hi[a]) * (q[a] - kt->root_hi[a]);
      distsq += dists[a];
    }
  }
  int count = 0;
  d2[k - 1] = 1.7976931348623157e308;  /* KNNResultSet::init */
  int sp = 0;
  st[0].node = 0;
  st[0].phase = 0;
  st[0].mind = distsq;
  st[0].dst = 0.0;
  while (sp >= 0) {
    const KdNode* nd = &kt->nodes[st[sp].node];
    if (nd->child1 < 0) {
      const double worst = d2[k - 1];
      for (int e = nd->left; e < nd->right; e++) {
        const int32_t i = kt->vind[e];
        double s = 0.0;
        for (int a = 0; a < 3; a++) {
          const double diff = q[a] - (double)kt->xyz[3 * i + a];
          s += diff * diff;
        }
        if (s < worst)
          knn_add(d2, idx, &count, k, s, i);
      }
      sp--;
      continue;
    }
    const int f = nd->feat;
    const double val = q[f];
    const double diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    const int first_is_1 = (diff1 + diff2) < 0;
    if (st[sp].phase == 0) {
      st[sp].phase = 1;
      if (sp + 1 >= KD_MAX_DEPTH)
        return -1;
      st[sp + 1].node = first_is_1 ? nd->child1 : nd->child2;
      st[sp + 1].phase = 0;
      st[sp + 1].mind = st[sp].mind;
      sp++;
    } else if (st[sp].phase == 1) {
      const double cut_dist = first_is_1 ? (val - nd->divhigh) * (val - nd->divhigh)
                                         : (val - nd->divlow) * (val - nd->divlow);
      const double dst = dists[f];
      const double mind = st[sp].mind + cut_dist - dst;
      dists[f] = cut_dist;
      st[sp].dst = dst;
      st[sp].phase = 2;
      if (mind <= d2[k - 1]) {
        st[sp + 1].node = first_is_1 ? nd->child2 : nd->child1;
        st[sp + 1].phase = 0;
        st[sp + 1].mind = mind;
        sp++;
      }
    } else {
      dists[f] = st[sp].dst;
      sp--;
    }
  }
  return count;
}

/* ---- std::sort(first, last, by distance) as libstdc++ 11 does it (bits/stl_algo.h:
 *      __introsort_loop, __unguarded_partition_pivot, __final_insertion_sort; bits/stl_heap.h) */
typedef struct {
  double dist;
  int32_t src;
} BwdEntry;

#define LESS(a, b) ((a).dist < (b).dist)
static void
ss_swap(BwdEntry* a, BwdEntry* b)
{
  const BwdEntry t = *a;
  *a = *b;
  *b = t;
}

static void
ss_adjust_heap(BwdEntry* first, int hole, int len, BwdEntry value)
{
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (LESS(first[child], first[child - 1]))
      child--;
    first[hole] = first[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    first[hole] = first[child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && LESS(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

int oracle_std_sort_heap_calls = 0;  /* (test support: the depth-limit path was exercised) */

static void
ss_heap_sort(BwdEntry* first, int len)
{
  oracle_std_sort_heap_calls++;
  if (len >= 2)
    for (int parent = (len - 2) / 2;; parent--) {
      ss_adjust_heap(first, parent, len, first[parent]);
      if (parent == 0)
        break;
    }
  for (int last = len - 1; last >= 1; last--) {
    const BwdEntry value = first[last];
    first[last] = first[0];
    ss_adjust_heap(first, 0, last, value);
  }
}

static void
ss_insertion(BwdEntry* a, int from, int n, int guarded)
{
  for (int i = from; i < n; i++) {
    const BwdEntry v = a[i];
    if (guarded && LESS(v, a[0])) {
      for (int j = i; j > 0; j--)
        a[j] = a[j - 1];
      a[0] = v;
    } else {
      int j = i;
      while (LESS(v, a[j - 1])) {
        a[j] = a[j - 1];
        j--;
      }
      a[j] = v;
    }
  }
}

void
oracle_std_sort_by_dist(BwdEntry* a, int n)
{
  if (n < 2)
    return;
  int depth = 0;
  for (int m = n; m > 1; m >>= 1)
    depth++;
  depth *= 2;
  /* the recursion on [cut, last) as a stack of (first, last, depth) */
  struct {
    int first, last, depth;
  } st[64];
  int sp = 0;
  st[0].first = 0;
  st[0].last = n;
  st[0].depth = depth;
  while (sp >= 0) {
    int first = st[sp].first, last = st[sp].last, dl = st[sp].depth;
    sp--;
    while (last - first > 16) {
      if (dl == 0) {
        ss_heap_sort(a + first, last - first);
        break;
      }
      dl--;
      BwdEntry *r = a + first, *x = a + first + 1, *y = a + first + (last - first) / 2, *z = a + last - 1;
      if (LESS(*x, *y)) {
        if (LESS(*y, *z))
          ss_swap(r, y);
        else if (LESS(*x, *z))
          ss_swap(r, z);
        else
          ss_swap(r, x);
      } else if (LESS(*x, *z))
        ss_swap(r, x);
      else if (LESS(*y, *z))
        ss_swap(r, z);
      else
        ss_swap(r, y);
      int i = first + 1, j = last;
      for (;;) {
        while (LESS(a[i], a[first]))
          i++;
        j--;
        while (LESS(a[first], a[j]))
          j--;
        if (!(i < j))
          break;
        ss_swap(a + i, a + j);
        i++;
      }
      /* the right part is sorted by the recursive call FIRST; the order of the two parts does
       * not matter (disjoint ranges), so it simply goes on the stack */
      sp++;
      st[sp].first = i;
      st[sp].last = last;
      st[sp].depth = dl;
      last = i;
    }
  }
  if (n > 16) {
    ss_insertion(a, 1, 16, 1);
    ss_insertion(a, 16, n, 0);
  } else {
    ss_insertion(a, 1, n, 1);
  }
}
#undef LESS

static double
clipd(double v, double lo, double hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}

int
oracle_recolour(
  const gpcc_recolour_params* p, const int32_t* src_xyz, const int32_t* src_attrs, int32_t ns,
  const int32_t* tgt_xyz, int32_t nt, int32_t c, float scale_f, const int32_t offset[3],
  int32_t* tgt_attrs)
{
  if (!p || ns <= 0 || nt <= 0 || (c != 1 && c != 3))
    return -1;
  const int kf = p->num_neighbours_fwd, kb = p->num_neighbours_bwd;
  if (kf < 1 || kf > RC_MAXK || kb < 1 || kb > RC_MAXK || ns < kf || nt < kb)
    return -2;
  /* A finite forward geometry limit (round 5): indicesFwd / sqrDistFwd live OUTSIDE the reference's loop
   * (:292-294) and its test (:304-313) looks at the farthest of the k neighbours FOUND
   * (sqrDistFwd[resultSetFwd.size() - 1]), not at the back of the shrinking vectors -- so the first target
   * point whose k-th neighbour lies beyond the limit pops the vectors down to ONE entry, and they stay
   * there: that point and every LATER one take the colour of their nearest source point (nNN =
   * indicesFwd.size() = 1, :329-334); the points before it are not limited at all.  (The result set still
   * writes k entries through the vectors' storage; only their size() has changed.) */
  const double max_g_f = p->max_geometry_dist2_fwd < 512 ? p->max_geometry_dist2_fwd : 1.7976931348623157e308;
  int fwd_size = kf;
  const double s2t = (double)scale_f;
  const double t2s = 1.0 / s2t;
  const double clip_max = (double)((1 << p->bitdepth) - 1);
  const double big = 1.7976931348623157e308;
  const double max_g_b = p->max_geometry_dist2_bwd < 512 ? p->max_geometry_dist2_bwd : big;
  const double max_a_f = p->max_attribute_dist2_fwd < 512 ? p->max_attribute_dist2_fwd : big;
  const double max_a_b = p->max_attribute_dist2_bwd < 512 ? p->max_attribute_dist2_bwd : big;

  KdTree gs, gt;
  if (kd_build(&gs, src_xyz, ns) || kd_build(&gt, tgt_xyz, nt))
    return -4;
  int32_t* ref1 = (int32_t*)malloc(sizeof(int32_t) * (size_t)nt * c);
  int32_t* cnt = (int32_t*)calloc((size_t)nt + 1, sizeof(int32_t));
  int32_t* bt = (int32_t*)malloc(sizeof(int32_t) * (size_t)ns * kb);
  double* bd = (double*)malloc(sizeof(double) * (size_t)ns * kb);

  /* ---- forward (pointset_processing.cpp:296-384 / 659-728) ---------------------- */
  for (int t = 0; t < nt; t++) {
    double q[3];
    for (int a = 0; a < 3; a++)
      q[a] = (double)(tgt_xyz[3 * t + a] + offset[a]) * t2s;
    double d2[RC_MAXK];
    int32_t idx[RC_MAXK];
    const int n = kd_search(&gs, q, kf, d2, idx);
    if (n < 0)
      return -5;
    if (fwd_size > 1 && d2[n - 1] > max_g_f)
      fwd_size = 1;
    int32_t* out = ref1 + (size_t)t * c;
    if (p->skip_avg_if_identical_fwd && d2[0] < 0.0001) {
      for (int k = 0; k < c; k++)
        out[k] = src_attrs[(size_t)idx[0] * c + k];
      continue;
    }
    for (int nn = n < fwd_size ? n : fwd_size; nn > 0; nn--) {
      if (nn == 1) {
        for (int k = 0; k < c; k++)
          out[k] = src_attrs[(size_t)idx[0] * c + k];
        break;
      }
      /* The forward test of the COLOUR path subtracts Vec3<attr_t> = uint16 vectors
       * (:341-349 with PCCMath.h:280): a negative component difference wraps to
       * 65536 - x, and both orders of every pair are visited, so with a finite limit
       * almost every set fails down to the nearest neighbour.  Reflectances are
       * subtracted as int (:692-699).  Reproduced as it is. */
      double maxa = 2.2250738585072014e-308;
      for (int i = 0; i < nn; i++)
        for (int j = 0; j < nn; j++) {
          double s = 0.0;
          for (int k = 0; k < c; k++) {
            const int32_t di = src_attrs[(size_t)idx[i] * c + k] - src_attrs[(size_t)idx[j] * c + k];
            const double d = c == 3 ? (double)(uint16_t)di : (double)di;
            s += d * d;
          }
          if (s > maxa)
            maxa = s;
        }
      if (maxa > max_a_f)
        continue;
      double acc[3] = {0.0, 0.0, 0.0};
      if (p->use_dist_weighted_avg_fwd) {
        double sumw = 0.0;
        for (int i = 0; i < nn; i++) {
          const double w = 1 / (d2[i] + p->dist_offset_fwd);
          for (int k = 0; k < c; k++)
            acc[k] += (double)src_attrs[(size_t)idx[i] * c + k] * w;
          sumw += w;
        }
        for (int k = 0; k < c; k++)
          acc[k] /= sumw;
      } else {
        for (int i = 0; i < nn; i++)
          for (int k = 0; k < c; k++)
            acc[k] += (double)src_attrs[(size_t)idx[i] * c + k];
        for (int k = 0; k < c; k++)
          acc[k] /= nn;
      }
      for (int k = 0; k < c; k++)
        out[k] = (int32_t)clipd(round(acc[k]), 0.0, clip_max);
      break;
    }
  }

  /* ---- backward: every source point joins the lists of its nearest targets
   *      (:386-424 / 730-766) -------------------------------------------------------- */
  for (int s = 0; s < ns; s++) {
    double q[3];
    for (int a = 0; a < 3; a++)
      q[a] = (double)src_xyz[3 * s + a] * s2t - (double)offset[a];
    double d2[RC_MAXK];
    int32_t idx[RC_MAXK];
    const int n = kd_search(&gt, q, kb, d2, idx);
    if (n < 0)
      return -5;
    for (int i = 0; i < kb; i++) {
      const int ok = i < n && d2[i] <= max_g_b;
      bt[(size_t)s * kb + i] = ok ? idx[i] : -1;
      bd[(size_t)s * kb + i] = ok ? d2[i] : 0.0;
      if (ok)
        cnt[idx[i] + 1]++;
    }
  }
  for (int t = 0; t < nt; t++)
    cnt[t + 1] += cnt[t];
  const int total = cnt[nt];
  BwdEntry* list = (BwdEntry*)malloc(sizeof(BwdEntry) * (size_t)(total > 0 ? total : 1));
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * (size_t)nt);
  memcpy(cur, cnt, sizeof(int32_t) * (size_t)nt);
  for (int s = 0; s < ns; s++)
    for (int i = 0; i < kb; i++) {
      const int t = bt[(size_t)s * kb + i];
      if (t >= 0) {
        list[cur[t]].dist = bd[(size_t)s * kb + i];
        list[cur[t]].src = s;
        cur[t]++;
      }
    }

  /* ---- blend and refinement (:426-592 / 768-914) -------------------------------- */
  const double r_source = 1.0 / (double)ns;
  const double r_target = 1.0 / (double)nt;
  for (int t = 0; t < nt; t++) {
    BwdEntry* l = list + cnt[t];
    int n = cnt[t + 1] - cnt[t];
    const int32_t* c1 = ref1 + (size_t)t * c;
    int32_t* out = tgt_attrs + (size_t)t * c;
    if (n == 0) {
      for (int k = 0; k < c; k++)
        out[k] = c1[k];
      continue;
    }
    /* filled in source order (the loop above), then std::sort by distance (:416-422) */
    oracle_std_sort_by_dist(l, n);
    double cen2[3] = {0.0, 0.0, 0.0};
    int done = 0;
    if (p->skip_avg_if_identical_bwd && l[0].dist < 0.0001) {
      n = 1;
      for (int k = 0; k < c; k++)
        cen2[k] = (double)src_attrs[(size_t)l[0].src * c + k];
      done = 1;
    }
    while (!done) {
      if (n == 1) {
        for (int k = 0; k < c; k++)
          cen2[k] = (double)src_attrs[(size_t)l[0].src * c + k];
        break;
      }
      double maxa = 2.2250738585072014e-308;
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          double s = 0.0;
          for (int k = 0; k < c; k++) {
            const double d = (double)src_attrs[(size_t)l[i].src * c + k] - (double)src_attrs[(size_t)l[j].src * c + k];
            s += d * d;
          }
          if (s > maxa)
            maxa = s;
        }
      if (maxa <= max_a_b) {
        for (int k = 0; k < c; k++)
          cen2[k] = 0.0;
        if (p->use_dist_weighted_avg_bwd) {
          double sumw = 0.0;
          for (int i = 0; i < n; i++) {
            const double w = 1 / (sqrt(l[i].dist) + p->dist_offset_bwd);
            for (int k = 0; k < c; k++)
              cen2[k] += (double)src_attrs[(size_t)l[i].src * c + k] * w;
            sumw += w;
          }
          for (int k = 0; k < c; k++)
            cen2[k] /= sumw;
        } else {
          for (int i = 0; i < n; i++)
            for (int k = 0; k < c; k++)
              cen2[k] += (double)src_attrs[(size_t)l[i].src * c + k];
          for (int k = 0; k < c; k++)
            cen2[k] /= (double)n;
        }
        break;
      }
      n--;  /* the farthest entry leaves */
    }
    /* fixWeight (m42538): w = 0, the start value is the backward centroid */
    double c0[3], best[3], col[3];
    for (int k = 0; k < c; k++) {
      c0[k] = clipd(round(0.0 * (double)c1[k] + 1.0 * cen2[k]), 0.0, clip_max);
      best[k] = c0[k];
    }
    double min_err = big;
    const int sr = p->search_range;
    const int n1 = c == 3 ? sr : 0;
    for (int s1 = -sr; s1 <= sr; s1++) {
      col[0] = clipd(c0[0] + s1, 0.0, clip_max);
      for (int s2 = -n1; s2 <= n1; s2++) {
        if (c == 3)
          col[1] = clipd(c0[1] + s2, 0.0, clip_max);
        for (int s3 = -n1; s3 <= n1; s3++) {
          if (c == 3)
            col[2] = clipd(c0[2] + s3, 0.0, clip_max);
          double e1 = 0.0;
          for (int k = 0; k < c; k++) {
            const double d = col[k] - (double)c1[k];
            e1 += d * d;
          }
          e1 *= r_target;
          double e2 = 0.0;
          for (int i = 0; i < n; i++)
            for (int k = 0; k < c; k++) {
              const double d = col[k] - (double)src_attrs[(size_t)l[i].src * c + k];
              e2 += d * d;
            }
          e2 *= r_source;
          const double err = e1 > e2 ? e1 : e2;
          if (err < min_err) {
            min_err = err;
            for (int k = 0; k < c; k++)
              best[k] = col[k];
          }
        }
      }
    }
    for (int k = 0; k < c; k++)
      out[k] = (int32_t)best[k];
  }
  free(cur);
  free(list);
  free(bd);
  free(bt);
  free(cnt);
  free(ref1);
  kd_free(&gs);
  kd_free(&gt);
  return 0;
}

/* the restated sort on separate arrays (test support: compared with std::sort itself) */
void
oracle_std_sort_pairs(double* dist, int32_t* src, int32_t n)
{
  BwdEntry* a = (BwdEntry*)malloc(sizeof(BwdEntry) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; i++) {
    a[i].dist = dist[i];
    a[i].src = src[i];
  }
  oracle_std_sort_by_dist(a, n);
  for (int i = 0; i < n; i++) {
    dist[i] = a[i].dist;
    src[i] = a[i].src;
  }
  free(a);
}
