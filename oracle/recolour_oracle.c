/* recolour_oracle.c -- TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of pcc::recolour (tmc3/pointset_processing.cpp:926-957:
 * recolourColour :253-594, recolourReflectance :618-916) in the form the device
 * kernels use: exact k-nearest-neighbour search over a uniform grid instead of the
 * nanoflann k-d tree, every floating-point expression of the reference evaluated
 * in double in the reference's order, and ONE rule where the reference's outcome
 * depends on container internals: among equidistant candidates the lower point
 * index comes first (nanoflann keeps whichever its tree visits first; std::sort
 * leaves equal keys in unspecified order).  The device path has to match this file
 * bit for bit; this file matches the compiled reference (oracle/_ref) wherever no
 * tie decides -- tests/test_oracle_recolour.py measures both.
 *
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may use it. */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"

#define RC_MAXK 8

typedef struct {
  int shift;           /* cell side = 1 << shift */
  int dim[3];          /* cells per axis */
  int lo[3];           /* first cell coordinate */
  int32_t* start;      /* [cells + 1] */
  int32_t* items;      /* [n] point indices, ascending inside a cell */
  const int32_t* xyz;
  int n;
} Grid;

static int
grid_build(Grid* g, const int32_t* xyz, int n)
{
  int mn[3], mx[3];
  for (int k = 0; k < 3; k++)
    mn[k] = mx[k] = xyz[k];
  for (int i = 1; i < n; i++)
    for (int k = 0; k < 3; k++) {
      if (xyz[3 * i + k] < mn[k])
        mn[k] = xyz[3 * i + k];
      if (xyz[3 * i + k] > mx[k])
        mx[k] = xyz[3 * i + k];
    }
  /* the smallest cell for which the table stays below ~4 n cells */
  int shift = 0;
  for (;; shift++) {
    double cells = 1;
    for (int k = 0; k < 3; k++)
      cells *= (double)((mx[k] >> shift) - (mn[k] >> shift) + 1);
    if (cells <= 4.0 * n + 64)
      break;
  }
  g->shift = shift;
  g->xyz = xyz;
  g->n = n;
  size_t cells = 1;
  for (int k = 0; k < 3; k++) {
    g->lo[k] = mn[k] >> shift;
    g->dim[k] = (mx[k] >> shift) - g->lo[k] + 1;
    cells *= (size_t)g->dim[k];
  }
  g->start = (int32_t*)calloc(cells + 1, sizeof(int32_t));
  g->items = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  if (!g->start || !g->items)
    return -1;
#define CELL(i)                                                                      \
  ((((size_t)((xyz[3 * (i)] >> shift) - g->lo[0])) * g->dim[1]                      \
    + (size_t)((xyz[3 * (i) + 1] >> shift) - g->lo[1]))                             \
     * g->dim[2]                                                                     \
   + (size_t)((xyz[3 * (i) + 2] >> shift) - g->lo[2]))
  for (int i = 0; i < n; i++)
    g->start[CELL(i) + 1]++;
  for (size_t c = 0; c < cells; c++)
    g->start[c + 1] += g->start[c];
  int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * cells);
  if (!cur)
    return -1;
  memcpy(cur, g->start, sizeof(int32_t) * cells);
  for (int i = 0; i < n; i++)
    g->items[cur[CELL(i)]++] = i;
#undef CELL
  free(cur);
  return 0;
}

static void
grid_free(Grid* g)
{
  free(g->start);
  free(g->items);
}

/* candidate (d2, idx) into the ascending list of at most k entries; ties by index */
static void
knn_insert(double* d2, int32_t* idx, int* count, int k, double d, int32_t i)
{
  int pos = *count;
  if (pos == k) {
    if (d > d2[k - 1] || (d == d2[k - 1] && i > idx[k - 1]))
      return;
    pos = k - 1;
  } else {
    (*count)++;
  }
  while (pos > 0 && (d2[pos - 1] > d || (d2[pos - 1] == d && idx[pos - 1] > i))) {
    d2[pos] = d2[pos - 1];
    idx[pos] = idx[pos - 1];
    pos--;
  }
  d2[pos] = d;
  idx[pos] = i;
}

/* the k nearest points of g to q: ring after ring of cells around q's cell until
 * the k-th distance is not larger than what any unvisited cell can offer */
static int
knn(const Grid* g, const double q[3], int k, double* d2, int32_t* idx)
{
  const int cs = 1 << g->shift;
  int cq[3];
  for (int a = 0; a < 3; a++)
    cq[a] = (int)floor(q[a] / cs) - g->lo[a];
  int count = 0;
  /* no point can be nearer than the distance of q to the grid's box */
  int maxr = 0;
  for (int a = 0; a < 3; a++) {
    int far = cq[a] > g->dim[a] - 1 - cq[a] ? cq[a] : g->dim[a] - 1 - cq[a];
    if (far < 0)
      far = -far;
    if (far > maxr)
      maxr = far;
  }
  for (int r = 0; r <= maxr; r++) {
    for (int dx = -r; dx <= r; dx++) {
      const int cx = cq[0] + dx;
      if (cx < 0 || cx >= g->dim[0])
        continue;
      for (int dy = -r; dy <= r; dy++) {
        const int cy = cq[1] + dy;
        if (cy < 0 || cy >= g->dim[1])
          continue;
        const int shell = (dx == -r || dx == r || dy == -r || dy == r);
        for (int dz = -r; dz <= r; dz += (shell || r == 0) ? 1 : 2 * r) {
          const int cz = cq[2] + dz;
          if (cz < 0 || cz >= g->dim[2])
            continue;
          const size_t c = ((size_t)cx * g->dim[1] + (size_t)cy) * g->dim[2] + (size_t)cz;
          for (int32_t e = g->start[c]; e < g->start[c + 1]; e++) {
            const int32_t i = g->items[e];
            /* nanoflann L2_Simple_Adaptor: result += diff * diff, x then y then z */
            double s = 0.0;
            for (int a = 0; a < 3; a++) {
              const double diff = q[a] - (double)g->xyz[3 * i + a];
              s += diff * diff;
            }
            knn_insert(d2, idx, &count, k, s, i);
          }
        }
      }
    }
    /* a point of ring r + 1 or beyond differs by more than r * cs in some axis */
    const double bound = (double)r * cs;
    if (count == k && d2[k - 1] <= bound * bound)
      break;
  }
  return count;
}

static double
clipd(double v, double lo, double hi)
{
  return v < lo ? lo : (v > hi ? hi : v);
}

typedef struct {
  double dist;
  int32_t src;
} BwdEntry;

static int
bwd_cmp(const void* a, const void* b)
{
  const BwdEntry* x = (const BwdEntry*)a;
  const BwdEntry* y = (const BwdEntry*)b;
  if (x->dist != y->dist)
    return x->dist < y->dist ? -1 : 1;
  return x->src < y->src ? -1 : (x->src > y->src ? 1 : 0);
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
  /* A finite forward geometry limit makes the reference shrink its result vectors
   * for every LATER target point as well (indicesFwd / sqrDistFwd live outside the
   * loop, :292-309): state that leaks from point to point, not restated. */
  if (p->max_geometry_dist2_fwd < 512)
    return -2;
  const double s2t = (double)scale_f;
  const double t2s = 1.0 / s2t;
  const double clip_max = (double)((1 << p->bitdepth) - 1);
  const double big = 1.7976931348623157e308;
  const double max_g_b = p->max_geometry_dist2_bwd < 512 ? p->max_geometry_dist2_bwd : big;
  const double max_a_f = p->max_attribute_dist2_fwd < 512 ? p->max_attribute_dist2_fwd : big;
  const double max_a_b = p->max_attribute_dist2_bwd < 512 ? p->max_attribute_dist2_bwd : big;

  Grid gs, gt;
  if (grid_build(&gs, src_xyz, ns) || grid_build(&gt, tgt_xyz, nt))
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
    const int n = knn(&gs, q, kf, d2, idx);
    int32_t* out = ref1 + (size_t)t * c;
    if (p->skip_avg_if_identical_fwd && d2[0] < 0.0001) {
      for (int k = 0; k < c; k++)
        out[k] = src_attrs[(size_t)idx[0] * c + k];
      continue;
    }
    for (int nn = n; nn > 0; nn--) {
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
    const int n = knn(&gt, q, kb, d2, idx);
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
    qsort(l, (size_t)n, sizeof(BwdEntry), bwd_cmp);
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
  grid_free(&gs);
  grid_free(&gt);
  return 0;
}

/* Where a TIE decides (the only places the reference may differ from this file):
 * flags[t] |= 1  the forward search of target t has equidistant candidates at the
 *                K-th place;
 *          |= 2  a source point has equidistant targets at the last place of its
 *                backward search and t is one of them;
 *          |= 4  the backward list of t holds equal distances (summation / truncation
 *                order).  Test support only. */
int
oracle_recolour_ties(
  const gpcc_recolour_params* p, const int32_t* src_xyz, int32_t ns, const int32_t* tgt_xyz,
  int32_t nt, float scale_f, const int32_t offset[3], uint8_t* flags)
{
  const int kf = p->num_neighbours_fwd, kb = p->num_neighbours_bwd;
  if (kf < 1 || kf > RC_MAXK || kb < 1 || kb > RC_MAXK)
    return -2;
  const double s2t = (double)scale_f;
  const double t2s = 1.0 / s2t;
  Grid gs, gt;
  if (grid_build(&gs, src_xyz, ns) || grid_build(&gt, tgt_xyz, nt))
    return -4;
  memset(flags, 0, (size_t)nt);
  enum { TIEK = 24 };
  double d2[TIEK];
  int32_t idx[TIEK];
  for (int t = 0; t < nt; t++) {
    double q[3];
    for (int a = 0; a < 3; a++)
      q[a] = (double)(tgt_xyz[3 * t + a] + offset[a]) * t2s;
    if (ns > kf) {
      const int n = knn(&gs, q, kf + 1, d2, idx);
      if (n == kf + 1 && d2[kf] == d2[kf - 1])
        flags[t] |= 1;
      /* equal distances inside the set: the order of the weighted sum */
      for (int i = 0; i + 1 < kf; i++)
        if (d2[i] == d2[i + 1])
          flags[t] |= 8;
    }
  }
  int32_t* et = (int32_t*)malloc(sizeof(int32_t) * ((size_t)ns * kb + 1));
  double* ed = (double*)malloc(sizeof(double) * ((size_t)ns * kb + 1));
  size_t ne = 0;
  for (int s = 0; s < ns; s++) {
    double q[3];
    for (int a = 0; a < 3; a++)
      q[a] = (double)src_xyz[3 * s + a] * s2t - (double)offset[a];
    /* (a source point between lattice positions has up to 8 equidistant targets) */
    const int want = nt > TIEK ? TIEK : nt;
    const int n = knn(&gt, q, want, d2, idx);
    if (n > kb && d2[kb] == d2[kb - 1])
      for (int i = 0; i < n; i++)
        if (d2[i] == d2[kb - 1])
          flags[idx[i]] |= 2;
    for (int i = 0; i < kb && i < n; i++) {
      et[ne] = idx[i];
      ed[ne] = d2[i];
      ne++;
    }
  }
  /* equal distances inside a target's list: bucket the entries by target */
  {
    int32_t* cnt = (int32_t*)calloc((size_t)nt + 1, sizeof(int32_t));
    for (size_t i = 0; i < ne; i++)
      cnt[et[i] + 1]++;
    for (int t = 0; t < nt; t++)
      cnt[t + 1] += cnt[t];
    double* by = (double*)malloc(sizeof(double) * (ne + 1));
    int32_t* cur = (int32_t*)malloc(sizeof(int32_t) * ((size_t)nt + 1));
    memcpy(cur, cnt, sizeof(int32_t) * (size_t)nt);
    for (size_t i = 0; i < ne; i++)
      by[cur[et[i]]++] = ed[i];
    for (int t = 0; t < nt; t++)
      for (int i = cnt[t]; i < cnt[t + 1]; i++)
        for (int j = i + 1; j < cnt[t + 1]; j++)
          if (by[i] == by[j])
            flags[t] |= 4;
    free(cur);
    free(by);
    free(cnt);
  }
  free(ed);
  free(et);
  grid_free(&gs);
  grid_free(&gt);
  return 0;
}
