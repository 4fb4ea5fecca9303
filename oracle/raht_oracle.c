/* raht_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the intra RAHT forward / inverse transform of
 * TMC13 (tmc3/RAHT.cpp:977-1976, `uraht_process`) in LEVEL-SYNCHRONOUS form:
 * instead of the reference's Lf/Hf stack shuffling (reduceLevel :157 /
 * expandLevel :210) every octree level is materialised once as a
 * structure-of-arrays (key = pos >> level, weight, attribute sum, qp,
 * child range) and the descent walks those arrays.  This is the blueprint
 * the HIP kernels follow and the checker the GPU results are compared
 * with.  It is pinned against the compiled reference
 * (oracle/_ref/libtmc3_ref.so) by tests/test_oracle_raht.py.
 *
 * Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
 * load this file's shared object.  Inter-frame branches of the reference
 * (enableAttrInterPred) are not restated: intra slices only.
 */
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"
#include "primitives.h"

/* one octree level of the RAHT tree, nodes in Morton order */
typedef struct {
  int shift;             /* key = pos >> shift                               */
  int m;                 /* node count                                       */
  int64_t* key;          /* [m]                                              */
  int32_t* weight;       /* [m]   number of points below the node           */
  int32_t* attr;         /* [m*c] attribute sum / integer-Haar low-pass      */
  int32_t* qp;           /* [m*2] region qp offsets, << 4 (RAHT.cpp:1045)    */
  int32_t* first_child;  /* [m+1] range in the next finer level (or points)  */
} level_t;

static void
level_alloc(level_t* l, int m, int c, int shift)
{
  l->shift = shift;
  l->m = m;
  l->key = (int64_t*)malloc(sizeof(int64_t) * (m + 1));
  l->weight = (int32_t*)malloc(sizeof(int32_t) * (m + 1));
  l->attr = (int32_t*)malloc(sizeof(int32_t) * (m + 1) * c);
  l->qp = (int32_t*)malloc(sizeof(int32_t) * (m + 1) * 2);
  l->first_child = (int32_t*)malloc(sizeof(int32_t) * (m + 2));
}

static void
level_free(level_t* l)
{
  free(l->key);
  free(l->weight);
  free(l->attr);
  free(l->qp);
  free(l->first_child);
}

static inline int32_t
wrap_add(int32_t a, int32_t b)
{
  return (int32_t)((uint32_t)a + (uint32_t)b);
}
static inline int32_t
wrap_sub(int32_t a, int32_t b)
{
  return (int32_t)((uint32_t)a - (uint32_t)b);
}

/* Level 0: merge exact duplicates (reduceUnique, RAHT.cpp:108-152). */
static void
build_leaf_level(
  level_t* l, const int64_t* pos, const int32_t* qp_off,
  const int32_t* attrs, int n, int c, int haar, int encoder)
{
  int m = 0;
  for (int i = 0; i < n; i++) {
    if (i == 0 || pos[i] != pos[i - 1]) {
      l->key[m] = pos[i];
      l->weight[m] = 1;
      l->qp[2 * m + 0] = qp_off ? qp_off[2 * i + 0] * 16 : 0;
      l->qp[2 * m + 1] = qp_off ? qp_off[2 * i + 1] * 16 : 0;
      for (int k = 0; k < c; k++)
        l->attr[m * c + k] = encoder ? attrs[i * c + k] : 0;
      l->first_child[m] = i;
      m++;
      continue;
    }
    l->weight[m - 1]++;
    if (!encoder)
      continue;
    for (int k = 0; k < c; k++) {
      int32_t* acc = &l->attr[(m - 1) * c + k];
      if (haar) {
        int32_t d = wrap_sub(attrs[i * c + k], *acc);
        *acc = wrap_add(*acc, d >> 1);
      } else {
        *acc = wrap_add(*acc, attrs[i * c + k]);
      }
    }
  }
  l->first_child[m] = n;
  l->m = m;
}

/* Three binary reduceLevel passes (RAHT.cpp:157-205) fused into one
 * 2x2x2 step: children are merged pairwise along z, then y, then x. */
static void
build_parent_level(level_t* up, const level_t* lo, int c, int haar)
{
  int m = 0;
  for (int i = 0; i < lo->m;) {
    int64_t pkey = lo->key[i] >> 3;
    int32_t w[8] = {0}, a[8][3] = {{0}}, q[8][2] = {{0}};
    int j = i;
    for (; j < lo->m && (lo->key[j] >> 3) == pkey; j++) {
      int idx = (int)(lo->key[j] & 7);
      w[idx] = lo->weight[j];
      q[idx][0] = lo->qp[2 * j];
      q[idx][1] = lo->qp[2 * j + 1];
      for (int k = 0; k < c; k++)
        a[idx][k] = lo->attr[j * c + k];
    }
    for (int step = 1; step < 8; step <<= 1) {
      for (int l = 0; l < 8; l += 2 * step) {
        int r = l + step;
        if (!w[r])
          continue;
        if (!w[l]) {
          w[l] = w[r];
          q[l][0] = q[r][0];
          q[l][1] = q[r][1];
          for (int k = 0; k < c; k++)
            a[l][k] = a[r][k];
          w[r] = 0;
          continue;
        }
        w[l] += w[r];
        q[l][0] = (q[l][0] + q[r][0]) >> 1;
        q[l][1] = (q[l][1] + q[r][1]) >> 1;
        for (int k = 0; k < c; k++) {
          if (haar) {
            int32_t d = wrap_sub(a[r][k], a[l][k]);
            a[l][k] = wrap_add(a[l][k], d >> 1);
          } else {
            a[l][k] = wrap_add(a[l][k], a[r][k]);
          }
        }
        w[r] = 0;
      }
    }
    up->key[m] = pkey;
    up->weight[m] = w[0];
    up->qp[2 * m] = q[0][0];
    up->qp[2 * m + 1] = q[0][1];
    for (int k = 0; k < c; k++)
      up->attr[m * c + k] = a[0][k];
    up->first_child[m] = i;
    m++;
    i = j;
  }
  up->first_child[m] = lo->m;
  up->m = m;
}

/* ---- 2x2x2 block transform ------------------------------------------- */

/* butterfly order of fwdTransformBlock222 (RAHT.cpp:676-677) */
static const int8_t kBflyL[12] = {0, 2, 4, 6, 0, 4, 1, 5, 0, 1, 2, 3};
static const int8_t kBflyR[12] = {1, 3, 5, 7, 2, 6, 3, 7, 4, 5, 6, 7};

typedef struct {
  int32_t wl[12], wr[12]; /* weights entering each butterfly             */
  int32_t cw[8];          /* weight of the coefficient left at position  */
} block_weights_t;

/* mkWeightTree (RAHT.cpp:742-771) expressed per butterfly */
static void
block_weights(const int32_t w[8], block_weights_t* bw)
{
  int32_t cw[8];
  memcpy(cw, w, sizeof(cw));
  for (int i = 0; i < 12; i++) {
    int l = kBflyL[i], r = kBflyR[i];
    bw->wl[i] = cw[l];
    bw->wr[i] = cw[r];
    if (cw[l] && cw[r]) {
      cw[l] = cw[r] = cw[l] + cw[r];
    } else {
      cw[l] = cw[l] + cw[r];
      cw[r] = 0;
    }
  }
  memcpy(bw->cw, cw, sizeof(cw));
}

/* RahtKernel ctor (RAHT.cpp:596-604) */
static void
raht_coeffs(int32_t wl, int32_t wr, int64_t* a, int64_t* b)
{
  uint64_t w = (uint64_t)wl + (uint64_t)wr;
  uint64_t rs = irsqrt_u64(w);
  *a = (int64_t)(((uint64_t)isqrt_u64((uint64_t)wl << 30) * rs) >> 40);
  *b = (int64_t)(((uint64_t)isqrt_u64((uint64_t)wr << 30) * rs) >> 40);
}

/* fwdTransformBlock222 (RAHT.cpp:671-701) over nbuf buffers */
static void
block_fwd(int nbuf, int64_t buf[][8], const block_weights_t* bw, int haar)
{
  for (int i = 0; i < 12; i++) {
    int l = kBflyL[i], r = kBflyR[i];
    int32_t wl = bw->wl[i], wr = bw->wr[i];
    if (!wl && !wr)
      continue;
    if (!wl || !wr) {
      if (!wl)
        for (int k = 0; k < nbuf; k++) {
          int64_t t = buf[k][l];
          buf[k][l] = buf[k][r];
          buf[k][r] = t;
        }
      continue;
    }
    if (haar) {
      /* HaarKernel::fwdTransform RAHT.cpp:653-658 */
      for (int k = 0; k < nbuf; k++) {
        int64_t hf = buf[k][r] - buf[k][l];
        buf[k][l] += (hf >> (1 + FP_FRAC)) << FP_FRAC;
        buf[k][r] = hf;
      }
      continue;
    }
    int64_t a, b;
    raht_coeffs(wl, wr, &a, &b);
    for (int k = 0; k < nbuf; k++) {
      /* RahtKernel::fwdTransform RAHT.cpp:606-623 */
      int64_t left = buf[k][l], right = buf[k][r];
      buf[k][l] = fp_mul(right, b) + fp_mul(a, left);
      buf[k][r] = fp_mul(right, a) - fp_mul(b, left);
    }
  }
}

/* invTransformBlock222 (RAHT.cpp:707-737) */
static void
block_inv(int nbuf, int64_t buf[][8], const block_weights_t* bw, int haar)
{
  for (int i = 11; i >= 0; i--) {
    int l = kBflyL[i], r = kBflyR[i];
    int32_t wl = bw->wl[i], wr = bw->wr[i];
    if (!wl && !wr)
      continue;
    if (!wl || !wr) {
      if (!wl)
        for (int k = 0; k < nbuf; k++) {
          int64_t t = buf[k][l];
          buf[k][l] = buf[k][r];
          buf[k][r] = t;
        }
      continue;
    }
    if (haar) {
      /* HaarKernel::invTransform RAHT.cpp:660-665 */
      for (int k = 0; k < nbuf; k++) {
        int64_t lf = buf[k][l], hf = buf[k][r];
        int64_t left = lf - ((hf >> (1 + FP_FRAC)) << FP_FRAC);
        buf[k][l] = left;
        buf[k][r] = hf + left;
      }
      continue;
    }
    int64_t a, b;
    raht_coeffs(wl, wr, &a, &b);
    for (int k = 0; k < nbuf; k++) {
      /* RahtKernel::invTransform RAHT.cpp:625-640 */
      int64_t lf = buf[k][l], hf = buf[k][r];
      buf[k][l] = fp_mul(lf, a) - fp_mul(b, hf);
      buf[k][r] = fp_mul(lf, b) + fp_mul(a, hf);
    }
  }
}

/* ---- quantiser selection (QpSet::quantizers quantization.cpp:165-174) -- */
static void
qpset_quantizers(
  const gpcc_raht_params* p, int layer, int off0, int off1, quantizer_t q[2])
{
  int qp0 = clip_int(p->layer_qp[layer][0] + off0, 4, p->max_qp);
  int qp1 = clip_int(p->layer_qp[layer][1] + off1 + qp0, 4, p->max_qp);
  q[0] = quantizer_make(qp0 + p->fixed_point_qp_offset);
  q[1] = quantizer_make(qp1 + p->fixed_point_qp_offset);
}

/* ---- neighbour search (findNeighbour/findNeighbours RAHT.cpp:272-416) -- */

static const uint8_t kNeighMask[19] = {255, 240, 204, 170, 192, 160, 136,
                                       3,   5,   15,  17,  51,  85,  10,
                                       34,  12,  68,  48,  80};
static const uint8_t kNeighOffset[19] = {0,  35, 21, 14, 49, 42, 28, 1,  2, 3,
                                         4,  5,  6,  10, 12, 17, 20, 33, 34};
static const uint8_t kOccuMask[12] = {3, 5, 15, 17, 51, 85, 10, 34, 12, 68,
                                      48, 80};
static const uint8_t kOccuShift[12] = {6, 5, 4, 3, 2, 1, 3, 1, 2, 1, 2, 3};

static int
find_in_window(const int64_t* key, int m, int from, int64_t value, int64_t d)
{
  int lo, end;
  if (d >= 0) {
    lo = from;
    end = (d + 1 < (int64_t)(m - from)) ? from + (int)(d + 1) : m;
  } else {
    end = from;
    lo = (-d < (int64_t)from) ? from - (int)(-d) : 0;
  }
  int hi = end;
  while (lo < hi) {
    int mid = lo + ((hi - lo) >> 1);
    if (key[mid] < value)
      lo = mid + 1;
    else
      hi = mid;
  }
  if (lo == end)
    return -1;
  return key[lo] == value ? lo : -1;
}

/* kDivisors of intraDcPred (RAHT.cpp:445-451) = round(32768 / (i + 1)) */
static inline int64_t
pred_divisor(int weight_sum_minus1)
{
  int d = weight_sum_minus1 + 1;
  return (32768 + d / 2) / d;
}

/* ---- the transform ---------------------------------------------------- */

typedef struct {
  int64_t* rec;     /* scaled reconstruction per node   [m*c] (attrRec)   */
  int64_t* rec_us;  /* unscaled reconstruction per node [m*c] (attrRecUs) */
  int32_t* nneigh;  /* numParentNeigh per node          [m]               */
  int32_t* dqp;     /* descent-time node qp (<< 4)       [m*2], see below  */
} recon_t;

static void
recon_alloc(recon_t* r, int m, int c)
{
  r->rec = (int64_t*)calloc((size_t)(m + 1) * c, sizeof(int64_t));
  r->rec_us = (int64_t*)calloc((size_t)(m + 1) * c, sizeof(int64_t));
  r->nneigh = (int32_t*)calloc((size_t)(m + 1), sizeof(int32_t));
  r->dqp = (int32_t*)calloc((size_t)(m + 1) * 2, sizeof(int32_t));
}
static void
recon_free(recon_t* r)
{
  free(r->rec);
  free(r->rec_us);
  free(r->nneigh);
  free(r->dqp);
}

/* Node QPs on the way DOWN the tree.  reduceLevel (RAHT.cpp:185-189)
 * overwrites the left node of a merged pair with the pair average and
 * expandLevel (:246-253) restores weight and attributes but not qp: after
 * expansion the LEFT node of every pair keeps its parent's (descent-time)
 * qp, the RIGHT node the value it had when it was pushed (its own ascent
 * average).  Given the ascent averages A[] of a block's children and the
 * descent qp of the block's parent, derive the children's descent qps
 * through the three binary stages x, y, z. */
static void
descend_block_qp(
  const int32_t w[8], const int32_t a[8][2], const int32_t parent[2],
  int32_t d[8][2])
{
  int32_t w1[4], a1[4][2], w2[2], a2[2][2], d1[4][2], d2[2][2];
  for (int t = 0; t < 4; t++) {
    int l = 2 * t, r = l + 1;
    w1[t] = w[l] + w[r];
    for (int k = 0; k < 2; k++)
      a1[t][k] = (w[l] && w[r]) ? (a[l][k] + a[r][k]) >> 1
                                : (w[l] ? a[l][k] : a[r][k]);
  }
  for (int u = 0; u < 2; u++) {
    int l = 2 * u, r = l + 1;
    w2[u] = w1[l] + w1[r];
    for (int k = 0; k < 2; k++)
      a2[u][k] = (w1[l] && w1[r]) ? (a1[l][k] + a1[r][k]) >> 1
                                  : (w1[l] ? a1[l][k] : a1[r][k]);
  }
  for (int k = 0; k < 2; k++) {
    d2[0][k] = parent[k];
    d2[1][k] = (w2[0] && w2[1]) ? a2[1][k] : parent[k];
    for (int u = 0; u < 2; u++) {
      int l = 2 * u, r = l + 1;
      d1[l][k] = d2[u][k];
      d1[r][k] = (w1[l] && w1[r]) ? a1[r][k] : d2[u][k];
    }
    for (int t = 0; t < 4; t++) {
      int l = 2 * t, r = l + 1;
      d[l][k] = d1[t][k];
      d[r][k] = (w[l] && w[r]) ? a[r][k] : d1[t][k];
    }
  }
}

/* children with weight > 1 are normalised by 1/sqrt(w)
 * (RAHT.cpp:1474-1481 and :1780-1787) */
static inline int64_t
scale_rsqrt(int64_t v, int32_t weight)
{
  uint64_t w = (uint64_t)weight;
  int shift = w > 1024 ? ilog2_u64(w - 1) >> 1 : 0;
  int64_t rs = (int64_t)(irsqrt_u64(w) >> (40 - shift - FP_FRAC));
  return fp_mul(v >> shift, rs);
}

/* PCCRAHTACCoefficientEntropyEstimate (RAHT.h:71-94, RAHT.cpp:54-91): the adaptive
 * rate model behind the per-layer inter / intra decision; doubles, summed in coding order */
typedef struct {
  int p0[3], p1[3];
  double bits;
} ac_estimate_t;

static void
ac_estimate_init(ac_estimate_t* e)
{
  for (int k = 0; k < 3; k++)
    e->p0[k] = e->p1[k] = (1 << 20) >> 1;
  e->bits = 0.;
}

static void
ac_estimate_cost(ac_estimate_t* e, int32_t value, int k)
{
  const unsigned scale = 1u << 20;
  const int lg = 20;
  double bits = 0;
  bits += value ? lg - log2((double)e->p0[k]) : lg - log2((double)(scale - (unsigned)e->p0[k]));
  const int mag = abs(value);
  if (mag) {
    bits += mag > 1 ? lg - log2((double)e->p1[k]) : lg - log2((double)(scale - (unsigned)e->p1[k]));
    bits += 1;
    if (mag > 1)
      bits += 2.0 * log2(mag - 1.0) + 1.0;
  }
  e->bits += bits;
}

static void
ac_estimate_update(ac_estimate_t* e, int32_t value, int k)
{
  const unsigned scale = 1u << 20;
  e->p0[k] += value ? (int)((scale - (unsigned)e->p0[k]) >> 6) : -(e->p0[k] >> 6);
  if (value)
    e->p1[k] += abs(value) > 1 ? (int)((scale - (unsigned)e->p1[k]) >> 6) : -(e->p1[k] >> 6);
}

/* attribute inter prediction (AttributeInterPredParams / paramsForInterRAHT,
 * PCCTMC3Common.h:236-298): the reference frame in Morton order and the tools */
typedef struct {
  const int64_t* pos;   /* [n] */
  const int32_t* attrs; /* [n][c] */
  int n;
  int depth_limit;      /* raht_inter_prediction_depth_minus1 + 1 */
  int layer_rdo;        /* raht_enable_inter_intra_layer_RDO */
  int filter_est;       /* enableFilterEstimation */
  int skip_layers;      /* skipInitLayersForFiltering */
  int32_t* layer_modes; /* attr_layer_code_mode: encoder out, decoder in */
  int32_t* num_modes;
  int32_t* filter_taps; /* FilterTaps: encoder out, decoder in */
  int32_t* num_taps;
} raht_inter_t;

/* the nodes of the reference frame's tree at bit level `lv` (unique pos >> lv): what
 * weightsLf_ref / attrsLf_ref hold when the descent has reached that level
 * (RAHT.cpp:1065-1120, 1185-1195); attribute sums wrap like the reference's ints */
typedef struct {
  int m;
  int64_t* key;
  int32_t* weight;
  int32_t* attr;
} ref_nodes_t;

static void
ref_nodes_build(ref_nodes_t* r, const raht_inter_t* ir, int lv, int c, int haar)
{
  r->key = (int64_t*)malloc(sizeof(int64_t) * (size_t)ir->n);
  r->weight = (int32_t*)malloc(sizeof(int32_t) * (size_t)ir->n);
  r->attr = (int32_t*)malloc(sizeof(int32_t) * (size_t)ir->n * c);
  int m = ir->n;
  for (int i = 0; i < m; i++) {
    r->key[i] = ir->pos[i];
    r->weight[i] = 1;
    for (int t = 0; t < c; t++)
      r->attr[i * c + t] = ir->attrs[(size_t)i * c + t];
  }
  /* the ascent one pass at a time: duplicates (reduceUnique :108-150), then one bit per
   * pass (reduceLevel :155-207).  A node that joins the one on its left adds its
   * attributes -- or, with the integer Haar kernel, half its difference -- so under Haar
   * the order of the passes is part of the result. */
  for (int pass = 0; pass <= lv; pass++) {
    int out = 0;
    for (int i = 0; i < m; i++) {
      if (out && ((r->key[out - 1] ^ r->key[i]) >> pass) == 0) {
        r->weight[out - 1] += r->weight[i];
        for (int t = 0; t < c; t++) {
          int32_t* left = &r->attr[(out - 1) * c + t];
          if (haar)
            *left = wrap_add(*left, wrap_sub(r->attr[i * c + t], *left) >> 1);
          else
            *left = wrap_add(*left, r->attr[i * c + t]);
        }
      } else {
        r->key[out] = r->key[i];
        r->weight[out] = r->weight[i];
        for (int t = 0; t < c; t++)
          r->attr[out * c + t] = r->attr[i * c + t];
        out++;
      }
    }
    m = out;
  }
  for (int i = 0; i < m; i++)
    r->key[i] >>= lv;
  r->m = m;
}

static void
ref_nodes_free(ref_nodes_t* r)
{
  free(r->key);
  free(r->weight);
  free(r->attr);
}

/* getFilterTap (RAHT.cpp:805-846): 128 * crosscorr / autocorr by subtraction and bisection */
static int
filter_tap_of(int64_t autocorr, int64_t crosscorr)
{
  if (crosscorr == 0)
    return 0;
  const int neg = crosscorr < 0;
  crosscorr = crosscorr < 0 ? -crosscorr : crosscorr;
  if (crosscorr == autocorr)
    return neg ? -128 : 128;
  int tapint = 0;
  while (crosscorr >= autocorr) {
    crosscorr -= autocorr;
    tapint += 128;
  }
  if (crosscorr == 0)
    return neg ? -tapint : tapint;
  int lo = 0, hi = 128;
  while (lo < hi - 1) {
    const int mid = (lo + hi) >> 1;
    const int64_t midval = (mid * autocorr) >> 7;
    if (crosscorr == midval)
      return neg ? -(tapint + mid) : (tapint + mid);
    if (crosscorr < midval)
      hi = mid;
    else
      lo = mid;
  }
  return neg ? -(tapint + lo) : (tapint + lo);
}

static int
bitlen_i64(int64_t v)
{
  int b = 0;
  while (v) {
    b++;
    v = (int64_t)((uint64_t)v >> 1);
  }
  return b;
}

static int
raht_process(
  int encoder, const gpcc_raht_params* p, const int64_t* pos,
  const int32_t* qp_off, int32_t* attributes, int32_t* coeffs, int n, int c,
  const raht_inter_t* ir)
{
  const int haar = p->integer_haar_enable_flag != 0;
  const int ext = p->raht_extension != 0;

  int32_t* coef_it[3] = {coeffs, coeffs + n, coeffs + 2 * (size_t)n};

  /* single point: RAHT.cpp:998-1017 */
  if (n == 1) {
    quantizer_t q[2];
    qpset_quantizers(
      p, 0, qp_off ? qp_off[0] : 0, qp_off ? qp_off[1] : 0, q);
    for (int k = 0; k < c; k++) {
      quantizer_t qk = q[k < 1 ? k : 1];
      int64_t coeff;
      if (encoder) {
        coeff = quantizer_quantize(qk, (int64_t)attributes[k] << 8);
        coeffs[(size_t)k * n] = (int32_t)coeff;
      } else {
        coeff = coeffs[(size_t)k * n];
      }
      attributes[k] =
        (int32_t)div_exp2_round_half_up(quantizer_scale(qk, coeff), 8);
    }
    return 0;
  }

  /* ---- ascend: materialise every octree level ---- */
  enum { kMaxLevels = 24 };
  level_t lv[kMaxLevels];
  int nlv = 0;
  level_alloc(&lv[0], n, c, 0);
  build_leaf_level(&lv[0], pos, qp_off, attributes, n, c, haar, encoder);
  nlv = 1;
  while (lv[nlv - 1].m > 1) {
    level_alloc(&lv[nlv], lv[nlv - 1].m, c, 3 * nlv);
    build_parent_level(&lv[nlv], &lv[nlv - 1], c, haar);
    nlv++;
  }
  /* lv[nlv-1] is the root (one node); the root block's children are
   * lv[nlv-2].  With a single unique position (nlv == 1) the level loop
   * of the reference does not run at all (RAHT.cpp:1165: level > 0). */
  const int num_unique = lv[0].m;

  recon_t cur, par;
  recon_alloc(&cur, num_unique, c);
  recon_alloc(&par, num_unique, c);

  /* the root keeps its ascent average */
  cur.dqp[0] = lv[nlv - 1].qp[0];
  cur.dqp[1] = lv[nlv - 1].qp[1];
  /* all points identical: the only node is also the root */
  int qp_layer = 0;
  int ac_layer = -1;
  int train_zeros = 0;
  const int max_ac_layers = p->num_ac_qp_layers - 1;
  int first = 1;
  int last_done = -1; /* index in lv[] of the most recent processed level */

  /* inter prediction: the two trees descend in lock step from their OWN tops
   * (RAHT.cpp:1165-1198): B / B_ref = number of bit levels below the single root */
  int tree_depth = 0;
  int depth = 0; /* index of the next per-layer mode */
  int intra_train_zeros = 0;
  ac_estimate_t est_cur, est_intra;
  ac_estimate_init(&est_cur);
  ac_estimate_init(&est_intra);
  /* the intra candidate of a level under the per-layer decision: its reconstruction
   * (intraAttrRec / intraAttrRecUs) and its coefficients (intraACCoeffcients) */
  int64_t *irec = NULL, *irec_us = NULL;
  int32_t* icoef = NULL;
  if (ir && ir->layer_rdo && encoder) {
    irec = (int64_t*)calloc((size_t)(num_unique + 1) * c, sizeof(int64_t));
    irec_us = (int64_t*)calloc((size_t)(num_unique + 1) * c, sizeof(int64_t));
    icoef = (int32_t*)calloc((size_t)n * c + 1, sizeof(int32_t));
  }
  if (ir && encoder) {
    *ir->num_modes = 0;
    *ir->num_taps = 0;
  }
  int bits_cur = 0, bits_ref = -1;
  if (ir) {
    bits_cur = bitlen_i64(pos[0] ^ pos[n - 1]);
    bits_ref = ir->n <= 1 ? -1 : bitlen_i64(ir->pos[0] ^ ir->pos[ir->n - 1]);
  }

  for (int li = nlv - 2; li >= 0; li--) {
    const level_t* ch = &lv[li];     /* children of the blocks           */
    const level_t* pa = &lv[li + 1]; /* parents = one block each         */
    /* "sumNodes == 0": nothing was added by the three binary expansions
     * (RAHT.cpp:1208-1209) */
    if (!first && ch->m == pa->m)
      continue;

    const int inherit_dc = !first;
    const int pred_in_level =
      inherit_dc && p->raht_prediction_enabled_flag != 0;
    first = 0;

    qp_layer = qp_layer + 1 < p->num_qp_layers ? qp_layer + 1
                                                 : p->num_qp_layers - 1;
    ac_layer++;

    /* inter prediction at this level: the reference frame's tree has nodes at bit level
     * lr (it ran out when lr would be negative, :1180-1181), the depth limit is not
     * reached (:1183-1184); both conditions only ever turn it off */
    const int lr = bits_ref - bits_cur + 3 * li;
    const int inter_on = ir && lr >= 0 && tree_depth < ir->depth_limit;
    /* the per-layer decision (:1256-1262): the encoder codes such a level twice, the
     * decoder follows the signalled mode */
    const int rdo_on = inter_on && ir->layer_rdo;
    const int cur_level = pred_in_level && rdo_on && (encoder || ir->layer_modes[depth]);
    const int dual = encoder && cur_level;
    /* a block is matched against the reference frame in such a level, and wherever the
     * level has no intra prediction (:1323-1335) */
    const int inter_blocks = inter_on && (cur_level || !pred_in_level);
    const int sum_nodes = ch->m - pa->m;
    int32_t* coef_begin[3] = {coef_it[0], coef_it[1], coef_it[2]};
    int32_t* icoef_it[3] = {icoef, icoef ? icoef + sum_nodes : NULL, icoef ? icoef + 2 * (size_t)sum_nodes : NULL};
    static const int kFixedTaps[7] = {128, 128, 128, 127, 125, 121, 115};
    int64_t filter_tap = 128;
    if (inter_on && !ir->filter_est)
      filter_tap = kFixedTaps[tree_depth < 7 ? tree_depth : 6];
    /* a filter tap of its own for the level (:1283-1305): estimated by the encoder over
     * EVERY block that lines up (whether or not the level then uses them), sent quantised */
    const int est_layer = inter_on && ir->filter_est && tree_depth >= ir->skip_layers;
    ref_nodes_t rn = {0, NULL, NULL, NULL};
    if (inter_blocks || (est_layer && encoder))
      ref_nodes_build(&rn, ir, lr, c, haar);
    if (est_layer) {
      quantizer_t tq[2];
      qpset_quantizers(p, qp_layer, 0, 0, tq);
      int64_t qtap;
      if (encoder) {
        /* estimate_layer_filter (:849-972) */
        int64_t autocorr = 0, crosscorr = 0;
        int jr = 0; /* its cursor: a block that finds it already on the last node sees no key */
        for (int jb = 0; jb < pa->m; jb++) {
          const int64_t want = pa->key[jb];
          int64_t rkey = jr < rn.m - 1 ? rn.key[jr] >> 3 : INT64_MAX;
          while (jr < rn.m - 1 && want > rkey) {
            jr++;
            rkey = rn.key[jr] >> 3;
          }
          if (want != rkey)
            continue;
          const int cs2 = pa->first_child[jb], ce2 = pa->first_child[jb + 1];
          if (ext && ce2 - cs2 == 1)
            continue;
          int32_t w2[8] = {0}, wr2[8] = {0};
          int64_t b2[3][8], r2[3][8];
          memset(b2, 0, sizeof(b2));
          memset(r2, 0, sizeof(r2));
          for (int t = jr; t < rn.m && (rn.key[t] >> 3) == want; t++) {
            const int idx = (int)(rn.key[t] & 7);
            wr2[idx] = rn.weight[t];
            r2[0][idx] = fp_from_int(rn.attr[t * c]);
          }
          for (int i = cs2; i < ce2; i++) {
            const int idx = (int)(ch->key[i] & 7);
            w2[idx] = ch->weight[i];
            b2[0][idx] = fp_from_int(ch->attr[i * c]);
          }
          for (int t = 0; t < 8; t++) {
            if (wr2[t] > 1)
              r2[0][t] = scale_rsqrt(r2[0][t], wr2[t]);
            if (w2[t] > 1)
              b2[0][t] = scale_rsqrt(b2[0][t], w2[t]);
          }
          block_weights_t bw2, bwr2;
          block_weights(w2, &bw2);
          block_weights(wr2, &bwr2);
          block_fwd(1, b2, &bw2, 0);
          block_fwd(1, r2, &bwr2, 0);
          static const int8_t kScan2[8] = {0, 4, 2, 1, 6, 5, 3, 7};
          for (int sc = 0; sc < 8; sc++) {
            const int idx = kScan2[sc];
            if (sc && !bw2.cw[idx])
              continue;
            if (inherit_dc && !idx)
              continue;
            const int64_t rv = r2[0][idx];
            if (rv) {
              autocorr += (rv * rv) >> FP_FRAC;
              crosscorr += (rv * b2[0][idx]) >> FP_FRAC;
            }
          }
        }
        const int orig = autocorr ? filter_tap_of(autocorr, crosscorr) : 128;
        qtap = quantizer_quantize(tq[0], (int64_t)(128 - orig) * 256);
        if (*ir->num_taps < 32)
          ir->filter_taps[(*ir->num_taps)++] = (int32_t)qtap;
        filter_tap = 128 - div_exp2_round_half_up(quantizer_scale(tq[0], qtap), 8);
      } else if (tree_depth - ir->skip_layers < *ir->num_taps) {
        qtap = ir->filter_taps[tree_depth - ir->skip_layers];
        filter_tap = 128 - div_exp2_round_half_up(quantizer_scale(tq[0], qtap), 8);
      }
    }

    /* previous reconstruction -> parent (RAHT.cpp:1275-1277) */
    {
      recon_t t = cur;
      cur = par;
      par = t;
    }
    /* child ranges of the parent-level nodes in the previously processed
     * level (only used by sub-node prediction) */
    const level_t* prev = last_done >= 0 ? &lv[last_done] : NULL;
    (void)prev;

    for (int j = 0; j < pa->m; j++) {
      const int cs = pa->first_child[j], ce = pa->first_child[j + 1];
      int64_t buf[6][8];
      memset(buf, 0, sizeof(buf));
      int64_t(*pred)[8] = &buf[c];
      int64_t ipred[3][8], ibuf2[3][8]; /* the intra candidate: prediction, coefficients */
      memset(ipred, 0, sizeof(ipred));
      memset(ibuf2, 0, sizeof(ibuf2));
      int32_t w[8] = {0};
      int32_t node_qp[8][2], asc_qp[8][2], dsc_qp[8][2];
      memset(node_qp, 0, sizeof(node_qp));
      memset(asc_qp, 0, sizeof(asc_qp));
      int occupancy = 0;
      int node_cnt = 0;

      for (int i = cs; i < ce; i++) {
        int idx = (int)(ch->key[i] & 7);
        w[idx] = ch->weight[i];
        asc_qp[idx][0] = ch->qp[2 * i];
        asc_qp[idx][1] = ch->qp[2 * i + 1];
        occupancy |= 1 << idx;
        if (ext)
          node_cnt++;
        if (encoder)
          for (int k = 0; k < c; k++)
            buf[k][idx] = fp_from_int(ch->attr[i * c + k]);
      }

      /* the block of the reference frame at the same place (:1322-1347) */
      int inter_node = 0;
      int32_t wr[8] = {0};
      int64_t ibuf[3][8];
      memset(ibuf, 0, sizeof(ibuf));
      if (inter_blocks) {
        const int64_t want = pa->key[j];
        int lo = 0, hi = rn.m;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((rn.key[mid] >> 3) < want)
            lo = mid + 1;
          else
            hi = mid;
        }
        const int jr = lo < rn.m - 1 ? lo : rn.m - 1; /* the cursor stops at the last node */
        if ((rn.key[jr] >> 3) == want) {
          inter_node = 1;
          for (int t = jr; t < rn.m && (rn.key[t] >> 3) == want; t++) {
            const int idx = (int)(rn.key[t] & 7);
            wr[idx] = rn.weight[t];
            for (int k = 0; k < c; k++)
              ibuf[k][idx] = fp_from_int(rn.attr[t * c + k]);
          }
        }
        if (ext && node_cnt == 1)
          inter_node = 0;
      }

      descend_block_qp(w, asc_qp, &par.dqp[2 * j], dsc_qp);
      for (int i = cs; i < ce; i++) {
        int idx = (int)(ch->key[i] & 7);
        cur.dqp[2 * i] = dsc_qp[idx][0];
        cur.dqp[2 * i + 1] = dsc_qp[idx][1];
        node_qp[idx][0] = dsc_qp[idx][0] >> 4;
        node_qp[idx][1] = dsc_qp[idx][1] >> 4;
      }

      block_weights_t bw;
      block_weights(w, &bw);

      if (!inherit_dc)
        for (int i = cs; i < ce; i++)
          cur.nneigh[i] = 19;

      /* ---- inter-level prediction (RAHT.cpp:1391-1432) ---- */
      int enable_pred = pred_in_level;
      if (pred_in_level) {
        int neigh_count = 0;
        int pn[19];
        int cn[12][8];
        if (ext && node_cnt == 1) {
          enable_pred = 0;
          neigh_count = 19;
        } else if (par.nneigh[j] < p->raht_prediction_threshold0) {
          enable_pred = 0;
        } else {
          /* findNeighbours */
          const int64_t cur_pos = pa->key[j];
          const int64_t base = (int64_t)morton3d_add(
            (uint64_t)cur_pos, (uint64_t)(int64_t)-1);
          pn[0] = j;
          for (int i = 1; i < 19; i++) {
            pn[i] = -1;
            if (!(occupancy & kNeighMask[i]))
              continue;
            int64_t np =
              (int64_t)morton3d_add((uint64_t)base, kNeighOffset[i]);
            int64_t d = np - cur_pos;
            const int64_t range = p->raht_prediction_search_range;
            if (d >= 0)
              d = d >= range ? range : d;
            else
              d = (-d) >= range ? -range : d;
            pn[i] = find_in_window(pa->key, pa->m, j, np, d);
          }
          if (p->raht_subnode_prediction_enabled_flag) {
            for (int i = 0; i < 12; i++)
              for (int t = 0; t < 8; t++)
                cn[i][t] = -1;
            for (int i = 0; i < 12; i++) {
              int q = pn[7 + i];
              if (q == -1)
                continue;
              /* a neighbour parent contributes children only once it has
               * been processed in this level (occupancy zeroed at level
               * start :1223, set at :1393): processed <=> q < j */
              if (q >= j)
                continue;
              int nocc = 0;
              for (int t = pa->first_child[q]; t < pa->first_child[q + 1];
                   t++)
                nocc |= 1 << (int)(ch->key[t] & 7);
              int mask = (i < 9 ? (nocc >> kOccuShift[i])
                                : ((nocc << kOccuShift[i]) & 0xff))
                & occupancy & kOccuMask[i];
              if (!mask)
                continue;
              for (int t = pa->first_child[q]; t < pa->first_child[q + 1];
                   t++) {
                int ni = (int)(ch->key[t] & 7)
                  + (i < 9 ? -(int)kOccuShift[i] : (int)kOccuShift[i]);
                if (ni >= 0 && ni < 8 && ((mask >> ni) & 1))
                  cn[i][ni] = t;
              }
            }
          }
          for (int i = 0; i < 19; i++)
            neigh_count += pn[i] != -1;
          if (neigh_count < p->raht_prediction_threshold1) {
            enable_pred = 0;
          } else {
            /* intraDcPred (RAHT.cpp:421-589) */
            int wsum[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
            int64_t limit_lo = 0, limit_hi = 0;
            const int sub = p->raht_subnode_prediction_enabled_flag != 0;
            const int parent_only = sub ? 7 : 19;
            for (int i = 0; i < parent_only; i++) {
              if (pn[i] == -1)
                continue;
              int64_t v[3];
              for (int k = 0; k < c; k++)
                v[k] = par.rec[(size_t)pn[i] * c + k];
              if (i) {
                if (10 * v[0] <= limit_lo || 10 * v[0] >= limit_hi)
                  continue;
              } else {
                limit_lo = 2 * v[0];
                limit_hi = 25 * v[0];
              }
              int64_t pw = p->pred_weight_parent[i];
              for (int k = 0; k < c; k++)
                v[k] *= ext ? pw : (pw << FP_FRAC);
              int mask = kNeighMask[i] & occupancy;
              for (int t = 0; mask; t++, mask >>= 1)
                if (mask & 1) {
                  wsum[t] += p->pred_weight_parent[i];
                  for (int k = 0; k < c; k++) {
                    pred[k][t] += v[k];
                    ipred[k][t] += v[k];
                  }
                }
            }
            if (sub) {
              for (int i = 0; i < 12; i++) {
                if (pn[7 + i] == -1)
                  continue;
                int64_t v[3];
                for (int k = 0; k < c; k++)
                  v[k] = par.rec[(size_t)pn[7 + i] * c + k];
                if (10 * v[0] <= limit_lo || 10 * v[0] >= limit_hi)
                  continue;
                int64_t pw = p->pred_weight_parent[7 + i];
                for (int k = 0; k < c; k++)
                  v[k] *= ext ? pw : (pw << FP_FRAC);
                int mask = kNeighMask[7 + i] & occupancy;
                for (int t = 0; mask; t++, mask >>= 1) {
                  if (!(mask & 1))
                    continue;
                  if (cn[i][t] != -1) {
                    int64_t cw = p->pred_weight_child[i];
                    wsum[t] += p->pred_weight_child[i];
                    for (int k = 0; k < c; k++) {
                      pred[k][t] += cur.rec[(size_t)cn[i][t] * c + k]
                        * (ext ? cw : (cw << FP_FRAC));
                      /* the intra candidate sees ITS reconstruction of the children (:549-561) */
                      if (dual)
                        ipred[k][t] += irec[(size_t)cn[i][t] * c + k] * (ext ? cw : (cw << FP_FRAC));
                    }
                  } else {
                    wsum[t] += p->pred_weight_parent[7 + i];
                    for (int k = 0; k < c; k++) {
                      pred[k][t] += v[k];
                      ipred[k][t] += v[k];
                    }
                  }
                }
              }
            }
            for (int t = 0; t < 8; t++) {
              if (!((occupancy >> t) & 1))
                continue;
              int64_t div = pred_divisor(wsum[t]);
              for (int k = 0; k < c; k++) {
                pred[k][t] = fp_mul(pred[k][t], div);
                ipred[k][t] = fp_mul(ipred[k][t], div);
                if (haar) {
                  pred[k][t] = (pred[k][t] >> FP_FRAC) << FP_FRAC;
                  ipred[k][t] = (ipred[k][t] >> FP_FRAC) << FP_FRAC;
                }
              }
            }
          }
        }
        for (int i = cs; i < ce; i++)
          cur.nneigh[i] = neigh_count;
      }

      /* the intra candidate has a prediction where the intra prediction succeeded (:1440-1442) */
      const int enable_intra = dual && enable_pred;

      /* ---- normalise (RAHT.cpp:1445-1499) ---- */
      if (!haar) {
        for (int t = 0; t < 8; t++) {
          if (w[t] <= 1)
            continue;
          if (encoder)
            for (int k = 0; k < c; k++)
              buf[k][t] = scale_rsqrt(buf[k][t], w[t]);
          if (enable_pred || enable_intra) {
            int64_t sq = (int64_t)isqrt_u64((uint64_t)w[t] << (2 * FP_FRAC));
            for (int k = 0; k < c; k++) {
              if (enable_pred)
                pred[k][t] = fp_mul(pred[k][t], sq);
              if (enable_intra)
                ipred[k][t] = fp_mul(ipred[k][t], sq);
            }
          }
        }
      }

      if (inter_node && !haar) {
        for (int t = 0; t < 8; t++)
          if (wr[t] > 1)
            for (int k = 0; k < c; k++)
              ibuf[k][t] = scale_rsqrt(ibuf[k][t], wr[t]);
        if (!encoder)
          enable_pred = 0;
      }

      /* ---- forward transform (RAHT.cpp:1504-1549) ---- */
      if (encoder && enable_pred)
        block_fwd(2 * c, buf, &bw, haar);
      else if (encoder)
        block_fwd(c, buf, &bw, haar);
      else if (enable_pred)
        block_fwd(c, pred, &bw, haar);
      if (inter_node) {
        /* the reference frame's block in ITS OWN weights, filtered, is the prediction
         * of every coefficient -- the DC as well where the level codes one (:1533-1545) */
        block_weights_t bwr;
        block_weights(wr, &bwr);
        block_fwd(c, ibuf, &bwr, haar);
        for (int t = 0; t < 8; t++)
          for (int k = 0; k < c; k++)
            /* (the integer Haar kernel takes the frame's coefficients as they are, :1520-1526) */
            pred[k][t] = haar || tree_depth < ir->skip_layers ? ibuf[k][t] : (ibuf[k][t] * filter_tap) >> 7;
        enable_pred = 1;
      }
      if (enable_intra)
        block_fwd(c, ipred, &bw, haar);
      if (dual)
        memcpy(ibuf2, buf, sizeof(int64_t) * 8 * (size_t)c);

      /* ---- per-coefficient scan (RAHT.cpp:1558-1724) ---- */
      static const int8_t kScan[8] = {0, 4, 2, 1, 6, 5, 3, 7};
      static const int kLutLog[16] = {0,   256, 406, 512, 594, 662, 719, 768,
                                      812, 850, 886, 918, 947, 975, 1000,
                                      1024};
      static const int kLutBins[11] = {1, 2, 3, 5, 5, 7, 7, 9, 9, 11, 11};
      for (int s = 0; s < 8; s++) {
        int idx = kScan[s];
        if (s && !bw.cw[idx])
          continue;
        if (inherit_dc && !idx)
          continue;

        if (encoder && enable_pred)
          for (int k = 0; k < c; k++)
            buf[k][idx] -= pred[k][idx];
        if (enable_intra)
          for (int k = 0; k < c; k++)
            ibuf2[k][idx] -= ipred[k][idx];

        int flag_rdoq = 0, iflag_rdoq = 0;
        if (encoder && !haar) {
          int64_t sum_coeff = 0, dist2 = 0, lambda0 = 0;
          int rate_coeff = 0;
          int64_t isum_coeff = 0, idist2 = 0;
          int irate_coeff = 0;
          quantizer_t q[2];
          qpset_quantizers(
            p, qp_layer, node_qp[idx][0], node_qp[idx][1], q);
          for (int k = 0; k < c; k++) {
            quantizer_t qk = q[k < 1 ? k : 1];
            int64_t coeff = fp_round(buf[k][idx]);
            dist2 += coeff * coeff;
            int64_t qc = quantizer_quantize(qk, coeff * 256);
            int64_t aq = qc < 0 ? -qc : qc;
            sum_coeff += aq;
            rate_coeff += aq < 15 ? kLutLog[aq] : kLutLog[15];
            if (!k)
              lambda0 = quantizer_scale(qk, 1);
            if (cur_level) {
              const int64_t ic = fp_round(ibuf2[k][idx]);
              idist2 += ic * ic;
              const int64_t iq = quantizer_quantize(qk, ic * 256);
              const int64_t iaq = iq < 0 ? -iq : iq;
              isum_coeff += iaq;
              irate_coeff += iaq < 15 ? kLutLog[iaq] : kLutLog[15];
            }
          }
          const int64_t lambda = lambda0 * lambda0 * (c == 1 ? 25 : 35);
          if (sum_coeff < 3) {
            int rate = kLutBins[train_zeros > 10 ? 10 : train_zeros];
            if (train_zeros > 10) {
              int temp = train_zeros - 11 + 1, a = 0;
              while (temp) {
                a++;
                temp >>= 1;
              }
              rate += 2 * a - 1 + 2;
            }
            rate += (rate_coeff + 128) >> 8;
            flag_rdoq = (int64_t)((uint64_t)dist2 << 26) < lambda * rate;
          }
          if (cur_level && isum_coeff < 3) {
            int rate = kLutBins[intra_train_zeros > 10 ? 10 : intra_train_zeros];
            if (intra_train_zeros > 10) {
              int temp = intra_train_zeros - 11 + 1, a = 0;
              while (temp) {
                a++;
                temp >>= 1;
              }
              rate += 2 * a - 1 + 2;
            }
            rate += (irate_coeff + 128) >> 8;
            iflag_rdoq = (int64_t)((uint64_t)idist2 << 26) < lambda * rate;
          }
          if (flag_rdoq || sum_coeff == 0)
            train_zeros++;
          else
            train_zeros = 0;
          if (cur_level) {
            if (iflag_rdoq || isum_coeff == 0)
              intra_train_zeros++;
            else
              intra_train_zeros = 0;
          }
        }

        int ac0 = 0, ac1 = 0;
        if (ac_layer <= max_ac_layers && idx) {
          ac0 = p->ac_qp_offset[ac_layer][idx - 1][0];
          ac1 = p->ac_qp_offset[ac_layer][idx - 1][1];
        }
        quantizer_t q[2];
        qpset_quantizers(
          p, qp_layer, node_qp[idx][0] + ac0, node_qp[idx][1] + ac1, q);
        for (int k = 0; k < c; k++) {
          quantizer_t qk = q[k < 1 ? k : 1];
          int64_t coeff;
          if (encoder) {
            if (flag_rdoq)
              buf[k][idx] = 0;
            if (iflag_rdoq)
              ibuf2[k][idx] = 0;
            coeff = quantizer_quantize(qk, fp_round(buf[k][idx]) * 256);
            if (cur_level)
              ac_estimate_cost(&est_cur, (int32_t)coeff, k);
            *coef_it[k]++ = (int32_t)coeff;
          } else {
            coeff = *coef_it[k]++;
          }
          /* FixedPoint += int64 goes through FixedPoint(int64)
           * (FixedPoint.h:63), i.e. a sign-symmetric << 15 */
          pred[k][idx] += fp_from_int(
            div_exp2_round_half_up(quantizer_scale(qk, coeff), 8));
          if (dual) {
            ac_estimate_update(&est_cur, (int32_t)coeff, k);
            const int64_t ic = quantizer_quantize(qk, fp_round(ibuf2[k][idx]) * 256);
            ac_estimate_cost(&est_intra, (int32_t)ic, k);
            *icoef_it[k]++ = (int32_t)ic;
            ipred[k][idx] += fp_from_int(div_exp2_round_half_up(quantizer_scale(qk, ic), 8));
            ac_estimate_update(&est_intra, (int32_t)ic, k);
          }
        }
      }

      /* ---- DC inheritance + inverse transform (RAHT.cpp:1727-1752) ---- */
      if (inherit_dc) {
        for (int k = 0; k < c; k++) {
          int64_t val = par.rec_us[(size_t)j * c + k];
          if (ext)
            pred[k][0] = val;
          else if (val > 0)
            pred[k][0] = val << (FP_FRAC - 2);
          else
            pred[k][0] = -((-val) << (FP_FRAC - 2));
          ipred[k][0] = pred[k][0];
        }
      }
      block_inv(c, pred, &bw, haar);
      if (dual)
        block_inv(c, ipred, &bw, haar);

      /* ---- store reconstruction (RAHT.cpp:1754-1806) ---- */
      for (int i = cs; i < ce; i++) {
        int idx = (int)(ch->key[i] & 7);
        for (int k = 0; k < c; k++) {
          int64_t v = pred[k][idx];
          cur.rec_us[(size_t)i * c + k] = ext ? v : fp_round(v * 4);
          if (!haar && w[idx] > 1)
            v = scale_rsqrt(v, w[idx]);
          cur.rec[(size_t)i * c + k] = ext ? v : fp_round(v);
          if (dual) {
            int64_t u = ipred[k][idx];
            irec_us[(size_t)i * c + k] = ext ? u : fp_round(u * 4);
            if (!haar && w[idx] > 1)
              u = scale_rsqrt(u, w[idx]);
            irec[(size_t)i * c + k] = ext ? u : fp_round(u);
          }
        }
      }
    }
    /* the level's decision (:1810-1829): the cheaper candidate's coefficients,
     * reconstruction, rate model and zero-run state go on */
    if (dual) {
      const int intra_wins = est_intra.bits < est_cur.bits;
      if (intra_wins) {
        for (int k = 0; k < c; k++)
          memcpy(coef_begin[k], icoef + (size_t)k * sum_nodes, sizeof(int32_t) * (size_t)sum_nodes);
        int64_t* t1 = irec;
        irec = cur.rec;
        cur.rec = t1;
        t1 = irec_us;
        irec_us = cur.rec_us;
        cur.rec_us = t1;
        est_cur = est_intra;
        train_zeros = intra_train_zeros;
      } else {
        est_intra = est_cur;
        intra_train_zeros = train_zeros;
      }
      if (*ir->num_modes < 32)
        ir->layer_modes[(*ir->num_modes)++] = !intra_wins;
      est_cur.bits = 0.;
      est_intra.bits = 0.;
    }
    if (pred_in_level && rdo_on)
      depth++;
    last_done = li;
    if (rn.key)
      ref_nodes_free(&rn);
    tree_depth++;
  }

  /* ---- duplicates (RAHT.cpp:1840-1964) and write-back (:1967-1975) ---- */
  int64_t* out = (int64_t*)calloc((size_t)n * c, sizeof(int64_t));
  if (num_unique != n) {
    const level_t* lf = &lv[0];
    for (int i = 0; i < lf->m; i++) {
      const int s = lf->first_child[i];
      const int weight = lf->weight[i];
      if (weight == 1) {
        for (int k = 0; k < c; k++)
          out[(size_t)s * c + k] = cur.rec[(size_t)i * c + k];
        continue;
      }
      int node_qp0 = cur.dqp[2 * i] >> 4, node_qp1 = cur.dqp[2 * i + 1] >> 4;
      int64_t sq = (int64_t)isqrt_u64((uint64_t)weight << (2 * FP_FRAC));
      int64_t attr_sum[3] = {0, 0, 0}, rec_dc[3];
      for (int k = 0; k < c; k++) {
        if (encoder)
          attr_sum[k] = fp_from_int(lf->attr[i * c + k]);
        int64_t v = cur.rec[(size_t)i * c + k];
        rec_dc[k] = ext ? v : fp_from_int(v);
        if (!haar)
          rec_dc[k] = fp_mul(rec_dc[k], sq);
      }
      /* the Hf entry of the w-th duplicate: its raw attribute, or for
       * integer Haar the running difference of reduceUnique :140-142 */
      int32_t hf[3];
      int32_t haar_acc[3];
      int32_t* haar_hf = NULL;
      if (encoder && haar) {
        haar_hf = (int32_t*)malloc(sizeof(int32_t) * (size_t)weight * c);
        for (int k = 0; k < c; k++)
          haar_acc[k] = attributes[(size_t)s * c + k];
        for (int wi = 1; wi < weight; wi++)
          for (int k = 0; k < c; k++) {
            int32_t d =
              wrap_sub(attributes[(size_t)(s + wi) * c + k], haar_acc[k]);
            haar_acc[k] = wrap_add(haar_acc[k], d >> 1);
            haar_hf[(size_t)wi * c + k] = d;
          }
      }
      quantizer_t q[2];
      qpset_quantizers(p, qp_layer, node_qp0, node_qp1, q);
      for (int wv = weight - 1; wv > 0; wv--) {
        int64_t a = 0, b = 0;
        if (!haar)
          raht_coeffs(wv, 1, &a, &b);
        for (int k = 0; k < c; k++) {
          quantizer_t qk = q[k < 1 ? k : 1];
          int64_t t0, t1;
          if (encoder) {
            hf[k] = haar ? haar_hf[(size_t)wv * c + k]
                         : attributes[(size_t)(s + wv) * c + k];
            t1 = fp_from_int(hf[k]);
            if (haar) {
              attr_sum[k] -= t1 >> 1;
              t1 += attr_sum[k];
              t0 = attr_sum[k];
              int64_t h = t1 - t0;
              t0 = t0 + ((h >> (1 + FP_FRAC)) << FP_FRAC);
              t1 = h;
            } else {
              attr_sum[k] -= t1;
              /* NB: no weight > 1 guard here (RAHT.cpp:1902-1903) */
              t0 = scale_rsqrt(attr_sum[k], wv);
              int64_t l = t0, r = t1;
              t0 = fp_mul(r, b) + fp_mul(a, l);
              t1 = fp_mul(r, a) - fp_mul(b, l);
            }
            int64_t coeff = quantizer_quantize(qk, fp_round(t1) * 256);
            *coef_it[k]++ = (int32_t)coeff;
            t1 = fp_from_int(
              div_exp2_round_half_up(quantizer_scale(qk, coeff), 8));
          } else {
            int64_t coeff = *coef_it[k]++;
            t1 = fp_from_int(
              div_exp2_round_half_up(quantizer_scale(qk, coeff), 8));
          }
          t0 = rec_dc[k];
          if (haar) {
            int64_t left = t0 - ((t1 >> (1 + FP_FRAC)) << FP_FRAC);
            t1 = t1 + left;
            t0 = left;
          } else {
            int64_t lfv = t0, hfv = t1;
            t0 = fp_mul(lfv, a) - fp_mul(b, hfv);
            t1 = fp_mul(lfv, b) + fp_mul(a, hfv);
          }
          rec_dc[k] = t0;
          out[(size_t)(s + wv) * c + k] = ext ? t1 : fp_round(t1);
          if (wv == 1)
            out[(size_t)s * c + k] = ext ? t0 : fp_round(t0);
        }
      }
      free(haar_hf);
    }
  } else {
    memcpy(out, cur.rec, sizeof(int64_t) * (size_t)n * c);
  }
  for (size_t i = 0; i < (size_t)n * c; i++)
    attributes[i] = ext ? (int32_t)((out[i] + FP_HALF) >> FP_FRAC)
                        : (int32_t)out[i];
  free(out);

  recon_free(&cur);
  recon_free(&par);
  free(irec);
  free(irec_us);
  free(icoef);
  for (int i = 0; i < nlv; i++)
    level_free(&lv[i]);
  return 0;
}

/* ---- exported entry points ------------------------------------------- */

int
oracle_raht_forward(
  const gpcc_raht_params* p, const int64_t* morton, const int32_t* qp_off,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3)
    return -1;
  return raht_process(1, p, morton, qp_off, attrs, coeffs, n, c, NULL);
}

int
oracle_raht_inverse(
  const gpcc_raht_params* p, const int64_t* morton, const int32_t* qp_off,
  int32_t* attrs, int32_t* coeffs, int32_t n, int32_t c)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3)
    return -1;
  return raht_process(0, p, morton, qp_off, attrs, coeffs, n, c, NULL);
}

/* RAHT with attribute inter prediction; arguments as ref_raht_inter (oracle/ref_harness.cpp) */
int
oracle_raht_inter(
  const gpcc_raht_params* p, int32_t fwd, const int64_t* morton, int32_t* attrs, int32_t* coeffs,
  int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  int32_t depth_minus1, int32_t layer_rdo, int32_t filter_est, int32_t skip_layers,
  int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3 || !morton_ref || !attrs_ref || n_ref <= 0)
    return -1;
  raht_inter_t ir = {morton_ref, attrs_ref, n_ref, depth_minus1 + 1, layer_rdo, filter_est, skip_layers,
                     layer_modes, num_modes, filter_taps, num_taps};
  return raht_process(fwd != 0, p, morton, NULL, attrs, coeffs, n, c, &ir);
}

/* ... with region QP offsets per point (QpSet::regionQpOffset), qp_off [n][2] */
int
oracle_raht_inter_qp(
  const gpcc_raht_params* p, int32_t fwd, const int64_t* morton, int32_t* attrs, int32_t* coeffs,
  int32_t n, int32_t c, const int64_t* morton_ref, const int32_t* attrs_ref, int32_t n_ref,
  int32_t depth_minus1, int32_t layer_rdo, int32_t filter_est, int32_t skip_layers,
  int32_t* layer_modes, int32_t* num_modes, int32_t* filter_taps, int32_t* num_taps, const int32_t* qp_off)
{
  if (!p || !morton || !attrs || !coeffs || n <= 0 || c < 1 || c > 3 || !morton_ref || !attrs_ref || n_ref <= 0)
    return -1;
  raht_inter_t ir = {morton_ref, attrs_ref, n_ref, depth_minus1 + 1, layer_rdo, filter_est, skip_layers,
                     layer_modes, num_modes, filter_taps, num_taps};
  return raht_process(fwd != 0, p, morton, qp_off, attrs, coeffs, n, c, &ir);
}

/* primitives exported for the pinning tests */
int64_t oracle_morton_addr(int32_t x, int32_t y, int32_t z) { return morton_addr(x, y, z); }
uint64_t oracle_morton3d_add(uint64_t a, uint64_t b) { return morton3d_add(a, b); }
uint32_t oracle_isqrt(uint64_t x) { return isqrt_u64(x); }
uint64_t oracle_irsqrt(uint64_t x) { return irsqrt_u64(x); }
int oracle_ilog2_u32(uint32_t x) { return ilog2_u32(x); }
int oracle_ilog2_u64(uint64_t x) { return ilog2_u64(x); }
int64_t oracle_fixedpoint_mul(int64_t a, int64_t b) { return fp_mul(a, b); }
int64_t oracle_fixedpoint_round(int64_t a) { return fp_round(a); }
int64_t oracle_fixedpoint_from_int(int64_t a) { return fp_from_int(a); }
int64_t oracle_quantize(int32_t qp, int64_t x) { return quantizer_quantize(quantizer_make(qp), x); }
int64_t oracle_scale(int32_t qp, int64_t x) { return quantizer_scale(quantizer_make(qp), x); }
int64_t oracle_div_exp2_round_half_up(int64_t x, int32_t s) { return div_exp2_round_half_up(x, s); }
int64_t oracle_div_exp2_round_half_inf(int64_t x, int32_t s) { return div_exp2_round_half_inf(x, s); }
int64_t oracle_div_approx(int64_t a, uint64_t b, int32_t s) { return div_approx(a, b, s); }
void
oracle_qpset_steps(
  const gpcc_raht_params* p, int32_t layer, int32_t off0, int32_t off1,
  int32_t out_step[2])
{
  quantizer_t q[2];
  qpset_quantizers(p, layer, off0, off1, q);
  out_step[0] = q[0].step;
  out_step[1] = q[1].step;
}

/* sort used by the Morton prologue restatement: (code, index) pairs */
typedef struct {
  int64_t code;
  int32_t index;
} code_index_t;
static int
cmp_code_index(const void* a, const void* b)
{
  const code_index_t* x = (const code_index_t*)a;
  const code_index_t* y = (const code_index_t*)b;
  if (x->code != y->code)
    return x->code < y->code ? -1 : 1;
  return x->index < y->index ? -1 : (x->index > y->index);
}
/* AttributeEncoder.cpp:1316-1321 + MortonCodeWithIndex::operator<
 * (PCCTMC3Common.h:184-190) */
int
oracle_attr_morton_sort(
  const int32_t* xyz, int32_t n, int64_t* morton, int32_t* order)
{
  code_index_t* v = (code_index_t*)malloc(sizeof(code_index_t) * (size_t)n);
  for (int i = 0; i < n; i++) {
    v[i].code = morton_addr(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    v[i].index = i;
  }
  qsort(v, (size_t)n, sizeof(code_index_t), cmp_code_index);
  for (int i = 0; i < n; i++) {
    morton[i] = v[i].code;
    order[i] = v[i].index;
  }
  free(v);
  return 0;
}
