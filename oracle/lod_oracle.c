/* lod_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), not product code.
 *
 * Plain-C restatement of the level-of-detail generation of TMC13 for intra
 * attribute coding of whole slices (scalable lifting included: octree
 * sub-sampling by LoD index, node-corner positions, pruning by range, the
 * repeated search of the finer layers :2377-2448):
 *   buildPredictorsFast          tmc3/PCCTMC3Common.h:2300-2469
 *   subsampleByDistance          :1984-2085   (MortonIndexMap3d :111-172)
 *   subsampleByDecimation        :2198-2214
 *   subsampleByOctree(+Centroid) :2089-2194
 *   computeNearestNeighbors      :1147-1953   (BoxHierarchy :58-107,
 *                                              updateNearestNeigh* :944-1143)
 *   updatePredictors             :2273-2296
 * written the way the kernels need it: the nearest-neighbour search of a
 * refinement point is a PURE FUNCTION of the point, the sorted retained
 * list and one per-LoD scalar (see `atlas_limit`), so all points of a LoD
 * are independent; the reference's sliding atlas (a 128^3-cell window that
 * is filled while walking the list) becomes binary searches in the
 * retained list.  Pinned against the compiled reference by
 * tests/test_oracle_lod.py.
 */
#include <limits.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "gpcc_attr_mi355.h"
#include "primitives.h"

typedef struct {
  int64_t code;
  int32_t pos[3];
  int32_t index; /* original point index */
} voxel_t;

static int
cmp_voxel(const void* a, const void* b)
{
  const voxel_t* x = (const voxel_t*)a;
  const voxel_t* y = (const voxel_t*)b;
  if (x->code != y->code)
    return x->code < y->code ? -1 : 1;
  return x->index < y->index ? -1 : (x->index > y->index);
}

enum { kAtlasLog2 = 7, kAtlasBits = 3 * kAtlasLog2 };

static inline int
min_i(int a, int b)
{
  return a < b ? a : b;
}
static inline int
max_i(int a, int b)
{
  return a > b ? a : b;
}

static inline int64_t
norm2(const int32_t* a, const int32_t* b)
{
  int64_t dx = (int64_t)a[0] - b[0], dy = (int64_t)a[1] - b[1], dz = (int64_t)a[2] - b[2];
  return dx * dx + dy * dy + dz * dz;
}
static inline int32_t
norm1(const int32_t* a, const int32_t* b)
{
  return abs(a[0] - b[0]) + abs(a[1] - b[1]) + abs(a[2] - b[2]);
}
/* Vec3::getDir of (a - b), PCCMath.h:105-109 */
static inline int
dir_of(const int32_t* a, const int32_t* b)
{
  return ((a[0] - b[0] >= 0) << 2) + ((a[1] - b[1] >= 0) << 1) + (a[2] - b[2] >= 0);
}

/* range [lo, hi) of entries of the sorted index list `list` (codes
 * ascending) with (code >> shift) == cell */
static void
cell_range(
  const voxel_t* pv, const int32_t* list, int n, int shift, int64_t cell,
  int* lo_out, int* hi_out)
{
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((pv[list[mid]].code >> shift) < cell)
      lo = mid + 1;
    else
      hi = mid;
  }
  int start = lo;
  hi = n;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if ((pv[list[mid]].code >> shift) <= cell)
      lo = mid + 1;
    else
      hi = mid;
  }
  *lo_out = start;
  *hi_out = lo;
}

/* ---- sub-sampling ------------------------------------------------------ */

static const uint8_t kSubNeigh[20] = {7, 3, 5, 6, 12, 10, 17, 20, 34, 33,
                                      4, 2, 1, 24, 40, 48, 32, 16, 8, 0};

/* subsampleByDistance :1984-2085.  Sequential greedy; the atlas of the
 * reference only ever holds retained points of the current 128^3-cell
 * block, i.e. the retained entries inserted since the block began. */
static void
subsample_by_distance(
  const voxel_t* pv, const int32_t* input, int n_in, int shift_bits0,
  int32_t* retained, int* n_ret, int32_t* refine, int* n_ref)
{
  *n_ret = 0;
  if (n_in == 1) {
    refine[(*n_ref)++] = input[0];
    return;
  }
  const int64_t radius2 = (int64_t)3 << (shift_bits0 << 1);
  const int shift3 = 3 * (shift_bits0 + 1);
  const int boundary = min_i(63, shift3 + kAtlasBits);
  int64_t cur_atlas = -1, last_cell = -1;
  int block_first = 0; /* first retained entry of the current atlas block */
  for (int t = 0; t < n_in; t++) {
    const int idx = input[t];
    const int64_t code = pv[idx].code;
    const int64_t atlas_id = code >> boundary;
    const int64_t cell = code >> shift3;
    if (cur_atlas != atlas_id) {
      cur_atlas = atlas_id;
      block_first = *n_ret;
    }
    if (*n_ret == 0) {
      retained[(*n_ret)++] = idx;
      last_cell = cell;
      continue;
    }
    if (last_cell == cell) {
      refine[(*n_ref)++] = idx;
      continue;
    }
    const uint64_t base = morton3d_add((uint64_t)cell, ~(uint64_t)0);
    int found = 0;
    for (int k = 0; k < 20 && !found; k++) {
      const int64_t nb = (int64_t)morton3d_add(base, kSubNeigh[k]);
      if ((nb >> kAtlasBits) != cur_atlas)
        continue;
      /* atlas.get(): the low 21 bits address the cell inside the block */
      int lo, hi;
      cell_range(pv, retained + block_first, *n_ret - block_first, shift3, nb, &lo, &hi);
      /* NB: the reference compares only the low 21 bits; inside one block
       * these identify the cell uniquely */
      for (int r = block_first + lo; r < block_first + hi; r++)
        if (norm2(pv[retained[r]].pos, pv[idx].pos) <= radius2) {
          found = 1;
          break;
        }
    }
    if (found) {
      refine[(*n_ref)++] = idx;
    } else {
      retained[(*n_ret)++] = idx;
      last_cell = cell;
    }
  }
}

/* subsampleByDecimation :2198-2214 */
static void
subsample_by_decimation(
  const int32_t* input, int n_in, int period, int32_t* retained, int* n_ret,
  int32_t* refine, int* n_ref)
{
  *n_ret = 0;
  for (int i = 0, j = 1; i < n_in; i++) {
    if (--j)
      refine[(*n_ref)++] = input[i];
    else {
      retained[(*n_ret)++] = input[i];
      j = period;
    }
  }
}

/* subsampleByOctree :2146-2194 with subsampleByOctreeWithCentroid
 * :2089-2142 (backward direction; clacIntermediatePosition masks the low
 * octreeNodeSizeLog2 bits) */
static void
subsample_by_octree(
  const voxel_t* pv, const int32_t* input, int n_in, int node_log2, int period,
  int backward, int32_t* retained, int* n_ret, int32_t* refine, int* n_ref)
{
  *n_ret = 0;
  if (n_in == 1) {
    refine[(*n_ref)++] = input[0];
    return;
  }
  const int quant = 3 * (node_log2 + 1);
  const uint32_t mask = node_log2 ? (uint32_t)(-1) << node_log2 : (uint32_t)(-1);
  int32_t* vox = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_in);
  int nv = 0;
  for (int i = 0; i < n_in; i++) {
    const uint64_t cur = (uint64_t)pv[input[i]].code >> quant;
    uint64_t next = cur;
    if (i < n_in - 1)
      next = (uint64_t)pv[input[i + 1]].code >> quant;
    vox[nv++] = input[i];
    if (i == n_in - 1 || cur < next) {
      if (nv < period && i != n_in - 1)
        continue;
      int32_t cen[3] = {0, 0, 0};
      for (int v = 0; v < nv; v++)
        for (int d = 0; d < 3; d++)
          cen[d] += (int32_t)((uint32_t)pv[vox[v]].pos[d] & mask);
      /* the first minimum met walking from the back (direction) or the front */
      int best = backward ? nv - 1 : 0;
      int64_t best_m = INT64_MAX;
      for (int w = 0; w < nv; w++) {
        const int v = backward ? nv - 1 - w : w;
        int64_t m = 0;
        for (int d = 0; d < 3; d++) {
          int32_t p = (int32_t)((uint32_t)pv[vox[v]].pos[d] & mask) * nv;
          m += abs(p - cen[d]);
        }
        m = (int32_t)m; /* getNorm1 on Vec3<int32_t> */
        if (best_m > m) {
          best_m = m;
          best = v;
        }
      }
      const int32_t picked = vox[best];
      for (int v = 0; v < nv; v++) {
        if (vox[v] == picked)
          retained[(*n_ret)++] = vox[v];
        else
          refine[(*n_ref)++] = vox[v];
      }
      nv = 0;
    }
  }
  free(vox);
}

/* ---- nearest-neighbour search ------------------------------------------ */

typedef struct {
  int32_t idx[6];
  int64_t dist[6];
  int idx2;
  /* inter-frame prediction: candidates of the reference frame travel with a flag
   * (localRef, :1316) -- `cur` is the flag of the candidate being visited */
  uint8_t ref[6];
  uint8_t cur;
} nn_state_t;

/* updateNearestNeigh :1030-1076 */
static void
nn_update(nn_state_t* s, int32_t d, int32_t index)
{
  if (d >= s->dist[2])
    return;
  if (d < s->dist[0]) {
    s->dist[2] = s->dist[1];
    s->dist[1] = s->dist[0];
    s->dist[0] = d;
    s->idx[2] = s->idx[1];
    s->idx[1] = s->idx[0];
    s->idx[0] = index;
    s->ref[2] = s->ref[1];
    s->ref[1] = s->ref[0];
    s->ref[0] = s->cur;
  } else if (d < s->dist[1]) {
    s->dist[2] = s->dist[1];
    s->dist[1] = d;
    s->idx[2] = s->idx[1];
    s->idx[1] = index;
    s->ref[2] = s->ref[1];
    s->ref[1] = s->cur;
  } else {
    s->dist[2] = d;
    s->idx[2] = index;
    s->ref[2] = s->cur;
  }
}

/* updateNearestNeighByDistanceAndDistribution :944-1027 */
static void
nn_update_dist(nn_state_t* s, int32_t d, int32_t index)
{
#define NN_SPILL()                    \
  if (s->idx[2] != -1) {              \
    s->ref[s->idx2] = s->ref[2];      \
    s->idx[s->idx2++] = s->idx[2];    \
  }
  if (d > s->dist[2]) {
    /* nothing */
  } else if (d < s->dist[0]) {
    NN_SPILL();
    s->dist[2] = s->dist[1];
    s->dist[1] = s->dist[0];
    s->dist[0] = d;
    s->idx[2] = s->idx[1];
    s->idx[1] = s->idx[0];
    s->idx[0] = index;
    s->ref[2] = s->ref[1];
    s->ref[1] = s->ref[0];
    s->ref[0] = s->cur;
  } else if (d < s->dist[1]) {
    NN_SPILL();
    s->dist[2] = s->dist[1];
    s->dist[1] = d;
    s->idx[2] = s->idx[1];
    s->idx[1] = index;
    s->ref[2] = s->ref[1];
    s->ref[1] = s->cur;
  } else if (d < s->dist[2]) {
    NN_SPILL();
    s->dist[2] = d;
    s->idx[2] = index;
    s->ref[2] = s->cur;
  } else if (s->idx[5] == -1) {
    s->ref[s->idx2] = s->cur;
    s->idx[s->idx2++] = index;
  }
#undef NN_SPILL
  if (s->idx2 == 6)
    s->idx2 = 3;
}

static void
nn_visit(nn_state_t* s, int distribution, int check, int32_t d, int32_t index)
{
  if (check) {
    const int lim = distribution ? 6 : 3;
    for (int h = 0; h < lim; h++)
      if (s->idx[h] == index && s->ref[h] == s->cur)
        return;
  }
  if (distribution)
    nn_update_dist(s, d, index);
  else
    nn_update(s, d, index);
}

static const uint8_t kNnNeigh[27] = {7,  3,  5,  6,  35, 21, 14, 28, 42,
                                     49, 12, 10, 17, 20, 34, 33, 4,  2,
                                     1,  56, 24, 40, 48, 32, 16, 8,  0};

/* three-level bounding boxes over buckets of 32 (BoxHierarchy<5,3>) */
typedef struct {
  int32_t* mn[3];
  int32_t* mx[3];
  int count[3];
} bbox_t;

static void
bbox_build(bbox_t* h, const int32_t* bpos /*[n][3] in list order*/, int n)
{
  int cnt = n;
  for (int l = 0; l < 3; l++) {
    cnt = (cnt + 31) >> 5;
    h->count[l] = cnt;
    h->mn[l] = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(cnt + 1));
    h->mx[l] = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(cnt + 1));
    for (int i = 0; i < 3 * cnt; i++) {
      h->mn[l][i] = INT32_MAX;
      h->mx[l][i] = INT32_MIN;
    }
  }
  for (int i = 0; i < n; i++)
    for (int d = 0; d < 3; d++) {
      int b = i >> 5;
      if (bpos[3 * i + d] < h->mn[0][3 * b + d])
        h->mn[0][3 * b + d] = bpos[3 * i + d];
      if (bpos[3 * i + d] > h->mx[0][3 * b + d])
        h->mx[0][3 * b + d] = bpos[3 * i + d];
    }
  for (int l = 0; l < 2; l++)
    for (int j = 0; j < h->count[l]; j++)
      for (int d = 0; d < 3; d++) {
        int b = j >> 5;
        if (h->mn[l][3 * j + d] < h->mn[l + 1][3 * b + d])
          h->mn[l + 1][3 * b + d] = h->mn[l][3 * j + d];
        if (h->mx[l][3 * j + d] > h->mx[l + 1][3 * b + d])
          h->mx[l + 1][3 * b + d] = h->mx[l][3 * j + d];
      }
}

static void
bbox_free(bbox_t* h)
{
  for (int l = 0; l < 3; l++) {
    free(h->mn[l]);
    free(h->mx[l]);
  }
}

/* Box3::getDist1 PCCMath.h:504-510 */
static inline int32_t
bbox_dist1(const bbox_t* h, int level, int b, const int32_t* p)
{
  int32_t s = 0;
  for (int d = 0; d < 3; d++) {
    int32_t a = h->mn[level][3 * b + d] - p[d];
    int32_t c = p[d] - h->mx[level][3 * b + d];
    int32_t m = a > 0 ? a : 0;
    s += m > c ? m : c;
  }
  return s;
}

/* the bucketed window scans of :1436-1522 / :1551-1604; dir +1: k0..k1
 * ascending, dir -1: descending */
static void
window_scan(
  nn_state_t* s, const bbox_t* h, const int32_t* bpos_list, const int32_t* bp,
  int k0, int k1, int dir, int distribution, int check, const int32_t* cand_id,
  int cand_base)
{
  if (k0 > k1)
    return;
  if (dir > 0) {
    for (int b2 = k0 >> 15; b2 <= (k1 >> 15); b2++) {
      if (s->idx[2] != -1 && bbox_dist1(h, 2, b2, bp) >= s->dist[2])
        continue;
      const int s1 = max_i(k0 >> 10, b2 << 5), e1 = min_i(k1 >> 10, (b2 << 5) + 31);
      for (int b1 = s1; b1 <= e1; b1++) {
        if (s->idx[2] != -1 && bbox_dist1(h, 1, b1, bp) >= s->dist[2])
          continue;
        const int s0 = max_i(k0 >> 5, b1 << 5), e0 = min_i(k1 >> 5, (b1 << 5) + 31);
        for (int b0 = s0; b0 <= e0; b0++) {
          if (s->idx[2] != -1 && bbox_dist1(h, 0, b0, bp) >= s->dist[2])
            continue;
          const int h0 = max_i(k0, b0 << 5), h1 = min_i(k1, (b0 << 5) + 31);
          for (int k = h0; k <= h1; k++)
            nn_visit(
              s, distribution, check, norm1(bp, &bpos_list[3 * k]),
              cand_id ? cand_id[cand_base + k] : k);
        }
      }
    }
  } else {
    for (int c2 = k1 >> 15; c2 >= (k0 >> 15); c2--) {
      if (s->idx[2] != -1 && bbox_dist1(h, 2, c2, bp) >= s->dist[2])
        continue;
      const int s1 = max_i(k0 >> 10, c2 << 5), e1 = min_i(k1 >> 10, (c2 << 5) + 31);
      for (int c1 = e1; c1 >= s1; c1--) {
        if (s->idx[2] != -1 && bbox_dist1(h, 1, c1, bp) >= s->dist[2])
          continue;
        const int s0 = max_i(k0 >> 5, c1 << 5), e0 = min_i(k1 >> 5, (c1 << 5) + 31);
        for (int c0 = e0; c0 >= s0; c0--) {
          if (s->idx[2] != -1 && bbox_dist1(h, 0, c0, bp) >= s->dist[2])
            continue;
          const int h0 = max_i(k0, c0 << 5), h1 = min_i(k1, (c0 << 5) + 31);
          for (int k = h1; k >= h0; k--)
            nn_visit(
              s, distribution, check, norm1(bp, &bpos_list[3 * k]),
              cand_id ? cand_id[cand_base + k] : k);
        }
      }
    }
  }
}

typedef struct {
  int32_t count;
  int32_t pidx[3];   /* neighbour POINT index (of the reference frame when ref[] is set) */
  uint64_t w[3];     /* squared distance */
  uint8_t ref[3];    /* PCCNeighborInfo::interFrameRef */
} raw_pred_t;

/* the reference frame of attribute inter prediction as computeNearestNeighbors sees it
 * (:1270-1292): the whole frame in Morton order, its biased positions and one box
 * hierarchy over the list; indexesRef is the identity, [startIndexRef, endIndexRef) the
 * whole list at every level of detail (buildPredictorsFast :2348-2376, :2396-2400) */
typedef struct {
  const voxel_t* pv;
  int32_t n;
  const int32_t* bias_pos; /* [n][3], sorted order */
  const int32_t* identity; /* 0 .. n-1 */
  bbox_t boxes;
  int32_t range;           /* abh.attrInterPredSearchRange: replaces BOTH LoD search ranges */
} inter_frame_t;

static void
compute_nearest_neighbours(
  const gpcc_lod_params* lp, const voxel_t* pv, int32_t n, const int32_t* bias_pos_in /*[n][3]*/,
  const int32_t* retained, int n_ret, int32_t* indexes /* in: packed idx, out: point idx */,
  int start, int end, int lod_index, raw_pred_t* preds, int32_t* pt2pred,
  int* pred_index, const inter_frame_t* ir)
{
  /* scalable lifting (:1174-1176, :1232-1236, clacIntermediatePosition :925-940): the
   * search cells are those of the octree level, and every position is replaced by the
   * corner of its node at that level before the bias is applied */
  const int scalable = lp->scalable_lifting_enabled_flag != 0;
  const int shift_bits =
    scalable ? 1 + lod_index : 1 + lp->dist2 + lp->attr_dist2_delta + lod_index;
  const uint32_t node_mask = scalable && lod_index ? (uint32_t)(-1) << lod_index : (uint32_t)(-1);
  int32_t* bias_pos_lod = NULL;
  if (scalable && lod_index) {
    bias_pos_lod = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
    for (int i = 0; i < n; i++)
      for (int d = 0; d < 3; d++)
        bias_pos_lod[3 * i + d] =
          (int32_t)((uint32_t)pv[i].pos[d] & node_mask) * lp->lod_neigh_bias[d];
  }
  const int32_t* bias_pos = bias_pos_lod ? bias_pos_lod : bias_pos_in;
  const int shift3 = 3 * shift_bits;
  const int boundary = min_i(63, shift3 + kAtlasBits);
  const int distribution = lp->prediction_with_distribution_enabled != 0;
  const int range_inter = ir ? ir->range : lp->inter_lod_search_range;
  const int range_intra = ir ? ir->range : lp->intra_lod_search_range;
  /* the inter-frame atlas holds 8^3 cells (:2391-2395) */
  const int inter_boundary = min_i(63, shift3 + 9);
  const int intra = lod_index >= lp->intra_lod_prediction_skip_layers;
  const int n_ref = end - start;

  /* biased positions in list order + hierarchies */
  int32_t* bret = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(n_ret + 1));
  for (int i = 0; i < n_ret; i++)
    memcpy(&bret[3 * i], &bias_pos[3 * retained[i]], 12);
  bbox_t hb, hi;
  bbox_build(&hb, bret, n_ret);
  int32_t* bref = NULL;
  if (intra) {
    bref = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(n_ref + 1));
    for (int i = 0; i < n_ref; i++)
      memcpy(&bref[3 * i], &bias_pos[3 * indexes[start + i]], 12);
    bbox_build(&hi, bref, n_ref);
  }

  /* The reference fills its atlas block by block from a cursor that only
   * advances over retained entries of the block being entered (:1349-1363).
   * A retained entry whose block holds no refinement point is never passed:
   * from that block on the atlas stays empty.  atlas_limit = id of the first
   * such block (INT64_MAX if none). */
  int64_t atlas_limit = INT64_MAX;
  {
    int r = 0;
    int64_t cur = -1;
    for (int i = start; i < end && atlas_limit == INT64_MAX; i++) {
      const int64_t id = pv[indexes[i]].code >> boundary;
      if (id == cur)
        continue;
      cur = id;
      if (r < n_ret && (pv[retained[r]].code >> boundary) < id) {
        atlas_limit = pv[retained[r]].code >> boundary;
        break;
      }
      while (r < n_ret && (pv[retained[r]].code >> boundary) == id)
        r++;
    }
  }

  /* packed indices of the refinement list survive in a copy: indexes[] is
   * rewritten with point indices as the walk proceeds */
  int32_t* packed = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n_ref + 1));
  memcpy(packed, &indexes[start], sizeof(int32_t) * (size_t)n_ref);

  for (int i = start; i < end; i++) {
    nn_state_t s;
    for (int h = 0; h < 6; h++) {
      s.idx[h] = -1;
      s.dist[h] = INT64_MAX;
      s.ref[h] = 0;
    }
    s.idx2 = 3;
    s.cur = 0;
    const int index = packed[i - start];
    const int64_t code = pv[index].code;
    const int64_t atlas_id = code >> boundary;
    const int64_t cell = code >> shift3;
    const int32_t* bp = &bias_pos[3 * index];
    indexes[i] = pv[index].index;
    raw_pred_t* pr = &preds[--(*pred_index)];
    pt2pred[pv[index].index] = *pred_index;

    if (n_ret) {
      /* j: first retained entry with a larger code, clipped (:1343-1346) */
      int lo = 0, hi2 = n_ret;
      while (lo < hi2) {
        int mid = (lo + hi2) >> 1;
        if (pv[retained[mid]].code <= code)
          lo = mid + 1;
        else
          hi2 = mid;
      }
      const int j = min_i(lo, n_ret - 1);

      if (atlas_id < atlas_limit) {
        const uint64_t base = morton3d_add((uint64_t)cell, ~(uint64_t)0);
        for (int nn = 0; nn < 27; nn++) {
          const int64_t nb = (int64_t)morton3d_add(base, kNnNeigh[nn]);
          if ((nb >> kAtlasBits) != atlas_id)
            continue;
          int r0, r1;
          cell_range(pv, retained, n_ret, shift3, nb, &r0, &r1);
          for (int k = r0; k < r1; k++)
            nn_visit(&s, distribution, 0, norm1(bp, &bret[3 * k]), k);
        }
      }

      if (s.idx[2] == -1) {
        const int center = s.idx[0] == -1 ? j : s.idx[0];
        const int k0 = max_i(0, center - range_inter);
        const int k1 = min_i(n_ret - 1, center + range_inter);
        nn_visit(&s, distribution, 1, norm1(bp, &bret[3 * center]), center);
        for (int nn = 1; nn <= 2; nn++) {
          const int kp = center + nn;
          if (kp <= k1)
            nn_visit(&s, distribution, 1, norm1(bp, &bret[3 * kp]), kp);
          const int kn = center - nn;
          if (kn >= k0)
            nn_visit(&s, distribution, 1, norm1(bp, &bret[3 * kn]), kn);
        }
        const int p1 = min_i(n_ret - 1, center + 3);
        const int p0 = max_i(0, center - 3);
        window_scan(&s, &hb, bret, bp, p1, k1, +1, distribution, 1, NULL, 0);
        window_scan(&s, &hb, bret, bp, k0, p0, -1, distribution, 1, NULL, 0);
      }
      /* retained-list indices -> packed indices */
      const int cnt = (s.idx[0] != -1) + (s.idx[1] != -1) + (s.idx[2] != -1);
      for (int h = 0; h < cnt; h++)
        s.idx[h] = retained[s.idx[h]];
      if (distribution) {
        const int cnt2 = (s.idx[3] != -1) + (s.idx[4] != -1) + (s.idx[5] != -1);
        for (int h = 3; h < 3 + cnt2; h++)
          s.idx[h] = retained[s.idx[h]];
      }
    }

    if (intra) {
      /* same-LoD candidates: the points that FOLLOW in the list (:1537-1604) */
      const int k00 = i + 1;
      const int k01 = min_i(end - 1, k00 + 2);
      for (int k = k00; k <= k01; k++)
        nn_visit(
          &s, distribution, 0, norm1(bp, &bref[3 * (k - start)]), packed[k - start]);
      const int w0 = k01 + 1 - start;
      const int w1 = min_i(end - 1, k00 + range_intra) - start;
      window_scan(&s, &hi, bref, bp, w0, w1, +1, distribution, 0, packed, 0);
    }

    if (ir) {
      /* candidates of the reference frame, no duplicate checks (:1606-1796) */
      s.cur = 1;
      /* (a) through the inter-frame atlas.  The test of a neighbour cell against the
       * atlas block shifts by the INTRA atlas' 21 bits (:1627) while the block id was
       * formed with 9: the two only agree in block 0, so the atlas contributes for the
       * cells of the first 8^3 block alone, and a neighbour cell outside that block
       * aliases into it (MortonIndexMap3d::get masks the address, :155-158) */
      if ((code >> inter_boundary) == 0) {
        const uint64_t base = morton3d_add((uint64_t)cell, ~(uint64_t)0);
        for (int nn = 0; nn < 27; nn++) {
          const int64_t nb = (int64_t)morton3d_add(base, kNnNeigh[nn]);
          if ((nb >> kAtlasBits) != 0)
            continue;
          int r0, r1;
          cell_range(ir->pv, ir->identity, ir->n, shift3, nb & 0x1ff, &r0, &r1);
          for (int k = r0; k < r1; k++)
            nn_visit(&s, distribution, 0, norm1(bp, &ir->bias_pos[3 * k]), k);
        }
      }
      /* (b) the window around the first entry that does not precede the point */
      if (ir->n > 0) {
        int lo = 0, hi2 = ir->n;
        while (lo < hi2) {
          int mid = (lo + hi2) >> 1;
          if (ir->pv[mid].code < code)
            lo = mid + 1;
          else
            hi2 = mid;
        }
        const int jr = min_i(lo, ir->n - 1);
        const int k1 = min_i(ir->n - 1, max_i(0, jr + ir->range));
        window_scan(&s, &ir->boxes, ir->bias_pos, bp, jr, k1, +1, distribution, 0, NULL, 0);
        const int l0 = min_i(ir->n - 1, max_i(0, jr - 1));
        const int l1 = min_i(ir->n - 1, max_i(0, l0 - ir->range));
        /* the left part is walked upwards too */
        window_scan(&s, &ir->boxes, ir->bias_pos, bp, l1, l0, +1, distribution, 0, NULL, 0);
      }
      s.cur = 0;
    }

    int count = (s.idx[0] != -1) + (s.idx[1] != -1) + (s.idx[2] != -1);
    count = min_i(lp->num_pred_nearest_neighbours_minus1 + 1, count);
#define NN_POS(h) (s.ref[h] ? &ir->bias_pos[3 * s.idx[h]] : &bias_pos[3 * s.idx[h]])
    if (distribution) {
      const int c1 = 3 + (s.idx[3] != -1) + (s.idx[4] != -1) + (s.idx[5] != -1);
      for (int m = 3; m < c1; m++)
        if (s.dist[m] == INT64_MAX)
          s.dist[m] = norm1(bp, NN_POS(m));
      for (int m = 3; m < c1; m++)
        for (int l = m + 1; l < c1; l++)
          if (s.dist[l] < s.dist[m]) {
            int32_t ti = s.idx[l];
            s.idx[l] = s.idx[m];
            s.idx[m] = ti;
            int64_t td = s.dist[l];
            s.dist[l] = s.dist[m];
            s.dist[m] = td;
            uint8_t tr = s.ref[l];
            s.ref[l] = s.ref[m];
            s.ref[m] = tr;
          }
      if (count >= 3) {
        /* third neighbour replaced by one on the far side (:1836-1902) */
        static const int8_t loose[8][3] = {{3, 5, 6}, {2, 4, 7}, {1, 4, 7},
                                           {0, 5, 6}, {1, 2, 7}, {0, 3, 6},
                                           {0, 3, 5}, {1, 2, 4}};
        int dir[6] = {-1, -1, -1, -1, -1, -1};
        int numend = 3;
        for (; numend < c1; numend++)
          if ((s.dist[numend] << 5) >= s.dist[2] * 54)
            break;
        for (int h = 0; h < numend; h++)
          dir[h] = dir_of(NN_POS(h), bp);
        int replace = 1, ridx = -1;
        if (dir[1] == 7 - dir[0] || dir[2] == 7 - dir[0] || dir[2] == 7 - dir[1])
          replace = 0;
        for (int h = 3; replace && h < numend; h++)
          if (dir[h] == 7 - dir[0] || dir[h] == 7 - dir[1]) {
            replace = 0;
            ridx = h;
          }
        const int e01 = dir[0] == dir[1], e02 = dir[0] == dir[2], e12 = dir[1] == dir[2];
        const int8_t* l0 = loose[dir[0]];
        if (replace) {
          if ((e02 || e12) && e01) {
            for (int h = 3; replace && h < numend; h++)
              if (dir[h] == l0[0] || dir[h] == l0[1] || dir[h] == l0[2]) {
                replace = 0;
                ridx = h;
              }
          } else if ((e02 || e12) && !e01) {
            if (!(dir[1] == l0[0] || dir[1] == l0[1] || dir[1] == l0[2]))
              for (int h = 3; replace && h < numend; h++)
                if (dir[h] != dir[0] && dir[h] != dir[1]) {
                  replace = 0;
                  ridx = h;
                }
          } else if (e01) {
            if (!(dir[2] == l0[0] || dir[2] == l0[1] || dir[2] == l0[2]))
              for (int h = 3; replace && h < numend; h++)
                if (dir[h] == l0[0] || dir[h] == l0[1] || dir[h] == l0[2]) {
                  replace = 0;
                  ridx = h;
                }
          }
        }
        if (ridx >= 0) {
          s.idx[2] = s.idx[ridx];
          s.ref[2] = s.ref[ridx];
        }
      }
    }
    pr->count = count;
    for (int h = 0; h < 3; h++) {
      pr->pidx[h] = 0;
      pr->w[h] = 0;
      pr->ref[h] = 0;
    }
    for (int h = 0; h < count; h++) {
      pr->ref[h] = s.ref[h];
      pr->pidx[h] = s.ref[h] ? ir->pv[s.idx[h]].index : pv[s.idx[h]].index;
      pr->w[h] = (uint64_t)norm2(NN_POS(h), bp);
    }
#undef NN_POS
    /* scalable lifting: neighbours further than the range are dropped, and all
     * that follow them (:1918-1939, pruneDistanceGt :695-703) */
    if (scalable) {
      const int64_t max_dist = (3ll * (lp->max_neigh_range_minus1 + 1)) << (2 * lod_index);
      const int unit_bias =
        lp->lod_neigh_bias[0] == 1 && lp->lod_neigh_bias[1] == 1 && lp->lod_neigh_bias[2] == 1;
      for (int h = 1; h < count; h++) {
        int64_t d2;
        if (unit_bias)
          d2 = (int64_t)pr->w[h];
        else {
          d2 = 0;
          for (int d = 0; d < 3; d++) {
            const int64_t a = (int64_t)(int32_t)((uint32_t)pv[index].pos[d] & node_mask)
              - (int64_t)(int32_t)((uint32_t)pv[s.idx[h]].pos[d] & node_mask);
            d2 += a * a;
          }
        }
        if (d2 > max_dist) {
          pr->count = count = h;
          break;
        }
      }
    }
    /* order by weight (:1941-1951) */
    if (count > 1) {
#define SWAP_PRED(a, b)                                                      \
  do {                                                                       \
    int32_t ti_ = pr->pidx[a];                                               \
    pr->pidx[a] = pr->pidx[b];                                               \
    pr->pidx[b] = ti_;                                                       \
    uint64_t tw_ = pr->w[a];                                                 \
    pr->w[a] = pr->w[b];                                                     \
    pr->w[b] = tw_;                                                          \
    uint8_t tr_ = pr->ref[a];                                                \
    pr->ref[a] = pr->ref[b];                                                 \
    pr->ref[b] = tr_;                                                        \
  } while (0)
      if (pr->w[0] > pr->w[1])
        SWAP_PRED(0, 1);
      if (count == 3 && pr->w[1] > pr->w[2]) {
        SWAP_PRED(1, 2);
        if (pr->w[0] > pr->w[1])
          SWAP_PRED(0, 1);
      }
#undef SWAP_PRED
    }
  }
  free(packed);
  free(bias_pos_lod);
  free(bret);
  bbox_free(&hb);
  if (intra) {
    free(bref);
    bbox_free(&hi);
  }
}

/* buildPredictorsFast.  raw != 0: stop before computeWeights (weights are
 * squared distances).  Outputs as ref_lod_generate in ref_lod_harness.inc. */
void oracle_compute_weights(int32_t n, int32_t* neigh_count, uint64_t* w);

static int
lod_generate(
  const gpcc_lod_params* lp, const int32_t* xyz, int32_t n, int32_t raw,
  int32_t* neigh_count, int32_t* neigh_index, uint64_t* weight64,
  int32_t* indexes_out, int32_t* num_points_in_lod, int32_t* num_lods,
  /* attribute inter prediction (NULL / 0: none) */
  const int32_t* xyz_ref, int32_t n_ref, int32_t search_range, int32_t frame_distance,
  int32_t* inter_ref)
{
  if (xyz_ref && (lp->scalable_lifting_enabled_flag || lp->canonical_point_order_flag
                  || lp->max_points_per_sort_log2_plus1 || n_ref <= 0))
    return -2; /* not restated together */
  voxel_t* pv = (voxel_t*)malloc(sizeof(voxel_t) * (size_t)n);
  for (int i = 0; i < n; i++) {
    pv[i].code = morton_addr(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]);
    memcpy(pv[i].pos, &xyz[3 * i], 12);
    pv[i].index = i;
  }
  /* canonical_point_order_flag / max_points_per_sort_log2_plus1 (:2322-2331): the points are
   * taken in the order they come (or sorted in chunks).  Restated for the case the octree
   * geometry coder produces -- the points ARE in (Morton code, index) order, so neither the
   * full sort nor the chunked one moves anything; any other order is not restated. */
  if (lp->canonical_point_order_flag || lp->max_points_per_sort_log2_plus1) {
    for (int i = 1; i < n; i++)
      if (pv[i - 1].code > pv[i].code) {
        free(pv);
        return -2;
      }
  }
  qsort(pv, (size_t)n, sizeof(voxel_t), cmp_voxel);
  int32_t* bias_pos = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
  for (int i = 0; i < n; i++)
    for (int d = 0; d < 3; d++)
      bias_pos[3 * i + d] = pv[i].pos[d] * lp->lod_neigh_bias[d];

  /* the reference frame: Morton order, biased, boxes over the whole list (:2352-2376, :1270-1292) */
  inter_frame_t irs;
  const inter_frame_t* ir = NULL;
  voxel_t* pvr = NULL;
  int32_t *bias_ref = NULL, *ident = NULL;
  if (xyz_ref) {
    pvr = (voxel_t*)malloc(sizeof(voxel_t) * (size_t)n_ref);
    for (int i = 0; i < n_ref; i++) {
      pvr[i].code = morton_addr(xyz_ref[3 * i], xyz_ref[3 * i + 1], xyz_ref[3 * i + 2]);
      memcpy(pvr[i].pos, &xyz_ref[3 * i], 12);
      pvr[i].index = i;
    }
    qsort(pvr, (size_t)n_ref, sizeof(voxel_t), cmp_voxel);
    bias_ref = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n_ref);
    ident = (int32_t*)malloc(sizeof(int32_t) * (size_t)n_ref);
    for (int i = 0; i < n_ref; i++) {
      ident[i] = i;
      for (int d = 0; d < 3; d++)
        bias_ref[3 * i + d] = pvr[i].pos[d] * lp->lod_neigh_bias[d];
    }
    irs.pv = pvr;
    irs.n = n_ref;
    irs.bias_pos = bias_ref;
    irs.identity = ident;
    irs.range = search_range;
    bbox_build(&irs.boxes, bias_ref, n_ref);
    ir = &irs;
  }

  int32_t* input = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* retained = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* indexes = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* pt2pred = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  raw_pred_t* preds = (raw_pred_t*)calloc((size_t)n, sizeof(raw_pred_t));
  int n_in = n, n_idx = 0, pred_index = n;
  for (int i = 0; i < n; i++)
    input[i] = i;
  int32_t npl[GPCC_MAX_LODS + 2];
  int nl = 0;
  npl[nl++] = n;
  /* AttributeParameterSet::maxNumDetailLevels hls.h:835-839 */
  const int max_levels = lp->scalable_lifting_enabled_flag ? 21 : lp->num_detail_levels_minus1 + 1;
  /* scalable lifting (:2377-2380, :2416-2448; whole slices only: minGeomNodeSizeLog2 = 0 and
   * no skipped points): while a new refinement layer is larger than all the finer ones
   * together, the finer layers are searched AGAIN, against the new (coarser) retained set */
  int concatenate = lp->scalable_lifting_enabled_flag != 0;
  int32_t* packed_of_layers = concatenate ? (int32_t*)malloc(sizeof(int32_t) * (size_t)n) : NULL;
  for (int lod = 0; n_in > 0 && lod < max_levels; lod++) {
    const int start = n_idx;
    int n_ret = 0;
    if (lod == max_levels - 1) {
      for (int i = 0; i < n_in; i++)
        indexes[n_idx++] = input[i];
    } else if (lp->scalable_lifting_enabled_flag) {
      /* subsample :2230-2235: octree level = LoD index, the direction alternates */
      subsample_by_octree(pv, input, n_in, lod, 0, lod & 1, retained, &n_ret, indexes, &n_idx);
    } else if (lp->lod_decimation_type == 1) {
      subsample_by_decimation(
        input, n_in, lp->lod_sampling_period[lod], retained, &n_ret, indexes, &n_idx);
    } else if (lp->lod_decimation_type == 2) {
      subsample_by_octree(
        pv, input, n_in, lp->dist2 + lp->attr_dist2_delta + lod,
        lp->lod_sampling_period[lod], 1, retained, &n_ret, indexes, &n_idx);
    } else {
      subsample_by_distance(
        pv, input, n_in, lp->dist2 + lp->attr_dist2_delta + lod, retained, &n_ret,
        indexes, &n_idx);
    }
    if (concatenate && start != n_idx) {
      memcpy(&packed_of_layers[start], &indexes[start], sizeof(int32_t) * (size_t)(n_idx - start));
      if (n_idx - start <= start)
        concatenate = 0;
      else {
        memcpy(indexes, packed_of_layers, sizeof(int32_t) * (size_t)start);
        pred_index = n;
        for (int l = 0; l < lod; l++)
          compute_nearest_neighbours(
            lp, pv, n, bias_pos, retained, n_ret, indexes, n - npl[l], n - npl[l + 1], l, preds,
            pt2pred, &pred_index, ir);
      }
    }
    compute_nearest_neighbours(
      lp, pv, n, bias_pos, retained, n_ret, indexes, start, n_idx, lod, preds, pt2pred,
      &pred_index, ir);
    if (n_ret && nl < GPCC_MAX_LODS + 1)
      npl[nl++] = n_ret;
    int32_t* t = input;
    input = retained;
    retained = t;
    n_in = n_ret;
  }
  /* reverse, updatePredictors :2273-2296 */
  for (int i = 0; i < n; i++)
    indexes_out[i] = indexes[n - 1 - i];
  for (int i = 0; i < n; i++) {
    raw_pred_t* p = &preds[i];
    if (p->count < 2) {
      p->w[0] = 1;
    } else if (p->w[0] == 0) {
      p->count = 1;
      p->w[0] = 1;
    }
    neigh_count[i] = p->count;
    for (int k = 0; k < 3; k++) {
      /* entries beyond the count keep their point index un-mapped, as the
       * reference leaves them; a neighbour in the reference frame keeps ITS point
       * index and is moved away by the frame distance (:2286-2293) */
      const int live = k < p->count;
      neigh_index[3 * i + k] = live && !p->ref[k] ? pt2pred[p->pidx[k]] : p->pidx[k];
      weight64[3 * i + k] = p->w[k] + (live && p->ref[k] ? (uint64_t)(int64_t)frame_distance : 0);
      if (inter_ref)
        inter_ref[3 * i + k] = p->ref[k];
    }
  }
  *num_lods = nl;
  for (int i = 0; i < nl; i++)
    num_points_in_lod[i] = npl[nl - 1 - i];
  if (!raw) {
    oracle_compute_weights(n, neigh_count, weight64);
    /* PCCPredictor::blendWeights :635-693 (predicting transform only,
     * AttributeCommon.cpp:66-69): positions of the three neighbours */
    if (lp->attr_encoding == 1 && lp->pred_weight_blending_enabled_flag) {
      for (int i = 0; i < n; i++) {
        if (neigh_count[i] != 3)
          continue;
        const int32_t* q[3];
        for (int k = 0; k < 3; k++)
          q[k] = preds[i].ref[k] ? &xyz_ref[3 * neigh_index[3 * i + k]]
                                 : &xyz[3 * indexes_out[neigh_index[3 * i + k]]];
        int64_t d01 = 0, d02 = 0, d12 = 0;
        for (int c = 0; c < 3; c++) {
          const int64_t a = (int64_t)q[0][c] - q[1][c], b = (int64_t)q[0][c] - q[2][c],
                        e = (int64_t)q[1][c] - q[2][c];
          d01 += a * a;
          d02 += b * b;
          d12 += e * e;
        }
        const int dd = 10, bb = 1, cc = 5;
        const int b1 = d01 <= d02 ? bb : cc;
        const int b2 = d01 <= d12 ? cc : bb;
        const int b3 = d02 <= d12 ? bb : cc;
        const int w0 = (int)weight64[3 * i], w1 = (int)weight64[3 * i + 1], w2 = (int)weight64[3 * i + 2];
        const int v0 = (w0 * dd + w1 * (16 - dd - b2) + w2 * b3) >> 4;
        const int v1 = (w0 * b1 + w1 * dd + w2 * (16 - dd - b3)) >> 4;
        weight64[3 * i] = (uint64_t)(int64_t)v0;
        weight64[3 * i + 1] = (uint64_t)(int64_t)v1;
        weight64[3 * i + 2] = (uint64_t)(int64_t)(256 - v0 - v1);
      }
    }
  }
  if (ir) {
    bbox_free(&irs.boxes);
    free(pvr);
    free(bias_ref);
    free(ident);
  }
  free(pv);
  free(packed_of_layers);
  free(bias_pos);
  free(input);
  free(retained);
  free(indexes);
  free(pt2pred);
  free(preds);
  return 0;
}

int
oracle_lod_generate(
  const gpcc_lod_params* lp, const int32_t* xyz, int32_t n, int32_t raw,
  int32_t* neigh_count, int32_t* neigh_index, uint64_t* weight64,
  int32_t* indexes_out, int32_t* num_points_in_lod, int32_t* num_lods)
{
  return lod_generate(
    lp, xyz, n, raw, neigh_count, neigh_index, weight64, indexes_out, num_points_in_lod, num_lods,
    NULL, 0, 0, 0, NULL);
}

/* ... with attribute inter prediction (AttributeInterPredParams::enableAttrInterPred): the
 * neighbour search also looks into the reference frame xyz_ref [n_ref][3]
 * (computeNearestNeighbors :1606-1796), search_range = abh.attrInterPredSearchRange.
 * inter_ref [n][3] out: PCCNeighborInfo::interFrameRef; for such a neighbour neigh_index is
 * the point index IN THE REFERENCE FRAME and the distance carries frame_distance. */
int
oracle_lod_generate_inter(
  const gpcc_lod_params* lp, const int32_t* xyz, int32_t n, const int32_t* xyz_ref, int32_t n_ref,
  int32_t search_range, int32_t frame_distance, int32_t raw, int32_t* neigh_count,
  int32_t* neigh_index, uint64_t* weight64, int32_t* indexes_out, int32_t* num_points_in_lod,
  int32_t* num_lods, int32_t* inter_ref)
{
  if (!xyz_ref)
    return -1;
  return lod_generate(
    lp, xyz, n, raw, neigh_count, neigh_index, weight64, indexes_out, num_points_in_lod, num_lods,
    xyz_ref, n_ref, search_range, frame_distance, inter_ref);
}

/* estimateDist2, tmc3/AttributeEncoder.cpp:1684-1720: every samplingPeriod-th
 * point (coded order), nearest other point inside a +-searchRange index
 * window, percentile of those squared distances -> smallest shift with
 * 3 << (2 shift) >= dist2.  dists_out (optional) receives the per-sample
 * minima in sample order. */
static int
cmp_i64(const void* a, const void* b)
{
  const int64_t x = *(const int64_t*)a, y = *(const int64_t*)b;
  return (x > y) - (x < y);
}

int
oracle_estimate_dist2(
  const int32_t* xyz, int32_t n, int32_t sampling_period, int32_t search_range,
  float percentile, int64_t* dists_out)
{
  if (n < 2)
    return 0;
  const int ns = (n + sampling_period - 1) / sampling_period;
  int64_t* d = (int64_t*)malloc(sizeof(int64_t) * (size_t)ns);
  int m = 0;
  for (int index = 0; index < n; index += sampling_period) {
    const int k0 = index - search_range > 0 ? index - search_range : 0;
    const int k1 = index + search_range < n - 1 ? index + search_range : n - 1;
    int64_t best = INT64_MAX;
    for (int k = k0; k <= k1; k++) {
      if (k == index)
        continue;
      int64_t s = 0;
      for (int c = 0; c < 3; c++) {
        const int64_t t = (int64_t)xyz[3 * index + c] - xyz[3 * k + c];
        s += t * t;
      }
      if (s < best)
        best = s;
    }
    d[m++] = best;
  }
  if (dists_out)
    memcpy(dists_out, d, sizeof(int64_t) * (size_t)m);
  /* int p = int(std::floor(dists.size() * percentileEstimate)): size_t -> float product */
  const int p = (int)floorf((float)(size_t)m * percentile);
  qsort(d, (size_t)m, sizeof(int64_t), cmp_i64);
  const int64_t dist2 = d[p];
  free(d);
  int shift = 0;
  while (((int64_t)3 << (shift << 1)) < dist2 && shift < 20)
    shift++;
  return shift;
}
