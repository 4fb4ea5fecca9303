// lod_scalable.hpp -- the level loop of the LoD build under
// aps.scalable_lifting_enabled_flag (buildPredictorsFast,
// tmc3/PCCTMC3Common.h:2300-2469 with the scalable branches :1174-1176,
// :1232-1236, :1918-1939, :2230-2235, :2377-2448; whole slices:
// minGeomNodeSizeLog2 = 0, no skipped points).
//
// What changes against the level loop of lod_build_core (gpcc_attr_mi355.hip):
//   * always 21 levels (AttributeParameterSet::maxNumDetailLevels, hls.h:835-839);
//   * sub-sampling is the octree one (lod_centroid_* kernels): node size = LoD
//     index, every node is a group, the walk direction alternates;
//   * the search of LoD l sees all positions at the corner of their node of
//     size 2^l (lod_node_corner_bpos_kernel), in cells of 2^(l+1), and drops
//     neighbours beyond max_neigh_range (lod_nn_search_kernel<true>);
//   * while a new refinement layer is larger than all finer layers together the
//     finer layers are searched AGAIN against the new retained set
//     ("concatenateLayers").  A predictor's slot depends on its position in the
//     coding order only, so a repeated search simply overwrites its results.
//
// Launches are written with hipLaunchKernelGGL and the HIP runtime calls are
// the plain ones, so the same text runs under the CPU wavefront emulator
// (tests/emu) -- the LoD structure of this file is pinned there against the
// oracle, which is pinned against the compiled reference.
#pragma once

#include <algorithm>
#include <vector>

#include "gpcc_attr_mi355.h"
#include "lod_kernels.hpp"

namespace gpcc {

// device workspace of one LoD build (carved by lod_build_core)
struct LodWork {
  int32_t n;
  const int64_t* code;   // [n] sorted Morton codes
  const int32_t* order;  // [n] point index of each sorted entry
  const int32_t* pos;    // [n][3] sorted positions
  const int32_t* bpos;   // [n][3] sorted positions * lodNeighBias
  int32_t* bpos_lod;     // [n][3] scratch: node-corner positions of the LoD searched
  int32_t *list_a, *list_b;  // [n + 1] input / retained, ping-pong
  int32_t* refine;       // [n + 1] coding-order list of packed indices
  uint8_t *flags, *heads;    // [n + 1]
  int32_t *nxt0, *nj0, *nj1; // [n + 2] group pointers of the octree sub-sampling
  int64_t* ret_key;      // [n + 1]
  int32_t* counts;       // [2]
  unsigned long long* scan;  // [1024]
  long long* atlas_limit;
  int32_t* box[2][3][2];
  int32_t *pred_count, *pred_point;
  uint64_t* pred_dist2;
  int32_t *pt2pred, *indexes;
};

inline int
lod_grid(int64_t items, int per_block)
{
  int64_t g = (items + per_block - 1) / per_block;
  g = std::min<int64_t>(std::max<int64_t>(g, 1), 1 << 16);
  return (int)g;
}

#define GPCC_LODS_TRY(expr)     \
  do {                          \
    hipError_t e_ = (expr);     \
    if (e_ != hipSuccess)       \
      return e_;                \
  } while (0)

// -> cumulative sizes in `npl` as the reference pushes them (n first, then the
// retained count of every level that retains); *scan_epoch is the partition
// kernel's epoch counter of the build
inline hipError_t
lod_scalable_levels(
  const gpcc_lod_params* lp, const LodWork& w, hipStream_t st, std::vector<int32_t>* npl,
  int* scan_epoch)
{
  constexpr int kLevels = 21;
  const int n = w.n;
  const bool unit_bias =
    lp->lod_neigh_bias[0] == 1 && lp->lod_neigh_bias[1] == 1 && lp->lod_neigh_bias[2] == 1;

  auto build_boxes = [&](int which, const int32_t* list, int cnt, const int32_t* bpos) {
    const int c0 = (cnt + 31) >> 5, c1 = (c0 + 31) >> 5;
    hipLaunchKernelGGL(
      lod_box0_kernel, dim3(lod_grid(std::max(c0, 1), 256)), dim3(256), 0, st, cnt, list, bpos,
      w.box[which][0][0], w.box[which][0][1]);
    hipLaunchKernelGGL(
      lod_box_up_kernel, dim3(lod_grid(std::max(c1, 1), 256)), dim3(256), 0, st, c0,
      (const int32_t*)w.box[which][0][0], (const int32_t*)w.box[which][0][1], w.box[which][1][0],
      w.box[which][1][1]);
    hipLaunchKernelGGL(
      lod_box_up_kernel, dim3(1), dim3(256), 0, st, c1, (const int32_t*)w.box[which][1][0],
      (const int32_t*)w.box[which][1][1], w.box[which][2][0], w.box[which][2][1]);
  };

  // nearest neighbours of the layer [s, e) of the coding-order list at LoD `l`
  // against the retained list `ret`
  auto search = [&](int l, int s, int e, const int32_t* ret, int n_ret) -> hipError_t {
    const int n_ref = e - s;
    if (n_ref <= 0)
      return hipSuccess;
    const uint32_t mask = l ? 0xffffffffu << l : 0xffffffffu;
    const int32_t* bpos = w.bpos;
    if (l) {
      hipLaunchKernelGGL(
        lod_node_corner_bpos_kernel, dim3(lod_grid(n, 256)), dim3(256), 0, st, n, w.pos, mask,
        lp->lod_neigh_bias[0], lp->lod_neigh_bias[1], lp->lod_neigh_bias[2], w.bpos_lod);
      bpos = w.bpos_lod;
    }
    NnCtx nc{};
    nc.n = n;
    nc.code = w.code;
    nc.order = w.order;
    nc.bpos = bpos;
    nc.retained = ret;
    nc.ret_key = w.ret_key;
    nc.n_ret = n_ret;
    nc.refine = w.refine + s;
    nc.n_ref = n_ref;
    nc.start = s;
    nc.shift3 = 3 * (1 + l);
    nc.boundary = std::min(63, nc.shift3 + kAtlasBits);
    nc.distribution = lp->prediction_with_distribution_enabled;
    nc.range_inter = lp->inter_lod_search_range;
    nc.range_intra = lp->intra_lod_search_range;
    nc.intra = l >= lp->intra_lod_prediction_skip_layers;
    nc.max_neigh = lp->num_pred_nearest_neighbours_minus1 + 1;
    for (int lev = 0; lev < 3; lev++)
      for (int m = 0; m < 2; m++) {
        nc.box_ret[lev][m] = w.box[0][lev][m];
        nc.box_ref[lev][m] = w.box[1][lev][m];
      }
    nc.atlas_limit = w.atlas_limit;
    nc.pred_count = w.pred_count;
    nc.pred_point = w.pred_point;
    nc.pred_dist2 = w.pred_dist2;
    nc.pt2pred = w.pt2pred;
    nc.indexes = w.indexes;
    nc.pos = w.pos;
    nc.node_mask = mask;
    nc.unit_bias = unit_bias;
    nc.prune_dist = (int64_t)(3ll * (lp->max_neigh_range_minus1 + 1)) << (2 * l);
    if (n_ret > 0) {
      hipLaunchKernelGGL(
        lod_ret_keys_kernel, dim3(lod_grid(n_ret, 256)), dim3(256), 0, st, n_ret, ret, w.code,
        nc.shift3, w.ret_key);
      build_boxes(0, ret, n_ret, bpos);
    }
    if (nc.intra)
      build_boxes(1, w.refine + s, n_ref, bpos);
    const long long inf = INT64_MAX;
    GPCC_LODS_TRY(hipMemcpyAsync(w.atlas_limit, &inf, sizeof(inf), hipMemcpyHostToDevice, st));
    // (the host value is read when the call returns only for pageable memory of
    // this size: a stack variable is copied before hipMemcpyAsync returns)
    if (n_ret > 0)
      hipLaunchKernelGGL(
        lod_atlas_limit_kernel, dim3(lod_grid(n_ret, 256)), dim3(256), 0, st, nc, w.atlas_limit);
    hipLaunchKernelGGL(
      lod_nn_search_kernel<true>, dim3(lod_grid(n_ref, 256)), dim3(256), 0, st, nc);
    return hipGetLastError();
  };

  npl->clear();
  npl->push_back(n);
  int32_t* d_input = w.list_a;
  int32_t* d_ret = w.list_b;
  int n_in = n, n_idx = 0;
  bool concatenate = true;
  for (int lod = 0; n_in > 0 && lod < kLevels; lod++) {
    const int start = n_idx;
    int n_ret = 0, n_ref = 0;
    if (lod == kLevels - 1 || n_in == 1) {
      GPCC_LODS_TRY(hipMemcpyAsync(
        w.refine + start, d_input, sizeof(int32_t) * n_in, hipMemcpyDeviceToDevice, st));
      n_ref = n_in;
    } else {
      // subsampleByOctree (:2146-2194) with octreeNodeSizeLog2 = lod, no minimum
      // group size (period 1 = every node closes a group), direction = lod & 1
      LodCtx lc{};
      lc.code = w.code;
      lc.pos = w.pos;
      lc.input = d_input;
      lc.n_in = n_in;
      lc.shift3 = 3 * (lod + 1);
      lc.flags = w.flags;
      int32_t* nj[2] = {w.nj0, w.nj1};
      hipLaunchKernelGGL(
        lod_centroid_next_kernel, dim3(lod_grid(n_in + 1, 256)), dim3(256), 0, st, lc, 1, w.nxt0);
      GPCC_LODS_TRY(hipMemsetAsync(w.heads, 0, (size_t)n_in + 1, st));
      GPCC_LODS_TRY(hipMemsetAsync(w.heads, 1, 1, st));
      const int32_t* cur = w.nxt0;
      for (int r = 0, reach = 1; reach < n_in; r++, reach *= 2) {
        hipLaunchKernelGGL(
          lod_centroid_jump_kernel, dim3(lod_grid(n_in + 1, 256)), dim3(256), 0, st, n_in, cur,
          nj[r & 1], w.heads);
        cur = nj[r & 1];
      }
      hipLaunchKernelGGL(
        lod_centroid_pick_kernel, dim3(lod_grid(n_in, 256)), dim3(256), 0, st, lc, lod,
        (const int32_t*)w.nxt0, (const uint8_t*)w.heads, lod & 1);
      // retained / refinement lists
      (*scan_epoch)++;
      const int grid = (int)std::min<int64_t>(1024, ((int64_t)n_in + 1023) / 1024);
      hipLaunchKernelGGL(
        lod_partition_kernel, dim3(std::max(grid, 1)), dim3(256), 0, st, n_in,
        (const uint8_t*)w.flags, (const int32_t*)d_input, d_ret, w.refine + start, w.counts, w.scan,
        *scan_epoch);
      int32_t h = 0;
      GPCC_LODS_TRY(hipMemcpyAsync(&h, w.counts, sizeof(int32_t), hipMemcpyDeviceToHost, st));
      GPCC_LODS_TRY(hipStreamSynchronize(st));
      n_ret = h;
      n_ref = n_in - n_ret;
    }
    n_idx += n_ref;

    if (concatenate && n_ref > 0) {
      if (n_ref <= start)
        concatenate = false;
      else
        for (int l = 0; l < lod; l++)
          GPCC_LODS_TRY(search(l, n - (*npl)[l], n - (*npl)[l + 1], d_ret, n_ret));
    }
    GPCC_LODS_TRY(search(lod, start, n_idx, d_ret, n_ret));
    if (n_ret > 0)
      npl->push_back(n_ret);
    std::swap(d_input, d_ret);
    n_in = n_ret;
  }
  return hipSuccess;
}

#undef GPCC_LODS_TRY

}  // namespace gpcc
