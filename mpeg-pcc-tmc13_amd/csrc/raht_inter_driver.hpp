// raht_inter_driver.hpp -- workspace layout and launch sequence of RAHT with attribute inter prediction
// (raht_inter.hpp, raht_tile.hpp with INTER): one slice, no sub-node prediction, no integer Haar, no region
// QP offsets.  Like cx_driver.hpp the sequence is a template over how the host learns the tree's shape, so
// the same code runs under the CPU wavefront emulator of the test tier (tests/emu).
//
//   tree_count / tree_scan / tree_emit     level arrays of the current frame
//   frame_sum / frame_scan / frame_prefix  modular prefix sums of the reference frame's attributes
//   schedule                               level plan (no coarse levels: every level is a launch)
//   per level, top down (tmc3/RAHT.cpp:1165-1345):
//     [inter_tap + inter_tap_finish]       encoder, estimated taps
//     encoder  tile<kAnalyze, INTER> -> rdoq_resolve [x 2 -> rate_pack / p1 / p0_bits / sum / decide -> commit]
//              -> tile<kSynthRec>
//     decoder  tile<kSynth, INTER>
//   with sub-node prediction (the reference's default flag) the level kernels are the dependency kernels
//   (raht_subnode.hpp) instead:
//     encoder  prepass -> level_sub<kLossySub, INTER> [+ level_sub<kLossySub> on a second workspace: the intra
//              candidate, whose reconstruction feeds ITS later blocks -> rate_* / decide -> commit_sub]
//     decoder  prepass -> level_sub<kSynth, INTER>
//   finish                                 duplicates, write-back
#pragma once
#include <algorithm>

#include "raht_edges.hpp"
#include "raht_inter.hpp"
#include "raht_links.hpp"
#include "raht_subnode.hpp"
#include "raht_tile.hpp"
#include "raht_tree.hpp"

namespace gpcc {

// AttributeInterPredParams / the APS fields of the tool (PCCTMC3Common.h:236-298, hls.h)
struct InterTools {
  int depth_limit = 0;   // raht_inter_prediction_depth_minus1 + 1
  int layer_rdo = 0;     // raht_enable_inter_intra_layer_RDO
  int filter_est = 0;    // enableFilterEstimation
  int skip_layers = 0;   // skipInitLayersForFiltering
  int bits_cur = 0;      // bit length of pos[0] ^ pos[n - 1] (the levels under the current tree's top) ...
  int bits_ref = -1;     // ... and of the frame's; -1: the frame has a single point
  // decoder: what the bitstream signalled
  const int32_t* modes = nullptr;
  int num_modes = 0;
  const int32_t* taps = nullptr;
  int num_taps = 0;
};

struct InterWork {
  int n = 0, c = 0, nlev = 0, n_ref = 0;
  bool encoder = false;
  TreeView tv{};
  int32_t* pt_off = nullptr;
  uint32_t* tile_cnt = nullptr;
  int32_t* tile_attr = nullptr;
  int32_t* attr_prefix = nullptr;
  SliceSched* sched = nullptr;
  gpcc_raht_params* params = nullptr;
  int64_t* rec[2] = {nullptr, nullptr};
  int64_t* rec_us[2] = {nullptr, nullptr};
  int32_t* nneigh[2] = {nullptr, nullptr};
  uint32_t *desc = nullptr, *idesc = nullptr;
  int64_t *ptrans = nullptr, *iptrans = nullptr;
  int32_t* icoeffs = nullptr;
  int32_t* rtile_base = nullptr;
  unsigned long long *rtile_state = nullptr, *irtile_state = nullptr;
  int32_t *slice_l = nullptr, *islice_l = nullptr;
  int num_rtiles = 0;
  int32_t* frame_tile = nullptr;
  int32_t* frame_prefix = nullptr;
  int32_t* tap_words = nullptr;  // [kMaxLevels]
  int32_t* modes = nullptr;      // [32] then num (unused), taps [32], num_taps
  int32_t* taps = nullptr;
  int32_t* num_taps = nullptr;
  unsigned long long* tap_acc = nullptr;
  RateState* rs = nullptr;
  int32_t* pb = nullptr;
  double* term = nullptr;
  unsigned long long *nzw = nullptr, *bigw = nullptr;
  int wstride = 0;
  // sub-node prediction: the dependency kernels' workspace, and a second one for the intra candidate
  bool sub = false;
  LinkView lv{};  // neighbour links of the dependency kernels (raht_links.hpp)
  bool links = false;
  bool f64 = false;  // the intra candidate's dependency kernel in ArithF64 (raht_arith.hpp; decided by the caller)
  int32_t* worklist = nullptr;
  int32_t* work_count = nullptr;  // [kMaxLevels] then tickets [kMaxLevels][8]
  int32_t* ticket2 = nullptr;     // [kMaxLevels][8]
  unsigned long long* scan_state = nullptr;
  uint8_t* pocc = nullptr;
  uint32_t *mbox = nullptr, *mbox2 = nullptr;
  unsigned long long *rdoq_state = nullptr, *rdoq_state2 = nullptr;
  int64_t *irec = nullptr, *irec_us = nullptr;
  // region QP offsets: the node QPs averaged up the tree (per level) and handed down (tmc3/RAHT.cpp:185-189, 246-253)
  bool has_qp = false;
  int32_t* asc_qp[kMaxLevels] = {};
  int32_t** asc_qp_tab = nullptr;
  int32_t* dqp[2] = {nullptr, nullptr};
  // integer Haar kernel: low-pass values of every level of the current frame (encoder) and of the reference
  // frame, which gets level arrays of its own; the pointer tables are filled by the caller (host copies beside)
  bool haar = false;
  int nlev_ref = 0;
  int32_t* haar_lf[kMaxLevels] = {};
  int32_t** haar_lf_tab = nullptr;
  int32_t* dup_hf = nullptr;
  TreeView tvr{};
  int32_t* pt_off_ref = nullptr;
  uint32_t* tile_cnt_ref = nullptr;
  int32_t* ref_lf[kMaxLevels] = {};
  int32_t** ref_lf_tab = nullptr;
  int32_t* ref_dup_hf = nullptr;
};

// Under the integer Haar kernel the frame's nodes come from level arrays of its own: the two trees have to line
// up on octree levels (they descend from their own tops, bit level B_ref - B + 3 level).
inline bool
inter_supported(const gpcc_raht_params* p, int64_t n, const InterTools& tl)
{
  if (n < 2)
    return false;
  if (p->integer_haar_enable_flag)
    return tl.bits_ref < 0 || (tl.bits_ref - tl.bits_cur) % 3 == 0;
  return true;
}

inline bool
inter_sub(const gpcc_raht_params* p)
{
  return p->raht_prediction_enabled_flag && p->raht_subnode_prediction_enabled_flag;
}

// `take(bytes)` hands out 256-byte aligned storage (or only counts)
template<class Take>
void
inter_carve(Take&& take, InterWork& w)
{
  const int n = w.n, c = w.c, nlev = w.nlev;
  auto arr = [&](size_t count, size_t elem) { return take(count * elem); };
  w.pt_off = (int32_t*)arr(2, 4);
  for (int li = 0; li < nlev; li++) {
    int64_t cap = n;
    const int up = nlev - 1 - li;
    if (up < 11) {
      const int64_t full = (int64_t)1 << (3 * up);
      cap = cap < full ? cap : full;
    }
    w.tv.cap[li] = (int32_t)cap;
    w.tv.key[li] = (int64_t*)arr(cap + 1, 8);
    w.tv.fp[li] = (int32_t*)arr(cap + 2, 4);
    w.tv.fc[li] = (int32_t*)arr(cap + 2, 4);
    w.tv.soff[li] = (int32_t*)arr(2, 4);
  }
  w.tv.nlev = nlev;
  w.tv.num_slices = 1;
  w.tv.n_total = n;
  w.tv.num_tiles = (n + kTilePoints - 1) / kTilePoints;
  w.tv.pt_off = w.pt_off;
  w.tile_cnt = (uint32_t*)arr((size_t)w.tv.num_tiles * nlev, 4);
  w.tile_attr = (int32_t*)arr((size_t)w.tv.num_tiles * c, 4);
  w.attr_prefix = w.encoder ? (int32_t*)arr(((size_t)n + 1) * c, 4) : nullptr;
  w.sched = (SliceSched*)arr(1, sizeof(SliceSched));
  w.params = (gpcc_raht_params*)arr(1, sizeof(gpcc_raht_params));
  for (int i = 0; i < 2; i++) {
    w.rec[i] = (int64_t*)arr((size_t)n * c, 8);
    w.rec_us[i] = (int64_t*)arr((size_t)n * c, 8);
    w.nneigh[i] = (int32_t*)arr((size_t)n, 4);
  }
  w.num_rtiles = (n + kRdoqTile - 1) / kRdoqTile;
  if (w.encoder) {
    w.desc = (uint32_t*)arr((size_t)n, 4);
    w.idesc = (uint32_t*)arr((size_t)n, 4);
    w.ptrans = (int64_t*)arr((size_t)n * c, 8);
    w.iptrans = (int64_t*)arr((size_t)n * c, 8);
    w.icoeffs = (int32_t*)arr((size_t)n * c, 4);
    w.rtile_base = (int32_t*)arr(2, 4);
    w.rtile_state = (unsigned long long*)arr((size_t)w.num_rtiles + 1, 8);
    w.irtile_state = (unsigned long long*)arr((size_t)w.num_rtiles + 1, 8);
    w.slice_l = (int32_t*)arr(2, 4);
    w.islice_l = (int32_t*)arr(2, 4);
    w.rs = (RateState*)arr(1, sizeof(RateState));
    w.pb = (int32_t*)arr((size_t)2 * c * n, 4);
    w.wstride = (n + 63) / 64 + 1;
    w.nzw = (unsigned long long*)arr((size_t)2 * c * w.wstride, 8);
    w.bigw = (unsigned long long*)arr((size_t)2 * c * w.wstride, 8);
    w.term = (double*)arr((size_t)2 * c * n, 8);
    w.tap_acc = (unsigned long long*)arr(2, 8);
  }
  if (w.has_qp) {
    for (int li = 0; li < nlev; li++)
      w.asc_qp[li] = (int32_t*)arr(((size_t)w.tv.cap[li] + 1) * 2, 4);
    w.asc_qp_tab = (int32_t**)arr(kMaxLevels, sizeof(void*));
    for (int i = 0; i < 2; i++)
      w.dqp[i] = (int32_t*)arr((size_t)n * 2, 4);
  }
  if (w.haar) {
    if (w.encoder) {
      for (int li = 0; li < nlev; li++)
        w.haar_lf[li] = (int32_t*)arr((size_t)w.tv.cap[li] * c + 1, 4);
      w.haar_lf_tab = (int32_t**)arr(kMaxLevels, sizeof(void*));
      w.dup_hf = (int32_t*)arr((size_t)n * c + 1, 4);
    }
    const int nr = w.n_ref, nlr = w.nlev_ref;
    w.pt_off_ref = (int32_t*)arr(2, 4);
    for (int li = 0; li < nlr; li++) {
      int64_t cap = nr;
      const int up = nlr - 1 - li;
      if (up < 11) {
        const int64_t full = (int64_t)1 << (3 * up);
        cap = cap < full ? cap : full;
      }
      w.tvr.cap[li] = (int32_t)cap;
      w.tvr.key[li] = (int64_t*)arr(cap + 1, 8);
      w.tvr.fp[li] = (int32_t*)arr(cap + 2, 4);
      w.tvr.fc[li] = (int32_t*)arr(cap + 2, 4);
      w.tvr.soff[li] = (int32_t*)arr(2, 4);
      w.ref_lf[li] = (int32_t*)arr((size_t)cap * c + 1, 4);
    }
    w.tvr.nlev = nlr;
    w.tvr.num_slices = 1;
    w.tvr.n_total = nr;
    w.tvr.num_tiles = (nr + kTilePoints - 1) / kTilePoints;
    w.tvr.pt_off = w.pt_off_ref;
    w.tile_cnt_ref = (uint32_t*)arr((size_t)w.tvr.num_tiles * nlr + 1, 4);
    w.ref_lf_tab = (int32_t**)arr(kMaxLevels, sizeof(void*));
    w.ref_dup_hf = (int32_t*)arr((size_t)nr * c + 1, 4);
    if (w.encoder && !w.sub) {
      w.irec = (int64_t*)arr((size_t)n * c, 8);
      w.irec_us = (int64_t*)arr((size_t)n * c, 8);
    }
  }
  w.links = w.sub && links_enabled();
  if (w.links)
    link_carve(take, w.lv, w.tv, n, 1, nlev);
  if (w.sub) {
    w.worklist = (int32_t*)arr((size_t)n + 1, 4);
    w.work_count = (int32_t*)arr(kMaxLevels * 9, 4);
    w.ticket2 = (int32_t*)arr(kMaxLevels * 8, 4);
    w.scan_state = (unsigned long long*)arr(1024, 8);
    w.pocc = (uint8_t*)arr((size_t)n + 1, 1);
    w.mbox = (uint32_t*)arr((size_t)n * c * 4, 4);
    if (w.encoder) {
      w.mbox2 = (uint32_t*)arr((size_t)n * c * 4, 4);
      w.rdoq_state = (unsigned long long*)arr((size_t)n + 1, 8);
      w.rdoq_state2 = (unsigned long long*)arr((size_t)n + 1, 8);
      w.irec = (int64_t*)arr((size_t)n * c, 8);
      w.irec_us = (int64_t*)arr((size_t)n * c, 8);
    }
  }
  const int ftiles = (w.n_ref + kTilePoints - 1) / kTilePoints;
  w.frame_tile = (int32_t*)arr((size_t)ftiles * c + 1, 4);
  w.frame_prefix = (int32_t*)arr(((size_t)w.n_ref + 1) * c, 4);
  w.tap_words = (int32_t*)arr(kMaxLevels, 4);
  w.modes = (int32_t*)arr(32, 4);
  w.taps = (int32_t*)arr(32, 4);
  w.num_taps = (int32_t*)arr(1, 4);
}

constexpr int kInterSubGrid = 1024;  // dependency kernels: resident workgroups (as the intra path's)

// A second stream for the encoder's intra candidate under sub-node prediction: the two candidates of a level are
// independent dependency sweeps, each of which leaves most of the device idle, so they run side by side (the
// waits inside a kernel are only ever for blocks a RUNNING workgroup has claimed: sharing the device cannot
// stall either).  `second` null: one after the other (the emulator harness).
struct InterStreams {
#ifndef GPCC_EMU
  hipStream_t second = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
#else
  void* second = nullptr;
#endif
};

// rows of a level's reconstruction from one workspace to the other: in front of the intra candidate's launch
// (what the prepass copied), and back when that candidate won (`when` null: always)
struct SubCopyCtx {
  TreeView tv;
  const int64_t *src, *src_us;
  int64_t *dst, *dst_us;
  int64_t count;
  const int32_t* when;
};

inline int
sub_copy_grid(int64_t count)
{
  return (int)std::min<int64_t>(std::max<int64_t>((count + 255) / 256, 1), 4096);
}

__global__ __launch_bounds__(256) void
inter_sub_copy_kernel(SubCopyCtx cx)
{
  if (tree_failed(cx.tv))
    return;
  if (cx.when && !*cx.when)
    return;
  for (int64_t x = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; x < cx.count; x += (int64_t)gridDim.x * blockDim.x) {
    cx.dst[x] = cx.src[x];
    cx.dst_us[x] = cx.src_us[x];
  }
}

__global__ __launch_bounds__(64) void
inter_pair_sync_kernel(int32_t* pair, int which)
{
  if (threadIdx.x == 0)
    pair[which ^ 1] = pair[which];
}

// Everything after the uploads of params / pt_off (= {0, n}) / rtile_base (= {0, tiles}) / the two frames.
// d_modes_out / d_taps_out: nothing -- the caller reads w.modes, w.rs->num_modes, w.taps, w.num_taps.
template<int C, class Prof, class Mark, class Wait>
hipError_t
inter_run(
  hipStream_t st, InterWork& w, const InterTools& tl, const gpcc_raht_params* hp, const SharedLut* d_lut,
  const double* d_log2tab, const int64_t* d_ref_pos, const int32_t* d_ref_attrs, int32_t* d_attrs, int32_t* d_coeffs,
  TreeStats* stats, Prof&& prof, Mark&& mark, Wait&& wait, const InterStreams& streams = InterStreams(),
  const int32_t* d_qp_off = nullptr)
{
  const TreeView tv = w.tv;
  const int n = w.n;
  const bool encoder = w.encoder;
  const bool haar = w.haar;
  const int32_t* sum_attrs = (encoder && !haar) ? d_attrs : nullptr;
  const int tgrid = std::min(std::max((tv.num_tiles + 3) / 4, 1), 2048);
  {
    auto t = prof("tree_count", -1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(tree_count_kernel<C>), dim3(tgrid), dim3(256), 0, st, tv, sum_attrs, w.tile_cnt, w.tile_attr);
  }
  {
    auto t = prof("tree_scan", -1);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(tree_scan_kernel<C>), dim3(1), dim3(1024), 0, st, tv, w.tile_cnt, w.tile_attr, w.attr_prefix,
      sum_attrs != nullptr);
  }
  {
    auto t = prof("tree_emit", -1);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(tree_emit_kernel<C>), dim3(tgrid), dim3(256), 0, st, tv, sum_attrs, w.tile_cnt, w.tile_attr,
      w.attr_prefix);
  }
  {
    auto t = prof("schedule", -1);
    hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(256), 0, st, tv, w.sched, (int)hp->num_qp_layers, 0, stats);
  }
  hipError_t e = mark();
  if (e != hipSuccess)
    return e;
  if (!haar) {
    auto t = prof("frame_prefix", -1);
    const int ftiles = (w.n_ref + kTilePoints - 1) / kTilePoints;
    const int fgrid = std::min(std::max((ftiles + 3) / 4, 1), 2048);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(frame_sum_kernel<C>), dim3(fgrid), dim3(256), 0, st, d_ref_attrs, w.n_ref, w.frame_tile);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(frame_scan_kernel<C>), dim3(1), dim3(64), 0, st, w.frame_tile, ftiles);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(frame_prefix_kernel<C>), dim3(fgrid), dim3(256), 0, st, d_ref_attrs, w.n_ref, (const int32_t*)w.frame_tile,
      w.frame_prefix);
  } else {
    // integer Haar: the frame's level arrays and its low-pass values level by level (reduceUnique / reduceLevel
    // with HaarKernel, tmc3/RAHT.cpp:108-205), likewise the current frame's for the encoder
    auto t = prof("frame_tree", -1);
    TreeView tvr = w.tvr;
    tvr.pos = d_ref_pos;
    tvr.error = tv.error;
    w.tvr = tvr;
    const int rgrid = std::min(std::max((tvr.num_tiles + 3) / 4, 1), 2048);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(tree_count_kernel<C>), dim3(rgrid), dim3(256), 0, st, tvr, (const int32_t*)nullptr, w.tile_cnt_ref,
      (int32_t*)nullptr);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(tree_scan_kernel<C>), dim3(1), dim3(1024), 0, st, tvr, w.tile_cnt_ref, (int32_t*)nullptr, (int32_t*)nullptr, 0);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(tree_emit_kernel<C>), dim3(rgrid), dim3(256), 0, st, tvr, (const int32_t*)nullptr,
      (const uint32_t*)w.tile_cnt_ref, (const int32_t*)nullptr, (int32_t*)nullptr);
    AscendCtx ar{};
    ar.tv = tvr;
    ar.attrs = d_ref_attrs;
    ar.haar_lf = w.ref_lf_tab;
    ar.dup_hf = w.ref_dup_hf;
    ar.li = 0;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ascend_leaf_kernel<C>), dim3(std::min(std::max((tvr.cap[0] + 255) / 256, 1), 4096)), dim3(256), 0, st, ar);
    for (int li = 1; li < tvr.nlev; li++) {
      ar.li = li;
      hipLaunchKernelGGL(HIP_KERNEL_NAME(ascend_level_kernel<C>), dim3(std::min(std::max((tvr.cap[li] + 255) / 256, 1), 4096)), dim3(256), 0, st, ar);
    }
  }
  // the current frame's ascent where it is not associative: Haar low-pass values (encoder), region QPs
  if ((haar && encoder) || w.has_qp) {
    auto t = prof("ascend", -1);
    AscendCtx ac{};
    ac.tv = tv;
    ac.attrs = (haar && encoder) ? d_attrs : nullptr;
    ac.qp_off = d_qp_off;
    ac.haar_lf = (haar && encoder) ? w.haar_lf_tab : nullptr;
    ac.asc_qp = w.has_qp ? w.asc_qp_tab : nullptr;
    ac.dup_hf = w.dup_hf;
    ac.dqp_root = w.dqp[1];
    ac.li = 0;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(ascend_leaf_kernel<C>), dim3(std::min(std::max((tv.cap[0] + 255) / 256, 1), 4096)), dim3(256), 0, st, ac);
    for (int li = 1; li < tv.nlev; li++) {
      ac.li = li;
      hipLaunchKernelGGL(HIP_KERNEL_NAME(ascend_level_kernel<C>), dim3(std::min(std::max((tv.cap[li] + 255) / 256, 1), 4096)), dim3(256), 0, st, ac);
    }
    if (w.has_qp)
      hipLaunchKernelGGL(qp_root_kernel, dim3(1), dim3(64), 0, st, ac, (const SliceSched*)w.sched);
  }
  if (encoder) {
    e = hipMemsetAsync(w.rtile_state, 0, ((size_t)w.num_rtiles + 1) * 8, st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.irtile_state, 0, ((size_t)w.num_rtiles + 1) * 8, st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.slice_l, 0xff, 8, st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.islice_l, 0xff, 8, st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.tap_acc, 0, 16, st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.num_taps, 0, 4, st);
    if (e != hipSuccess)
      return e;
    hipLaunchKernelGGL(rate_init_kernel, dim3(1), dim3(64), 0, st, w.rs);
  }
  if (w.sub) {
    e = hipMemsetAsync(w.work_count, 0, kMaxLevels * 9 * sizeof(int32_t), st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.ticket2, 0, kMaxLevels * 8 * sizeof(int32_t), st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.scan_state, 0, 1024 * sizeof(unsigned long long), st);
    if (e != hipSuccess)
      return e;
    e = hipMemsetAsync(w.mbox, 0, (size_t)n * C * 16, st);
    if (e != hipSuccess)
      return e;
    if (encoder) {
      e = hipMemsetAsync(w.mbox2, 0, (size_t)n * C * 16, st);
      if (e != hipSuccess)
        return e;
      e = hipMemsetAsync(w.rdoq_state, 0, ((size_t)n + 1) * 8, st);
      if (e != hipSuccess)
        return e;
      e = hipMemsetAsync(w.rdoq_state2, 0, ((size_t)n + 1) * 8, st);
      if (e != hipSuccess)
        return e;
    }
  }
  e = wait();
  if (e != hipSuccess)
    return e;
  const TreeStats ts = *stats;
  const int top = ts.max_top;

  LevelCtx lc{};
  lc.tv = tv;
  lc.params = w.params;
  lc.sched = w.sched;
  lc.attr_prefix = w.attr_prefix;
  lc.haar_lf = (haar && encoder) ? w.haar_lf_tab : nullptr;
  lc.asc_qp = w.has_qp ? w.asc_qp_tab : nullptr;
  for (int i = 0; i < 2; i++) {
    lc.rec[i] = w.rec[i];
    lc.rec_us[i] = w.rec_us[i];
    lc.nneigh[i] = w.nneigh[i];
    lc.dqp[i] = w.dqp[i];
  }
  lc.coeffs = d_coeffs;
  lc.desc = w.desc;
  lc.ptrans = w.ptrans;
  lc.lut = d_lut;
  lc.error = tv.error;
  lc.slice_l = w.slice_l;
  lc.inter.pos = d_ref_pos;
  lc.inter.prefix = w.frame_prefix;
  lc.inter.n_ref = w.n_ref;
  lc.inter.idesc = w.idesc;
  lc.inter.iptrans = w.iptrans;
  lc.inter.icoeffs = w.icoeffs;
  if (w.sub) {
    lc.worklist = w.worklist;
    lc.work_count = w.work_count;
    lc.scan_state = w.scan_state;
    lc.pocc = w.pocc;
    lc.mbox = w.mbox;
    lc.ticket = w.work_count + kMaxLevels;
    lc.rdoq_state = w.rdoq_state;
  }

  RdoqCtx rc{};
  rc.tv = tv;
  rc.sched = w.sched;
  rc.tile_base = w.rtile_base;
  rc.num_tiles = w.num_rtiles;
  rc.c = C;

  RateCtx rt{};
  rt.tv = tv;
  rt.plane[0] = d_coeffs;
  rt.plane[1] = w.icoeffs;
  rt.n = n;
  rt.c = C;
  rt.pb = w.pb;
  rt.nzw = w.nzw;
  rt.bigw = w.bigw;
  rt.wstride = w.wstride;
  rt.term = w.term;
  rt.rs = w.rs;
  rt.log2tab = d_log2tab;
  rt.slice_l = w.slice_l;
  rt.islice_l = w.islice_l;
  rt.islice_pair = w.islice_l;
  rt.modes = w.modes;
  rt.coeffs = d_coeffs;
  rt.icoeffs = w.icoeffs;
  rt.ptrans = w.ptrans;
  rt.iptrans = w.iptrans;

  LinkSchedule links;
  // (without the RAHT extension a parent with one child -- and one point -- searches too: the links cover
  // the nodes with more than one point only)
  const bool use_links = w.links && top >= 1 && hp->raht_extension != 0;
  if (use_links) {
    links.tv = tv;
    links.lv = w.lv;
    links.begin(st, ts.nodes, top, prof);
  }
  static const int kFixedTaps[7] = {128, 128, 128, 127, 125, 121, 115};
  int tree_depth = 0, depth = 0, qp_layer = 0, coeff = 0, parity = 1;
  for (int li = top - 1; li >= 0; li--) {
    const bool root = li == top - 1;
    if (!root && ts.nodes[li] == ts.nodes[li + 1])
      continue;
    // (the plan of schedule_kernel, restated for the host's decisions: tmc3/RAHT.cpp:1165-1217, 1264-1265)
    qp_layer = qp_layer + 1 < hp->num_qp_layers ? qp_layer + 1 : hp->num_qp_layers - 1;
    const int a = coeff;
    coeff += root ? ts.nodes[li] : ts.nodes[li] - ts.nodes[li + 1];
    const int b = coeff;
    parity ^= 1;  // (the reconstruction buffer this level writes, schedule_kernel)
    const bool pred_in_level = !root && hp->raht_prediction_enabled_flag != 0;
    const int lr = tl.bits_ref - tl.bits_cur + 3 * li;
    const bool inter_on = tl.bits_ref >= 0 && lr >= 0 && lr <= 62 && tree_depth < tl.depth_limit
      && (!haar || lr / 3 < w.nlev_ref);
    const bool rdo_on = inter_on && tl.layer_rdo;
    const bool cur_level =
      pred_in_level && rdo_on && (encoder || (depth < tl.num_modes ? tl.modes[depth] != 0 : false));
    const bool dual = encoder && cur_level;
    const bool inter_blocks = inter_on && (cur_level || !pred_in_level);
    const bool est_layer = inter_on && tl.filter_est && tree_depth >= tl.skip_layers;

    lc.li = li;
    lc.inter.lr = lr;
    lc.inter.blocks = inter_blocks;
    lc.inter.dual = dual;
    lc.inter.filtered = tree_depth >= tl.skip_layers;
    lc.inter.tap = w.tap_words + li;
    if (haar && inter_on) {
      const int lref = lr / 3;  // (the caller has checked that the two trees line up on octree levels)
      lc.inter.hkey = w.tvr.key[lref];
      lc.inter.hfp = w.tvr.fp[lref];
      lc.inter.hlf = w.ref_lf[lref];
      lc.inter.hsoff = w.tvr.soff[lref];
    } else {
      lc.inter.hkey = nullptr;
    }
    // ---- the level's filter tap (:1283-1305) ---------------------------------------------
    if (est_layer && encoder) {
      auto t = prof("inter_tap", li);
      TapCtx tc{};
      tc.lc = lc;
      tc.acc = w.tap_acc;
      tc.taps = w.taps;
      tc.num_taps = w.num_taps;
      tc.tap_out = w.tap_words + li;
      tc.qp_layer = qp_layer;
      const int64_t parents = ts.nodes[li + 1];
      const int grid = (int)std::min<int64_t>(std::max<int64_t>((parents + 31) / 32, 1), 1024);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(inter_tap_kernel<C>), dim3(grid), dim3(256), 0, st, tc);
      hipLaunchKernelGGL(inter_tap_finish_kernel, dim3(1), dim3(64), 0, st, tc);
    } else if (est_layer && tree_depth - tl.skip_layers < tl.num_taps) {
      hipLaunchKernelGGL(
        inter_tap_decode_kernel, dim3(1), dim3(64), 0, st, (const gpcc_raht_params*)w.params, qp_layer,
        tl.taps[tree_depth - tl.skip_layers], w.tap_words + li);
    } else {
      const int tap = (inter_on && !tl.filter_est) ? kFixedTaps[tree_depth < 7 ? tree_depth : 6] : 128;
      hipLaunchKernelGGL(inter_set_word_kernel, dim3(1), dim3(64), 0, st, w.tap_words + li, tap);
    }

    if (w.sub) {
      // ---- the dependency kernels (raht_subnode.hpp) ------------------------------------------
      const int64_t parents = ts.nodes[li + 1];
      lc.mtag = (uint32_t)(li + 1);
      if (use_links) {
        links.produce(st, li + 1, prof);
        lc.link_rec = w.lv.rec[(li + 1) & 1];
        lc.link_lrec = w.lv.lrec[(li + 1) & 1];
      }
      {
        auto t = prof("level_prepass", li);
        hipLaunchKernelGGL(
          HIP_KERNEL_NAME(raht_level_prepass_kernel<C>), dim3((int)std::min<int64_t>((parents + 1023) / 1024, 1024)), dim3(256),
          0, st, lc);
      }
      const int sgrid = (int)std::min<int64_t>(kInterSubGrid, std::max<int64_t>(8, (parents / 64 + 7) / 8 * 8));
      // (claims of several consecutive rounds, raht_subnode.hpp: an opt-in experiment -- measured slower --
      // that the emulator tier keeps pinned by setting GPCC_SUB_CLAIM itself)
      lc.claim_rounds = sub_claim_rounds(encoder, haar, parents);
#ifdef GPCC_EMU  // (the workgroups of a dependency kernel wait for one another: eight run together)
#define GPCC_EMU_CONCURRENT(n) emu::set_concurrent_blocks(n)
#else
#define GPCC_EMU_CONCURRENT(n) ((void)0)
#endif
      if (!encoder) {
        auto t = prof("inter_sub_synth", li);
        GPCC_EMU_CONCURRENT(8);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kSynth, ArithI64, true>), dim3(sgrid), dim3(256), 0, st, lc);
        GPCC_EMU_CONCURRENT(1);
      } else {
        rt.a = a;
        rt.b = b;
        rt.rows = ts.nodes[li];
        // the zero-run state the kernels read is the entry of the OTHER level parity (the prepass copies it
        // over, raht_levels.hpp): both entries carry the state in front of the level
        rt.slice_l = w.slice_l + (li & 1);
        rt.islice_l = w.islice_l + (li & 1);
        LevelCtx lb = lc;
        if (dual) {
          // the intra candidate: the same kernel without the frame, on a workspace of its own -- the children's
          // reconstruction (what the prepass copied included), coefficients, mailbox, zero-run words, tickets
          lb.rec[parity] = w.irec;
          lb.rec_us[parity] = w.irec_us;
          lb.coeffs = w.icoeffs;
          lb.mbox = w.mbox2;
          lb.rdoq_state = w.rdoq_state2;
          lb.slice_l = w.islice_l;
          lb.ticket = w.ticket2;
          lb.inter.blocks = 0;
          SubCopyCtx sc{tv, w.rec[parity], w.rec_us[parity], w.irec, w.irec_us, (int64_t)ts.nodes[li] * C, nullptr};
          hipLaunchKernelGGL(rate_level_begin_kernel, dim3(1), dim3(64), 0, st, rt);
          hipLaunchKernelGGL(inter_sub_copy_kernel, dim3(sub_copy_grid(sc.count)), dim3(256), 0, st, sc);
        }
        bool forked = false;
#ifndef GPCC_EMU
        if (dual && streams.second) {
          // the intra candidate on the second stream, behind everything enqueued so far
          hipError_t ef = hipEventRecord(streams.fork, st);
          if (ef == hipSuccess)
            ef = hipStreamWaitEvent(streams.second, streams.fork, 0);
          if (ef != hipSuccess)
            return ef;
          if (haar)
            hipLaunchKernelGGL(
              HIP_KERNEL_NAME(raht_level_sub_kernel<C, kFused, ArithI64, false>), dim3(sgrid), dim3(256), 0, streams.second, lb);
          else if (w.f64)
            hipLaunchKernelGGL(
              HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithF64, false>), dim3(sgrid), dim3(256), 0, streams.second, lb);
          else
            hipLaunchKernelGGL(
              HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithI64, false>), dim3(sgrid), dim3(256), 0, streams.second, lb);
          ef = hipEventRecord(streams.join, streams.second);
          if (ef != hipSuccess)
            return ef;
          forked = true;
        }
#endif
        {
          auto t = prof("inter_sub_lossy", li);
          GPCC_EMU_CONCURRENT(8);
          if (haar)
            hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kFused, ArithI64, true>), dim3(sgrid), dim3(256), 0, st, lc);
          else
            hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithI64, true>), dim3(sgrid), dim3(256), 0, st, lc);
          GPCC_EMU_CONCURRENT(1);
        }
        if (dual) {
          if (!forked) {
            auto t = prof("inter_sub_lossy_intra", li);
            GPCC_EMU_CONCURRENT(8);
            if (haar)
              hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kFused, ArithI64, false>), dim3(sgrid), dim3(256), 0, st, lb);
            else if (w.f64)
              hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithF64, false>), dim3(sgrid), dim3(256), 0, st, lb);
            else
              hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithI64, false>), dim3(sgrid), dim3(256), 0, st, lb);
            GPCC_EMU_CONCURRENT(1);
          }
#ifndef GPCC_EMU
          if (forked) {
            const hipError_t ej = hipStreamWaitEvent(st, streams.join, 0);
            if (ej != hipSuccess)
              return ej;
          }
#endif
          {
            auto t = prof("rate_states", li);
            const int words = (b - a + 63) / 64;
            const int pgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * words + 3) / 4, 1), 4096);
            hipLaunchKernelGGL(rate_pack_kernel, dim3(pgrid), dim3(256), 0, st, rt);
            hipLaunchKernelGGL(rate_p1_kernel, dim3(2 * C), dim3(64), 0, st, rt);
            const int chunks = (b - a + kAcRateChunk - 1) / kAcRateChunk;
            const int bgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * chunks + 255) / 256, 1), 4096);
            hipLaunchKernelGGL(rate_p0_bits_kernel, dim3(bgrid), dim3(256), 0, st, rt);
          }
          {
            auto t = prof("rate_sum", li);
            hipLaunchKernelGGL(rate_sum_kernel, dim3(2), dim3(kAcSumThreads), 0, st, rt);
          }
          {
            auto t = prof("rate_decide", li);
            hipLaunchKernelGGL(rate_decide_kernel, dim3(1), dim3(64), 0, st, rt);
            // the winner's coefficients (inter_commit with no prediction record) and reconstruction
            RateCtx rcm = rt;
            rcm.rows = 0;
            const int64_t work = (int64_t)(b - a) * C;
            hipLaunchKernelGGL(inter_commit_kernel, dim3((int)std::min<int64_t>(std::max<int64_t>((work + 255) / 256, 1), 4096)), dim3(256), 0, st, rcm);
            SubCopyCtx sc{tv, w.irec, w.irec_us, w.rec[parity], w.rec_us[parity], (int64_t)ts.nodes[li] * C, &w.rs->intra_wins};
            hipLaunchKernelGGL(inter_sub_copy_kernel, dim3(sub_copy_grid(sc.count)), dim3(256), 0, st, sc);
          }
        }
        // the state in front of the next level in both entries (a level no slice processes is not launched here)
        hipLaunchKernelGGL(inter_pair_sync_kernel, dim3(1), dim3(64), 0, st, w.slice_l, li & 1);
      }
      if (pred_in_level && rdo_on)
        depth++;
      tree_depth++;
      continue;
    }

    const int64_t parents = ts.nodes[li + 1];
    const bool sparse = (int64_t)ts.nodes[li] * 8 <= parents * 9 && parents >= 64 * kTileTSparse;
    const int tile_t = sparse ? kTileTSparse : kTileT;
    const int ntiles = (int)((parents + tile_t - 1) / tile_t);
    const int grid = std::min((ntiles + 7) / 8 * 8, 8192);
#define GPCC_INTER_TILE(MODE, INTER)                                                                                       \
  do {                                                                                                                     \
    if (sparse)                                                                                                            \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_tile_kernel<C, MODE, kTileTSparse, INTER>), dim3(grid), dim3(kTileThreads), 0, st, lc); \
    else                                                                                                                   \
      hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_tile_kernel<C, MODE, kTileT, INTER>), dim3(grid), dim3(kTileThreads), 0, st, lc);       \
  } while (0)
    if (!encoder) {
      auto t = prof("inter_synth", li);
      GPCC_INTER_TILE(kSynth, true);
    } else if (haar) {
      // integer Haar: no RDOQ, coefficients and reconstruction in one pass (kFused); the intra candidate of a
      // level is a second launch without the frame on a workspace of its own, as under sub-node prediction
      rt.a = a;
      rt.b = b;
      rt.rows = 0;
      rt.slice_l = w.slice_l;
      rt.islice_l = w.islice_l;
      {
        auto t = prof("inter_fused", li);
        GPCC_INTER_TILE(kFused, true);
      }
      if (dual) {
        const LevelCtx la = lc;
        lc.rec[parity] = w.irec;
        lc.rec_us[parity] = w.irec_us;
        lc.coeffs = w.icoeffs;
        lc.inter.blocks = 0;
        hipLaunchKernelGGL(rate_level_begin_kernel, dim3(1), dim3(64), 0, st, rt);
        {
          auto t = prof("inter_fused_intra", li);
          GPCC_INTER_TILE(kFused, false);
        }
        lc = la;
        {
          auto t = prof("rate_states", li);
          const int words = (b - a + 63) / 64;
          const int pgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * words + 3) / 4, 1), 4096);
          hipLaunchKernelGGL(rate_pack_kernel, dim3(pgrid), dim3(256), 0, st, rt);
          hipLaunchKernelGGL(rate_p1_kernel, dim3(2 * C), dim3(64), 0, st, rt);
          const int chunks = (b - a + kAcRateChunk - 1) / kAcRateChunk;
          const int bgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * chunks + 255) / 256, 1), 4096);
          hipLaunchKernelGGL(rate_p0_bits_kernel, dim3(bgrid), dim3(256), 0, st, rt);
        }
        {
          auto t = prof("rate_sum", li);
          hipLaunchKernelGGL(rate_sum_kernel, dim3(2), dim3(kAcSumThreads), 0, st, rt);
        }
        {
          auto t = prof("rate_decide", li);
          hipLaunchKernelGGL(rate_decide_kernel, dim3(1), dim3(64), 0, st, rt);
          const int64_t work = (int64_t)(b - a) * C;
          hipLaunchKernelGGL(inter_commit_kernel, dim3((int)std::min<int64_t>(std::max<int64_t>((work + 255) / 256, 1), 4096)), dim3(256), 0, st, rt);
          SubCopyCtx sc{tv, w.irec, w.irec_us, w.rec[parity], w.rec_us[parity], (int64_t)ts.nodes[li] * C, &w.rs->intra_wins};
          hipLaunchKernelGGL(inter_sub_copy_kernel, dim3(sub_copy_grid(sc.count)), dim3(256), 0, st, sc);
        }
      }
    } else {
      rt.a = a;
      rt.b = b;
      rt.rows = ts.nodes[li];
      if (dual)
        hipLaunchKernelGGL(rate_level_begin_kernel, dim3(1), dim3(64), 0, st, rt);
      {
        auto t = prof("inter_analyze", li);
        GPCC_INTER_TILE(kAnalyze, true);
      }
      rc.li = li;
      {
        auto t = prof("rdoq_resolve", li);
        rc.desc = w.desc;
        rc.coeffs = d_coeffs;
        rc.slice_l = w.slice_l;
        rc.state = w.rtile_state;
        hipLaunchKernelGGL(rdoq_resolve_kernel, dim3((w.num_rtiles + 3) / 4), dim3(256), 0, st, rc);
        if (dual) {
          rc.desc = w.idesc;
          rc.coeffs = w.icoeffs;
          rc.slice_l = w.islice_l;
          rc.state = w.irtile_state;
          hipLaunchKernelGGL(rdoq_resolve_kernel, dim3((w.num_rtiles + 3) / 4), dim3(256), 0, st, rc);
        }
      }
      if (dual) {
        {
          auto t = prof("rate_states", li);
          const int words = (b - a + 63) / 64;
          const int pgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * words + 3) / 4, 1), 4096);
          hipLaunchKernelGGL(rate_pack_kernel, dim3(pgrid), dim3(256), 0, st, rt);
          hipLaunchKernelGGL(rate_p1_kernel, dim3(2 * C), dim3(64), 0, st, rt);
          const int chunks = (b - a + kAcRateChunk - 1) / kAcRateChunk;
          const int bgrid = (int)std::min<int64_t>(std::max<int64_t>(((int64_t)2 * C * chunks + 255) / 256, 1), 4096);
          hipLaunchKernelGGL(rate_p0_bits_kernel, dim3(bgrid), dim3(256), 0, st, rt);
        }
        {
          auto t = prof("rate_sum", li);
          hipLaunchKernelGGL(rate_sum_kernel, dim3(2), dim3(kAcSumThreads), 0, st, rt);
        }
        {
          auto t = prof("rate_decide", li);
          hipLaunchKernelGGL(rate_decide_kernel, dim3(1), dim3(64), 0, st, rt);
          const int64_t work = (int64_t)(b - a) * C + (int64_t)rt.rows * C;
          const int cgrid = (int)std::min<int64_t>(std::max<int64_t>((work + 255) / 256, 1), 4096);
          hipLaunchKernelGGL(inter_commit_kernel, dim3(cgrid), dim3(256), 0, st, rt);
        }
      }
      {
        auto t = prof("inter_synth_rec", li);
        GPCC_INTER_TILE(kSynthRec, false);
      }
    }
#undef GPCC_INTER_TILE
    if (pred_in_level && rdo_on)
      depth++;
    tree_depth++;
  }

  FinishCtx fc{};
  fc.tv = tv;
  fc.params = w.params;
  fc.sched = w.sched;
  fc.attr_prefix = w.attr_prefix;
  fc.haar_lf = (haar && encoder) ? w.haar_lf_tab : nullptr;
  fc.dup_hf = haar ? w.dup_hf : nullptr;
  fc.asc_qp = w.has_qp ? w.asc_qp_tab : nullptr;
  fc.qp_off = d_qp_off;
  for (int i = 0; i < 2; i++) {
    fc.rec[i] = w.rec[i];
    fc.dqp[i] = w.dqp[i];
  }
  fc.attrs = d_attrs;
  fc.coeffs = d_coeffs;
  fc.encoder = encoder;
  fc.lut = d_lut;
  {
    auto t = prof("finish", -1);
    const int fgrid = std::min(std::max((tv.cap[0] + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(finish_kernel<C>), dim3(fgrid), dim3(256), 0, st, fc);
  }
  return hipGetLastError();
}

}  // namespace gpcc
