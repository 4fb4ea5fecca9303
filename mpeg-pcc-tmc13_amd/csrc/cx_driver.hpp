// cx_driver.hpp -- workspace layout and launch sequence of the compact level pass
// (cx_tree.hpp, cx_level.hpp):
//
//   cx_count -> cx_scan -> cx_scan_fin -> cx_emit      level arrays + block lists
//   schedule                                            per-slice level plan
//   cx_level<C, ENC> x (levels with blocks)             ONE launch per level
//   finish                                              duplicates, write-back
//
// Used for batches without sub-node prediction, with the RAHT extension, without
// integer Haar and without region QP offsets; everything else keeps the tile
// kernels (raht_tile.hpp).  The sequence is a template over how the host learns
// the tree's shape (an event on pinned memory in the library) so that the same
// code runs under the CPU wavefront emulator of the test tier (tests/emu).
#pragma once
#include <cstdlib>

#include <algorithm>

#include "cx_level.hpp"
#include "raht_edges.hpp"
#include "raht_links.hpp"

// workgroups (four wavefront-tiles each) of the count / emit launches at most
#ifndef GPCC_CX_TGRID_CAP
#define GPCC_CX_TGRID_CAP 2048
#endif

namespace gpcc {

struct CxWork {
  int n = 0, s = 0, c = 0, nlev = 0;
  bool encoder = false;
  bool f64 = false;  // the level kernels in ArithF64 (raht_arith.hpp; decided by the caller)
  TreeView tv{};
  CxLists cl{};
  int32_t* pt_off = nullptr;
  SliceSched* sched = nullptr;
  gpcc_raht_params* params = nullptr;
  int32_t* attr_prefix = nullptr;
  int64_t* val = nullptr;
  int64_t* rec = nullptr;
  int32_t* nn = nullptr;
  unsigned long long* tstate = nullptr;
  int32_t* slice_l = nullptr;
  int max_tiles = 0;
  LinkView lv{};  // neighbour links (raht_links.hpp), opt-in
  bool links = false;  // decided by the caller before the workspace is carved (links_enabled())
};

inline bool
cx_supported(const gpcc_raht_params* p, bool has_qp, int64_t n)
{
  return p->raht_extension != 0 && !p->integer_haar_enable_flag && !has_qp
    && !(p->raht_prediction_enabled_flag && p->raht_subnode_prediction_enabled_flag)
    && n <= kCxMaxPoints;
}

// `take(bytes)` hands out 256-byte aligned storage (or only counts)
template<class Take>
void
cx_carve(Take&& take, CxWork& w)
{
  const int n = w.n, s = w.s, c = w.c, nlev = w.nlev;
  auto arr = [&](size_t count, size_t elem) { return take(count * elem); };
  w.pt_off = (int32_t*)arr(s + 1, 4);
  for (int li = 0; li < nlev; li++) {
    int64_t cap = n;
    const int up = nlev - 1 - li;
    if (up < 11) {
      const int64_t full = (int64_t)s << (3 * up);
      cap = cap < full ? cap : full;
    }
    w.tv.cap[li] = (int32_t)cap;
    w.tv.key[li] = (int64_t*)arr(cap + 1, 8);
    w.tv.fp[li] = (int32_t*)arr(cap + 2, 4);
    // (the block lists carry the first children; the neighbour links descend through fc)
    w.tv.fc[li] = w.links ? (int32_t*)arr(cap + 2, 4) : nullptr;
    w.tv.soff[li] = (int32_t*)arr(s + 1, 4);
    w.cl.hold[li] = (uint32_t*)arr(cap + 1, 4);
  }
  w.tv.nlev = nlev;
  w.tv.num_slices = s;
  w.tv.n_total = n;
  w.tv.num_tiles = (n + kTilePoints - 1) / kTilePoints;
  w.tv.pt_off = w.pt_off;
  const int ncol = 3 * nlev + c;
  w.cl.h = (uint8_t*)arr((size_t)n + 1, 1);
  w.cl.bp = (int32_t*)arr((size_t)n + 1, 4);
  w.cl.bq = (int32_t*)arr((size_t)n + nlev + 1, 4);
  w.cl.rb = (int32_t*)arr(2 * (size_t)n + 1, 4);
  w.cl.bc = (int32_t*)arr((size_t)n + 1, 4);
  w.cl.tab = (CxLevelTab*)arr(1, sizeof(CxLevelTab));
  w.cl.tile_tab = (uint32_t*)arr((size_t)w.tv.num_tiles * ncol, 4);
  w.cl.col_total = (uint32_t*)arr(ncol, 4);
  w.sched = (SliceSched*)arr(s, sizeof(SliceSched));
  w.params = (gpcc_raht_params*)arr(1, sizeof(gpcc_raht_params));
  w.attr_prefix = w.encoder ? (int32_t*)arr(((size_t)n + 1) * c, 4) : nullptr;
  w.val = (int64_t*)arr(2 * (size_t)n * c, 8);
  w.rec = (int64_t*)arr(2 * (size_t)n * c, 8);
  w.nn = (int32_t*)arr(2 * (size_t)n, 4);
  w.max_tiles = n / kCxG + 2;
  w.tstate = (unsigned long long*)arr((size_t)w.max_tiles + 1, 8);
  w.slice_l = (int32_t*)arr(2 * (size_t)s, 4);
  if (w.links)
    link_carve(take, w.lv, w.tv, n, s, nlev);
}

// Everything after the uploads of params / pt_off.  `prof(name, level)` returns a scoped
// timer object; `mark()` records an event on the stream and `wait()` blocks the host on it:
// the level table is copied to the host right behind the scan, the wait comes after the
// emit and schedule kernels have been enqueued, so the host sizes and enqueues the level
// kernels while the device is still emitting the tree.
template<int C, class Prof, class Mark, class Wait>
hipError_t
cx_run(
  hipStream_t st, CxWork& w, const SharedLut* d_lut, int num_qp_layers, int32_t* d_attrs,
  int32_t* d_coeffs, TreeStats* stats, CxLevelTab* tab, Prof&& prof, Mark&& mark, Wait&& wait)
{
  const TreeView tv = w.tv;
  const CxLists cl = w.cl;
  const int ncol = 3 * w.nlev + C;
  const int32_t* sum_attrs = w.encoder ? d_attrs : nullptr;
  const int tgrid = std::min(std::max((tv.num_tiles + 3) / 4, 1), GPCC_CX_TGRID_CAP);
  {
    auto t = prof("cx_count", -1);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_count_kernel<C>), dim3(tgrid), dim3(256), 0, st, tv, sum_attrs, cl);
  }
  {
    auto t = prof("cx_scan", -1);
    hipLaunchKernelGGL(cx_scan_kernel, dim3(ncol), dim3(256), 0, st, tv, cl, ncol);
    // (the level table also goes to the host's pinned copy `tab`, written by the kernel itself: no copy engine
    // between the scan and the event the host waits for)
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(cx_scan_fin_kernel<C>), dim3(1), dim3(64), 0, st, tv, cl, w.attr_prefix,
      sum_attrs != nullptr, tab);
  }
  hipError_t e = mark();
  if (e != hipSuccess)
    return e;
  {
    auto t = prof("cx_emit", -1);
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(cx_emit_kernel<C>), dim3(tgrid), dim3(256), 0, st, tv, sum_attrs, cl, w.attr_prefix);
  }
  {
    auto t = prof("schedule", -1);
    hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(256), 0, st, tv, w.sched, num_qp_layers, 0, stats);
  }
  e = hipMemsetAsync(w.tstate, 0, ((size_t)w.max_tiles + 1) * sizeof(unsigned long long), st);
  if (e != hipSuccess)
    return e;
  e = hipMemsetAsync(w.slice_l, 0xff, 2 * (size_t)w.s * sizeof(int32_t), st);
  if (e != hipSuccess)
    return e;
  e = wait();
  if (e != hipSuccess)
    return e;

  CxCtx cx{};
  cx.tv = tv;
  cx.cl = cl;
  cx.params = w.params;
  cx.sched = w.sched;
  cx.attr_prefix = w.attr_prefix;
  cx.val = w.val;
  cx.rec = w.rec;
  cx.nn = w.nn;
  cx.coeffs = d_coeffs;
  cx.lut = d_lut;
  cx.tstate = w.tstate;
  cx.slice_l = w.slice_l;
  // (levels above the tallest slice's root hold no block: nr = 0)
  int li_start = w.nlev - 2;
  while (li_start >= 0 && tab->nr[li_start] <= 0)
    li_start--;
  // the top levels, as long as a level is a few tiles: one launch (cx_top_kernel)
  {
    int li_lo = li_start + 1;
    while (li_lo - 1 >= 0 && tab->nr[li_lo - 1] <= kCxTopTiles * kCxG)
      li_lo--;
    static const bool top_on = [] {
      const char* e = getenv("GPCC_CX_TOP");
      return !(e && e[0] == '0');
    }();
    if (top_on && li_start - li_lo + 1 >= 2) {
      auto t = prof(w.encoder ? "cx_top_enc" : "cx_top_dec", -1);
      if (w.encoder && w.f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_top_kernel<C, true, ArithF64>), dim3(1), dim3(256), 0, st, cx, li_start, li_lo);
      else if (w.encoder)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_top_kernel<C, true, ArithI64>), dim3(1), dim3(256), 0, st, cx, li_start, li_lo);
      else if (w.f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_top_kernel<C, false, ArithF64>), dim3(1), dim3(256), 0, st, cx, li_start, li_lo);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_top_kernel<C, false, ArithI64>), dim3(1), dim3(256), 0, st, cx, li_start, li_lo);
      li_start = li_lo - 1;
    }
  }
  LinkSchedule links;
  const bool use_links = w.links && li_start >= 0;
  if (use_links) {
    links.tv = tv;
    links.lv = w.lv;
    links.begin(st, tab->nodes, li_start + 1, prof);
  }
  for (int li = li_start; li >= 0; li--) {
    const int nr = tab->nr[li];
    if (nr <= 0)
      continue;
    cx.li = li;
    if (use_links) {
      links.produce(st, li + 1, prof);
      cx.link_rec = w.lv.rec[(li + 1) & 1];
      cx.link_lrec = w.lv.lrec[(li + 1) & 1];
    }
    const int ntiles = (nr + kCxG - 1) / kCxG;
    auto t = prof(w.encoder ? "cx_level_enc" : "cx_level_dec", li);
    if (w.encoder && w.f64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_level_kernel<C, true, ArithF64>), dim3((ntiles + 3) / 4), dim3(256), 0, st, cx);
    else if (w.encoder)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_level_kernel<C, true, ArithI64>), dim3((ntiles + 3) / 4), dim3(256), 0, st, cx);
    else if (w.f64)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_level_kernel<C, false, ArithF64>), dim3((ntiles + 3) / 4), dim3(256), 0, st, cx);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_level_kernel<C, false, ArithI64>), dim3((ntiles + 3) / 4), dim3(256), 0, st, cx);
  }

  FinishCtx fc{};
  fc.tv = tv;
  fc.params = w.params;
  fc.sched = w.sched;
  fc.attr_prefix = w.attr_prefix;
  fc.slot_rec = w.rec;
  fc.slot_f64 = w.f64 ? 1 : 0;
  fc.hold0 = cl.hold[0];
  fc.attrs = d_attrs;
  fc.coeffs = d_coeffs;
  fc.encoder = w.encoder;
  fc.lut = d_lut;
  {
    auto t = prof("finish", -1);
    const int fgrid = std::min(std::max((tv.cap[0] + 255) / 256, 1), 2048);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(finish_kernel<C>), dim3(fgrid), dim3(256), 0, st, fc);
  }
  return hipGetLastError();
}

}  // namespace gpcc
