// tests/emu/sweep_emu_harness.cpp -- TEST INFRASTRUCTURE: the block loop with sub-node prediction
// (mpeg-pcc-tmc13_amd/csrc/raht_sweep.hpp for the coarse levels, raht_subnode.hpp for the others) under
// the CPU wavefront emulator, in the launch order of the gfx950 library's launch_transform
// (gpcc_attr_mi355.hip): tree, schedule, sweep, per-level prepass + dependency kernel, finish.
// Fixed-point transform (no integer Haar), no region QPs.
#include <algorithm>
#include <vector>

#include "hip/hip_runtime.h"

#include "raht_tree.hpp"
#include "raht_edges.hpp"
#include "raht_sweep.hpp"

using namespace gpcc;

namespace {

struct Carver {
  std::vector<void*> blocks;
  template<class T>
  T* take(size_t count)
  {
    const size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
    void* p = malloc(bytes + 256);
    memset(p, 0xCD, bytes + 256);  // the arena of the library is not cleared either
    blocks.push_back(p);
    return (T*)p;
  }
  ~Carver()
  {
    for (void* p : blocks)
      free(p);
  }
};

template<int C>
int
run(
  const gpcc_raht_params* params, bool encoder, bool f64, bool use_rec, int sweep_parents, int S, const int64_t* offsets,
  const int64_t* morton, int32_t* attrs, int32_t* coeffs, int bits, int32_t* levels_swept)
{
  Carver ar;
  const int n = (int)offsets[S];
  const int nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  TreeView tv{};
  int32_t error = 0;
  int32_t* pt_off = ar.take<int32_t>(S + 1);
  for (int i = 0; i <= S; i++)
    pt_off[i] = (int32_t)offsets[i];
  std::vector<int64_t> cap(nlev);
  for (int li = 0; li < nlev; li++) {
    int64_t c = n;
    const int up = nlev - 1 - li;
    if (up < 11)
      c = std::min<int64_t>(c, (int64_t)S << (3 * up));
    cap[li] = c;
    tv.cap[li] = (int32_t)c;
    tv.key[li] = ar.take<int64_t>(c + 1);
    tv.fp[li] = ar.take<int32_t>(c + 2);
    tv.fc[li] = ar.take<int32_t>(c + 2);
    tv.soff[li] = ar.take<int32_t>(S + 1);
  }
  tv.nlev = nlev;
  tv.num_slices = S;
  tv.n_total = n;
  tv.num_tiles = (n + kTilePoints - 1) / kTilePoints;
  tv.pt_off = pt_off;
  tv.pos = morton;
  tv.error = &error;
  uint32_t* tile_cnt = ar.take<uint32_t>((size_t)tv.num_tiles * nlev);
  int32_t* tile_attr = ar.take<int32_t>((size_t)tv.num_tiles * C);
  SliceSched* sched = ar.take<SliceSched>(S);
  int32_t* worklist = ar.take<int32_t>((size_t)n + 1);
  int32_t* work_count = ar.take<int32_t>(kMaxLevels * 9);
  unsigned long long* scan_state = ar.take<unsigned long long>(1024);
  uint8_t* pocc = ar.take<uint8_t>((size_t)n + 1);
  uint32_t* mbox = ar.take<uint32_t>((size_t)n * C * 4);
  unsigned long long* rdoq_state = ar.take<unsigned long long>((size_t)n + 1);
  gpcc_raht_params* dparams = ar.take<gpcc_raht_params>(1);
  memcpy(dparams, params, sizeof(*params));
  int64_t *rec[2], *rec_us[2];
  int32_t* nneigh[2];
  // (both buffers of a pair from one block: par2 addresses the second relative to the first)
  for (int i = 0; i < 1; i++) {
    int64_t* r = ar.take<int64_t>((size_t)n * C * 2);
    rec[0] = r;
    rec[1] = r + (size_t)n * C;
    int64_t* u = ar.take<int64_t>((size_t)n * C * 2);
    rec_us[0] = u;
    rec_us[1] = u + (size_t)n * C;
    int32_t* q = ar.take<int32_t>((size_t)n * 2);
    nneigh[0] = q;
    nneigh[1] = q + n;
  }
  int32_t* attr_prefix = encoder ? ar.take<int32_t>(((size_t)n + 1) * C) : nullptr;
  int32_t* slice_l = ar.take<int32_t>(2 * (size_t)S);
  SharedLut* lut = ar.take<SharedLut>(1);
  hipLaunchKernelGGL(lut_init_kernel, dim3(1), dim3(256), 0, nullptr, lut);

  const int32_t* sum_attrs = encoder ? attrs : nullptr;
  const int tgrid = std::max((tv.num_tiles + 3) / 4, 1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(tree_count_kernel<C>), dim3(tgrid), dim3(256), 0, nullptr, tv, sum_attrs, tile_cnt, tile_attr);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(tree_scan_kernel<C>), dim3(1), dim3(1024), 0, nullptr, tv, tile_cnt, tile_attr, attr_prefix, sum_attrs != nullptr);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(tree_emit_kernel<C>), dim3(tgrid), dim3(256), 0, nullptr, tv, sum_attrs, tile_cnt, tile_attr, attr_prefix);
  TreeStats ts{};
  TreeStats* tsp = &ts;
  hipLaunchKernelGGL(schedule_kernel, dim3(1), dim3(256), 0, nullptr, tv, sched, (int)params->num_qp_layers, sweep_parents, tsp);
  if (error)
    return -100 - error;

  LevelCtx lc{};
  lc.tv = tv;
  lc.params = dparams;
  lc.sched = sched;
  lc.attr_prefix = attr_prefix;
  for (int i = 0; i < 2; i++) {
    lc.rec[i] = rec[i];
    lc.rec_us[i] = rec_us[i];
    lc.nneigh[i] = nneigh[i];
    lc.dqp[i] = nullptr;
  }
  lc.coeffs = coeffs;
  lc.lut = lut;
  lc.worklist = worklist;
  lc.work_count = work_count;
  lc.scan_state = scan_state;
  lc.pocc = pocc;
  lc.mbox = mbox;
  lc.ticket = work_count + kMaxLevels;
  lc.error = &error;
  lc.rdoq_state = encoder ? rdoq_state : nullptr;
  lc.slice_l = encoder ? slice_l : nullptr;
  memset(work_count, 0, kMaxLevels * 9 * sizeof(int32_t));
  memset(scan_state, 0, 1024 * sizeof(unsigned long long));
  memset(mbox, 0, (size_t)n * C * 16);
  memset(rdoq_state, 0, ((size_t)n + 1) * 8);
  memset(slice_l, 0xff, 2 * (size_t)S * 4);

  const int first_level = std::min(nlev - 1, ts.max_top);
  int sweep_lo = first_level;
  if (sweep_parents > 0)
    sweep_lo = std::min(first_level, std::max(0, ts.fine_levels));
  if (levels_swept)
    *levels_swept = first_level - sweep_lo;
  if (sweep_lo < first_level) {
    const SweepCtx sw{first_level - 1, sweep_lo};
    SweepRec rec{};
    const int64_t parents = sweep_rec_layout(&rec, ts.nodes, sw.li_hi, sw.li_lo);
    sweep_rec_carve(&rec, ar.take<char>(sweep_rec_bytes(parents, C)), parents, C);
    sweep_launch<C>(nullptr, lc, sw, rec, ts.nodes, S, encoder, f64);
  }
  for (int li = sweep_lo - 1; li >= 0; li--) {
    lc.li = li;
    lc.mtag = (uint32_t)(li + 1);
    const int64_t parents = ts.nodes[li + 1];
    hipLaunchKernelGGL(
      HIP_KERNEL_NAME(raht_level_prepass_kernel<C>), dim3((int)std::min<int64_t>((parents + 1023) / 1024, 1024)), dim3(256), 0,
      nullptr, lc);
    const int sgrid = (int)std::min<int64_t>(1024, std::max<int64_t>(8, (parents / 64 + 7) / 8 * 8));
    if (use_rec) {
      const int64_t mb = level_max_blocks(ts.nodes, li, params->raht_extension != 0);
      lc.brec = level_record_launch<C>(nullptr, lc, li, ar.take<char>(sweep_rec_bytes(mb, C)), mb, encoder, f64);
    }
    emu::set_concurrent_blocks(8);  // (the workgroups of a dependency kernel wait for one another)
    if (use_rec && !encoder) {
      if (f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kSynth, ArithF64, false, true>), dim3(sgrid), dim3(256), 0, nullptr, lc);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kSynth, ArithI64, false, true>), dim3(sgrid), dim3(256), 0, nullptr, lc);
    } else if (use_rec) {
      if (f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithF64, false, true>), dim3(sgrid), dim3(256), 0, nullptr, lc);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithI64, false, true>), dim3(sgrid), dim3(256), 0, nullptr, lc);
    } else if (!encoder) {
      if (f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kSynth, ArithF64>), dim3(sgrid), dim3(256), 0, nullptr, lc);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kSynth>), dim3(sgrid), dim3(256), 0, nullptr, lc);
    } else {
      if (f64)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub, ArithF64>), dim3(sgrid), dim3(256), 0, nullptr, lc);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(raht_level_sub_kernel<C, kLossySub>), dim3(sgrid), dim3(256), 0, nullptr, lc);
    }
    emu::set_concurrent_blocks(1);
  }

  FinishCtx fc{};
  fc.tv = tv;
  fc.params = dparams;
  fc.sched = sched;
  fc.attr_prefix = attr_prefix;
  for (int i = 0; i < 2; i++) {
    fc.rec[i] = rec[i];
    fc.dqp[i] = nullptr;
  }
  fc.attrs = attrs;
  fc.coeffs = coeffs;
  fc.encoder = encoder;
  fc.lut = lut;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(finish_kernel<C>), dim3((int)std::min<int64_t>(std::max<int64_t>((cap[0] + 255) / 256, 1), 4096)), dim3(256), 0, nullptr, fc);
  return error ? -100 - error : 0;
}

}  // namespace

// attrs: in source (encoder) / out reconstruction; coeffs: planar per slice.
// flags: bit 0 = encoder, bit 1 = ArithF64, bit 2 = block records in the per-level kernels (raht_level_sub_kernel<.., REC>).  sweep_parents: a slice's levels with at most so many
// parents go to raht_sub_sweep_kernel (0: none -- the per-level kernels only).
extern "C" int
sweep_emu_transform(
  const gpcc_raht_params* params, int flags, int32_t sweep_parents, int32_t num_slices, const int64_t* offsets,
  const int64_t* morton, int32_t* attrs, int32_t* coeffs, int32_t c, int32_t morton_bits, int32_t* levels_swept)
{
  const int bits = morton_bits > 0 ? std::min(morton_bits, 63) : 63;
  const bool enc = (flags & 1) != 0, f64 = (flags & 2) != 0, rec = (flags & 4) != 0;
  if (sweep_parents > kSweepMaxParents)
    return -3;
  switch (c) {
  case 1: return run<1>(params, enc, f64, rec, sweep_parents, num_slices, offsets, morton, attrs, coeffs, bits, levels_swept);
  case 2: return run<2>(params, enc, f64, rec, sweep_parents, num_slices, offsets, morton, attrs, coeffs, bits, levels_swept);
  case 3: return run<3>(params, enc, f64, rec, sweep_parents, num_slices, offsets, morton, attrs, coeffs, bits, levels_swept);
  }
  return -2;
}
