// tests/emu/rc_emu_harness.cpp -- TEST INFRASTRUCTURE: gpcc_recolour's kernels
// (mpeg-pcc-tmc13_amd/csrc/recolour_kdtree.hpp, recolour_kernels.hpp) under the CPU wavefront
// emulator.  What recolour_impl (gpcc_attr_mi355.hip) does around them -- buffers, the two tree
// builds, the launch sequence -- is repeated here with the same kernels and the same level loop
// (kd_build_levels); also exports the tree so that a test can compare it node by node.
#include <vector>

#include "hip/hip_runtime.h"

#include "recolour_kernels.hpp"

using namespace gpcc;

namespace {
template<class T>
T*
carve(std::vector<void*>* blocks, size_t count)
{
  const size_t bytes = (sizeof(T) * std::max<size_t>(count, 1) + 255) & ~size_t(255);
  void* p = malloc(bytes + 256);
  memset(p, 0xCD, bytes + 256);  // the pool of the library is not cleared either
  blocks->push_back(p);
  return (T*)p;
}

int
build_tree(std::vector<void*>* blocks, const int32_t* xyz, int n, const int32_t* box, KdTree* out, int* depth)
{
  const size_t N = (size_t)n, M = 2 * N + 2;
  KdBuild b{};
  b.t.xyz = xyz;
  b.t.n = n;
  b.t.vind = carve<int32_t>(blocks, N);
  b.node_cap = (int32_t)kd_node_capacity(N);
  b.t.nodes = carve<KdNode>(blocks, kd_node_capacity(N));
  b.pnode = carve<int32_t>(blocks, N);
  b.rng = carve<int32_t>(blocks, 2 * M);
  b.parent = carve<int32_t>(blocks, M);
  b.box = carve<double>(blocks, 6 * M);
  b.mm = carve<int32_t>(blocks, 6 * M);
  b.cut = carve<double>(blocks, M);
  b.lim = carve<int32_t>(blocks, 2 * M);
  b.split = carve<int32_t>(blocks, M);
  b.flag = carve<int32_t>(blocks, N + 1);
  b.tmp_l = carve<int32_t>(blocks, N);
  b.tmp_r = carve<int32_t>(blocks, N);
  b.sums = carve<long long>(blocks, (N + 1) / kKdScanBlock + 2);
  b.counters = carve<int32_t>(blocks, 4);
  b.sub_list = carve<int32_t>(blocks, 2 * (N / 11 + 2));
  int nodes = 0;
  if (kd_build_levels(b, box, nullptr, depth, &nodes) != hipSuccess)
    return -1;
  *out = b.t;
  for (int k = 0; k < 3; k++) {
    out->root_lo[k] = (double)box[k];
    out->root_hi[k] = (double)box[3 + k];
  }
  return nodes;
}
}  // namespace

// -> node ids handed out (or < 0; ids have gaps: size the arrays by 2n + n/4 + 64); vind[n], and per node
// {a, b, feat} + divlow / divhigh
extern "C" int
rc_emu_kdtree(const int32_t* xyz, int32_t n, int32_t* vind, int32_t* node_abf, double* node_div)
{
  std::vector<void*> blocks;
  int32_t* box = carve<int32_t>(&blocks, 6);
  for (int k = 0; k < 3; k++) {
    box[k] = 0x7fffffff;
    box[3 + k] = -0x7fffffff;
  }
  hipLaunchKernelGGL(rc_bbox_kernel, dim3(2), dim3(256), 0, nullptr, xyz, (int)n, box);
  KdTree t{};
  int depth = 0;
  const int nodes = build_tree(&blocks, xyz, n, box, &t, &depth);
  if (nodes > 0) {
    memcpy(vind, t.vind, sizeof(int32_t) * (size_t)n);
    for (int k = 0; k < nodes; k++) {
      node_abf[3 * k] = t.nodes[k].a;
      node_abf[3 * k + 1] = t.nodes[k].b;
      node_abf[3 * k + 2] = t.nodes[k].feat;
      node_div[2 * k] = t.nodes[k].divlow;
      node_div[2 * k + 1] = t.nodes[k].divhigh;
    }
  }
  for (void* p : blocks)
    free(p);
  return depth > kKdMaxDepth ? -2 : nodes;
}

extern "C" int
rc_emu_recolour(
  const gpcc_recolour_params* p, const int32_t* src_xyz, const int32_t* src_attrs, int32_t ns,
  const int32_t* tgt_xyz, int32_t nt, int32_t c, float scale, const int32_t offset[3], int32_t* tgt_attrs)
{
  const int kf = p->num_neighbours_fwd, kb = p->num_neighbours_bwd;
  if (ns < kf || nt < kb || (c != 1 && c != 3))
    return -2;
  std::vector<void*> blocks;
  int32_t* box = carve<int32_t>(&blocks, 12);
  for (int k = 0; k < 3; k++) {
    box[k] = box[6 + k] = 0x7fffffff;
    box[3 + k] = box[9 + k] = -0x7fffffff;
  }
  hipLaunchKernelGGL(rc_bbox_kernel, dim3(2), dim3(256), 0, nullptr, src_xyz, (int)ns, box);
  hipLaunchKernelGGL(rc_bbox_kernel, dim3(2), dim3(256), 0, nullptr, tgt_xyz, (int)nt, box + 6);
  RcCtx cx{};
  cx.p = *p;
  cx.c = c;
  cx.s2t = (double)scale;
  cx.t2s = 1.0 / (double)scale;
  for (int k = 0; k < 3; k++)
    cx.off[k] = offset[k];
  cx.src_attrs = src_attrs;
  int depth = 0;
  int rc = 0;
  if (build_tree(&blocks, src_xyz, ns, box, &cx.src, &depth) < 0 || depth > kKdMaxDepth
      || build_tree(&blocks, tgt_xyz, nt, box + 6, &cx.tgt, &depth) < 0 || depth > kKdMaxDepth)
    rc = -3;
  if (rc == 0) {
    const size_t total_cap = (size_t)ns * kb;
    cx.ref1 = carve<int32_t>(&blocks, (size_t)c * nt);
    cx.bt = carve<int32_t>(&blocks, total_cap);
    cx.bd = carve<double>(&blocks, total_cap);
    cx.lstart = carve<int32_t>(&blocks, (size_t)nt + 1);
    cx.lcur = carve<int32_t>(&blocks, (size_t)nt);
    cx.ldist = carve<double>(&blocks, total_cap);
    cx.lsrc = carve<int32_t>(&blocks, total_cap);
    cx.out = tgt_attrs;
    if (p->max_geometry_dist2_fwd < 512) {
      cx.nearest = carve<int32_t>(&blocks, (size_t)nt + 1);
      cx.fwd_first = cx.nearest + nt;
      *cx.fwd_first = 0x7f7f7f7f;
    }
    long long* sums = carve<long long>(&blocks, ((size_t)nt + 1) / kKdScanBlock + 2);
    const bool alimit = p->max_attribute_dist2_fwd < 512;
    const int fgrid = (nt + 255) / 256, bgrid = (ns + 255) / 256;
    // (the emulator runs one instantiation per list capacity: 8 covers every k)
    if (c == 3) {
      if (alimit)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_forward_kernel<3, 8, true>), dim3(fgrid), dim3(256), 0, nullptr, cx);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_forward_kernel<3, 8, false>), dim3(fgrid), dim3(256), 0, nullptr, cx);
    } else {
      if (alimit)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_forward_kernel<1, 8, true>), dim3(fgrid), dim3(256), 0, nullptr, cx);
      else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_forward_kernel<1, 8, false>), dim3(fgrid), dim3(256), 0, nullptr, cx);
    }
    if (cx.nearest)
      hipLaunchKernelGGL(rc_forward_limit_kernel, dim3(fgrid), dim3(256), 0, nullptr, cx);
    memset(cx.lstart, 0, sizeof(int32_t) * ((size_t)nt + 1));
    memset(cx.lcur, 0, sizeof(int32_t) * (size_t)nt);
    if (kb <= 1)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_backward_kernel<1>), dim3(bgrid), dim3(256), 0, nullptr, cx);
    else if (kb <= 4)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_backward_kernel<4>), dim3(bgrid), dim3(256), 0, nullptr, cx);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_backward_kernel<8>), dim3(bgrid), dim3(256), 0, nullptr, cx);
    kd_scan(nullptr, cx.lstart, (size_t)nt + 1, sums);
    hipLaunchKernelGGL(rc_list_fill_kernel, dim3(bgrid), dim3(256), 0, nullptr, cx);
    if (c == 3)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_blend_kernel<3>), dim3(fgrid), dim3(256), 0, nullptr, cx);
    else
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rc_blend_kernel<1>), dim3(fgrid), dim3(256), 0, nullptr, cx);
  }
  for (void* q : blocks)
    free(q);
  return rc;
}

// kd_scan alone (the inclusive prefix sum the tree build runs ~60 times per tree): in place on a[n]
extern "C" int
rc_emu_scan(int32_t* a, int64_t n)
{
  std::vector<long long> sums((size_t)n / kKdScanBlock + 2);
  const hipError_t e = kd_scan(nullptr, a, (size_t)n, sums.data());
  return e == hipSuccess ? 0 : -1;
}
