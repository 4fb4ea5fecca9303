// tests/emu/cx_emu_harness.cpp -- TEST INFRASTRUCTURE: runs the compact level pass
// (mpeg-pcc-tmc13_amd/csrc/cx_*.hpp) under the CPU wavefront emulator, through the
// same launch sequence (cx_driver.hpp) the gfx950 library uses.
#include <algorithm>
#include <vector>

#include "hip/hip_runtime.h"

#include "cx_driver.hpp"

using namespace gpcc;

namespace {
struct NoProf {
  int operator()(const char*, int) const { return 0; }
};
}  // namespace

extern "C" int
cx_emu_supported(const gpcc_raht_params* p, int has_qp, int64_t n)
{
  return cx_supported(p, has_qp != 0, n) ? 1 : 0;
}

// attrs: in source (encoder) / out reconstruction; coeffs: planar per slice
// encoder: bit 0 = encoder, bit 1 = the level kernels in ArithF64 (raht_arith.hpp)
extern "C" int
cx_emu_transform(
  const gpcc_raht_params* params, int encoder, int32_t num_slices, const int64_t* offsets,
  const int64_t* morton, int32_t* attrs, int32_t* coeffs, int32_t c, int32_t morton_bits,
  int32_t* debug_tab)
{
  CxWork w;
  w.n = (int)offsets[num_slices];
  w.s = num_slices;
  w.c = c;
  const int bits = morton_bits > 0 ? std::min(morton_bits, 63) : 63;
  w.nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  w.encoder = (encoder & 1) != 0;
  w.f64 = (encoder & 2) != 0;
  w.links = (encoder & 4) != 0 || links_enabled();  // bit 2: with the neighbour links (raht_links.hpp)
  std::vector<void*> blocks;
  cx_carve(
    [&](size_t bytes) {
      bytes = (bytes + 255) & ~size_t(255);
      void* p = malloc(bytes + 256);
      memset(p, 0xCD, bytes + 256);  // the arena of the library is not cleared either
      blocks.push_back(p);
      return (char*)p;
    },
    w);
  int32_t error = 0;
  w.tv.pos = morton;
  w.tv.error = &error;
  std::vector<int32_t> off(num_slices + 1);
  for (int i = 0; i <= num_slices; i++)
    off[i] = (int32_t)offsets[i];
  memcpy(w.pt_off, off.data(), off.size() * 4);
  memcpy(w.params, params, sizeof(*params));
  SharedLut* lut = (SharedLut*)malloc(sizeof(SharedLut));
  hipLaunchKernelGGL(lut_init_kernel, dim3(1), dim3(256), 0, nullptr, lut);
  TreeStats stats{};
  CxLevelTab tab{};
  hipError_t e;
  auto fetch = [&]() { return hipSuccess; };
  switch (c) {
  case 1: e = cx_run<1>(nullptr, w, lut, params->num_qp_layers, attrs, coeffs, &stats, &tab, NoProf(), fetch, fetch); break;
  case 2: e = cx_run<2>(nullptr, w, lut, params->num_qp_layers, attrs, coeffs, &stats, &tab, NoProf(), fetch, fetch); break;
  default: e = cx_run<3>(nullptr, w, lut, params->num_qp_layers, attrs, coeffs, &stats, &tab, NoProf(), fetch, fetch); break;
  }
  if (debug_tab)
    for (int l = 0; l < kMaxLevels; l++) {
      debug_tab[l] = tab.nb[l];
      debug_tab[kMaxLevels + l] = tab.nr[l];
    }
  for (void* p : blocks)
    free(p);
  free(lut);
  if (e != hipSuccess)
    return -5;
  return error ? -100 - error : 0;
}

// ---- tree / list check against a direct computation ---------------------------------
extern "C" int
cx_emu_check_tree(int32_t num_slices, const int64_t* offsets, const int64_t* morton, int32_t morton_bits)
{
  CxWork w;
  w.n = (int)offsets[num_slices];
  w.s = num_slices;
  w.c = 1;
  const int bits = morton_bits > 0 ? std::min(morton_bits, 63) : 63;
  w.nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  w.encoder = false;
  std::vector<void*> blocks;
  cx_carve(
    [&](size_t bytes) {
      bytes = (bytes + 255) & ~size_t(255);
      void* p = malloc(bytes + 256);
      memset(p, 0xCD, bytes + 256);
      blocks.push_back(p);
      return (char*)p;
    },
    w);
  int32_t error = 0;
  w.tv.pos = morton;
  w.tv.error = &error;
  for (int i = 0; i <= num_slices; i++)
    w.pt_off[i] = (int32_t)offsets[i];
  const TreeView tv = w.tv;
  const CxLists cl = w.cl;
  const int ncol = 3 * w.nlev + 1;
  const int tgrid = std::max((tv.num_tiles + 3) / 4, 1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_count_kernel<1>), dim3(tgrid), dim3(256), 0, nullptr, tv, (const int32_t*)nullptr, cl);
  hipLaunchKernelGGL(cx_scan_kernel, dim3(ncol), dim3(256), 0, nullptr, tv, cl, ncol);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_scan_fin_kernel<1>), dim3(1), dim3(64), 0, nullptr, tv, cl, (int32_t*)nullptr, 0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_emit_kernel<1>), dim3(tgrid), dim3(256), 0, nullptr, tv, (const int32_t*)nullptr, cl, (int32_t*)nullptr);
  const int n = w.n, nlev = w.nlev;
  int bad = 0;
  auto fail = [&](const char* what, int l, int i, long long got, long long want) {
    if (bad++ < 12)
      fprintf(stderr, "tree check: %s level %d index %d: got %lld want %lld\n", what, l, i, got, want);
  };
  // head levels
  std::vector<int> h(n + 1);
  for (int s = 0; s < num_slices; s++)
    for (int i = (int)offsets[s]; i < (int)offsets[s + 1]; i++) {
      if (i == offsets[s]) {
        h[i] = nlev;
        continue;
      }
      const uint64_t x = (uint64_t)(morton[i] ^ morton[i - 1]);
      int hh = x ? (64 - __builtin_clzll(x) + 2) / 3 : 0;
      h[i] = hh < nlev ? hh : nlev;
    }
  h[n] = nlev;
  for (int i = 0; i <= n; i++)
    if (cl.h[i] != h[i])
      fail("h", 0, i, cl.h[i], h[i]);
  std::vector<std::vector<int>> heads(nlev);
  for (int l = 0; l < nlev; l++) {
    for (int i = 0; i < n; i++)
      if (h[i] > l)
        heads[l].push_back(i);
    const int m = (int)heads[l].size();
    if (tv.soff[l][num_slices] != m)
      fail("node count", l, 0, tv.soff[l][num_slices], m);
    for (int q = 0; q < m; q++) {
      const int a = heads[l][q], b = q + 1 < m ? heads[l][q + 1] : n;
      if (tv.fp[l][q] != a)
        fail("fp", l, q, tv.fp[l][q], a);
      if (tv.key[l][q] != (morton[a] >> (3 * l)))
        fail("key", l, q, tv.key[l][q], morton[a] >> (3 * l));
      const int slot = h[a] <= h[b] ? a : n + b;
      const int t = std::min(h[a], h[b]) - 1;
      const uint32_t want = (uint32_t)slot | ((uint32_t)t << kCxSlotBits);
      if (cl.hold[l][q] != want)
        fail("hold", l, q, cl.hold[l][q], want);
    }
    if (tv.fp[l][m] != n)
      fail("fp sentinel", l, m, tv.fp[l][m], n);
  }
  // first child of every node (the build itself keeps it for the blocks only)
  std::vector<std::vector<int>> fcs(nlev);
  for (int l = 1; l < nlev; l++) {
    const int m = (int)heads[l].size();
    size_t c = 0;
    for (int q = 0; q <= m; q++) {
      const int a = q < m ? heads[l][q] : n;
      while (c < heads[l - 1].size() && heads[l - 1][c] < a)
        c++;
      fcs[l].push_back((int)c);
    }
  }
  // block lists of every children level
  for (int l = 0; l + 1 < nlev; l++) {
    const int mp = (int)heads[l + 1].size();
    std::vector<int> ebp, ebq, erb;
    int rank = 0;
    std::vector<int> ebc;
    for (int j = 0; j < mp; j++) {
      const int k = fcs[l + 1][j + 1] - fcs[l + 1][j];
      if (k < 2)
        continue;
      ebp.push_back(j);
      ebc.push_back(fcs[l + 1][j]);
      ebq.push_back(rank);
      for (int u = 0; u < k; u++)
        erb.push_back((int)ebp.size() - 1);
      rank += k;
    }
    ebq.push_back(rank);
    const CxLevelTab* tab = cl.tab;
    if (tab->nb[l] != (int)ebp.size())
      fail("nb", l, 0, tab->nb[l], (long long)ebp.size());
    if (tab->nr[l] != rank)
      fail("nr", l, 0, tab->nr[l], rank);
    if (tab->nb[l] != (int)ebp.size() || tab->nr[l] != rank)
      continue;
    const int32_t* bp = cl.bp + tab->boff[l];
    const int32_t* bq = cl.bq + tab->boff[l] + l;
    const int32_t* rb = cl.rb + tab->roff[l];
    for (size_t b = 0; b < ebp.size(); b++) {
      if (bp[b] != ebp[b])
        fail("bp", l, (int)b, bp[b], ebp[b]);
      if (bq[b] != ebq[b])
        fail("bq", l, (int)b, bq[b], ebq[b]);
      if (cl.bc[tab->boff[l] + b] != ebc[b])
        fail("bc", l, (int)b, cl.bc[tab->boff[l] + b], ebc[b]);
    }
    if (bq[ebp.size()] != rank)
      fail("bq sentinel", l, (int)ebp.size(), bq[ebp.size()], rank);
    for (int r = 0; r < rank; r++)
      if (rb[r] != erb[r])
        fail("rb", l, r, rb[r], erb[r]);
  }
  for (void* p : blocks)
    free(p);
  return bad;
}


// ---- neighbour links (raht_links.hpp) against a direct computation --------------------------------
// use_top = 0: the seed, then every level by its own launch, each level checked; 1: the levels the
// driver would give the single-workgroup loop run there (its last two levels checked), the rest by launch
extern "C" int
cx_emu_check_links(int32_t num_slices, const int64_t* offsets, const int64_t* morton, int32_t morton_bits, int32_t use_top)
{
  CxWork w;
  w.n = (int)offsets[num_slices];
  w.s = num_slices;
  w.c = 1;
  const int bits = morton_bits > 0 ? std::min(morton_bits, 63) : 63;
  w.nlev = std::min((bits + 2) / 3 + 1, (int)kMaxLevels);
  w.encoder = false;
  w.links = true;
  std::vector<void*> blocks;
  cx_carve(
    [&](size_t bytes) {
      bytes = (bytes + 255) & ~size_t(255);
      void* p = malloc(bytes + 256);
      memset(p, 0xCD, bytes + 256);
      blocks.push_back(p);
      return (char*)p;
    },
    w);
  int32_t error = 0;
  w.tv.pos = morton;
  w.tv.error = &error;
  for (int i = 0; i <= num_slices; i++)
    w.pt_off[i] = (int32_t)offsets[i];
  const TreeView tv = w.tv;
  const CxLists cl = w.cl;
  const LinkView lv = w.lv;
  const int ncol = 3 * w.nlev + 1;
  const int tgrid = std::max((tv.num_tiles + 3) / 4, 1);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_count_kernel<1>), dim3(tgrid), dim3(256), 0, nullptr, tv, (const int32_t*)nullptr, cl);
  hipLaunchKernelGGL(cx_scan_kernel, dim3(ncol), dim3(256), 0, nullptr, tv, cl, ncol);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_scan_fin_kernel<1>), dim3(1), dim3(64), 0, nullptr, tv, cl, (int32_t*)nullptr, 0);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(cx_emit_kernel<1>), dim3(tgrid), dim3(256), 0, nullptr, tv, (const int32_t*)nullptr, cl, (int32_t*)nullptr);
  const int n = w.n, nlev = w.nlev;
  int bad = 0;
  auto fail = [&](const char* what, int l, int i, long long got, long long want) {
    if (bad++ < 12)
      fprintf(stderr, "link check: %s level %d index %d: got %lld want %lld\n", what, l, i, got, want);
  };
  // the levels, directly
  std::vector<std::vector<int64_t>> keys(nlev);
  std::vector<std::vector<int>> fps(nlev), sl_of(nlev);
  for (int l = 0; l < nlev; l++)
    for (int s = 0; s < num_slices; s++)
      for (int i = (int)offsets[s]; i < (int)offsets[s + 1]; i++) {
        const int64_t k = morton[i] >> (3 * l);
        if (i == offsets[s] || k != (morton[i - 1] >> (3 * l))) {
          keys[l].push_back(k);
          fps[l].push_back(i);
          sl_of[l].push_back(s);
        }
      }
  for (int l = 0; l < nlev; l++) {
    if ((int)keys[l].size() != tv.soff[l][num_slices])
      fail("node count", l, 0, tv.soff[l][num_slices], (long long)keys[l].size());
    fps[l].push_back(n);
    if ((int)keys[l].size() != cl.tab->nodes[l])
      fail("tab.nodes", l, 0, cl.tab->nodes[l], (long long)keys[l].size());
  }
  if (bad)
    return bad;
  // first children and occupancies
  for (int l = 1; l < nlev; l++) {
    const int m = (int)keys[l].size();
    size_t c = 0;
    for (int q = 0; q < m; q++) {
      while (c < keys[l - 1].size() && fps[l - 1][c] < fps[l][q])
        c++;
      if (tv.fc[l][q] != (int)c)
        fail("fc", l, q, tv.fc[l][q], (long long)c);
    }
    if (tv.fc[l][m] != (int)keys[l - 1].size())
      fail("fc sentinel", l, m, tv.fc[l][m], (long long)keys[l - 1].size());
  }
  if (bad)
    return bad;
  memset(lv.cnt, 0, kMaxLevels * sizeof(int32_t));
  hipLaunchKernelGGL(link_occ_kernel, dim3(3), dim3(256), 0, nullptr, tv, lv);
  for (int l = 1; l < nlev; l++)
    for (int q = 0; q < (int)keys[l].size(); q++) {
      uint32_t occ = 0;
      for (int c = tv.fc[l][q]; c < tv.fc[l][q + 1]; c++)
        occ |= 1u << (int)(keys[l - 1][c] & 7);
      if (lv.occ[l][q] != occ)
        fail("occ", l, q, lv.occ[l][q], occ);
    }
  // a level's records against a direct look-up of the 18 keys inside the node's slice
  auto step = [](int64_t k, int axis, int d) {  // one axis of the Morton key by +-1 (no wrap: out of range = absent)
    int64_t v = 0;
    for (int b = 0; b < 21; b++)
      v |= ((k >> (3 * b + (2 - axis))) & 1) << b;
    v += d;
    if (v < 0 || v >= (1 << 21))
      return (int64_t)-1;
    int64_t r = k;
    for (int b = 0; b < 21; b++) {
      r &= ~((int64_t)1 << (3 * b + (2 - axis)));
      r |= ((v >> b) & 1) << (3 * b + (2 - axis));
    }
    return r;
  };
  auto check_level = [&](int L) {
    const int m = (int)keys[L].size();
    int expect = 0;
    for (int q = 0; q < m; q++)
      if (fps[L][q + 1] - fps[L][q] > 1 || L == nlev - 1)
        expect++;
    if (lv.cnt[L] != expect)
      fail("records", L, 0, lv.cnt[L], expect);
    std::vector<char> seen(expect > 0 ? expect : 1, 0);
    for (int q = 0; q < m; q++) {
      if (!(fps[L][q + 1] - fps[L][q] > 1 || L == nlev - 1))
        continue;
      const int r = lv.lrec[L & 1][q];
      if (r < 0 || r >= expect || seen[r]) {
        fail("lrec", L, q, r, -1);
        continue;
      }
      seen[r] = 1;
      const int32_t* rec = lv.rec[L & 1] + (size_t)r * kLinkRec;
      if (rec[18] != q)
        fail("rec.node", L, q, rec[18], q);
      const int s = sl_of[L][q];
      for (int id = 1; id < 19; id++) {
        int64_t k = keys[L][q];
        for (int axis = 0; axis < 3 && k >= 0; axis++)
          if (link_axis(id, axis))
            k = step(k, axis, link_axis(id, axis));
        int want = -1;
        if (k >= 0) {
          const int lo = tv.soff[L][s], hi = tv.soff[L][s + 1];
          const auto it = std::lower_bound(keys[L].begin() + lo, keys[L].begin() + hi, k);
          if (it != keys[L].begin() + hi && *it == k)
            want = (int)(it - keys[L].begin());
        }
        if (rec[id - 1] != want)
          fail("link", L, q * 100 + id, rec[id - 1], want);
      }
    }
  };
  if (!use_top) {
    hipLaunchKernelGGL(link_top_kernel, dim3(1), dim3(1024), 0, nullptr, tv, lv, nlev - 1);
    check_level(nlev - 1);
    for (int L = nlev - 2; L >= 1; L--) {
      hipLaunchKernelGGL(link_level_kernel, dim3(5), dim3(256), 0, nullptr, tv, lv, L);
      check_level(L);
    }
  } else {
    NoProf prof;
    LinkSchedule ls;
    ls.tv = tv;
    ls.lv = lv;
    std::vector<int32_t> nodes(kMaxLevels, 0);
    for (int l = 0; l < nlev; l++)
      nodes[l] = (int32_t)keys[l].size();
    const int first_need = std::max(1, nlev - 4);
    ls.begin(nullptr, nodes.data(), first_need, prof);
    check_level(ls.next + 1);
    if (ls.next + 2 <= nlev - 1)
      check_level(ls.next + 2);
    for (int need = first_need; need >= 1; need--) {
      ls.produce(nullptr, need, prof);
      check_level(need);
    }
  }
  for (void* p : blocks)
    free(p);
  if (error)
    return 1000 + error;
  return bad;
}

// ---- the one-irsqrt forms of the weight constants against the reference forms ---------
extern "C" int
cx_emu_check_coeffs(int64_t max_exhaustive, int64_t num_random, uint64_t seed)
{
  SharedLut* lut = (SharedLut*)malloc(sizeof(SharedLut));
  hipLaunchKernelGGL(lut_init_kernel, dim3(1), dim3(256), 0, nullptr, lut);
  int bad = 0;
  auto fail = [&](const char* what, long long w1, long long w2, long long got, long long want) {
    if (bad++ < 10)
      fprintf(stderr, "coeff check: %s (%lld, %lld): got %lld want %lld\n", what, w1, w2, got, want);
  };
  for (int64_t w = 1; w <= max_exhaustive; w++) {
    const CxNorm nm = cx_norm((int32_t)w, *lut);
    if (nm.sq != sqrt_weight((int32_t)w, *lut))
      fail("sqrt_weight", w, 0, nm.sq, sqrt_weight((int32_t)w, *lut));
    const int64_t v = 123456789 + 7919 * w;
    if (w > 1 && cx_scale<ArithI64>(v, nm, nm.rs) != scale_rsqrt(v, (int32_t)w, *lut))
      fail("scale_rsqrt", w, 0, cx_scale<ArithI64>(v, nm, nm.rs), scale_rsqrt(v, (int32_t)w, *lut));
    if (w > 1 && cx_scale<ArithI64>(-v, nm, nm.rs) != scale_rsqrt(-v, (int32_t)w, *lut))
      fail("scale_rsqrt(-)", w, 0, cx_scale<ArithI64>(-v, nm, nm.rs), scale_rsqrt(-v, (int32_t)w, *lut));
  }
  uint64_t x = seed | 1;
  auto rnd = [&]() {
    x ^= x << 13;
    x ^= x >> 7;
    x ^= x << 17;
    return x;
  };
  for (int64_t it = 0; it < num_random; it++) {
    // weights of every magnitude up to 2^28, small ones often
    const int sh1 = (int)(rnd() % 28), sh2 = (int)(rnd() % 28);
    const int32_t wl = (int32_t)(1 + rnd() % ((1ull << sh1) + 1));
    const int32_t wr = (int32_t)(1 + rnd() % ((1ull << sh2) + 1));
    if ((int64_t)wl + wr >= (1ll << 29))
      continue;
    int64_t a0, b0, a1, b1, sq;
    raht_coeffs(wl, wr, *lut, &a0, &b0);
    cx_coeffs(wl, wr, sqrt_weight(wl, *lut), sqrt_weight(wr, *lut), *lut, &a1, &b1, &sq);
    if (a0 != a1 || b0 != b1)
      fail("a/b", wl, wr, a1 * 100000 + b1, a0 * 100000 + b0);
    if (sq != sqrt_weight(wl + wr, *lut))
      fail("sqrt of the sum", wl, wr, sq, sqrt_weight(wl + wr, *lut));
  }
  free(lut);
  return bad;
}
