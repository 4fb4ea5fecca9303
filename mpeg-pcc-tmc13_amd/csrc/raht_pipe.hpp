// raht_pipe.hpp -- the DECODER with sub-node prediction as ONE dependency-
// ordered launch over all levels.
//
// raht_subnode.hpp walks the dependency DAG of one level per launch; a level's
// launch cannot start before the previous one has finished, so the frame pays
// the SUM of the per-level chains (4 496 hops between blocks with two or more
// children for the 1 M-point lidar frame, tools/chain_model_xlevel.py).  But a
// block of level li-1 only needs its own node and the parent-level neighbours
// it predicts from -- a handful of nodes of level li -- not the whole level:
// walked across levels the longest path of the same frame is 1 214 hops.  The
// encoder cannot follow (its RDOQ state is carried in coding order, all of a
// level before the next: DESIGN.md section 4.2); the decoder can, and this is
// it.
//
// What changes against the per-level kernel:
//  * every node value lives in a store of its own (no ping-pong by level
//    parity): two 16-byte granules per node and component, G = {rec lo, hi,
//    tag, numParentNeigh} and U = {rec_us lo, hi, tag, 0}, indexed by a
//    UNIFIED node index uoff[level] + node; the data is the flag;
//  * a parent with a single child is not copied down level by level: `src`
//    maps every node to the node that holds its value (itself, or the nearest
//    ancestor that is a child of a real block), built top-down by the
//    tree-only prepasses; a reader polls src[node];
//  * the parent-level terms (the block's own node: DC, numParentNeigh; up to
//    18 neighbour parents) are polled like the same-level children, in two
//    stages in front of the existing loop;
//  * wavefronts claim rounds of 8 blocks over the whole frame, levels in
//    descending order, blocks in Morton order: every dependency has an earlier
//    ticket, so the lowest unfinished round can always run.
// Bit-exact with the per-level path (and so with the reference): the same
// arithmetic on the same values, only the waiting differs.
#pragma once

#include "raht_arith.hpp"
#include "raht_subnode.hpp"

namespace gpcc {

// Where a round's wave time goes (experiment builds only, -DGPCC_PIPE_PROF:
// s_memtime at the stages of a round, summed per level into g_pipe_prof and
// read back with gpcc_debug_pipe_prof).  Empty otherwise.
#ifdef GPCC_PIPE_PROF
__device__ unsigned long long g_pipe_prof[32 * 8];
struct PipeProf {
  unsigned long long last = 0;
  int li = 0;
  __device__ void begin(int l) { li = l; last = __builtin_amdgcn_s_memtime(); }
  __device__ void mark(int stage, int lane)
  {
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    if (lane == 0)
      atomicAdd(&g_pipe_prof[li * 8 + stage], t - last);
    last = t;
  }
  __device__ void round(int lane) { if (lane == 0) atomicAdd(&g_pipe_prof[li * 8 + 7], 1ull); }
};
#else
struct PipeProf {
  __device__ void begin(int) {}
  __device__ void mark(int, int) {}
  __device__ void round(int) {}
};
#endif

struct PipeCtx {
  int32_t* src;        // [total] unified index of the node that holds a node's value
  uint32_t* g;         // [total * C][4]
  uint32_t* u;         // [total * C][4]
  uint8_t* pocc;       // [total] child occupancy of every parent (unified index of the parent)
  int32_t* worklist;   // [total] blocks of level li at uoff[li + 1] ..
  int32_t uoff[kMaxLevels + 1];
  int32_t total;
  int32_t top;         // the launch covers children levels top-1 .. 0
  uint32_t tag;
  int32_t* ticket;     // [8]
};

// src entries: unified index of the node that holds the value; kPipeCopied is
// set when a single-child copy lies on the way (numParentNeigh is then 19)
constexpr int32_t kPipeCopied = (int32_t)0x80000000;
constexpr int32_t kPipeIndex = 0x7fffffff;

__global__ __launch_bounds__(256) void
pipe_iota_kernel(int32_t* src, int n)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    src[i] = i;
}

// The tree-only half of raht_level_prepass_kernel for one level: worklist of
// the real blocks in ascending order, child occupancy of every parent, and the
// value source of every child (its own node, or its parent's source when the
// parent has this one child).  Launched level after level, top down.
template<int C>
__global__ __launch_bounds__(256) void
raht_pipe_prepass_kernel(LevelCtx ctx, PipeCtx px)
{
  __shared__ int wave_cnt[4];
  __shared__ int base_s;
  if (tree_failed(ctx.tv))
    return;
  const TreeView& tv = ctx.tv;
  const int li = ctx.li;
  const bool ext = ctx.params->raht_extension != 0;
  const int num_parents = tv.soff[li + 1][tv.num_slices];
  const int lane = lane_id(), wave = threadIdx.x >> 6;
  const int pbase = px.uoff[li + 1], cbase = px.uoff[li];
  const int64_t chunks = ((int64_t)num_parents + 255) >> 8;
  const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
  const int64_t gbeg = min((int64_t)blockIdx.x * per, chunks);
  const int64_t gend = min(gbeg + per, chunks);

  int mine = 0;
  for (int64_t chunk = gbeg; chunk < gend; chunk++) {
    const int j = (int)(chunk * 256) + threadIdx.x;
    if (j < num_parents) {
      const int s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      if (ctx.sched[s].lvl[li].processed) {
        const int nchild = tv.fc[li + 1][j + 1] - tv.fc[li + 1][j];
        mine += !(ext && nchild == 1);
      }
    }
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1)
    mine += __shfl_xor(mine, d);
  if (lane == 0)
    wave_cnt[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long ep = (unsigned long long)(li + 1) << 48;
    const unsigned total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    unsigned excl = 0;
    if (blockIdx.x > 0) {
      __hip_atomic_store(
        &ctx.scan_state[blockIdx.x], ep | (1ull << 32) | total, __ATOMIC_RELAXED,
        __HIP_MEMORY_SCOPE_AGENT);
      int k = (int)blockIdx.x - 1;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(
          &ctx.scan_state[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 48) != (unsigned long long)(li + 1)) {
          __builtin_amdgcn_s_sleep(2);
          continue;
        }
        excl += (unsigned)v;
        if (((v >> 32) & 0xffff) == 2)
          break;
        k--;
      }
    }
    __hip_atomic_store(
      &ctx.scan_state[blockIdx.x], ep | (2ull << 32) | (excl + total),
      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == gridDim.x - 1)
      ctx.work_count[li] = (int)(excl + total);
    base_s = (int)excl;
  }
  __syncthreads();
  int running = base_s;

  for (int64_t chunk = gbeg; chunk < gend; chunk++) {
    const int j = (int)(chunk * 256) + threadIdx.x;
    bool real = false;
    if (j < num_parents) {
      const int s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      const LevelSched e = ctx.sched[s].lvl[li];
      if (e.processed) {
        const int c0 = tv.fc[li + 1][j];
        const int nchild = tv.fc[li + 1][j + 1] - c0;
        uint32_t o = 0;
        for (int u = 0; u < nchild; u++)
          o |= 1u << (int)(tv.key[li][c0 + u] & 7);
        px.pocc[pbase + j] = (uint8_t)o;
        if (ext && nchild == 1) {
          // the copy of the per-level path: value of the parent, numParentNeigh 19
          px.src[cbase + c0] = px.src[pbase + j] | kPipeCopied;
        } else {
          real = true;
          for (int u = 0; u < nchild; u++)
            px.src[cbase + c0 + u] = cbase + c0 + u;
        }
      } else {
        // a level the slice does not process (nothing merges there): the node
        // IS its parent, numParentNeigh included
        const int c0 = tv.fc[li + 1][j];
        if (tv.fc[li + 1][j + 1] - c0 == 1)
          px.src[cbase + c0] = px.src[pbase + j];
      }
    }
    const unsigned long long m = __ballot(real);
    __syncthreads();
    if (lane == 0)
      wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; w++)
      off += wave_cnt[w];
    if (real)
      px.worklist[pbase + off + __popcll(m & ((1ull << lane) - 1))] = j;
    running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
  }
}

// The leaf values where finish_kernel expects them.
template<int C>
__global__ __launch_bounds__(256) void
pipe_leaf_kernel(LevelCtx ctx, PipeCtx px)
{
  if (tree_failed(ctx.tv))
    return;
  const TreeView& tv = ctx.tv;
  const int m = tv.soff[0][tv.num_slices];
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < m; j += gridDim.x * blockDim.x) {
    const int s = find_slice(tv.soff[0], tv.num_slices, j);
    const SliceSched* sc = &ctx.sched[s];
    if (sc->num_unique <= 1)
      continue;
    const int64_t row = (int64_t)tv.pt_off[s] + (j - tv.soff[0][s]);
    const int from = px.src[px.uoff[0] + j] & kPipeIndex;
    const u32x4* gp = reinterpret_cast<const u32x4*>(px.g);
#pragma unroll
    for (int k = 0; k < C; k++) {
      const u32x4 gv = gp[(size_t)from * C + k];
      par2(ctx.rec, sc->final_parity)[row * C + k] = (int64_t)(((uint64_t)gv.y << 32) | gv.x);
    }
  }
}

// A = the arithmetic back end (raht_arith.hpp).  The granules keep int64 values whatever A is
// (the prepasses, the coarse levels and pipe_leaf read and write them): a value is converted
// where it is loaded or stored, hand-offs inside a wavefront stay in A's registers.
template<int C, class A = ArithI64>
__global__ __launch_bounds__(256, C == 3 ? GPCC_SUB_SYNTH3_WAVES : 4) void
raht_pipe_synth_kernel(LevelCtx ctx, PipeCtx px)
{
  typedef typename A::T VT;
  typedef typename A::Coef VC;
  __shared__ SharedLut lut_s;
  if (tree_failed(ctx.tv))
    return;
  load_lut(&lut_s, ctx.lut);
  const SharedLut& lut = lut_s;

  const TreeView& tv = ctx.tv;
  const ParamsConst prm = (ParamsConst)ctx.params;
  const int t = threadIdx.x & 7;
  const int lane = lane_id();
  const int gbase = threadIdx.x & 56;
  const bool haar = !A::kF64 && prm->integer_haar_enable_flag != 0;
  const bool ext = A::kF64 || prm->raht_extension != 0;
  const int cls = blockIdx.x & 7;
  bool in_range = true;  // (ArithF64: the magnitudes that bound every product, raht_arith.hpp)
  const auto grsrc = __builtin_amdgcn_make_buffer_rsrc(
    px.g, 0, (int)((size_t)px.total * C * 16), 0x00020000);
  const auto ursrc = __builtin_amdgcn_make_buffer_rsrc(
    px.u, 0, (int)((size_t)px.total * C * 16), 0x00020000);

  for (;;) {
    int tk = 0;
    if (lane == 0)
      tk = atomicAdd(&px.ticket[cls], 1);
    tk = __shfl(tk, 0);
    const int64_t ground = (int64_t)tk * 8 + cls;
    // which level does this round belong to?  (levels in descending order)
    int li = -1;
    int64_t rbase = 0;
    for (int l = px.top - 1; l >= 0; l--) {
      const int64_t r = ((int64_t)ctx.work_count[l] + 7) >> 3;
      if (ground < rbase + r) {
        li = l;
        break;
      }
      rbase += r;
    }
    if (li < 0)
      break;
    if (__hip_atomic_load(ctx.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      break;
    PipeProf prof;
    prof.begin(li);
    prof.round(lane);
    const int num_work = ctx.work_count[li];
    const int wi = (int)((ground - rbase) * 8) + (lane >> 3);
    const bool live = wi < num_work;
    const int pbase = px.uoff[li + 1], cbase = px.uoff[li];
    const int j = live ? px.worklist[pbase + wi] : 0;
    int s = 0;
    LevelSched e;
    e.processed = 0;
    if (live) {
      s = find_slice(tv.soff[li + 1], tv.num_slices, j);
      e = ctx.sched[s].lvl[li];
    }
    const bool on = live && e.processed;
    const int sp0 = on ? tv.soff[li + 1][s] : 0;
    const int sp1 = on ? tv.soff[li + 1][s + 1] : 0;
    const int sc0 = on ? tv.soff[li][s] : 0;
    const int pt0 = on ? tv.pt_off[s] : 0;
    const int n_s = on ? tv.pt_off[s + 1] - pt0 : 0;
    const int c0 = on ? tv.fc[li + 1][j] : 0;
    const int nchild = on ? tv.fc[li + 1][j + 1] - c0 : 0;
    const int pj = j - sp0;

    // ---- children -> positions ---------------------------------------
    const int64_t ckey = t < nchild ? tv.key[li][c0 + t] : 0;
    const uint32_t occ = group8_or(t < nchild ? 1u << (int)(ckey & 7) : 0u);
    const bool has = (occ >> t) & 1;
    const int child = c0 + popc32(occ & ((1u << t) - 1));
    int32_t w = 0;
    if (has)
      w = tv.fp[li][child + 1] - tv.fp[li][child];

    prof.mark(0, lane);  // block located, children, weights
    // ---- butterfly weights + coefficients (mkWeightTree :742) ----------
    int32_t wl[3], wr[3];
    VC ca[3], cb[3];
    int32_t cw = w;
#pragma unroll
    for (int st = 0; st < 3; st++) {
      const int bit = 1 << st;
      const int32_t pw = lane_xor8(cw, bit);
      const bool left = !(t & bit);
      wl[st] = left ? cw : pw;
      wr[st] = left ? pw : cw;
      int64_t ia = 0, ib = 0;
      if (wl[st] && wr[st]) {
        if (!haar)
          raht_coeffs(wl[st], wr[st], lut, &ia, &ib);
        cw = wl[st] + wr[st];
      } else {
        cw = left ? wl[st] + wr[st] : 0;
      }
      ca[st] = A::coef(ia);
      cb[st] = A::coef(ib);
    }

    // ---- inter-level prediction gating (tmc3/RAHT.cpp:1391-1432): the
    //      neighbour search depends on the tree alone and runs before the
    //      parent's numParentNeigh is there; the test follows stage A1 --------
    const bool inherit_dc = !e.is_root;
    const bool pred_in_level = on && inherit_dc && prm->raht_prediction_enabled_flag != 0;
    bool enable_pred = pred_in_level;
    int neigh_count = 0;
    VT pred[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pred[k] = A::zero();
    bool want_search = false;
    if (pred_in_level) {
      if (ext && nchild == 1) {
        enable_pred = false;
        neigh_count = 19;
      } else {
        want_search = true;
      }
    }
    // A first look at the block's own node: when it is there already (the usual
    // case away from the front of the walk) numParentNeigh decides at once
    // whether the neighbours have to be searched at all (threshold0)
    const int pbase_j = pbase + j;
    const int psrc_raw = (on && inherit_dc && t == 0) ? px.src[pbase_j] : 0;
    const int psrc = psrc_raw & kPipeIndex;
    int64_t own_v[C], own_us[C];
    int own_nn = 0;
    bool a1_done = false;
    {
      const bool need = on && inherit_dc && t == 0;
      u32x4 a[C], b[C];
#pragma unroll
      for (int k = 0; k < C; k++) {
        own_v[k] = own_us[k] = 0;
        a[k] = b[k] = u32x4{0, 0, 0, 0};
        if (need) {
          a[k] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (psrc * C + k) * 16, 0, /*sc1*/ 16);
          b[k] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, (psrc * C + k) * 16, 0, /*sc1*/ 16);
        }
      }
      bool ok = need;
#pragma unroll
      for (int k = 0; k < C; k++)
        ok = ok && a[k].z == px.tag && b[k].z == px.tag;
      if (ok) {
#pragma unroll
        for (int k = 0; k < C; k++) {
          own_v[k] = (int64_t)(((uint64_t)a[k].y << 32) | a[k].x);
          own_us[k] = (int64_t)(((uint64_t)b[k].y << 32) | b[k].x);
        }
        own_nn = (int)a[0].w;
        a1_done = true;
      }
    }
    {
      const int early = __shfl(a1_done ? ((psrc_raw & kPipeCopied) ? 19 : own_nn) : 1 << 20, gbase);
      if (want_search && early < prm->raht_prediction_threshold0)
        want_search = false;  // (the test after stage A1 turns prediction off)
    }
    int pn[3] = {-1, -1, -1};  // neighbour i = 1 + t + 8*slot
    {
      int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, end[3] = {0, 0, 0};
      int64_t want[3] = {0, 0, 0};
      if (want_search) {
        const int64_t cur_pos = tv.key[li + 1][j];
        const uint64_t base = morton3d_add((uint64_t)cur_pos, ~0ull);
        const int64_t range = prm->raht_prediction_search_range;
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          const int i = 1 + t + 8 * slot;
          if (i < 19 && (occ & neigh_mask(i))) {
            const int64_t np = (int64_t)morton3d_add(base, neigh_offset(i));
            int64_t d = np - cur_pos;
            if (d >= 0) {
              d = d >= range ? range : d;
              lo[slot] = j;
              end[slot] = (d + 1 < (int64_t)(sp1 - j)) ? j + (int)(d + 1) : sp1;
            } else {
              d = (-d) >= range ? range : -d;
              end[slot] = j;
              lo[slot] = (d < (int64_t)(j - sp0)) ? j - (int)d : sp0;
            }
            hi[slot] = end[slot];
            want[slot] = np;
          }
        }
      }
      const int64_t* __restrict__ pkey = tv.key[li + 1];
      while (__any((lo[0] < hi[0]) | (lo[1] < hi[1]) | (lo[2] < hi[2]))) {
        int mid[3];
        int64_t kv[3];
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          mid[slot] = lo[slot] + ((hi[slot] - lo[slot]) >> 1);
          kv[slot] = lo[slot] < hi[slot] ? pkey[mid[slot]] : 0;
        }
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          if (lo[slot] < hi[slot]) {
            if (kv[slot] < want[slot])
              lo[slot] = mid[slot] + 1;
            else
              hi[slot] = mid[slot];
          }
        }
      }
#pragma unroll
      for (int slot = 0; slot < 3; slot++) {
        if (lo[slot] < end[slot] && pkey[lo[slot]] == want[slot])
          pn[slot] = lo[slot];
      }
    }

    prof.mark(1, lane);  // early look + neighbour search
    // ---- coefficient slot of this position (scanBlock :776-791) --------
    const uint32_t present = group8_bits(on && cw != 0) | (on ? 1u : 0u);
    const int spos = (0x74516230u >> (4 * t)) & 7;
    const uint32_t pscan = ((present >> 0) & 1) | (((present >> 4) & 1) << 1)
      | (((present >> 2) & 1) << 2) | (((present >> 1) & 1) << 3)
      | (((present >> 6) & 1) << 4) | (((present >> 5) & 1) << 5)
      | (((present >> 3) & 1) << 6) | (((present >> 7) & 1) << 7);
    const int rank = popc32(pscan & ((1u << spos) - 1));
    const bool coded = on && ((present >> t) & 1) && (t != 0 || !inherit_dc);
    const int cidx = e.coeff_base + (inherit_dc ? (c0 - sc0) - pj + rank - 1 : rank);
    const int32_t* __restrict__ cplane = ctx.coeffs + (size_t)pt0 * C + cidx;
    Quantizer qa[2] = {{1, 1}, {1, 1}};
    if (coded) {
      int ac0 = 0, ac1 = 0;
      if (e.ac_layer < prm->num_ac_qp_layers && t) {
        ac0 = prm->ac_qp_offset[e.ac_layer][t - 1][0];
        ac1 = prm->ac_qp_offset[e.ac_layer][t - 1][1];
      }
      qpset_quantizers(prm, e.qp_layer, ac0, ac1, qa);
    }
    int32_t nrm_sq_i = 0, nrm_rs_i = 0, nrm_shift = 0;
    if (!haar && w > 1) {
      nrm_sq_i = (int32_t)sqrt_weight(w, lut);
      if (w < kSmallN) {
        nrm_rs_i = lut.norm_rs[w];
      } else {
        const uint64_t w64 = (uint64_t)w;
        nrm_shift = w64 > 1024 ? ilog2_u64(w64 - 1) >> 1 : 0;
        nrm_rs_i = (int32_t)(irsqrt(w64, lut.rsqrt) >> (40 - nrm_shift - kFpFrac));
      }
    }
    const VC nrm_sq = A::coef(nrm_sq_i), nrm_rs = A::coef(nrm_rs_i);
    const typename A::Quant qaa[2] = {A::quant(qa[0]), A::quant(qa[1])};
    int32_t qc[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      qc[k] = coded ? cplane[(size_t)k * n_s] : 0;

    // value sources of what this lane will wait for (tree only)
    int nsrc[3] = {0, 0, 0};
#pragma unroll
    for (int slot = 0; slot < 3; slot++)
      if (pn[slot] >= 0)
        nsrc[slot] = px.src[pbase + pn[slot]] & kPipeIndex;

    prof.mark(2, lane);  // slots, quantisers, coefficients, sources
    unsigned spins = 0;
    bool failed = false;
    // ---- stage A1: the block's own node (lane 0 of the group) ------------
    {
      bool need = on && inherit_dc && t == 0 && !a1_done;
      while (__any(need)) {
        u32x4 a[C], b[C];
#pragma unroll
        for (int k = 0; k < C; k++) {
          a[k] = b[k] = u32x4{0, 0, 0, 0};
          if (need) {
            a[k] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (psrc * C + k) * 16, 0, /*sc1*/ 16);
            b[k] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, (psrc * C + k) * 16, 0, /*sc1*/ 16);
          }
        }
        bool ok = need;
#pragma unroll
        for (int k = 0; k < C; k++)
          ok = ok && a[k].z == px.tag && b[k].z == px.tag;
        if (ok) {
#pragma unroll
          for (int k = 0; k < C; k++) {
            own_v[k] = (int64_t)(((uint64_t)a[k].y << 32) | a[k].x);
            own_us[k] = (int64_t)(((uint64_t)b[k].y << 32) | b[k].x);
          }
          own_nn = (int)a[0].w;
          need = false;
        }
        if (!__any(ok)) {
          if (++spins > (1u << 21)) {
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(GPCC_SUB_SLEEP);
        }
      }
    }
    // a node whose parent has this one child has numParentNeigh 19 (the copy
    // of the per-level path, tmc3/RAHT.cpp:1382-1401)
    const int pneigh = __shfl((psrc_raw & kPipeCopied) ? 19 : own_nn, gbase);
    bool do_search = false;
    if (pred_in_level && !(ext && nchild == 1)) {
      if (pneigh < prm->raht_prediction_threshold0)
        enable_pred = false;
      else
        do_search = true;  // (then the early look did not cancel the search)
    }
    if (!do_search) {
      pn[0] = pn[1] = pn[2] = -1;
    }
    {
      int found = (pn[0] >= 0) + (pn[1] >= 0) + (pn[2] >= 0);
      found = group8_sum(found);
      if (do_search) {
        neigh_count = found + 1;
        if (neigh_count < prm->raht_prediction_threshold1)
          enable_pred = false;
      }
    }
    const bool run = do_search && enable_pred;

    prof.mark(3, lane);  // stage A1
    // ---- stage A2: the neighbour parents this lane searched ----------------
    int64_t nbv[3][C];
#pragma unroll
    for (int slot = 0; slot < 3; slot++)
#pragma unroll
      for (int k = 0; k < C; k++)
        nbv[slot][k] = 0;
    if (!failed) {
      uint32_t need = 0;
#pragma unroll
      for (int slot = 0; slot < 3; slot++)
        if (run && pn[slot] >= 0)
          need |= 1u << slot;
      while (__any(need != 0)) {
        u32x4 a[3][C];
#pragma unroll
        for (int slot = 0; slot < 3; slot++)
#pragma unroll
          for (int k = 0; k < C; k++) {
            a[slot][k] = u32x4{0, 0, 0, 0};
            if ((need >> slot) & 1)
              a[slot][k] = __builtin_amdgcn_raw_buffer_load_b128(
                grsrc, (nsrc[slot] * C + k) * 16, 0, /*sc1*/ 16);
          }
        bool any_ok = false;
#pragma unroll
        for (int slot = 0; slot < 3; slot++) {
          bool ok = (need >> slot) & 1;
#pragma unroll
          for (int k = 0; k < C; k++)
            ok = ok && a[slot][k].z == px.tag;
          if (ok) {
#pragma unroll
            for (int k = 0; k < C; k++)
              nbv[slot][k] = (int64_t)(((uint64_t)a[slot][k].y << 32) | a[slot][k].x);
            need &= ~(1u << slot);
            any_ok = true;
          }
        }
        if (!__any(any_ok)) {
          if (++spins > (1u << 21)) {
            failed = true;
            break;
          }
          __builtin_amdgcn_s_sleep(GPCC_SUB_SLEEP);
        }
      }
    }
    if (__any(failed)) {
      if (lane == 0)
        atomicExch(ctx.error, 1);
      break;
    }

    prof.mark(4, lane);  // stage A2
    VT dc[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
      const int64_t val = own_us[k];
      dc[k] = A::from_i64(
        (on && inherit_dc && t == 0)
          ? (ext ? val : (val > 0 ? val << (kFpFrac - 2) : -((-val) << (kFpFrac - 2))))
          : 0);
      in_range = in_range && A::below(dc[k], A::kInvLimit);
    }

    int wsum = 0;
    VT lim_lo = A::zero(), lim_hi = A::zero();
    // intraDcPred, the seven neighbours that never use child values
    // (tmc3/RAHT.cpp:463-502 with parentOnlyCheckMaxIdx = 7)
#pragma unroll
    for (int i = 0; i < 7; i++) {
      int q;
      VT v[C];
      if (i == 0) {
        q = j;
#pragma unroll
        for (int k = 0; k < C; k++)
          v[k] = A::from_i64(shfl_i64(own_v[k], gbase));
      } else {
        q = __shfl(pn[0], gbase | (i - 1));
#pragma unroll
        for (int k = 0; k < C; k++)
          v[k] = A::from_i64(shfl_i64(nbv[0][k], gbase | (i - 1)));
      }
#pragma unroll
      for (int k = 0; k < C; k++)
        in_range = in_range && A::below(v[k], A::kRecLimit);
      if (!run || q < 0)
        continue;
      if (i) {
        if (A::muli(v[0], 10) <= lim_lo || A::muli(v[0], 10) >= lim_hi)
          continue;
      } else {
        lim_lo = A::muli(v[0], 2);
        lim_hi = A::muli(v[0], 25);
      }
      if (has && ((neigh_mask(i) >> t) & 1)) {
        const int pw = prm->pred_weight_parent[i];
        wsum += pw;
        const int mul = ext ? pw : (pw << kFpFrac);
#pragma unroll
        for (int k = 0; k < C; k++)
          pred[k] += A::muli(v[k], mul);
      }
    }

    // ---- neighbours 7..18 (tmc3/RAHT.cpp:503-565) --------------------------
    int nb_c0[3] = {0, 0, 0};
    uint32_t nb_occ[3] = {0, 0, 0};
#pragma unroll
    for (int slot = 0; slot < 3; slot++) {
      const int i = 1 + t + 8 * slot;
      if (run && i >= 7 && i < 19 && pn[slot] >= 0) {
        const int q = pn[slot];
        if (q < j) {  // processed before this block: its children count
          nb_c0[slot] = tv.fc[li + 1][q];
          nb_occ[slot] = px.pocc[pbase + q];
        }
      }
    }
    uint32_t pend = 0;       // neighbours whose child granule is awaited
    int32_t nrow12[12];      // unified index of the node that holds that child's value
    uint32_t inw = 0, wsrc_a = 0, wsrc_b = 0;
    int jg[8];
#pragma unroll
    for (int g = 0; g < 8; g++)
      jg[g] = __shfl(on ? j : -1, g << 3);
    // pass 1 (no memory): which (lane, neighbour) pairs use a child, and which
    uint32_t own = 0;  // of those: produced by a block of THIS wavefront's round
#pragma unroll
    for (int i12 = 0; i12 < 12; i12++) {
      nrow12[i12] = 0;
      const int i = 7 + i12;
      const int owner = gbase | ((i - 1) & 7);
      const int sl = (i - 1) >> 3;
      const int q = __shfl(pn[sl], owner);
      const int qc0 = __shfl(nb_c0[sl], owner);
      const uint32_t qocc = __shfl(nb_occ[sl], owner);
      VT v[C];
#pragma unroll
      for (int k = 0; k < C; k++) {
        v[k] = A::from_i64(shfl_i64(nbv[sl][k], owner));
        in_range = in_range && A::below(v[k], A::kRecLimit);
      }
      if (!run || q < 0)
        continue;
      if (A::muli(v[0], 10) <= lim_lo || A::muli(v[0], 10) >= lim_hi)
        continue;
      if (has && ((neigh_mask(i) >> t) & 1)) {
        const int sh = occu_shift(i12);
        const int cpos = i12 < 9 ? t + sh : t - sh;
        const bool child_ok = cpos >= 0 && cpos < 8 && ((qocc >> cpos) & 1);
        if (child_ok) {
          nrow12[i12] = cbase + qc0 + popc32(qocc & ((1u << cpos) - 1));
          wsum += (int)prm->pred_weight_child[i12];
          pend |= 1u << i12;
          int pg = -1;
#pragma unroll
          for (int g = 0; g < 7; g++)
            pg = q == jg[g] ? g : pg;
          if (pg >= 0) {
            own |= 1u << i12;
            if (i12 < 10)
              wsrc_a |= (uint32_t)pg << (3 * i12);
            else
              wsrc_b |= (uint32_t)pg << (3 * (i12 - 10));
          }
        } else {
          const int pwp = prm->pred_weight_parent[i];
          wsum += pwp;
          const int mul = ext ? pwp : (pwp << kFpFrac);
#pragma unroll
          for (int k = 0; k < C; k++)
            pred[k] += A::muli(v[k], mul);
        }
      }
    }
    // pass 2: the value sources of all of them, loads in flight together
    {
      int32_t from[12];
#pragma unroll
      for (int i12 = 0; i12 < 12; i12++)
        from[i12] = ((pend >> i12) & 1) ? px.src[nrow12[i12]] : 0;
#pragma unroll
      for (int i12 = 0; i12 < 12; i12++) {
        if (!((pend >> i12) & 1))
          continue;
        const int f = from[i12] & kPipeIndex;
        // a child of a block of this wavefront's round comes from registers
        if (((own >> i12) & 1) && f == nrow12[i12])
          inw |= 1u << i12;
        nrow12[i12] = f;
      }
    }
    // pass 3: one look at every awaited granule, four in flight at a time --
    // most are there already (values of coarser levels, blocks long done); the
    // loop below then waits for the rest one per iteration
#pragma unroll
    for (int b4 = 0; b4 < 12; b4 += 4) {
      u32x4 gq[4][C];
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++)
#pragma unroll
        for (int k = 0; k < C; k++) {
          gq[q4][k] = u32x4{0, 0, 0, 0};
          if (((pend & ~inw) >> (b4 + q4)) & 1)
            gq[q4][k] = __builtin_amdgcn_raw_buffer_load_b128(
              grsrc, (nrow12[b4 + q4] * C + k) * 16, 0, /*sc1*/ 16);
        }
#pragma unroll
      for (int q4 = 0; q4 < 4; q4++) {
        const int i12 = b4 + q4;
        bool ok = ((pend & ~inw) >> i12) & 1;
#pragma unroll
        for (int k = 0; k < C; k++)
          ok = ok && gq[q4][k].z == px.tag;
        if (ok) {
          const int pwc = prm->pred_weight_child[i12];
          const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
          for (int k = 0; k < C; k++) {
            const VT cv = A::from_i64((int64_t)(((uint64_t)gq[q4][k].y << 32) | gq[q4][k].x));
            in_range = in_range && A::below(cv, A::kRecLimit);
            pred[k] += A::muli(cv, mul);
          }
          pend &= ~(1u << i12);
        }
      }
    }
    const VC pdiv = A::coef(pred_divisor(wsum > 0 ? wsum : 1));

    prof.mark(5, lane);  // parent-level terms, children sources, first polls
    // ---- the dependency loop (raht_subnode.hpp, decoder) --------------------
    int stage = on ? 0 : 3;
    VT pt[C];
#pragma unroll
    for (int k = 0; k < C; k++)
      pt[k] = A::zero();
    while (__any(stage != 3)) {
      bool progressed = false;
      if (__any(stage == 0 && inw)) {
#pragma unroll
        for (int i12 = 0; i12 < 12; i12++) {
          const bool mine = stage == 0 && ((inw >> i12) & 1);
          if (!__any(mine))
            continue;
          const int pg = (i12 < 10 ? wsrc_a >> (3 * i12) : wsrc_b >> (3 * (i12 - 10))) & 7;
          const int sh = occu_shift(i12);
          const int srcl = (pg << 3) | ((i12 < 9 ? t + sh : t - sh) & 7);
          const int pst = __shfl(stage, srcl);
          VT v[C];
#pragma unroll
          for (int k = 0; k < C; k++)
            v[k] = shfl_v(pt[k], srcl);
          if (mine && pst == 3) {
            const int pwc = prm->pred_weight_child[i12];
            const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
            for (int k = 0; k < C; k++)
              pred[k] += A::muli(v[k], mul);
            pend &= ~(1u << i12);
            inw &= ~(1u << i12);
          }
        }
      }
      const uint32_t pm = stage == 0 ? (pend & ~inw) : 0u;
      if (pm) {
        const int slot = __ffs(pm) - 1;
        int32_t row = 0;
        int pwc = 0;
#pragma unroll
        for (int i12 = 0; i12 < 12; i12++) {
          if (slot == i12) {
            row = nrow12[i12];
            pwc = prm->pred_weight_child[i12];
          }
        }
        u32x4 g[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          g[k] = __builtin_amdgcn_raw_buffer_load_b128(grsrc, (row * C + k) * 16, 0, /*sc1*/ 16);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < C; k++)
          ok = ok && g[k].z == px.tag;
        if (ok) {
          const int mul = ext ? pwc : (pwc << kFpFrac);
#pragma unroll
          for (int k = 0; k < C; k++) {
            const VT cv = A::from_i64((int64_t)(((uint64_t)g[k].y << 32) | g[k].x));
            in_range = in_range && A::below(cv, A::kRecLimit);
            pred[k] += A::muli(cv, mul);
          }
          pend &= ~(1u << slot);
        }
      }
      const bool blocked = group8_any(stage == 0 && pend);
      const bool nready = stage == 0 && !blocked;

      if (__any(nready)) {
        progressed = true;
        // ---- (P) normalise the prediction, transform -----------------------
        VT pw_[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          pw_[k] = pred[k];
        if (run && has) {
#pragma unroll
          for (int k = 0; k < C; k++) {
            pw_[k] = A::mulc(pw_[k], pdiv);
            if constexpr (!A::kF64) {
              if (haar)
                pw_[k] = (pw_[k] >> kFpFrac) << kFpFrac;
            }
          }
        }
        if (!haar && w > 1 && enable_pred) {
#pragma unroll
          for (int k = 0; k < C; k++)
            pw_[k] = A::mulc(pw_[k], nrm_sq);
        }
#pragma unroll
        for (int k = 0; k < C; k++)
          in_range = in_range && A::below(pw_[k], A::kFwdLimit);
#pragma unroll
        for (int st = 0; st < 3; st++) {
          const int bit = 1 << st;
          const bool left = !(t & bit);
          const bool both = wl[st] && wr[st];
          const bool swap = !wl[st] && wr[st];
#pragma unroll
          for (int k = 0; k < C; k++) {
            const VT own = pw_[k], oth = shfl_xor_v(own, bit);
            if (enable_pred) {
              if (both) {
                if (haar) {
                  if constexpr (!A::kF64) {
                    const int64_t hf = left ? oth - own : own - oth;
                    pw_[k] = left ? own + ((hf >> (1 + kFpFrac)) << kFpFrac) : hf;
                  }
                } else {
                  pw_[k] = left ? A::mulc(oth, cb[st]) + A::mulc(own, ca[st])
                                : A::mulc(own, ca[st]) - A::mulc(oth, cb[st]);
                }
              } else if (swap) {
                pw_[k] = oth;
              }
            }
          }
        }
        if (nready) {
#pragma unroll
          for (int k = 0; k < C; k++)
            pt[k] = pw_[k];
          stage = 1;
        }
      }

      const bool can = stage == 1;
      if (__any(can)) {
        progressed = true;
        // ---- (W) coefficients, DC, inverse transform, commit ---------------
        VT pw_[C];
#pragma unroll
        for (int k = 0; k < C; k++)
          pw_[k] = pt[k];
        if (coded && can) {
#pragma unroll
          for (int k = 0; k < C; k++)
            pw_[k] += A::dequant_fp(qaa[k ? 1 : 0], qc[k]);
        }
        if (on && inherit_dc && t == 0) {
#pragma unroll
          for (int k = 0; k < C; k++)
            pw_[k] = dc[k];
        }
#pragma unroll
        for (int k = 0; k < C; k++)
          in_range = in_range && A::below(pw_[k], A::kInvLimit);
#pragma unroll
        for (int st = 2; st >= 0; st--) {
          const int bit = 1 << st;
          const bool left = !(t & bit);
          const bool both = wl[st] && wr[st];
          const bool swap = !wl[st] && wr[st];
#pragma unroll
          for (int k = 0; k < C; k++) {
            const VT own = pw_[k], oth = shfl_xor_v(own, bit);
            if (both) {
              if (haar) {
                if constexpr (!A::kF64) {
                  const int64_t lf = left ? own : oth, hf = left ? oth : own;
                  const int64_t lv = lf - ((hf >> (1 + kFpFrac)) << kFpFrac);
                  pw_[k] = left ? lv : hf + lv;
                }
              } else {
                pw_[k] = left ? A::mulc(own, ca[st]) - A::mulc(oth, cb[st])
                              : A::mulc(oth, cb[st]) + A::mulc(own, ca[st]);
              }
            } else if (swap) {
              pw_[k] = oth;
            }
          }
        }
        if (can && has) {
          const uint32_t nn = (uint32_t)(inherit_dc ? neigh_count : 19);
          const int uc = cbase + child;
#pragma unroll
          for (int k = 0; k < C; k++) {
            VT v = pw_[k];
            if (!haar && w > 1)
              v = A::mulc(A::shr(v, nrm_shift), nrm_rs);
            v = ext ? v : A::round_int(v);
            pt[k] = v;  // read by later groups of this wavefront once stage == 3
            const int64_t vi = A::to_i64(v);
            const u32x4 gr = {(uint32_t)vi, (uint32_t)((uint64_t)vi >> 32), px.tag, nn};
            __builtin_amdgcn_raw_buffer_store_b128(gr, grsrc, (uc * C + k) * 16, 0, /*sc1*/ 16);
          }
#pragma unroll
          for (int k = 0; k < C; k++) {
            const int64_t us = A::to_i64(ext ? pw_[k] : A::round_int(A::muli(pw_[k], 4)));
            const u32x4 ur = {(uint32_t)us, (uint32_t)((uint64_t)us >> 32), px.tag, 0u};
            __builtin_amdgcn_raw_buffer_store_b128(ur, ursrc, (uc * C + k) * 16, 0, /*sc1*/ 16);
          }
        }
        if (can)
          stage = 3;
      }

      if (!progressed) {
        if (++spins > (1u << 21)) {
          if (lane == 0)
            atomicExch(ctx.error, 1);
          break;
        }
        __builtin_amdgcn_s_sleep(GPCC_SUB_SLEEP);
      }
    }
    prof.mark(6, lane);  // the loop
  }
  // ArithF64: a value left the exact range -- see raht_subnode.hpp
  if (A::kF64 && __any(!in_range) && lane == 0)
    atomicCAS(ctx.error, 0, 3);
}

}  // namespace gpcc
